"""One-process-per-GPU data parallelism for the UNet training step (new functionality: the reference trains
on a single device, st_water_seg/fit.py:87-88, so the semantics are defined here and in DESIGN.md):

  * default = DDP semantics: per-rank BatchNorm statistics and per-rank 1/N_valid loss normaliser, gradients averaged
    over ranks (sum all-reduce, 1/world folded into the Adam kernel's grad_scale).
  * exact=True: SyncBN statistics (forward and backward sums) and a global N_valid through fu_set_exact_sync, gradients
    summed: W ranks x B tiles reproduce one device with W*B tiles up to fp32 summation order (the parity mode).
  * the only exchange is the gradient all-reduce over RCCL (torch.distributed backend "nccl").  Backward runs
    block by block (fu_backward_block); as soon as the blocks of a bucket are final, the bucket -- one contiguous
    slice of the flat gradient buffer -- is all-reduced asynchronously on RCCL's stream while the remaining
    backward kernels keep the compute stream busy.  5 buckets for the full-width net at the default 16 MB cap
    (7.8 / 23.6 / 18.9 / 14.2 / 4.6 MB; a block larger than the cap is a bucket by itself): xGMI is point-to-point, so
    few large messages amortise the per-collective latency, and only the last 4.6 MB (down2, down1, inc) cannot overlap
    with backward (with a 25 MB cap down3's 14.2 MB waited for the end of backward too).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os

import torch
import torch.distributed as dist

# rehearsal knob: drive the block-wise backward (the multi-GPU code path) in a single process
_FORCE_BLOCKS = os.environ.get("FU_DP_FORCE_BLOCKS") == "1"
_DIAG_MODE = None      # tools/dp_diag.py: override of the side-stream mode


def plan_buckets(block_ranges: Sequence[Tuple[int, int]], cap_bytes: int = 16 << 20,
                 elem_bytes: int = 4) -> List[Tuple[int, int, int]]:
    """Merge consecutive backward blocks into buckets.  block_ranges: (offset, numel) per block in backward
    order; consecutive blocks are adjacent in the flat buffer.  Returns (last_block_index, offset, numel)."""
    buckets: List[Tuple[int, int, int]] = []
    cur_lo = cur_hi = None
    for b, (off, n) in enumerate(block_ranges):
        lo, hi = off, off + n
        if cur_lo is None:
            cur_lo, cur_hi = lo, hi
        else:
            new_lo, new_hi = min(cur_lo, lo), max(cur_hi, hi)
            if (new_hi - new_lo) != (cur_hi - cur_lo) + n:
                raise ValueError("backward blocks are not adjacent in the flat gradient buffer")
            if (new_hi - new_lo) * elem_bytes > cap_bytes:
                buckets.append((b - 1, cur_lo, cur_hi - cur_lo))
                cur_lo, cur_hi = lo, hi
            else:
                cur_lo, cur_hi = new_lo, new_hi
        if b == len(block_ranges) - 1:
            buckets.append((b, cur_lo, cur_hi - cur_lo))
    return buckets


class BucketedReducer:
    """Asynchronous bucketed sum all-reduce of a flat gradient buffer, driven by block completion."""

    def __init__(self, block_ranges: Sequence[Tuple[int, int]], world_size: int, group=None,
                 cap_bytes: int = 16 << 20, timing: bool = False):
        self.world_size = world_size
        self.group = group
        self.buckets = plan_buckets(block_ranges, cap_bytes)
        self._by_last = {last: (off, n) for last, off, n in self.buckets}
        self._pending = []
        # diagnostics (bench.py, world > 1): per step an event pair on the compute stream around the waits of finish() --
        # the time the compute stream stands still for all-reduces that backward did not cover -- and the host time there
        self.timing = timing
        self._wait_events = []
        self._host_wait_s = []

    def bucket_bytes(self, elem_bytes: int = 4) -> List[int]:
        return [n * elem_bytes for _, _, n in self.buckets]

    def block_done(self, flat_grad: torch.Tensor, block: int, launch_stream=None) -> None:
        if self.world_size <= 1 or block not in self._by_last:
            return
        off, n = self._by_last[block]
        # all_reduce(async_op=True) orders itself after the work already enqueued on the current stream and
        # runs on the process group's own stream: the next backward blocks overlap with it.  launch_stream: the stream
        # that is "current" for the call -- one that waits for the bucket's gradients (fu_backward_fence) while the compute
        # stream does not.
        if launch_stream is not None:
            with torch.cuda.stream(launch_stream):
                w = dist.all_reduce(flat_grad[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            w = dist.all_reduce(flat_grad[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._pending.append(w)

    def finish(self) -> None:
        timed = self.timing and self._pending and torch.cuda.is_available()
        if timed:
            import time
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
        for w in self._pending:
            w.wait()          # nccl: the current stream waits for the collective's stream; gloo: the host waits
        if timed:
            e1.record()
            self._host_wait_s.append(time.perf_counter() - t0)
            self._wait_events.append((e0, e1))
        self._pending.clear()

    def exposed_wait_ms(self) -> Optional[dict]:
        """Per-step time the compute stream was blocked in finish() (device events) and the host spent there; call after a
        device synchronise.  None when nothing was timed."""
        if not self._wait_events:
            return None
        dev = [a.elapsed_time(b) for a, b in self._wait_events]
        host = [1e3 * t for t in self._host_wait_s]
        dev_s, host_s = sorted(dev), sorted(host)
        return {"steps": len(dev), "device_mean": sum(dev) / len(dev), "device_median": dev_s[len(dev) // 2],
                "device_max": dev_s[-1], "host_mean": sum(host) / len(host), "host_median": host_s[len(host) // 2]}

    def reset_timing(self) -> None:
        self._wait_events.clear()
        self._host_wait_s.clear()


class DataParallelTrainer:
    """fwd + CE + block-wise bwd (+ overlapped all-reduce) + fused Adam on a HipUNet."""

    def __init__(self, net, lr: float, world_size: int = 1, rank: int = 0, betas=(0.9, 0.999), eps: float = 1e-8,
                 group=None, cap_bytes: int = 16 << 20, exact: bool = False, time_waits: bool = False,
                 graph: bool = False):
        """exact=False: DDP semantics (per-rank BN statistics and 1/N_valid, gradients averaged).
        exact=True: SyncBN statistics and a global N_valid (HipUNet.enable_exact_sync); the ranks together reproduce
        one device with world_size x the batch, gradients are summed (SURVEY.md 8(e) "exact mode")."""
        self.net, self.lr, self.world_size, self.rank = net, lr, world_size, rank
        self.betas, self.eps, self.group, self.cap_bytes = betas, eps, group, cap_bytes
        self.exact = bool(exact) and world_size > 1
        if self.exact:
            net.enable_exact_sync(world_size, group)
        self.step_count = 0
        self.time_waits = time_waits
        # graph=True (single device): the whole step -- fu_forward, fu_loss_ce, fu_backward with its side-stream fork / join,
        # fu_adam_step_dev -- is captured once into a hipGraph (torch.cuda.CUDAGraph on the capture stream the C calls are
        # issued on) and replayed; only the seven Adam scalars, which depend on the step count, are refreshed per step.  The
        # captured launches are valid for ONE (x, target) pair of device buffers: other tensors are copied into them.
        self.graph = bool(graph) and world_size <= 1
        self._graph = None
        self._gx = self._gt = self._gloss = self._gscal = self._gscal_host = None
        self._g_ignore = None
        self._reducer: Optional[BucketedReducer] = None
        self._launch_stream = None     # the stream the bucket all-reduces are launched from (mode 2, fu_backward_fence)
        self._synced = False

    def _side_mode(self) -> int:
        if _DIAG_MODE is not None:
            return _DIAG_MODE
        env = os.environ.get("FU_DP_SIDE_MODE")
        if env is not None:
            return int(env)
        if self.world_size <= 1:
            return 2
        return 2 if dist.get_backend(self.group) == "nccl" else 0

    def _sync_initial_state(self, device):
        net = self.net
        if not net._flat_valid:
            net._flatten(device)
        for t in (net._flat, net._flat_rm, net._flat_rv, net._flat_nbt):
            dist.broadcast(t, src=0, group=self.group)
        net._mark_dirty()
        self._synced = True

    # ------------------------------------------------------------------ captured step
    def _adam_scalars_to_device(self, step: int):
        import ctypes as C
        from . import _lib
        buf = (C.c_float * 7)()
        _lib.check(_lib.load().fu_adam_scalars(float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                               int(step), 1.0, buf))
        # A ring of pinned slots, each guarded by an event recorded behind its host-to-device copy: nothing in a replayed step
        # synchronises, so the host runs steps ahead of the GPU, and ONE pinned buffer would be rewritten for step k+1, k+2, ...
        # before the copy of step k has executed (the replay of step k would then take a later step's bias corrections).
        slot = step % len(self._gscal_ring)
        host, ev = self._gscal_ring[slot], self._gscal_ev[slot]
        if ev is not None:
            ev.synchronize()                                          # the copy that last read this slot has executed
        host.copy_(torch.tensor(list(buf), dtype=torch.float32))
        self._gscal.copy_(host, non_blocking=True)                    # ordered on the current stream, ahead of the replay
        ev = torch.cuda.Event()
        ev.record()
        self._gscal_ev[slot] = ev

    def _capture(self, x, target, ignore_index):
        from . import _lib
        net, dev = self.net, x.device
        # private static buffers: capturing the caller's first batch in place would overwrite it on every later step (and skip
        # the copy -- i.e. train on the previous batch -- whenever that first tensor came round again)
        self._gx, self._gt, self._g_ignore = torch.empty_like(x), torch.empty_like(target), int(ignore_index)
        self._gx.copy_(x)
        self._gt.copy_(target)
        self._gscal = torch.zeros(7, dtype=torch.float32, device=dev)
        self._gscal_ring = [torch.zeros(7, dtype=torch.float32).pin_memory() for _ in range(8)]
        self._gscal_ev = [None] * 8
        lib = _lib.load()
        torch.cuda.synchronize(dev)
        dot = os.environ.get("FU_GRAPH_DOT")                          # diagnostics: hipGraphDebugDotPrint of the captured step (tools/graph_dot.py)
        g = torch.cuda.CUDAGraph(keep_graph=True) if dot else torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            net._forward_raw(self._gx, True, want_logits=False)
            self._gloss = net._loss_raw(self._gt, self._g_ignore, dev)
            net._backward_raw(None, dev)
            _lib.check(lib.fu_adam_step_dev(net._ctx, self._gscal.data_ptr(), net._stream(dev)))
        net._generation -= 1          # (the capture enqueued nothing; the bookkeeping of _forward_raw is redone per replay)
        if dot:       # (torch's own debug_dump writes nothing on this ROCm build)
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
            rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(g.raw_cuda_graph()), dot.encode(), 0)
            if rc != 0:
                raise RuntimeError(f"hipGraphDebugDotPrint -> {rc}")
        self._graph = g

    def _graph_step(self, x, target, ignore_index):
        net = self.net
        if self._graph is not None and (int(ignore_index) != self._g_ignore or x.shape != self._gx.shape):
            self._graph = None                                        # another workload: capture again
        if self._graph is None:
            if net._ctx is None or net._ctx_key[1:3] != tuple(x.shape[2:]) or net._ctx_key[3] < x.shape[0]:
                return None                                           # no context for this shape yet: one eager step first
            self._capture(x.detach().contiguous().float(), target.contiguous().long(), ignore_index)
        self._gx.copy_(x)
        self._gt.copy_(target)
        self.step_count += 1
        self._adam_scalars_to_device(self.step_count)
        self._graph.replay()
        net._generation += 1
        net._eval_dirty = True
        net.attach_grads()
        return self._gloss.clone()       # (the graph's own loss tensor is rewritten by the next replay)

    def step(self, x: torch.Tensor, target: torch.Tensor, ignore_index: int) -> torch.Tensor:
        from . import _lib
        net = self.net
        if self.graph and not _FORCE_BLOCKS and not self.exact and not getattr(self, "_graph_off", False):
            out = self._graph_step(x, target, ignore_index)
            if out is not None:
                return out
        if self.world_size > 1 and not self._synced:
            self._sync_initial_state(x.device)
        net._forward_raw(x, True, want_logits=False)
        loss = net._loss_raw(target, ignore_index, x.device)
        if self.world_size <= 1 and not _FORCE_BLOCKS:
            net._backward_raw(None, x.device)
        else:
            if self._reducer is None:
                self._reducer = BucketedReducer(net.block_ranges(), self.world_size, self.group, self.cap_bytes,
                                                timing=self.time_waits)
            lib = _lib.load()
            stream = net._stream(x.device)
            flat = net.flat_grads()
            # weight gradients run on the context's side stream: join it only where a bucket ends (mode 2), not after
            # every block, so that the two chains stay concurrent inside a bucket.  Only with RCCL, whose collectives
            # are kernels on a stream of their own: gloo's worker thread synchronises streams from the host, and next
            # to the two-stream backward that was measured 30-80x slower (tools/dp_diag.py), so any
            # other backend gets the serial chain (mode 0).
            _lib.check(lib.fu_set_side_stream(net._ctx, self._side_mode()))
            nb = lib.fu_num_blocks(net._ctx)
            # Mode 2 (RCCL): at a bucket's end only the stream the all-reduce is launched from waits for the two backward
            # chains (fu_backward_fence); the compute stream goes straight on with the next block.  Joining the compute
            # stream there instead made it stand still until the weight-gradient chain had caught up, four times per step:
            # 5.63 against 5.45 ms per step in one process without any collective (tools/r3_blocks.sh).  The last block
            # joins: the optimizer step reads every gradient on the compute stream.
            fenced = self._side_mode() == 2 and x.is_cuda and os.environ.get("FU_DP_JOIN_AT_BUCKETS") != "1"
            if fenced and self._launch_stream is None:
                self._launch_stream = torch.cuda.Stream(device=x.device)
            for b in range(nb):
                _lib.check(lib.fu_backward_block(net._ctx, b, None, stream))
                if b == nb - 1:
                    _lib.check(lib.fu_backward_join(net._ctx, stream))
                    self._reducer.block_done(flat, b)
                elif b in self._reducer._by_last:
                    if fenced:
                        _lib.check(lib.fu_backward_fence(net._ctx, stream, self._launch_stream.cuda_stream))
                        self._reducer.block_done(flat, b, self._launch_stream)
                    else:
                        _lib.check(lib.fu_backward_join(net._ctx, stream))
                        self._reducer.block_done(flat, b)
            _lib.check(lib.fu_set_side_stream(net._ctx, 1))
            self._reducer.finish()
        self.step_count += 1
        net.adam_step(self.lr, self.step_count, self.betas, self.eps,
                      grad_scale=1.0 if self.exact else 1.0 / self.world_size)
        return loss
