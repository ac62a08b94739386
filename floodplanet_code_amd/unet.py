"""HipUNet: the reference UNet (st_water_seg/models/unet.py:80-111) executed by libfloodunet.so.

The module owns nn.Parameters / BN buffers with EXACTLY the reference's state_dict keys and OIHW
shapes (``inc.double_conv.0.weight`` ... ``outc.conv.bias``), so checkpoints, ``optim.Adam(self.parameters())``
and ``load_state_dict`` work unchanged.  All arithmetic (forward, loss, backward, optional Adam) happens in
hand-written gfx950 kernels behind the C ABI; torch only supplies device memory, streams and autograd glue.

Storage: every parameter is a view into one flat fp32 buffer in state_dict order (the same for gradients
and BN running statistics), which is what the C side binds to and what data-parallel all-reduce buckets slice.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch.nn import init

from . import _lib
from ._lib import FuConfig, check, ptr


# --------------------------------------------------------------------------------------------------
# parameter holders (names mirror nn.Conv2d / nn.BatchNorm2d inside the reference's nn.Sequential)
# --------------------------------------------------------------------------------------------------
class _ConvParams(nn.Module):
    def __init__(self, cin: int, cout: int, k: int, transposed: bool = False):
        super().__init__()
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(cout))
        # nn.Conv2d.reset_parameters (same RNG draws, so a seeded construction matches the reference)
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
        bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
        init.uniform_(self.bias, -bound, bound)


class _BNParams(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Holder(nn.Module):
    """Plain container; children are attached with add_module so keys match the reference."""


def _double_conv(cin: int, cmid: int, cout: int) -> nn.Module:
    seq = _Holder()
    seq.add_module("0", _ConvParams(cin, cmid, 3))
    seq.add_module("1", _BNParams(cmid))
    seq.add_module("3", _ConvParams(cmid, cout, 3))
    seq.add_module("4", _BNParams(cout))
    h = _Holder()
    h.add_module("double_conv", seq)
    return h


def channel_plan(n_channels: int, base: int, bilinear: bool):
    """unet.py:88-98 generalised to base_feat_channels (unet.py:143-149,176-183)."""
    factor = 2 if bilinear else 1
    e = [base, base * 2, base * 4, base * 8, base * 16 // factor]
    outs = [base * 8 // factor, base * 4 // factor, base * 2 // factor, base]
    return e, outs


def _build_encoder(root: nn.Module, n_channels: int, base: int, bilinear: bool):
    """inc, down1..down4 of UNetEncoder (unet.py:143-149) attached to `root`; construction order == reference so that
    torch.manual_seed(s) gives identical weights."""
    e, _ = channel_plan(n_channels, base, bilinear)
    root.inc = _double_conv(n_channels, e[0], e[0])
    for i in range(1, 5):
        d = _Holder()
        mp = _Holder()
        mp.add_module("1", _double_conv(e[i - 1], e[i], e[i]))
        d.add_module("maxpool_conv", mp)
        root.add_module(f"down{i}", d)


def _build_decoder(root: nn.Module, n_classes: int, base: int, bilinear: bool):
    """up1..up4, outc of UNetDecoder (unet.py:176-184, channel_factor 1) attached to `root`."""
    e, outs = channel_plan(1, base, bilinear)
    low = e[4]
    for k in range(4):
        u = _Holder()
        skip = e[3 - k]
        if bilinear:
            cin = low + skip
            u.add_module("conv", _double_conv(cin, cin // 2, outs[k]))
        else:
            u.add_module("up", _ConvParams(low, low // 2, 2, transposed=True))
            u.add_module("conv", _double_conv(low // 2 + skip, outs[k], outs[k]))
        root.add_module(f"up{k + 1}", u)
        low = outs[k]
    oc = _Holder()
    oc.add_module("conv", _ConvParams(base, n_classes, 1))
    root.outc = oc


class HipUNet(nn.Module):
    """Drop-in for ``UNet(n_channels, n_classes, bilinear=True)`` running on MI355X HIP kernels.

    precision: 'fp32' (exact-f32 MFMA; logits within 1e-4 of the reference) or 'bf16'.
    """

    def __init__(self, n_channels: int, n_classes: int, bilinear: bool = True, base_channels: int = 64,
                 precision: str = "fp32"):
        super().__init__()
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}")
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        self.base_channels, self.precision = base_channels, precision
        self._build_tree()

        self._table: List[Tuple[str, nn.Parameter, int, int]] = []  # name, param, offset, numel
        off = 0
        for name, p in self.named_parameters():
            self._table.append((name, p, off, p.numel()))
            off += p.numel()
        self._total = off
        self._bn: List[Tuple[str, _BNParams, int]] = []
        boff = 0
        for name, m in self.named_modules():
            if isinstance(m, _BNParams):
                self._bn.append((name, m, boff))
                boff += m.weight.numel()
        self._total_bn = boff
        self._flat: Optional[torch.Tensor] = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._flat_rm = self._flat_rv = self._flat_nbt = None
        # Adam moments (torch.optim.Adam's exp_avg / exp_avg_sq) as flat caller-owned buffers: bound into every context
        # (fu_bind_adam_state), so they survive context re-creation (another tile size, a larger batch, .to(device))
        self._flat_m: Optional[torch.Tensor] = None
        self._flat_v: Optional[torch.Tensor] = None
        self._generation = 0          # counts training forwards: an autograd node may only backward the latest one
        self._flat_valid = False
        self._ctx = None
        self._ctx_key = None
        self._eval_dirty = True
        self._confusion: Optional[torch.Tensor] = None
        self._exact = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._mark_dirty())

    def _build_tree(self):
        n_channels, base_channels, bilinear, n_classes = self.n_channels, self.base_channels, self.bilinear, self.n_classes
        _build_encoder(self, n_channels, base_channels, bilinear)
        _build_decoder(self, n_classes, base_channels, bilinear)

    def _enc_config(self):
        """(n_encoders, enc_channels) of fu_config: 0 = the plain UNet."""
        return 0, []

    def _c_param_name(self, name: str) -> str:
        """state_dict key -> the C side's canonical name (identity for the plain UNet)."""
        return name

    # ---------------------------------------------------------------- flat storage
    def _mark_dirty(self):
        self._eval_dirty = True

    def _apply(self, fn, *args, **kwargs):
        self._flat_valid = False
        self._eval_dirty = True
        return super()._apply(fn, *args, **kwargs)

    def _flatten(self, device: torch.device):
        with torch.no_grad():
            flat = torch.empty(self._total, dtype=torch.float32, device=device)
            for _, p, off, n in self._table:
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
            rm = torch.empty(self._total_bn, dtype=torch.float32, device=device)
            rv = torch.empty(self._total_bn, dtype=torch.float32, device=device)
            nbt = torch.empty(len(self._bn), dtype=torch.int64, device=device)
            for i, (_, m, off) in enumerate(self._bn):
                c = m.weight.numel()
                rm[off:off + c].copy_(m.running_mean)
                rv[off:off + c].copy_(m.running_var)
                nbt[i].copy_(m.num_batches_tracked)
                m.running_mean = rm[off:off + c]
                m.running_var = rv[off:off + c]
                m.num_batches_tracked = nbt[i]
            self._flat, self._flat_rm, self._flat_rv, self._flat_nbt = flat, rm, rv, nbt
            self._flat_grad = torch.zeros(self._total, dtype=torch.float32, device=device)
            for p_ in (p for _, p, _, _ in self._table):
                p_.grad = None          # (a .grad that aliased the previous flat gradient buffer would go stale)
            if self._flat_m is not None and self._flat_m.numel() == self._total:
                self._flat_m = self._flat_m.to(device)      # the optimiser state moves with the module, never resets
                self._flat_v = self._flat_v.to(device)
            else:
                self._flat_m = torch.zeros(self._total, dtype=torch.float32, device=device)
                self._flat_v = torch.zeros(self._total, dtype=torch.float32, device=device)
        self._flat_valid = True
        self._destroy_ctx()

    def flat_parameters(self) -> torch.Tensor:
        return self._flat

    def flat_grads(self) -> torch.Tensor:
        return self._flat_grad

    def grad_views(self) -> List[torch.Tensor]:
        g = self._flat_grad
        return [g[off:off + n].view(p.shape) for _, p, off, n in self._table]

    def adam_state(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(exp_avg, exp_avg_sq) flat fp32 buffers in parameter order (torch.optim.Adam's state, for checkpoints)."""
        return self._flat_m, self._flat_v

    def reset_adam_state(self):
        if self._flat_m is not None:
            self._flat_m.zero_()
            self._flat_v.zero_()

    def attach_grads(self):
        """Point every parameter's .grad at its slice of the flat gradient buffer."""
        for (_, p, off, n), gv in zip(self._table, self.grad_views()):
            p.grad = gv

    # ---------------------------------------------------------------- context
    def _destroy_ctx(self):
        if self._ctx is not None:
            _lib.load().fu_destroy(self._ctx)
            self._ctx = None
            self._ctx_key = None

    def __del__(self):
        try:
            self._destroy_ctx()
        except Exception:
            pass

    def _get_ctx(self, device: torch.device, B: int, H: int, W: int):
        if device.type != "cuda":
            raise RuntimeError("HipUNet runs only on a ROCm GPU (device type 'cuda'); there is no CPU fallback")
        if not self._flat_valid or self._flat is None or self._flat.device != device:
            self._flatten(device)
        key = (device.index, H, W)
        if self._ctx is not None and self._ctx_key[:3] == key and self._ctx_key[3] >= B:
            return self._ctx
        self._destroy_ctx()
        lib = _lib.load()
        n_enc, enc_ch = self._enc_config()
        cfg = FuConfig(C.sizeof(FuConfig), self.n_channels, self.n_classes, self.base_channels, int(self.bilinear),
                       B, H, W, _lib.PRECISIONS[self.precision], device.index if device.index is not None else 0,
                       n_enc, (C.c_int32 * 6)(*(list(enc_ch) + [0] * (6 - len(enc_ch)))))
        h = C.c_void_p()
        check(lib.fu_create(C.byref(cfg), C.byref(h)))
        self._ctx = h
        self._ctx_key = key + (B,)
        self._verify_table()
        check(lib.fu_bind_buffers(h, ptr(self._flat), ptr(self._flat_grad), ptr(self._flat_rm), ptr(self._flat_rv),
                                  ptr(self._flat_nbt)))
        check(lib.fu_bind_adam_state(h, ptr(self._flat_m), ptr(self._flat_v)))
        self._eval_dirty = True
        self._install_exact_sync(device)
        return h

    # ---------------------------------------------------------------- exact data-parallel mode
    def enable_exact_sync(self, world_size: int, group=None, enabled: bool = True):
        """SyncBN statistics + global N_valid over the ranks of `group` (include/floodunet.h, fu_set_exact_sync):
        W ranks x B tiles then reproduce one device with W*B tiles.  Gradients hold each rank's share of the global
        gradient afterwards: all-reduce them with SUM and use grad_scale 1 (DataParallelTrainer(exact=True))."""
        self._exact = (int(world_size), group) if enabled and world_size > 1 else None
        if self._ctx is not None:
            self._install_exact_sync(self._flat.device)

    def _install_exact_sync(self, device):
        lib = _lib.load()
        exact = getattr(self, "_exact", None)
        if exact is None:
            if self._ctx is not None:
                check(lib.fu_set_exact_sync(self._ctx, _lib.SYNC_HOOK(0), None, 1, None, 0))   # null hook: off
            return
        import torch.distributed as dist
        world, group = exact
        nbytes = int(lib.fu_exact_sync_bytes(self._ctx))
        self._xbuf = torch.zeros(nbytes // 8 + 1, dtype=torch.float64, device=device)
        xbuf = self._xbuf

        def hook(_user, n_elems, is_double):
            try:
                t = xbuf[:n_elems] if is_double else xbuf.view(torch.float32)[:n_elems]
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)   # ordered on the current stream
                return 0
            except Exception:   # never let an exception cross the C ABI
                import traceback
                traceback.print_exc()
                return 1

        self._sync_cb = _lib.SYNC_HOOK(hook)     # keep the callback object alive as long as the context
        check(lib.fu_set_exact_sync(self._ctx, self._sync_cb, None, world, ptr(xbuf), nbytes))

    def _verify_table(self):
        lib = _lib.load()
        n = lib.fu_num_params(self._ctx)
        if n != len(self._table) or lib.fu_total_param_elems(self._ctx) != self._total:
            raise RuntimeError("parameter table mismatch between HipUNet and libfloodunet")
        name = C.c_char_p()
        ndim = C.c_int32()
        shape = (C.c_int64 * 4)()
        off = C.c_int64()
        for i, (pname, p, poff, _) in enumerate(self._table):
            check(lib.fu_param_info(self._ctx, i, C.byref(name), C.byref(ndim), shape, C.byref(off)))
            if name.value.decode() != self._c_param_name(pname) or off.value != poff or \
                    tuple(shape[:ndim.value]) != tuple(p.shape):
                raise RuntimeError(f"parameter {i}: python {pname}{tuple(p.shape)}@{poff} vs C "
                                   f"{name.value.decode()}{tuple(shape[:ndim.value])}@{off.value}")

    @staticmethod
    def _stream(device) -> int:
        return torch.cuda.current_stream(device).cuda_stream

    # ---------------------------------------------------------------- raw calls
    def _forward_raw(self, x, training: bool, want_logits: bool = True) -> Optional[torch.Tensor]:
        """x: the input tensor [B, n_channels, H, W], or a list / tuple of tensors [B, C_k, H, W] that the model sees side by
        side along the channel axis (ef_model.py:28-44 concatenates them; here fu_forward_srcs gathers them inside the
        NCHW -> NHWC conversion: no concatenated copy)."""
        srcs = list(x) if isinstance(x, (list, tuple)) else [x]
        if not srcs or any(t.dim() != 4 for t in srcs) or sum(t.shape[1] for t in srcs) != self.n_channels:
            raise ValueError(f"expected input [B,{self.n_channels},H,W] (in one tensor or split along C), got "
                             f"{[tuple(t.shape) for t in srcs]}")
        srcs = [t.detach().contiguous().float() for t in srcs]
        B, _, H, W = srcs[0].shape
        dev = srcs[0].device
        if any(t.shape[0] != B or t.shape[2:] != (H, W) or t.device != dev for t in srcs):
            raise ValueError("all input tensors must share batch, tile size and device")
        ctx = self._get_ctx(dev, B, H, W)
        lib = _lib.load()
        if training or self._eval_dirty:
            check(lib.fu_params_changed(ctx))
            self._eval_dirty = training  # a training step is followed by an optimiser update
        logits = torch.empty(B, self.n_classes, H, W, dtype=torch.float32, device=dev) if want_logits else None
        if len(srcs) == 1:
            check(lib.fu_forward(ctx, ptr(srcs[0]), B, int(training), ptr(logits), self._stream(dev)))
        else:
            arr = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
            chs = (C.c_int32 * len(srcs))(*[t.shape[1] for t in srcs])
            check(lib.fu_forward_srcs(ctx, arr, chs, len(srcs), B, int(training), ptr(logits), self._stream(dev)))
        if training:
            self._generation += 1
        return logits

    def _loss_raw(self, target: torch.Tensor, ignore_index: int, device, kind: str = "ce",
                  dice_weight: float = 1.0) -> torch.Tensor:
        lib = _lib.load()
        target = target.contiguous().long()
        loss = torch.empty((), dtype=torch.float32, device=device)
        if kind == "bce_dice":
            check(lib.fu_loss_bce_dice(self._ctx, ptr(target), int(ignore_index), float(dice_weight), ptr(loss),
                                       self._stream(device)))
            return loss
        if kind != "ce":
            raise ValueError(f"unknown loss kind {kind!r}")
        if self._confusion is None or self._confusion.device != device:
            self._confusion = torch.zeros(self.n_classes * self.n_classes, dtype=torch.int64, device=device)
        check(lib.fu_loss_ce(self._ctx, ptr(target), int(ignore_index), ptr(loss), ptr(self._confusion), None,
                             self._stream(device)))
        return loss

    def _backward_raw(self, dlogits: Optional[torch.Tensor], device):
        check(_lib.load().fu_backward(self._ctx, ptr(dlogits), self._stream(device)))

    def pop_confusion(self) -> Optional[torch.Tensor]:
        """[n_classes, n_classes] int64 confusion counts (target row, argmax column) accumulated by the fused
        loss calls since the last pop."""
        if self._confusion is None:
            return None
        out = self._confusion.view(self.n_classes, self.n_classes).clone()
        self._confusion.zero_()
        return out

    # ---------------------------------------------------------------- public API
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """logits = UNet(x).  Differentiable w.r.t. the parameters when grad mode is on and the module trains."""
        if self.training and torch.is_grad_enabled():
            params = [p for _, p, _, _ in self._table]
            return _UNetFn.apply(self, x, *params)
        return self._forward_raw(x, self.training)

    def loss(self, x: torch.Tensor, target: torch.Tensor, ignore_index: int,
             return_logits: bool = False, kind: str = "ce", dice_weight: float = 1.0):
        """Fused forward + CrossEntropyLoss(ignore_index) (+ NaN guard) of water_seg_model.py:101-106
        (kind='ce', the reference's loss) or the BCE + soft-Dice extension (kind='bce_dice').
        The returned loss is differentiable: ``loss.backward()`` runs the HIP backward."""
        if self.training and torch.is_grad_enabled():
            params = [p for _, p, _, _ in self._table]
            out = _UNetLossFn.apply(self, x, target, int(ignore_index), bool(return_logits), kind,
                                    float(dice_weight), *params)
            return out if return_logits else out[0]
        logits = self._forward_raw(x, self.training, want_logits=return_logits)
        loss = self._loss_raw(target, ignore_index, _device_of(x), kind, dice_weight)
        return (loss, logits) if return_logits else loss

    def train_step(self, x: torch.Tensor, target: torch.Tensor, ignore_index: int, kind: str = "ce",
                   dice_weight: float = 1.0) -> torch.Tensor:
        """forward + loss + backward without autograd; gradients land in the flat buffer / p.grad."""
        self._forward_raw(x, True, want_logits=False)
        loss = self._loss_raw(target, ignore_index, _device_of(x), kind, dice_weight)
        self._backward_raw(None, _device_of(x))
        self.attach_grads()
        return loss

    def adam_step(self, lr: float, step: int, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
        """Native fused Adam on the flat buffers (torch.optim.Adam semantics, water_seg_model.py:200)."""
        dev = self._flat.device
        check(_lib.load().fu_adam_step(self._ctx, float(lr), float(betas[0]), float(betas[1]), float(eps), int(step),
                                       float(grad_scale), self._stream(dev)))
        self._eval_dirty = True

    def fp16_guard_state(self) -> Tuple[int, int]:
        """(optimiser steps skipped because a gradient was not finite, current loss-scale back-off exponent) -- fp16 mode;
        (0, 0) otherwise.  Synchronises the device."""
        n, e = C.c_int64(), C.c_int32()
        check(_lib.load().fu_fp16_guard_state(self._ctx, C.byref(n), C.byref(e)))
        return n.value, e.value

    def flops_per_tile(self) -> Tuple[float, float]:
        f, t = C.c_double(), C.c_double()
        check(_lib.load().fu_flops_per_tile(self._ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

    def block_ranges(self) -> List[Tuple[int, int]]:
        """(offset, numel) of each backward block's gradients in the flat buffer, in backward order."""
        lib = _lib.load()
        out = []
        o, n = C.c_int64(), C.c_int64()
        for b in range(lib.fu_num_blocks(self._ctx)):
            check(lib.fu_block_param_range(self._ctx, b, C.byref(o), C.byref(n)))
            out.append((o.value, n.value))
        return out


def _device_of(x):
    return (x[0] if isinstance(x, (list, tuple)) else x).device


def _check_generation(ctx):
    m = ctx.module
    if ctx.generation != m._generation:
        raise RuntimeError(
            "HipUNet: backward() of a forward pass that is no longer the latest training forward of this module. The "
            "saved activations live in the module's single device context (one forward in flight): call backward() "
            "before the next training forward.")


def _save_accumulated(m: "HipUNet"):
    """Before fu_backward overwrites the flat gradient buffer: if some p.grad ALIASES its slice of that buffer (the state
    the fast path of _return_param_grads leaves behind, also after zero_grad(set_to_none=False)), the values about to be
    overwritten are gradients autograd is expected to accumulate into -- keep a copy.  Returns (saved flat buffer or None,
    per-parameter alias flags)."""
    views = m.grad_views()
    alias = [p.grad is not None and p.grad.data_ptr() == v.data_ptr() for (_, p, _, _), v in zip(m._table, views)]
    return (m._flat_grad.clone() if any(alias) else None), alias


def _return_param_grads(m: "HipUNet", saved, alias):
    """Gradients of the HIP backward as autograd results.
    * p.grad is None (right after optimizer.zero_grad(set_to_none=True), what Lightning and torch >= 2.0 do): point p.grad at
      the parameter's slice of the flat gradient buffer the kernels have just written and hand autograd nothing to
      accumulate -- no 74 clones, no 74 accumulations, and HipAdam finds the gradients where fu_adam_step reads them.
    * p.grad aliases that slice (a second backward before the optimiser step: gradient accumulation): the kernels have just
      overwritten the accumulated value; add the saved copy back on the device, p.grad then holds old + new, and again
      nothing is returned for autograd to add (returning a clone here would make autograd add it INTO the aliased slice:
      2 * new instead of old + new).
    * p.grad lives elsewhere (set by the user): return a copy and let autograd accumulate into it."""
    views = m.grad_views()
    out = []
    for (_, p, off, n), v, al in zip(m._table, views, alias):
        if p.grad is None:
            p.grad = v
            out.append(None)
        elif al:
            v.add_(saved[off:off + n].view(p.shape))
            out.append(None)
        else:
            out.append(v.clone())
    return tuple(out)


class _UNetFn(torch.autograd.Function):
    """logits = f(x; params): forward through fu_forward, backward through fu_backward(dlogits)."""

    @staticmethod
    def forward(ctx, module: HipUNet, x, *params):
        ctx.module = module
        ctx.device = _device_of(x)
        out = module._forward_raw(x, True)
        ctx.generation = module._generation
        return out

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.module
        _check_generation(ctx)
        saved, alias = _save_accumulated(m)
        m._backward_raw(dlogits.contiguous().float(), ctx.device)
        return (None, None) + _return_param_grads(m, saved, alias)


class _UNetLossFn(torch.autograd.Function):
    """(loss, logits) = CE(f(x; params), target): the fused training_step path."""

    @staticmethod
    def forward(ctx, module: HipUNet, x, target, ignore_index, want_logits, kind, dice_weight, *params):
        ctx.module = module
        ctx.device = _device_of(x)
        logits = module._forward_raw(x, True, want_logits=want_logits)
        ctx.generation = module._generation
        loss = module._loss_raw(target, ignore_index, ctx.device, kind, dice_weight)
        if logits is None:
            logits = torch.empty(0, device=ctx.device)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        m = ctx.module
        _check_generation(ctx)
        # the upstream gradient of the loss (ones for a plain loss.backward()) scales dL/dlogits once, on the device
        dl = dloss.detach().reshape(()).to(device=ctx.device, dtype=torch.float32)
        check(_lib.load().fu_scale_loss_grad(m._ctx, ptr(dl), m._stream(ctx.device)))
        saved, alias = _save_accumulated(m)
        m._backward_raw(None, ctx.device)
        return (None, None, None, None, None, None, None) + _return_param_grads(m, saved, alias)


class HipAdam(torch.optim.Adam):
    """A torch.optim.Adam (isinstance holds; same param_groups / defaults) (water_seg_model.py:198-205: lr, default betas / eps, no weight decay, no amsgrad) whose
    step() is ONE launch of libfloodunet's fused Adam kernel on the module's flat parameter / gradient / moment
    buffers (fu_adam_step).  param_groups[0] carries lr / betas / eps as torch's Adam does (LR schedulers work);
    state_dict() holds {step, exp_avg, exp_avg_sq} per parameter like torch.optim.Adam's, the moments being views of the
    module's flat buffers."""

    def __init__(self, net: HipUNet, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if not isinstance(net, HipUNet):
            raise TypeError("HipAdam drives a HipUNet's flat buffers")
        super().__init__([p for _, p, _, _ in net._table], lr=lr, betas=betas, eps=eps)
        self._net = net
        self._step = 0

    def _views(self):
        net = self._net
        return [(p, net._flat_m[off:off + n].view(p.shape), net._flat_v[off:off + n].view(p.shape))
                for _, p, off, n in net._table]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        net = self._net
        if net._ctx is None:
            raise RuntimeError("HipAdam.step() before any forward / backward of the module")
        grads = net.grad_views()
        if any(p.grad is None for _, p, _, _ in net._table):
            raise RuntimeError("HipAdam.step(): a parameter has no gradient (the fused kernel updates all of them)")
        # gradients that autograd accumulated outside the flat buffer (the slow path of _return_param_grads) come home
        outside = [(g, p.grad) for (_, p, _, _), g in zip(net._table, grads) if p.grad.data_ptr() != g.data_ptr()]
        if outside:
            torch._foreach_copy_([a for a, _ in outside], [b for _, b in outside])
        g = self.param_groups[0]
        self._step += 1
        net.adam_step(g["lr"], self._step, g["betas"], g["eps"])
        return loss

    def state_dict(self):
        sd = super().state_dict()
        if self._net._flat_m is not None:
            sd["state"] = {i: {"step": torch.tensor(float(self._step)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
                           for i, (_, m, v) in enumerate(self._views())}
        return sd

    def load_state_dict(self, state_dict):
        st = state_dict.get("state", {})
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        if st:
            if self._net._flat_m is None:
                raise RuntimeError("HipAdam.load_state_dict(): move the module to its device first")
            for i, (_, m, v) in enumerate(self._views()):
                e = st[i] if i in st else st[str(i)]
                m.copy_(e["exp_avg"])
                v.copy_(e["exp_avg_sq"])
                self._step = int(e["step"])
