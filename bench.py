#!/usr/bin/env python
"""bench.py -- training tiles/sec of the MI355X UNet hot path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic tiles already resident in HBM:
forward (train-mode BN) + CrossEntropy(ignore_index) + backward + [gradient all-reduce when N>1] + Adam.
Workload at N=1: BASELINE.json configs[1] -- UNet depth-4, 8-band 256x256 tiles, batch 16 per GPU.

One JSON line on stdout (rank 0).  See DESIGN.md "Measurement" for the definitions of `roofline`
(HIP events around every launch of the dominant kernel, live in the timed region) and `cpu_baseline`
(the oracle's restatement of the reference's torch-CPU step, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default=os.environ.get("FU_BENCH_DTYPE", "bf16"), choices=["f32", "bf16"],
                    help="bf16 = BASELINE.json configs[1] (default); f32 = the 1e-4 parity mode")
    ap.add_argument("--model", default="unet", choices=["unet", "lf"],
                    help="unet = the headline workload (ms_model / ef_model); lf = the late-fusion net (lf_model.py): "
                         "--channels image bands + --aux one-band auxiliary inputs, one encoder each")
    ap.add_argument("--aux", type=int, default=1, help="--model lf: number of one-band auxiliary inputs (dem, slope, ...)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serial-pass", action="store_true",
                    help="skip the extra un-timed pass that measures the dominant kernel without the side stream")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=20)
    return ap.parse_args()


def host_cores():
    """CPU cores this process may actually use: cgroup quota if there is one, else the affinity mask, capped at
    the GPU box's per-GPU CPU share (16) so that torch does not oversubscribe a 256-thread host."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, int(os.environ.get("FU_BENCH_CPU_THREADS", "16")))


AUX_NAMES = ["dem", "slope", "preflood", "pre_post_difference", "hand"]


def lf_in_channels(channels, aux):
    from collections import OrderedDict
    return OrderedDict([("ms_image", channels)] + [(k, 1) for k in AUX_NAMES[:aux]])


def cpu_baseline(channels, size, cpu_batch, cpu_steps, model="unet", aux=1):
    """The oracle (kind 'port': torch-CPU restatement of the reference step, pinned bit-exactly against the
    reference in the dev container) on this box's host cores, bounded sample."""
    from oracle import unet_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    if model == "lf":
        in_ch = lf_in_channels(channels, aux)
        st = O.lf_make_state(in_ch, 3, 64, seed=0)
        batch = O.make_batch(cpu_batch, channels, size, size, seed=1, extra=tuple(AUX_NAMES[:aux]))
        step = lambda: O.lf_train_step(st, opt, batch, in_ch, 0, 1e-4)      # noqa: E731
    else:
        st = O.make_state(channels, 3, 64, True, seed=0, nontrivial_bn=False)
        batch = O.make_batch(cpu_batch, channels, size, size, seed=1)
        step = lambda: O.train_step(st, opt, batch, 0, 1e-4)                # noqa: E731
    opt = O.new_adam_state(st)
    step()  # warm-up
    times = []
    for _ in range(cpu_steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(cpu_batch / best, 4), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"{cpu_steps} steps of batch {cpu_batch} ({channels}ch {size}x{size}, fp32 torch-CPU "
                      f"{torch.__version__}, {torch.get_num_threads()} threads), best step {best:.3f}s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with python -m torch.distributed.run --nproc-per-node N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(1, n_dev))  # (ranks share a device only in the gloo rehearsal)
    torch.cuda.set_device(dev)

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FU_DIST_BACKEND", "nccl")   # "gloo" = rehearsal of the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from floodplanet_code_amd import _lib
    from floodplanet_code_amd.distributed import DataParallelTrainer
    from floodplanet_code_amd.unet import HipUNet
    import ctypes as C

    precision = "fp32" if args.dtype == "f32" else "bf16"
    if os.environ.get("FU_BENCH_GENERAL_CONV") == "1":   # A/B knob: general bf16 conv kernel instead of the fast one
        _lib.load().fu_test_force_general_conv(1)
    torch.manual_seed(0)
    if args.model == "lf":
        from floodplanet_code_amd.latefusion import HipLateFusion
        net = HipLateFusion(lf_in_channels(args.channels, args.aux), 3, base_channels=64, precision=precision)
        net = net.to(dev).train()
    else:
        net = HipUNet(args.channels, 3, bilinear=True, base_channels=64, precision=precision).to(dev).train()
    trainer = DataParallelTrainer(net, lr=1e-4, world_size=world, rank=rank)

    B, Cc, S = args.batch, args.channels, args.size
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    x = torch.rand(B, net.n_channels, S, S, device=dev, generator=g)   # (lf: the image and the aux inputs side by side)
    # blob-like binary labels, ignore_index 0 as in conf/config.yaml:26
    yy, xx = torch.meshgrid(torch.arange(S, device=dev), torch.arange(S, device=dev), indexing="ij")
    ph = torch.rand(B, 3, device=dev, generator=g) * 6.28
    f = (torch.sin(yy[None] * 0.07 + ph[:, 0, None, None]) + torch.cos(xx[None] * 0.05 + ph[:, 1, None, None])
         + torch.sin((xx + yy)[None] * 0.03 + ph[:, 2, None, None]))
    target = (f > 0.3).long()

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    if world > 1:   # communicator set-up + parameter broadcast outside the steps (also with --warmup 0)
        trainer._sync_initial_state(dev)
    for _ in range(args.warmup):
        trainer.step(x, target, 0)
    lib = _lib.load()
    # HIP events around every conv / wgrad launch serialise the kernel boundaries (~2 us per pair: 6.39 -> 6.66 ms per
    # step when every step is timed, same box), so the live roofline measurement samples every EVENT_STRIDE-th step of
    # the timed region; FU_BENCH_EVENT_STRIDE=1 times them all, 0 none
    stride = int(os.environ.get("FU_BENCH_EVENT_STRIDE", "8"))
    sampled = 0
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        on = stride > 0 and i % stride == 0
        if on:
            _lib.check(lib.fu_profile_enable(net._ctx, 1 if sampled == 0 else 2))
            sampled += 1
        loss = trainer.step(x, target, 0)
        if on:
            _lib.check(lib.fu_profile_enable(net._ctx, 0))
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    tiles = args.steps * B * world
    value = tiles / dt
    fwd_fl, train_fl = net.flops_per_tile()

    # ---- roofline of the dominant kernel (HIP events recorded around every launch in the timed region)
    best = None
    for cls in (0, 1):
        n, ms, fl, name = C.c_int64(), C.c_double(), C.c_double(), C.c_char_p()
        _lib.check(lib.fu_profile_read(net._ctx, cls, C.byref(n), C.byref(ms), C.byref(fl), C.byref(name)))
        if n.value and (best is None or ms.value > best["ms"]):
            best = {"kernel": name.value.decode(), "launches": n.value, "ms": ms.value, "flops": fl.value}
    peak = PEAK_TFLOPS[args.dtype]
    roof = None
    # HBM bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs of this same command; counters cannot be read live): profiles/r1_pmc_<dtype>.json
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", f"r1_pmc_{args.dtype}.json")) as fh:
            pm = json.load(fh)
        if best and best["kernel"] in pm and B == 16 and S == 256 and Cc == 8 and args.model == "unet":
            traffic = pm[best["kernel"]]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    # The weight-gradient chain runs on a side stream, concurrently with the dgrad launches of the dominant kernel: their
    # event-timed durations in the timed region include that sharing.  A short extra pass with the side stream off
    # (fu_set_side_stream) gives the same kernel's un-shared rate as `achieved_serial` (not part of `value`).
    serial = None
    if best and world == 1 and not args.no_serial_pass:
        _lib.check(lib.fu_set_side_stream(net._ctx, 0))
        trainer.step(x, target, 0)
        _lib.check(lib.fu_profile_enable(net._ctx, 1))
        for _ in range(3):
            trainer.step(x, target, 0)
        torch.cuda.synchronize(dev)
        _lib.check(lib.fu_profile_enable(net._ctx, 0))
        n, ms, fl, name = C.c_int64(), C.c_double(), C.c_double(), C.c_char_p()
        cls = 0 if best["kernel"].startswith("k_conv3x3") else 1
        _lib.check(lib.fu_profile_read(net._ctx, cls, C.byref(n), C.byref(ms), C.byref(fl), C.byref(name)))
        if n.value:
            serial = fl.value / (ms.value * 1e-3) / 1e12
        _lib.check(lib.fu_set_side_stream(net._ctx, 1))
    if best:
        achieved = best["flops"] / (best["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": best["kernel"], "achieved": round(achieved, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "launches": best["launches"], "event_timed_steps": sampled,
                "avg_launch_ms": round(best["ms"] / best["launches"], 4),
                "avg_launch_gflop": round(best["flops"] / best["launches"] / 1e9, 3),
                "whole_step_frac_of_conv_roofline": round(value / world * train_fl / 1e12 / peak, 4)}
        if serial is not None:
            roof["achieved_serial"] = round(serial, 3)
            roof["frac_serial"] = round(serial / peak, 4)
        if traffic:   # HBM bytes per launch (PMC passes) over the live launch duration, against the 8 TB/s HBM3E peak
            gbps = traffic / (best["ms"] / best["launches"] * 1e-3) / 1e9
            roof["hbm_gbps"] = round(gbps, 1)
            roof["hbm_frac_of_8tbps"] = round(gbps / 8000.0, 4)

    out = {
        "metric": (f"training tiles/sec ({S}x{S}x{Cc}ch UNet)" if args.model == "unet" else
                   f"training tiles/sec ({S}x{S}, {Cc}ch image + {args.aux} aux, late fusion)"),
        "value": round(value, 3), "unit": "tiles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": (f"UNet depth-4 (17.27M params), {Cc}-band {S}x{S} tiles, batch {B}/GPU, "
                                f"fwd+CE(ignore_index=0)+bwd+Adam, train-mode BN" if args.model == "unet" else
                                f"LateFusion ({1 + args.aux} UNet encoders + 1x1 fusion + decoder, "
                                f"{net._total / 1e6:.2f}M params), {Cc}-band image + {args.aux} aux {S}x{S} tiles, "
                                f"batch {B}/GPU, fwd+CE(ignore_index=0)+bwd+Adam, train-mode BN"),
                   "global_batch": B * world, "parallelism": f"dp{world}", "train_gflop_per_tile": round(train_fl / 1e9, 3)},
        "loss": round(float(loss.item()), 6),
        "roofline": roof,
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(Cc, S, args.cpu_batch, args.cpu_steps, args.model, args.aux)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
