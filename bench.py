#!/usr/bin/env python
"""bench.py -- training tiles/sec of the MI355X UNet hot path (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic tiles already resident in HBM:
forward (train-mode BN) + CrossEntropy(ignore_index) + backward + [gradient all-reduce when N>1] + Adam.
Workload at N=1: BASELINE.json configs[1] -- UNet depth-4, 8-band 256x256 tiles, batch 16 per GPU.

`python3 bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (one fresh child
process per GPU, RCCL over `torch.distributed`, rendezvous on 127.0.0.1) before anything touches the GPU, relays rank
0's JSON line and exits non-zero if any rank fails; under `python -m torch.distributed.run` it is simply one rank.

One JSON line on stdout (rank 0).  See DESIGN.md "Measurement" for the definitions of `roofline`
(HIP events around every launch of the dominant kernel, live in the timed region), `cpu_baseline`
(the oracle's restatement of the reference's torch-CPU step, timed on this box's host cores) and `miou_vs_ref`.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PRECISION = {"f32": "fp32", "bf16": "bf16", "f16": "fp16"}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default=os.environ.get("FU_BENCH_DTYPE", "bf16"), choices=["f32", "bf16", "f16"],
                    help="bf16 = BASELINE.json configs[1] (default); f16 = configs[3]'s mixed mode; f32 = the 1e-4 parity mode")
    ap.add_argument("--model", default="unet", choices=["unet", "lf"],
                    help="unet = the headline workload (ms_model / ef_model); lf = the late-fusion net (lf_model.py): "
                         "--channels image bands + --aux one-band auxiliary inputs, one encoder each")
    ap.add_argument("--aux", type=int, default=1, help="--model lf: number of one-band auxiliary inputs (dem, slope, ...)")
    ap.add_argument("--path", default="cabi", choices=["cabi", "plugin", "loader"],
                    help="cabi = DataParallelTrainer over the C ABI (fused fwd+CE, block-wise bwd, fused Adam); plugin = what "
                         "a fit.py user runs: build_model('ms_model') -> configure_optimizers() -> opt.zero_grad(); "
                         "training_step(); loss.backward(); opt.step() (N=1 only); loader = the DATA path alone (not part of "
                         "`value`): tiles/s of TileLoader over a synthetic Sentinel-1 tree whose image rasters are smaller than "
                         "their label rasters (so the Lanczos-4 resample runs), host resample against device resample")
    ap.add_argument("--loader-workers", type=int, default=4, help="--path loader: DataLoader workers (fit.py's n_workers: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-loader", action="store_true", help="skip the data-path measurement (`data_path` object; outside the timed "
                                                                "region, never part of `value`)")
    ap.add_argument("--no-miou", action="store_true", help="skip the small HIP-vs-oracle training comparison (miou_vs_ref)")
    ap.add_argument("--no-eval", action="store_true", help="skip the eval-forward / stitching side measurement")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("FU_STEP_GRAPH", "0")), choices=[0, 1],
                    help="1: the whole step is captured into a hipGraph once and replayed (N = 1, C-ABI path); the steps whose "
                         "conv launches are event-timed for the roofline run eagerly either way; with --graph 1 the eager "
                         "launch mode is measured in a short extra pass and reported as `step_other_mode` (the default line "
                         "no longer replays a graph: DESIGN.md section 5, hipGraph)")
    ap.add_argument("--no-serial-pass", action="store_true",
                    help="skip the extra un-timed pass that measures the dominant kernel without the side stream")
    ap.add_argument("--cpu-batch", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--cpu-steps-b16", type=int, default=3, help="steps of the batch-16 CPU sample (0 = skip)")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous test of the multi-rank launcher on the CPU (gloo): every rank all-reduces its rank")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, started before this process makes any torch.cuda / HIP call
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start n children of this script, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relay rank 0's stdout
    (the JSON line) and every rank's stderr; return non-zero if any rank failed.  The parent never touches the GPU and
    never re-execs itself."""
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
        env.setdefault("OMP_NUM_THREADS", "4")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("FU_BENCH_LAUNCH_TIMEOUT", "1500"))
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
            failed = True                 # one rank is gone (or the job hangs): the others would wait in a collective
            time.sleep(1.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()              # the exact children started above, never a pattern
            break
        time.sleep(0.1)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    for line in b"".join(chunks).decode("utf-8", "replace").splitlines():
        # stdout carries exactly the JSON line(s); anything else a rank printed there (gloo's connection banner) goes to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad or failed:
        sys.stderr.write(f"bench.py launcher: ranks failed (rank, exit code): {bad}\n")
        return 1
    return 0


def launch_check_rank():
    """--launch-check: the rank side of the launcher test.  gloo on the CPU, no GPU, no HIP library."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("FU_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "rank_sum": t.item(),
                          "local_rank": int(os.environ["LOCAL_RANK"]), "master_addr": os.environ["MASTER_ADDR"]}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
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


def _cpu_sample(channels, size, batch_size, steps, model, aux):
    import torch
    from oracle import unet_oracle as O
    if model == "lf":
        in_ch = lf_in_channels(channels, aux)
        st = O.lf_make_state(in_ch, 3, 64, seed=0)
        batch = O.make_batch(batch_size, channels, size, size, seed=1, extra=tuple(AUX_NAMES[:aux]))
        step = lambda: O.lf_train_step(st, opt, batch, in_ch, 0, 1e-4)      # noqa: E731
    else:
        st = O.make_state(channels, 3, 64, True, seed=0, nontrivial_bn=False)
        batch = O.make_batch(batch_size, channels, size, size, seed=1)
        step = lambda: O.train_step(st, opt, batch, 0, 1e-4)                # noqa: E731
    opt = O.new_adam_state(st)
    step()  # warm-up
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2] if len(times) % 2 else 0.5 * (times[len(times) // 2 - 1] + times[len(times) // 2])
    return {"batch": batch_size, "steps": steps, "median_step_s": round(med, 4), "best_step_s": round(times[0], 4),
            "tiles_per_s_median": round(batch_size / med, 4), "tiles_per_s_best": round(batch_size / times[0], 4)}


def cpu_baseline(args):
    """The oracle (kind 'port': torch-CPU restatement of the reference step, pinned bit-exactly against the reference in
    the dev container) on this box's host cores, bounded samples at batch 2 (BASELINE configs[0] shape) and batch 16
    (configs[1] shape), as BASELINE.md section 4 prescribes: warm-up + timed steps, median and best."""
    import torch
    cores = host_cores()
    torch.set_num_threads(cores)
    b2 = _cpu_sample(args.channels, args.size, args.cpu_batch, args.cpu_steps, args.model, args.aux)
    out = {"value": b2["tiles_per_s_median"], "unit": "tiles/s", "cores": cores, "kind": "port",
           "sample": f"median of {args.cpu_steps} steps of batch {args.cpu_batch} after 1 warm-up ({args.channels}ch "
                     f"{args.size}x{args.size}, fp32 torch-CPU {torch.__version__}, {torch.get_num_threads()} threads): "
                     f"median {b2['median_step_s']}s, best {b2['best_step_s']}s per step",
           "best": b2["tiles_per_s_best"], "batch2": b2}
    if args.cpu_steps_b16 > 0:
        out["batch16"] = _cpu_sample(args.channels, args.size, 16, args.cpu_steps_b16, args.model, args.aux)
    return out


def csrc_sha():
    """Hash of the kernel sources the loaded library was built from (profiles/*_pmc_*.json records the same hash)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "floodplanet_code_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(dtype, kernel, standard_workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs of this same command; counters cannot be read live).  Only valid for the binary they were
    taken from: the newest profiles/r*_pmc_<dtype>.json whose `csrc_sha` equals the hash of the sources in this tree."""
    import glob
    import re
    if not standard_workload:
        return None, "traffic: PMC passes exist for the default workload only"
    sha = csrc_sha()
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{dtype}.json")),
                   key=lambda p: int(re.search(r"r(\d+)_pmc", os.path.basename(p)).group(1)), reverse=True)
    for p in cands:
        try:
            pm = json.load(open(p))
        except Exception:
            continue
        if pm.get("csrc_sha") == sha and kernel in pm:
            return pm[kernel]["hbm_bytes_per_launch"], f"{os.path.basename(p)} (csrc_sha {sha})"
    return None, f"traffic: no PMC summary under profiles/ was taken from these kernel sources (csrc_sha {sha})"


MIOU_GOLDEN = os.path.join(ROOT, "tests", "golden", "miou_golden.json")


def miou_vs_ref(dev, dtype, precisions=None):
    """`mIoU vs ref` half of BASELINE.json's metric ("the HIP path vs the reference path trained on identical
    tiles/seeds").  The reference side is tests/golden/miou_golden.json: the oracle (= the reference's arithmetic, pinned bit
    for bit) trained at FULL width on 32 seeded 8-band 128x128 tiles of a task that has to be learnt for 100 Adam steps and
    evaluated, in eval mode, on 16 HELD-OUT tiles (oracle/make_miou_golden.py, ~4 CPU-minutes: run in the build container).
    Here the HIP path trains from the same state on the same tiles (this run's dtype, and fp32) and predicts the same
    held-out tiles; reported: micro Jaccard over argmax with ignore_index (water_seg_model.py:46-63) and the gaps."""
    import torch
    from floodplanet_code_amd.metrics import SegmentationMetrics
    from floodplanet_code_amd.unet import HipUNet
    from oracle import unet_oracle as O
    gold = json.load(open(MIOU_GOLDEN))
    c = gold["config"]
    ii = c["ignore_index"]
    st0 = O.make_state(c["channels"], 3, c["base"], True, seed=c["param_seed"], nontrivial_bn=False)
    train = O.make_task_tiles(c["n_train"], c["channels"], c["size"], c["train_seed"], c["signal"])
    held = O.make_task_tiles(c["n_heldout"], c["channels"], c["size"], c["heldout_seed"], c["signal"])
    xt, tt = train["image"].to(dev), train["target"].to(dev)
    xh, th = held["image"].to(dev), held["target"]
    nb, bs = c["n_train"] // c["batch"], c["batch"]

    def jac(pred, target):
        return SegmentationMetrics(3, ignore_index=ii)(pred.cpu(), target.cpu())["MulticlassJaccardIndex"].item()

    out, losses, track = {}, {}, {}
    names = precisions or list(dict.fromkeys(["fp32", PRECISION[dtype]]))
    for name in names:
        net = HipUNet(c["channels"], 3, base_channels=c["base"], precision=name)
        net.load_state_dict(st0)
        net.to(dev).train()
        last, curve = None, []
        for step in range(c["steps"]):
            k = step % nb
            last = net.train_step(xt[k * bs:(k + 1) * bs], tt[k * bs:(k + 1) * bs], ii)
            curve.append(last)
            net.adam_step(c["lr"] if step < c["lr_switch"] else c["lr_late"], step + 1)
        net.eval()
        with torch.no_grad():
            ph = torch.cat([net(xh[k:k + bs]).argmax(1) for k in range(0, xh.shape[0], bs)])
            pt = torch.cat([net(xt[k:k + bs]).argmax(1) for k in range(0, xt.shape[0], bs)])
        out[f"hip_{name}"] = {"heldout": jac(ph, th), "train": jac(pt, tt)}
        losses[f"hip_{name}"] = float(last.item())
        # the first 40 steps, where two correct implementations still walk the same trajectory: per-step loss ratio to the
        # reference curve (a systematic error in forward, loss, backward, BatchNorm statistics or Adam shows here at once)
        r = [float(v.item()) / g for v, g in zip(curve[:40], gold["loss_curve"][:40])]
        track[f"hip_{name}"] = {"max_abs_dev": round(max(abs(v - 1.0) for v in r), 4),
                                "mean_abs_dev": round(sum(abs(v - 1.0) for v in r) / len(r), 4)}
    ref_h, ref_t = gold["jaccard_heldout"], gold["jaccard_train"]
    return {"workload": f"{c['steps']} Adam steps (lr {c['lr']}, {c['lr_late']} from step {c['lr_switch'] + 1}; batch {bs}) on {c['n_train']} seeded {c['channels']}ch "
                        f"{c['size']}x{c['size']} tiles, full width (base {c['base']}), ignore_index {ii}; micro Jaccard of the "
                        f"eval-mode argmax on {c['n_heldout']} held-out tiles (and on the training tiles)",
            "reference": "tests/golden/miou_golden.json (oracle/make_miou_golden.py: the reference's arithmetic, torch-CPU fp32)",
            "jaccard_heldout": {"oracle_fp32": round(ref_h, 4), **{k: round(v["heldout"], 4) for k, v in out.items()}},
            "jaccard_train": {"oracle_fp32": round(ref_t, 4), **{k: round(v["train"], 4) for k, v in out.items()}},
            "final_train_loss": {"oracle_fp32": round(gold["loss_curve"][-1], 5), **{k: round(v, 5) for k, v in losses.items()}},
            "loss_curve_dev_first_40_steps": track,
            "gap_vs_ref": {k: round(v["heldout"] - ref_h, 4) for k, v in out.items()}}


def eval_forward_bench(net, dev, x, iters=20):
    """Inference beside the training number (predict.py:198-347): eval-mode forward of the bench batch
    (BatchNorm folded into the packed conv weights once, fu_forward(training = 0)) and the same followed by the
    overlap-average stitching of every tile's softmax into a canvas (fu_stitch_add) + the final divide / argmax
    (fu_stitch_finalize).  Not part of `value`."""
    import torch
    from floodplanet_code_amd.stitch import GpuImageStitcher
    B, _, S, _ = x.shape
    net.eval()
    res = {}
    with torch.no_grad():
        for _ in range(3):
            net._forward_raw(x, False, want_logits=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            net._forward_raw(x, False, want_logits=False)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / iters
        res["forward_tiles_per_s"] = round(B / dt, 1)
        res["forward_ms_per_batch"] = round(dt * 1e3, 3)
        # tiles laid out on a 4 x 4 grid with half-tile overlap (stride S/2), as predict.py crops a raster
        side = int(B ** 0.5)
        H = W = S // 2 * (side + 1)
        boxes = [((i // side) * S // 2, (i % side) * S // 2) for i in range(side * side)]
        st = GpuImageStitcher(net, dev)
        def once():
            st.image_canvas.clear()
            st.weight_canvas.clear()
            net._forward_raw(x, False, want_logits=False)
            for i, (h0, w0) in enumerate(boxes):
                st.add_image(i, "r", (h0, w0, h0 + S, w0 + S), H, W)
            return st.combine("r")
        once()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            once()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / iters
        res["forward_plus_stitch_tiles_per_s"] = round(len(boxes) / dt, 1)
        res["forward_plus_stitch_ms_per_batch"] = round(dt * 1e3, 3)
        res["stitch_canvas"] = [H, W]
    net.train()
    return res


class _PluginStepper:
    """The drop-in path as fit.py / Lightning's automatic optimisation drives it (fit.py:95-97):
    opt.zero_grad(); loss = model.training_step(batch, i); loss.backward(); opt.step()."""

    def __init__(self, args, dev, precision):
        from floodplanet_code_amd.models import build_model
        import torch
        torch.manual_seed(0)
        self.model = build_model("ms_model", {"ms_image": args.channels}, 3, 1e-4, log_image_iter=50, to_rgb_fcn=None,
                                 ignore_index=0, optimizer_name="adam", precision=precision).to(dev)
        self.opt = self.model.configure_optimizers()
        self.net = self.model.model
        self.i = 0

    def step(self, x, target, ignore_index):
        batch = {"image": x, "target": target}
        self.opt.zero_grad()
        loss = self.model.training_step(batch, self.i)
        loss.backward()
        self.opt.step()
        self.i += 1
        return loss.detach()


def loader_bench(args, dev, host_budget=20.0, device_epochs=4):
    """The data path of SURVEY 8(f) ranks 1 / 4 measured alone: FloodplanetTiles (TIFF decode, 360 -> 1024 Lanczos-4 resample,
    256 x 256 crops, S1 scaling) -> TileLoader(device_assembly=True, transforms={}) -> batches of 16 augmented tiles in HBM.
    `resize: host` = the per-raster resample in the DataLoader workers (cached per raster; the reference resamples per item,
    st_water_seg/datasets/floodplanet.py:338-340); `resize: device` = window cut-out in the workers + fu_resize_lanczos4_tiles."""
    import tempfile, time, torch
    from floodplanet_code_amd.datasets import FloodplanetTiles, TileLoader, generate_image_slice_object
    from floodplanet_code_amd.datasets.synthetic import make_s1_tree
    root = tempfile.mkdtemp(prefix="fu_loader_")
    n_img = make_s1_tree(root, images_per_region=24, label_size=1024, s1_size=360)
    sp = generate_image_slice_object(args.size, args.size, args.size)
    ds = FloodplanetTiles(root, "train", sp, eval_region=["RegC"], sensor="S1", ignore_index=0)
    out = {"workload": f"synthetic CSDAP tree: {n_img} Sentinel-1 rasters 2 x 360 x 360 f32 -> labels 1024 x 1024 u8, "
                       f"{len(ds)} tiles of {args.size} x {args.size}, batch {args.batch}, shuffle, hflip / vflip / rotate on "
                       "the device", "workers": args.loader_workers, "batch": args.batch, "cores": host_cores()}
    for mode in ("host", "device"):
        ld = TileLoader(ds, args.batch, dev, shuffle=True, seed=0, drop_last=True, num_workers=args.loader_workers,
                        transforms={}, ignore_index=0, device_assembly=True, device_resize=(mode == "device"))
        n, t0, it = 0, None, iter(ld)
        for _ in range(8):                       # warm-up: worker start, page cache, first kernels
            next(it)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        budget = host_budget if mode == "host" else 1e9   # the host path is ~50x slower: a bounded sample of it
        for ep in range(device_epochs):
            for b in it:
                n += b["image"].shape[0]
                if time.perf_counter() - t0 > budget:
                    break
            if time.perf_counter() - t0 > budget:
                break
            it = iter(ld)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        out[f"tiles_timed_resize_{mode}"] = n
        out[f"tiles_per_s_resize_{mode}"] = round(n / dt, 1)
        if mode == "device":
            # which stage caps the device path: ONE collated host batch pushed through the device stage (H2D copies of the
            # windows + tables, Lanczos, normalisation / padding, augmentation) over and over -- what is left of the loader when
            # the DataLoader workers cost nothing
            from floodplanet_code_amd import augment
            import numpy as np
            raw = next(iter(ld._dl))
            rng = np.random.RandomState(0)
            def device_stage():
                o = ld._assemble(raw)
                flags, angles = augment.sample_transforms(o["image"].shape[0], {}, rng)
                return augment.apply(o["image"], o["target"], flags, angles, target_fill=0)
            for _ in range(5):
                device_stage()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            reps = 60
            for _ in range(reps):
                device_stage()
            torch.cuda.synchronize(dev)
            rate = reps * args.batch / (time.perf_counter() - t1)
            out["tiles_per_s_device_stage_alone"] = round(rate, 1)
            # ... and the host pipeline alone: the same DataLoader iterated without the device stage (workers: TIFF decode of
            # the source windows, cut-outs, 8-tap tables, collation; main process: queue hand-off + pinning)
            m, t2 = 0, time.perf_counter()
            for b in ld._dl:
                m += b["target"].shape[0]
            host_rate = m / (time.perf_counter() - t2)
            out["tiles_per_s_host_pipeline_alone"] = round(host_rate, 1)
            out["capped_by"] = (f"the host pipeline (DataLoader with {args.loader_workers} workers: {host_rate:.0f} tiles/s alone, "
                                f"{1e3 * args.batch / host_rate:.1f} ms per batch); the device stage alone sustains {rate:.0f} tiles/s"
                                if rate > 1.15 * host_rate else "the device stage (H2D copies + kernels)")
        del ld
    return out


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if args.launch_check and world_env is not None:
        return launch_check_rank()
    if args.gpus > 1 and world_env is None:
        # BEFORE any torch.cuda / HIP call: this process only starts the ranks and relays their output
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.launch_check:
        raise SystemExit("--launch-check needs --gpus N > 1")

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.path == "plugin" and world > 1:
        raise SystemExit("--path plugin measures the single-device Lightning path (fit.py:87-88: devices=1)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(1, n_dev))  # (ranks share a device only in the gloo rehearsal)
    torch.cuda.set_device(dev)

    if args.path == "loader":
        if world > 1:
            raise SystemExit("--path loader measures one process's data path")
        res = loader_bench(args, dev)
        print(json.dumps({"metric": "data-path tiles/sec (not the training metric)", "unit": "tiles/s", "n_gpus": 1,
                          "data_path": res, "higher_is_better": True}))
        return

    import torch.distributed as dist
    backend = None
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

    precision = PRECISION[args.dtype]
    if os.environ.get("FU_BENCH_GENERAL_CONV") == "1":   # A/B knob: general bf16 conv kernel instead of the fast one
        _lib.load().fu_test_force_general_conv(1)
    torch.manual_seed(0)
    if args.path == "plugin":
        trainer = _PluginStepper(args, dev, precision)
        net = trainer.net
    else:
        if args.model == "lf":
            from floodplanet_code_amd.latefusion import HipLateFusion
            net = HipLateFusion(lf_in_channels(args.channels, args.aux), 3, base_channels=64, precision=precision)
            net = net.to(dev).train()
        else:
            net = HipUNet(args.channels, 3, bilinear=True, base_channels=64, precision=precision).to(dev).train()
        trainer = DataParallelTrainer(net, lr=1e-4, world_size=world, rank=rank, time_waits=world > 1,
                                      graph=bool(args.graph) and world == 1)

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
    lib = _lib.load()
    stride = int(os.environ.get("FU_BENCH_EVENT_STRIDE", "8"))
    for w in range(args.warmup):
        # the LAST warm-up step runs with the launch events on: the first event-timed step creates ~100 events and pays their first
        # records -- 2.2 ms on a 5.3 ms step (FU_BENCH_DUMP_STEPS=1: 7.57 / 5.39 / 5.37 ...), a one-time cost that belongs to the
        # warm-up, not to the K timed steps (it was 3 % of `value` at K = 20 through round 3)
        warm_events = stride > 0 and w == args.warmup - 1 and net._ctx is not None
        if warm_events:
            _lib.check(lib.fu_profile_enable(net._ctx, 1))
            trainer._graph_off = True
        trainer.step(x, target, 0)
        if warm_events:
            _lib.check(lib.fu_profile_enable(net._ctx, 0))
            trainer._graph_off = False
    if net._ctx is None:      # --warmup 0: the context is created by the first forward; the event profiler needs one
        trainer.step(x, target, 0)
    # HIP events around every conv / wgrad launch serialise the kernel boundaries (~2 us per pair: 6.39 -> 6.66 ms per
    # step when every step is timed, same box), so the live roofline measurement samples every EVENT_STRIDE-th step of
    # the timed region; FU_BENCH_EVENT_STRIDE=1 times them all, 0 none
    sampled = 0
    # event-timed steps: every stride-th, starting mid-stride (not the first step behind the synchronisation: its launches meet an
    # empty queue); a run shorter than that samples its last step
    sample_steps = set(i for i in range(args.steps) if stride > 0 and i % stride == stride // 2)
    if stride > 0 and not sample_steps and args.steps > 0:
        sample_steps = {args.steps - 1}
    if world > 1 and getattr(trainer, "_reducer", None) is not None:
        trainer._reducer.reset_timing()     # the exposed all-reduce waits of the TIMED steps only
    # one event per step boundary on the compute stream (a record is ~1 us of host time and serialises nothing): the
    # per-step durations behind `ms_per_step_median`; `value` stays the wall clock around all K steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync_all()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        on = i in sample_steps
        if on:
            _lib.check(lib.fu_profile_enable(net._ctx, 1 if sampled == 0 else 2))
            sampled += 1
            trainer._graph_off = True          # event-timed steps are launched eagerly (a replay records no events)
        loss = trainer.step(x, target, 0)
        if on:
            _lib.check(lib.fu_profile_enable(net._ctx, 0))
            trainer._graph_off = False
        marks[i + 1].record()
    sync_all()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if os.environ.get("FU_BENCH_DUMP_STEPS"):      # diagnostics: the per-step durations in launch order
        print("step_ms " + " ".join(f"{v:.3f}" for v in step_ms), file=sys.stderr)
    step_ms = sorted(step_ms)
    step_ms_median = (step_ms[len(step_ms) // 2] if len(step_ms) % 2 else
                      0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])) if step_ms else None
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
    traffic, traffic_src = (None, None)
    if best:
        traffic, traffic_src = pmc_traffic(args.dtype, best["kernel"],
                                           B == 16 and S == 256 and Cc == 8 and args.model == "unet")
    # The weight-gradient chain runs on a side stream, concurrently with the dgrad launches of the dominant kernel: their
    # event-timed durations in the timed region include that sharing.  A short extra pass with the side stream off
    # (fu_set_side_stream) gives the same kernel's un-shared rate as `achieved_serial` (not part of `value`).
    serial = None
    serial_conv_only = None
    if best and world == 1 and not args.no_serial_pass and args.path == "cabi":
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
        # In the 16-bit modes 9 of the 35 conv launches of a step (the dgrads into a block's first BatchNorm) also compute
        # that BatchNorm's backward sums in their epilogue -- non-conv work inside the dominant kernel's
        # time.  Same pass once more with those sums back in their own kernel: the conv kernels' rate for conv work only.
        if args.dtype != "f32" and cls == 0:
            lib.fu_test_bnb_separate(1)
            try:
                trainer.step(x, target, 0)
                _lib.check(lib.fu_profile_enable(net._ctx, 1))
                for _ in range(3):
                    trainer.step(x, target, 0)
                torch.cuda.synchronize(dev)
                _lib.check(lib.fu_profile_enable(net._ctx, 0))
                _lib.check(lib.fu_profile_read(net._ctx, cls, C.byref(n), C.byref(ms), C.byref(fl), C.byref(name)))
                if n.value:
                    serial_conv_only = fl.value / (ms.value * 1e-3) / 1e12
            finally:
                lib.fu_test_bnb_separate(0)
        _lib.check(lib.fu_set_side_stream(net._ctx, 1))
    other_mode = None
    if world == 1 and args.path == "cabi" and args.model == "unet" and not args.no_serial_pass and args.graph:
        # the step launched the other way (captured hipGraph -> eager), 20 steps after 3 of warm-up; not part of `value`.
        # Only with --graph 1: round 4 measured the replay of the ONE-chain capture 2 % slower than its eager launches (5.81
        # against 5.67 ms) and the replay of the two-stream capture 60 % slower (8.70 against 5.39) -- a runtime property, not
        # a mode worth a place in every line.
        was = trainer.graph
        trainer.graph = not was
        try:
            for _ in range(3):
                trainer.step(x, target, 0)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(20):
                trainer.step(x, target, 0)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t1) / 20 * 1e3
            other_mode = {"mode": "hipGraph replay" if trainer.graph else "eager launches", "ms_per_step": round(ms, 3),
                          "tiles_per_s": round(B / ms * 1e3, 1)}
        finally:
            trainer.graph = was
    if best:
        achieved = best["flops"] / (best["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": best["kernel"], "achieved": round(achieved, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches": best["launches"], "event_timed_steps": sampled,
                "avg_launch_ms": round(best["ms"] / best["launches"], 4),
                "avg_launch_gflop": round(best["flops"] / best["launches"] / 1e9, 3),
                "whole_step_frac_of_conv_roofline": round(value / world * train_fl / 1e12 / peak, 4)}
        if serial is not None:
            roof["achieved_serial"] = round(serial, 3)
            roof["frac_serial"] = round(serial / peak, 4)
        if serial_conv_only is not None:
            roof["achieved_serial_conv_only"] = round(serial_conv_only, 3)
            roof["frac_serial_conv_only"] = round(serial_conv_only / peak, 4)
            roof["fused_work"] = ("9 of the 35 launches per step (dgrad into a block's first BatchNorm) also emit that "
                                  "BatchNorm's backward sums from their epilogue; *_conv_only = the same pass with those sums "
                                  "in their own kernel (fu_test_bnb_separate)")
        if traffic:   # HBM bytes per launch (PMC passes) over the live launch duration, against the 8 TB/s HBM3E peak
            gbps = traffic / (best["ms"] / best["launches"] * 1e-3) / 1e9
            roof["hbm_gbps"] = round(gbps, 1)
            roof["hbm_frac_of_8tbps"] = round(gbps / 8000.0, 4)

    wl = (f"UNet depth-4 (17.27M params), {Cc}-band {S}x{S} tiles, batch {B}/GPU, fwd+CE(ignore_index=0)+bwd+Adam, "
          f"train-mode BN" if args.model == "unet" else
          f"LateFusion ({1 + args.aux} UNet encoders + 1x1 fusion + decoder, {net._total / 1e6:.2f}M params), {Cc}-band "
          f"image + {args.aux} aux {S}x{S} tiles, batch {B}/GPU, fwd+CE(ignore_index=0)+bwd+Adam, train-mode BN")
    cfg = {"workload": wl, "global_batch": B * world, "parallelism": f"dp{world}",
           "train_gflop_per_tile": round(train_fl / 1e9, 3),
           "path": ("C ABI: DataParallelTrainer.step" if args.path == "cabi" else
                    "plugin: build_model('ms_model').training_step + loss.backward() + configure_optimizers().step()")}
    # every FU_* variable of this process's environment: several of them switch the dispatch of the library that is being timed
    # (FU_CONV_PP, FU_BNB_SEPARATE, FU_WGRAD_MODE, FU_NO_SIDE_STREAM, FU_LIB_PATH ...), and a line that does not name them cannot
    # be compared with another
    cfg["env_FU"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("FU_")}
    if world > 1:
        # what actually ran, for whoever reads an N > 1 record later: the world torch.distributed reports after
        # init_process_group (not the flag), the transport, the bucket plan and how long the compute stream stood still
        # for all-reduces that backward did not cover
        cfg["backend"] = backend
        cfg["world_size_seen"] = dist.get_world_size()
        cfg["devices_visible"] = n_dev
        cfg["backward_chain"] = {0: "serial (one stream)", 1: "side stream, join per block",
                                 2: ("side stream; per all-reduce bucket the stream the collective is launched from waits "
                                     "for both backward chains (fu_backward_fence), the compute stream joins once, after the "
                                     "last block" if os.environ.get("FU_DP_JOIN_AT_BUCKETS") != "1" else
                                     "side stream, compute stream joined per all-reduce bucket")}[trainer._side_mode()]
        if os.environ.get("FU_DP_SIDE_MODE") is not None:
            cfg["backward_chain_forced_by_env"] = "FU_DP_SIDE_MODE=" + os.environ["FU_DP_SIDE_MODE"]
        red = getattr(trainer, "_reducer", None)
        if red is not None:
            cfg["allreduce_bucket_bytes"] = red.bucket_bytes()
            w = red.exposed_wait_ms()
            if w is not None:
                stat = torch.tensor([w["device_mean"], w["device_max"], w["host_mean"]], device=dev, dtype=torch.float64)
                dist.all_reduce(stat, op=dist.ReduceOp.MAX)
                cfg["allreduce_exposed_wait_ms"] = {
                    "rank0": {k: round(v, 4) if isinstance(v, float) else v for k, v in w.items()},
                    "max_over_ranks": {"device_mean": round(stat[0].item(), 4), "device_max": round(stat[1].item(), 4),
                                       "host_mean": round(stat[2].item(), 4)},
                    "meaning": "per step: HIP events on the compute stream around BucketedReducer.finish() (the stream "
                               "waits there for the buckets' all-reduces: what backward did not overlap), and the host time "
                               "in the same call"}
    out = {
        "metric": (f"training tiles/sec ({S}x{S}x{Cc}ch UNet)" if args.model == "unet" else
                   f"training tiles/sec ({S}x{S}, {Cc}ch image + {args.aux} aux, late fusion)"),
        "value": round(value, 3), "unit": "tiles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "ms_per_step_median": round(step_ms_median, 3) if step_ms_median is not None else None,
        "ms_per_step_min_max": [round(step_ms[0], 3), round(step_ms[-1], 3)] if step_ms else None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": cfg,
        "step_launch_mode": "hipGraph replay" if (args.graph and world == 1 and args.path == "cabi") else "eager launches",
        "step_other_mode": other_mode,
        "loss": round(float(loss.item()), 6),
        "roofline": roof,
    }
    if rank == 0:
        if world == 1 and args.model == "unet" and args.path == "cabi" and not args.no_eval:
            out["eval"] = eval_forward_bench(net, dev, x)
        if world == 1 and not args.no_miou:
            out["miou_vs_ref"] = miou_vs_ref(dev, args.dtype)
        if world == 1 and not args.no_loader and args.path == "cabi" and args.model == "unet":
            # the data path beside the train number ("at >= 1,000 tiles/s/GPU the CPU DataLoader starves
            # the GPU"): tiles/s of TileLoader with the Lanczos-4 resample in the workers / on the device; `--path loader` runs
            # the longer sample
            out["data_path"] = loader_bench(args, dev, host_budget=6.0, device_epochs=2)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
