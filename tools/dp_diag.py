"""Rehearsal diagnostics of the N>1 code path: 2 gloo ranks sharing one GPU, side-stream mode x reducer variants.
usage: FU_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
       --master-port P tools/dp_diag.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
backend = os.environ.get("FU_DIST_BACKEND", "gloo")
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
from floodplanet_code_amd import _lib
from floodplanet_code_amd.distributed import DataParallelTrainer, BucketedReducer
from floodplanet_code_amd.unet import HipUNet
import floodplanet_code_amd.distributed as D

torch.manual_seed(0)
net = HipUNet(8, 3, bilinear=True, base_channels=64, precision="bf16").to(dev).train()
x = torch.rand(16, 8, 256, 256, device=dev); t = (torch.rand(16, 256, 256, device=dev) > 0.5).long()
lib = _lib.load()

def run(tag, mode, reduce_kind):
    tr = DataParallelTrainer(net, lr=1e-4, world_size=world, rank=rank)
    if world == 1:      # one nccl rank: ProcessGroupNCCL's stream/event/watchdog machinery without a peer
        D._FORCE_BLOCKS = True
        tr._reducer = BucketedReducer(net.block_ranges(), 2, None, tr.cap_bytes)
    D._DIAG_MODE = mode
    orig = BucketedReducer.block_done
    if reduce_kind == "none":
        BucketedReducer.block_done = lambda self, flat, b: None
    elif reduce_kind == "sync":
        def bd(self, flat, b):
            if b in self._by_last:
                off, n = self._by_last[b]
                dist.all_reduce(flat[off:off + n])
        BucketedReducer.block_done = bd
    ts = []
    for i in range(5):
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        tr.step(x, t, 0)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    BucketedReducer.block_done = orig
    if rank == 0:
        print(f"{tag:28s} " + " ".join(f"{v:8.1f}" for v in ts), flush=True)

DataParallelTrainer(net, lr=1e-4).step(x, t, 0) if world == 1 else None   # creates the context
for mode in (0, 1, 2):
    for rk in ("async", "sync", "none"):
        run(f"mode{mode} reduce={rk}", mode, rk)
dist.destroy_process_group()
