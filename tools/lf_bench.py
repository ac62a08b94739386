"""Training step time of the late-fusion network (lf_model.py) at the bench shape: 8-band image + 1-band DEM, 256x256,
batch 16, bf16; beside the plain 9-band UNet (= the early-fusion model on the same inputs).
usage: python tools/lf_bench.py [steps]"""
import sys, time
sys.path.insert(0, '.')
import torch
from floodplanet_code_amd.distributed import DataParallelTrainer
from floodplanet_code_amd.latefusion import HipLateFusion
from floodplanet_code_amd.unet import HipUNet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(16, 9, 256, 256, device=dev)
t = (torch.rand(16, 256, 256, device=dev) > 0.6).long()
for name, net in (("ef  UNet(9 ch)", HipUNet(9, 3, precision="bf16")),
                  ("lf  image 8 + dem 1", HipLateFusion({"ms_image": 8, "dem": 1}, 3, precision="bf16"))):
    net = net.to(dev).train()
    tr = DataParallelTrainer(net, lr=1e-4)
    for _ in range(3):
        tr.step(x, t, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(x, t, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fwd, train = net.flops_per_tile()
    print(f"{name:22s} {dt * 1e3:7.2f} ms/step  {16 / dt:8.1f} tiles/s  loss {loss.item():.4f}  "
          f"algorithmic {train / 1e9:.1f} GFLOP/tile -> {16 * train / dt / 1e12:.0f} TFLOP/s", flush=True)
    del tr, net
