"""Per-tensor gradient differences between dispatch variants of the bf16 step at the bench shape (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from floodplanet_code_amd import _lib
from floodplanet_code_amd.unet import HipUNet
from oracle import unet_oracle as O

DEV = "cuda:0"
torch.manual_seed(0)
net = HipUNet(8, 3, precision="bf16").to(DEV).train()
batch = O.make_batch(16, 8, 256, 256, seed=11)
x, t = batch["image"].to(DEV), batch["target"].to(DEV)
lib = _lib.load()

def run(tile, lock, side):
    lib.fu_test_conv_tile_mode(tile); lib.fu_test_force_lockstep_wgrad(lock)
    if net._ctx is not None:
        _lib.check(lib.fu_set_side_stream(net._ctx, side))
    l = net.train_step(x, t, 0).item()
    g = net.flat_grads().clone()
    torch.cuda.synchronize()
    lib.fu_test_conv_tile_mode(0); lib.fu_test_force_lockstep_wgrad(0)
    return l, g

l0, g0 = run(0, 0, 1)
def rel(a, b): return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
for name, cfg in [("side off", (0, 0, 0)), ("lockstep", (0, 1, 1)), ("square tiles", (1, 0, 1)), ("all tall", (2, 0, 1))]:
    l, g = run(*cfg)
    print(name, "loss", l0, l, "bit-identical" if torch.equal(g, g0) else "differs")
    if not torch.equal(g, g0):
        for (k, p, off, n) in net._table:
            a, b = g[off:off+n], g0[off:off+n]
            if b.norm() > 1e-7 and (k.endswith("weight") and p.dim() == 4):
                print(f"   {k:45s} rel {rel(a,b):.4f}  norm {b.norm().item():.3e}")
