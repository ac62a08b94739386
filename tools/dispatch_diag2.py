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
g = torch.Generator(device=DEV).manual_seed(3)
dl = torch.randn(16, 3, 256, 256, device=DEV, generator=g) * 1e-6
def rel(a, b): return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
res = {}
for mode in (0, 1, 0):
    lib.fu_test_conv_tile_mode(mode)
    logits = net._forward_raw(x, True, want_logits=True)
    net._backward_raw(dl, x.device)
    torch.cuda.synchronize()
    res.setdefault(mode, []).append((logits.clone(), net.flat_grads().clone()))
lib.fu_test_conv_tile_mode(0)
(l0, g0), (l0b, g0b) = res[0]
(l1, g1), = res[1]
print("mode0 repeat identical:", torch.equal(l0, l0b), torch.equal(g0, g0b))
d = (l0 - l1)
print("logits: frac differing", (d != 0).float().mean().item(), "max", d.abs().max().item(), "rel", rel(l1, l0))
for (k, p, off, n) in net._table:
    a, b = g1[off:off+n], g0[off:off+n]
    if b.norm() > 1e-12 and (k.endswith("weight") and p.dim() == 4):
        print(f"   {k:45s} rel {rel(a,b):.5f}  norm {b.norm().item():.3e}")
