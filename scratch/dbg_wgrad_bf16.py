import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
from test_gpu_ops import nhwc_bf, bf, stream, DEV
lib = _lib.load()
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
def run(B, C0, Cout, H, W, xfun, dyfun):
    x = xfun(); dy = dyfun()
    ref = torch.nn.grad.conv2d_weight(bf(x), (Cout, C0, 3, 3), bf(dy), padding=1)
    dw = torch.full((Cout, C0, 3, 3), float('nan'), device=DEV)
    d0, ddy = nhwc_bf(x), nhwc_bf(dy)
    check(lib.fu_op_conv3x3_wgrad(_lib.FU_BF16, ptr(d0), C0, None, None, None, 0, ptr(ddy), Cout, ptr(dw), B, H, W, stream()))
    torch.cuda.synchronize()
    return dw.cpu(), ref
# case 1: x = delta at (pixel (1,5), channel 2), dy = delta at (pixel (1,5), channel 3) -> only center tap (4) of dw[3][2] = 1
B, C0, Cout, H, W = 1, 8, 8, 4, 16
for (py, px) in [(1, 5), (0, 0), (2, 9), (3, 15), (1, 12)]:
    def xf():
        x = torch.zeros(B, C0, H, W); x[0, 2, py, px] = 1.0; return x
    def dyf():
        d = torch.zeros(B, Cout, H, W); d[0, 3, py, px] = 1.0; return d
    got, ref = run(B, C0, Cout, H, W, xf, dyf)
    nz = got.nonzero().tolist()
    print("pixel", (py, px), "ref nz", ref.nonzero().tolist(), "got nz", nz[:10], "vals", [got[tuple(i)].item() for i in nz[:10]])
# case 2: random, single tile
g = torch.Generator().manual_seed(0)
got, ref = run(1, 8, 8, 4, 16, lambda: torch.randn(1, 8, 4, 16, generator=g), lambda: torch.randn(1, 8, 4, 16, generator=g))
print("single tile random: max err", (got - ref).abs().max().item(), "ref max", ref.abs().max().item())
print((got - ref)[0, 0], ref[0, 0])
