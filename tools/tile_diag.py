"""Do the square and the tall bf16 conv tiles produce the same output on one big layer? (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
DEV = "cuda:0"
lib = _lib.load()
BF16 = _lib.FU_BF16
g = torch.Generator().manual_seed(0)
for (B, C0, Cout, H, W, bn) in [(16, 64, 64, 256, 256, True), (16, 64, 64, 256, 256, False), (4, 64, 64, 64, 64, True), (16, 8, 64, 256, 256, False)]:
    x = torch.randn(B, H, W, C0, generator=g).to(DEV).to(torch.bfloat16)
    a = (torch.rand(C0, generator=g) + 0.5).to(DEV) if bn else None
    b = (torch.randn(C0, generator=g) * 0.3).to(DEV) if bn else None
    w = (torch.randn(Cout, C0, 3, 3, generator=g) / (3.0 * C0 ** 0.5)).to(DEV)
    bias = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    outs = []
    for mode in (1, 2):
        lib.fu_test_conv_tile_mode(mode)
        y = torch.empty(B, H, W, Cout, device=DEV, dtype=torch.bfloat16)
        ssum = torch.empty(Cout, device=DEV); ssq = torch.empty(Cout, device=DEV)
        check(lib.fu_op_conv3x3_fwd(BF16, ptr(x), C0, ptr(a), ptr(b), None, 0, ptr(w), ptr(bias), ptr(y), Cout, B, H, W,
                                    ptr(ssum), ptr(ssq), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        outs.append((y.float(), ssum.clone(), ssq.clone()))
    lib.fu_test_conv_tile_mode(0)
    (y1, s1, q1), (y2, s2, q2) = outs
    d = (y1 - y2)
    print((B, C0, Cout, H, W, bn), "differing elems", (d != 0).float().mean().item(), "max", d.abs().max().item(),
          "rel", (d.norm() / y1.norm()).item(), "stats rel", ((s1 - s2).norm() / s1.norm()).item(), ((q1 - q2).norm() / q1.norm()).item())
