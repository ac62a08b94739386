import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
from test_gpu_ops import make_conv_case, nhwc, nchw, stream, CONV_SHAPES, DEV, F32
lib = _lib.load()
for shape in CONV_SHAPES:
    B, C0, C1, Cout, H, W, bn = shape
    x0, x1, a, b, w, bias, xin = make_conv_case(*shape)
    ref = F.conv2d(xin, w, bias, padding=1)
    y = torch.empty(B, H, W, Cout, device=DEV)
    check(lib.fu_op_conv3x3_fwd(F32, ptr(nhwc(x0)), C0, ptr(a.to(DEV)) if bn else None, ptr(b.to(DEV)) if bn else None,
          ptr(nhwc(x1)) if x1 is not None else None, C1, ptr(w.to(DEV)), ptr(bias.to(DEV)), ptr(y), Cout, B, H, W, None, None, stream()))
    torch.cuda.synchronize()
    out = nchw(y)
    err = (out - ref).abs()
    bad = err > 1e-4
    print(shape, "max err", err.max().item(), "bad frac", bad.float().mean().item())
    if bad.any():
        # which rows / cols / channels / batch are bad
        print("  bad by batch", bad.float().mean((1,2,3)).tolist())
        print("  bad by channel", [round(v,2) for v in bad.float().mean((0,2,3)).tolist()][:40])
        print("  bad by row", [round(v,2) for v in bad.float().mean((0,1,3)).tolist()])
        print("  bad by col", [round(v,2) for v in bad.float().mean((0,1,2)).tolist()])
