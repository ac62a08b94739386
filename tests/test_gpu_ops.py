"""GPU: single HIP operators (through the C ABI's fu_op_* entry points) against torch-CPU fp32 references of
the same op as used in unet.py (F.conv2d / batch-norm+relu prologue / max_pool2d / bilinear upsample)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32 = _lib.FU_F32


def nhwc(x):  # [B,C,H,W] cpu -> device NHWC
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(x):  # device NHWC -> cpu [B,C,H,W]
    return x.permute(0, 3, 1, 2).contiguous().cpu()


def stream():
    return torch.cuda.current_stream().cuda_stream


def rel_err(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


CONV_SHAPES = [
    # B, C0, C1, Cout, H, W, bn_prologue
    (2, 8, 0, 24, 20, 37, False),
    (1, 64, 64, 64, 32, 32, True),
    (2, 4, 0, 4, 16, 16, True),
    (1, 16, 8, 40, 9, 11, True),
    (2, 128, 0, 72, 18, 18, False),
    (3, 32, 32, 128, 64, 64, True),
]


def make_conv_case(B, C0, C1, Cout, H, W, bn, seed=0):
    g = torch.Generator().manual_seed(seed)
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    a = (torch.rand(C0, generator=g) + 0.5) if bn else None
    b = (torch.randn(C0, generator=g) * 0.3) if bn else None
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / (3.0 * (C0 + C1) ** 0.5)
    bias = torch.randn(Cout, generator=g) * 0.1
    xin = torch.relu(x0 * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)) if bn else x0
    if x1 is not None:
        xin = torch.cat([xin, x1], 1)
    return x0, x1, a, b, w, bias, xin


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_forward_and_stats(shape):
    B, C0, C1, Cout, H, W, bn = shape
    lib = _lib.load()
    x0, x1, a, b, w, bias, xin = make_conv_case(*shape)
    ref = F.conv2d(xin, w, bias, padding=1)
    d0 = nhwc(x0)
    d1 = nhwc(x1) if x1 is not None else None
    da = a.to(DEV) if bn else None
    db = b.to(DEV) if bn else None
    y = torch.empty(B, H, W, Cout, device=DEV)
    ssum = torch.empty(Cout, device=DEV)
    ssq = torch.empty(Cout, device=DEV)
    dw_, dbias = w.to(DEV), bias.to(DEV)   # keep every device tensor alive across the call
    check(lib.fu_op_conv3x3_fwd(F32, ptr(d0), C0, ptr(da), ptr(db), ptr(d1), C1, ptr(dw_), ptr(dbias),
                                ptr(y), Cout, B, H, W, ptr(ssum), ptr(ssq), stream()))
    torch.cuda.synchronize()
    out = nchw(y)
    assert (out - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    nb = ref - bias.view(1, -1, 1, 1)
    assert rel_err(ssum.cpu(), nb.sum((0, 2, 3))) < 1e-4 or (ssum.cpu() - nb.sum((0, 2, 3))).abs().max() < 1e-3
    assert rel_err(ssq.cpu(), (nb * nb).sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_dgrad(shape):
    B, C0, C1, Cout, H, W, _ = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 3.0
    ref = torch.nn.grad.conv2d_input((B, C0 + C1, H, W), w, dy, padding=1)
    dx0 = torch.full((B, H, W, C0), float("nan"), device=DEV)
    dx1 = torch.full((B, H, W, C1), float("nan"), device=DEV) if C1 else None
    ddy, dw_ = nhwc(dy), w.to(DEV)
    check(lib.fu_op_conv3x3_dgrad(F32, ptr(ddy), Cout, ptr(dw_), ptr(dx0), C0, ptr(dx1), C1, B, H, W, stream()))
    torch.cuda.synchronize()
    got = nchw(dx0) if dx1 is None else torch.cat([nchw(dx0), nchw(dx1)], 1)
    assert rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_wgrad(shape):
    B, C0, C1, Cout, H, W, bn = shape
    lib = _lib.load()
    x0, x1, a, b, w, bias, xin = make_conv_case(*shape, seed=2)
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(B, Cout, H, W, generator=g)
    ref = torch.nn.grad.conv2d_weight(xin, w.shape, dy, padding=1)
    dw = torch.full(w.shape, float("nan"), device=DEV)
    d0, d1, ddy = nhwc(x0), (nhwc(x1) if x1 is not None else None), nhwc(dy)
    da, db = (a.to(DEV), b.to(DEV)) if bn else (None, None)
    check(lib.fu_op_conv3x3_wgrad(F32, ptr(d0), C0, ptr(da), ptr(db), ptr(d1), C1, ptr(ddy), Cout, ptr(dw), B, H, W,
                                  stream()))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), ref) < 1e-5


@pytest.mark.parametrize("B,C,H,W,bn", [(2, 8, 16, 16, True), (1, 64, 37, 45, True), (2, 4, 9, 8, False)])
def test_maxpool(B, C, H, W, bn):
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, C, H, W, generator=g)
    a = torch.rand(C, generator=g) + 0.5
    b = torch.randn(C, generator=g) * 0.2
    z = torch.relu(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)) if bn else x
    ref = F.max_pool2d(z, 2)
    out = torch.empty(B, H // 2, W // 2, C, device=DEV)
    dx = nhwc(x)
    da, db = (a.to(DEV), b.to(DEV)) if bn else (None, None)
    check(lib.fu_op_maxpool2(F32, ptr(dx), ptr(da), ptr(db), ptr(out), B, H, W, C, stream()))
    torch.cuda.synchronize()
    assert (nchw(out) - ref).abs().max().item() < 1e-6


@pytest.mark.parametrize("B,C,H,W,oh,ow", [(2, 8, 16, 16, 32, 32), (1, 16, 4, 5, 9, 11), (2, 4, 1, 2, 2, 5),
                                           (1, 64, 18, 18, 37, 37)])
def test_upsample_bilinear_align_corners_with_pad(B, C, H, W, oh, ow):
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g)
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    dy, dx = oh - up.shape[2], ow - up.shape[3]
    ref = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    out = torch.full((B, oh, ow, C), float("nan"), device=DEV)
    dxx = nhwc(x)
    check(lib.fu_op_upsample2(F32, ptr(dxx), None, None, ptr(out), B, H, W, C, oh, ow, stream()))
    torch.cuda.synchronize()
    # fp32 interpolation weights and two lerps, evaluated in another association than ATen's: a few ulps of the largest
    # value involved.  Measured over these shapes: max 2.1e-6 on |x| <= 4.6 (= 4.4 ulp of 4), mean 6e-8; the bound is 8 ulp
    # of the tensor's range (the earlier absolute 2e-6 sat on top of the measured maximum), the bulk is bounded tightly.
    err = (nchw(out) - ref).abs()
    assert err.max().item() <= 8 * 2.0 ** -23 * max(1.0, ref.abs().max().item())
    assert err.mean().item() <= 2e-7


def test_layout_roundtrip():
    lib = _lib.load()
    x = torch.randn(2, 9, 13, 17)
    d = torch.empty(2, 13, 17, 12, device=DEV)
    xd = x.to(DEV)
    check(lib.fu_op_nchw_to_nhwc(F32, ptr(xd), ptr(d), 2, 9, 13, 17, 12, stream()))
    back = torch.empty(2, 9, 13, 17, device=DEV)
    check(lib.fu_op_nhwc_to_nchw(F32, ptr(d), ptr(back), 2, 9, 13, 17, 12, stream()))
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), x)
    assert float(d[..., 9:].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------
# bf16 kernels: operands are rounded to bf16 first, the fp32 torch result of those rounded operands is the
# reference; the kernel's fp32 accumulator is rounded once to bf16 on store (relative error <= 2^-8).
# ---------------------------------------------------------------------------------------------------
BF16 = _lib.FU_BF16
BF_SHAPES = [
    # B, C0, C1, Cout, H, W, bn      (tile configuration hit)
    (1, 64, 64, 64, 32, 32, True),   # 128x64 tiles
    (2, 16, 8, 40, 9, 11, True),     # ragged, partial channel tile
    (2, 8, 0, 64, 224, 224, False),  # 256x64 tiles, single ragged K chunk
    (2, 32, 0, 128, 192, 192, True), # 256x128 tiles
    (3, 32, 32, 136, 50, 38, True),  # odd sizes
    (2, 64, 32, 96, 48, 80, True),   # interior + border tiles, two sources (switch after 2 chunks), 64 + 32 channel tiles
    (1, 96, 0, 32, 16, 16, False),   # one tile, narrow (32-channel) workgroups
    (2, 128, 64, 64, 40, 24, True),  # second destination at a 64-channel boundary (dgrad: D0 = 128)
    (4, 32, 0, 64, 208, 200, True),  # >= 512 64-channel tiles with ragged rows AND columns: the tall 16x32 tile's masked epilogue
    # shapes the row-stationary kernel takes (Cin % 32 == 0, N % 64 == 0, H % 32 == 0, W % 16 == 0): border + interior tiles,
    # two sources, two destinations (dgrad), several channel tiles, one and many K chunks
    (2, 64, 0, 64, 64, 48, True),
    (1, 64, 64, 128, 96, 32, True),
    # ... and the persistent ping-pong kernel (the same + H % 32 == 0, >= 8 tiles): odd step counts (3 chunks), more tiles
    # than workgroups on a small grid, two sources switching inside a tile, a second destination
    (3, 96, 0, 64, 64, 32, False),
    (4, 64, 64, 128, 64, 64, True),
    (2, 128, 64, 64, 32, 16, True),
    (1, 256, 0, 192, 32, 32, False),
    (3, 32, 0, 64, 160, 80, True),
    # 8 input channels, no BN prologue, N % 64 == 0, H, W % 16 == 0: the first-conv kernel (k_conv3x3_*_c8) under "auto":
    # 16 x 64 tiles (H % 64 == 0), 16 x 16 tiles, two channel tiles, the bench's tile size
    (2, 8, 0, 64, 64, 48, False),
    (1, 8, 0, 128, 48, 32, False),
    (4, 8, 0, 64, 256, 256, False),
]


# the 16-bit kernels exist for two element types (same sources, fu_conv_bf16.h): every test below runs on both.
# eps = the element type's rounding unit (bf16: 8 significant bits, fp16: 11)
LOWP = {"bf16": dict(code=_lib.FU_BF16, dt=torch.bfloat16, eps=2.0 ** -8),
        "fp16": dict(code=_lib.FU_F16, dt=torch.float16, eps=2.0 ** -11)}
_cur = dict(LOWP["bf16"])


@pytest.fixture(params=["bf16", "fp16"])
def lowp(request):
    _cur.update(LOWP[request.param])
    yield request.param
    _cur.update(LOWP["bf16"])


def bf(x):
    return x.to(_cur["dt"]).float()


def nhwc_bf(x):
    return x.permute(0, 2, 3, 1).contiguous().to(_cur["dt"]).to(DEV)


def nchw_bf(x):
    return x.float().permute(0, 3, 1, 2).contiguous().cpu()


@pytest.fixture(params=["auto", "general", "tall", "square", "rs", "pp"])
def conv_path(request):
    """bf16 forward/dgrad have an aligned-shape fast kernel (16x16 and tall 16x32 workgroup tiles), a general one, the
    row-stationary kernel ("rs") and the persistent ping-pong kernel ("pp": forced wherever the shape is eligible, i.e. from 8
    tiles of 16 x 32 pixels x 64 channels on -- the default dispatch asks for 256): run every shape on all of them."""
    lib = _lib.load()
    lib.fu_test_force_general_conv(1 if request.param == "general" else 0)
    lib.fu_test_conv_tile_mode({"tall": 2, "square": 1, "rs": 3, "pp": 4}.get(request.param, 0))
    yield request.param
    lib.fu_test_force_general_conv(0)
    lib.fu_test_conv_tile_mode(0)


@pytest.mark.parametrize("shape", BF_SHAPES)
def test_conv3x3_bf16_forward_and_stats(shape, conv_path, lowp):
    B, C0, C1, Cout, H, W, bn = shape
    lib = _lib.load()
    x0, x1, a, b, w, bias, _ = make_conv_case(*shape)
    x0r = bf(x0)
    xin = bf(torch.relu(x0r * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))) if bn else x0r
    if x1 is not None:
        xin = torch.cat([xin, bf(x1)], 1)
    ref = F.conv2d(xin, bf(w), None, padding=1)
    d0, d1 = nhwc_bf(x0), (nhwc_bf(x1) if x1 is not None else None)
    da, db = (a.to(DEV), b.to(DEV)) if bn else (None, None)
    dw_, dbias = w.to(DEV), bias.to(DEV)
    y = torch.empty(B, H, W, Cout, device=DEV, dtype=_cur["dt"])
    ssum = torch.empty(Cout, device=DEV)
    ssq = torch.empty(Cout, device=DEV)
    check(lib.fu_op_conv3x3_fwd(_cur["code"], ptr(d0), C0, ptr(da), ptr(db), ptr(d1), C1, ptr(dw_), ptr(dbias), ptr(y), Cout,
                                B, H, W, ptr(ssum), ptr(ssq), stream()))
    torch.cuda.synchronize()
    out = nchw_bf(y)
    full = ref + bias.view(1, -1, 1, 1)
    assert (out - full).abs().max().item() <= 2 * _cur["eps"] * max(1.0, full.abs().max().item())
    assert rel_err(out, full) < _cur["eps"]
    assert rel_err(ssq.cpu(), (ref * ref).sum((0, 2, 3))) < 1e-4
    assert (ssum.cpu() - ref.sum((0, 2, 3))).abs().max() < 1e-3 * (ref.abs().sum((0, 2, 3)).max() + 1)


@pytest.mark.parametrize("shape", BF_SHAPES)
def test_conv3x3_bf16_dgrad(shape, conv_path, lowp):
    B, C0, C1, Cout, H, W, _ = shape
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 3.0
    ref = torch.nn.grad.conv2d_input((B, C0 + C1, H, W), bf(w), bf(dy), padding=1)
    dx0 = torch.full((B, H, W, C0), float("nan"), device=DEV, dtype=_cur["dt"])
    dx1 = torch.full((B, H, W, C1), float("nan"), device=DEV, dtype=_cur["dt"]) if C1 else None
    ddy, dw_ = nhwc_bf(dy), w.to(DEV)
    check(lib.fu_op_conv3x3_dgrad(_cur["code"], ptr(ddy), Cout, ptr(dw_), ptr(dx0), C0, ptr(dx1), C1, B, H, W, stream()))
    torch.cuda.synchronize()
    got = nchw_bf(dx0) if dx1 is None else torch.cat([nchw_bf(dx0), nchw_bf(dx1)], 1)
    assert rel_err(got, ref) < _cur["eps"]


@pytest.fixture(params=["auto", "lockstep"])
def wgrad_path(request):
    """bf16 wgrad of c_in > 64 has a ping-pong kernel and a lock-step one: run every shape on both."""
    lib = _lib.load()
    lib.fu_test_force_lockstep_wgrad(1 if request.param == "lockstep" else 0)
    yield request.param
    lib.fu_test_force_lockstep_wgrad(0)


@pytest.mark.parametrize("shape", [(2, 128, 64, 64, 40, 24, True), (1, 96, 0, 32, 16, 16, False),
                                   (16, 128, 0, 128, 32, 32, True), (3, 256, 0, 72, 19, 50, True),
                                   (1, 128, 128, 64, 8, 16, True), (2, 64, 64, 64, 32, 64, True),
                                   (2, 256, 0, 128, 32, 64, False), (1, 128, 0, 64, 24, 48, True),
                                   (3, 64, 192, 64, 16, 16, True)])
def test_bf16_wgrad_pingpong_kernel_is_bit_identical_to_lockstep(shape, lowp):
    """Three routes to the same sums: ping-pong kernel with whole-tile staging (where eligible), ping-pong kernel with the
    general staging, lock-step kernel.  Shapes cover interior and border tiles, a channel tile straddling two sources, no BN."""
    B, C0, C1, Cout, H, W, bn = shape
    lib = _lib.load()
    x0, x1, a, b, w, bias, _ = make_conv_case(*shape, seed=5)
    g = torch.Generator().manual_seed(6)
    ddy = nhwc_bf(torch.randn(B, Cout, H, W, generator=g))
    d0, d1 = nhwc_bf(x0), (nhwc_bf(x1) if x1 is not None else None)
    da, db = (a.to(DEV), b.to(DEV)) if bn else (None, None)
    outs = []
    for lock in (0, 1, 2):
        lib.fu_test_force_lockstep_wgrad(lock)
        dw = torch.full(w.shape, float("nan"), device=DEV)
        try:
            check(lib.fu_op_conv3x3_wgrad(_cur["code"], ptr(d0), C0, ptr(da), ptr(db), ptr(d1), C1, ptr(ddy), Cout, ptr(dw),
                                          B, H, W, stream()))
            torch.cuda.synchronize()
        finally:
            lib.fu_test_force_lockstep_wgrad(0)
        outs.append(dw.cpu())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("shape", [(1, 8, 0, 64, 8, 32, False), (2, 8, 0, 64, 16, 64, False), (3, 8, 0, 64, 40, 96, False),
                                   (5, 8, 0, 64, 64, 32, False), (16, 8, 0, 64, 256, 256, False)])
def test_bf16_wgrad_first_conv_kernel(shape, lowp):
    """k_wgrad_bf16_c8 (8 un-normalised input channels -> 64: the K = 72 stream over dy) against torch on the rounded operands
    and against the general weight-gradient kernel on the same operands: one all-border tile, border + interior tiles, a
    split that leaves some workgroups one tile short, and the bench shape (512 workgroups x 8 tiles)."""
    B, C0, C1, Cout, H, W, bn = shape
    lib = _lib.load()
    x0, x1, a, b, w, bias, _ = make_conv_case(*shape, seed=7)
    g = torch.Generator().manual_seed(8)
    dy = torch.randn(B, Cout, H, W, generator=g)
    d0, ddy = nhwc_bf(x0), nhwc_bf(dy)
    outs = []
    for lock in (0, 1):
        lib.fu_test_force_lockstep_wgrad(lock)
        dw = torch.full(w.shape, float("nan"), device=DEV)
        try:
            check(lib.fu_op_conv3x3_wgrad(_cur["code"], ptr(d0), C0, None, None, None, 0, ptr(ddy), Cout, ptr(dw), B, H, W,
                                          stream()))
            torch.cuda.synchronize()
        finally:
            lib.fu_test_force_lockstep_wgrad(0)
        outs.append(dw.cpu())
    assert torch.isfinite(outs[0]).all()
    assert rel_err(outs[0], outs[1]) < 2e-6          # same exact products, another fp32 summation order
    if B * H * W <= 1 << 16:
        ref = torch.nn.grad.conv2d_weight(bf(x0), w.shape, bf(dy), padding=1)
        assert rel_err(outs[0], ref) < 1e-4


@pytest.mark.parametrize("shape", BF_SHAPES)
def test_conv3x3_bf16_wgrad(shape, wgrad_path, lowp):
    B, C0, C1, Cout, H, W, bn = shape
    lib = _lib.load()
    x0, x1, a, b, w, bias, _ = make_conv_case(*shape, seed=2)
    x0r = bf(x0)
    xin = bf(torch.relu(x0r * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))) if bn else x0r
    if x1 is not None:
        xin = torch.cat([xin, bf(x1)], 1)
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(B, Cout, H, W, generator=g)
    ref = torch.nn.grad.conv2d_weight(xin, w.shape, bf(dy), padding=1)
    dw = torch.full(w.shape, float("nan"), device=DEV)
    d0, d1, ddy = nhwc_bf(x0), (nhwc_bf(x1) if x1 is not None else None), nhwc_bf(dy)
    da, db = (a.to(DEV), b.to(DEV)) if bn else (None, None)
    check(lib.fu_op_conv3x3_wgrad(_cur["code"], ptr(d0), C0, ptr(da), ptr(db), ptr(d1), C1, ptr(ddy), Cout, ptr(dw), B, H, W,
                                  stream()))
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), ref) < 1e-4   # exact products of bf16 operands, fp32 accumulation


def test_bf16_memory_bound_ops(lowp):
    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 20, 22, generator=g)
    a = torch.rand(16, generator=g) + 0.5
    b = torch.randn(16, generator=g) * 0.2
    xr = bf(x)
    z = torch.relu(xr * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
    dx, da, db = nhwc_bf(x), a.to(DEV), b.to(DEV)
    out = torch.empty(2, 10, 11, 16, device=DEV, dtype=_cur["dt"])
    check(lib.fu_op_maxpool2(_cur["code"], ptr(dx), ptr(da), ptr(db), ptr(out), 2, 20, 22, 16, stream()))
    up = torch.empty(2, 41, 45, 16, device=DEV, dtype=_cur["dt"])
    check(lib.fu_op_upsample2(_cur["code"], ptr(dx), ptr(da), ptr(db), ptr(up), 2, 20, 22, 16, 41, 45, stream()))
    torch.cuda.synchronize()
    assert rel_err(nchw_bf(out), F.max_pool2d(z, 2)) < _cur["eps"]
    u = F.interpolate(z, scale_factor=2, mode="bilinear", align_corners=True)
    ref = F.pad(u, [0, 1, 0, 1])
    assert rel_err(nchw_bf(up), ref) < _cur["eps"]


def test_gpu_augmentation_matches_oracle_and_flip_identities():
    from floodplanet_code_amd import augment
    from oracle import unet_oracle as O
    g = torch.Generator().manual_seed(6)
    B, Cc, H, W = 6, 5, 40, 56
    img = torch.rand(B, Cc, H, W, generator=g)
    tgt = torch.randint(0, 3, (B, H, W), generator=g)
    flags = [0, 1, 2, 3, 4, 7]
    angles = [0.0, 0.0, 0.0, 0.0, 33.0, 211.5]
    many = list(np.random.RandomState(0).uniform(0.0, 360.0, size=24)) + [45.0, 90.0, 180.0, 270.0, 359.999, 1e-3]
    out, tout = augment.apply(img.to(DEV), tgt.to(DEV), flags, angles, target_fill=0)
    torch.cuda.synchronize()
    out, tout = out.cpu(), tout.cpu()
    assert torch.equal(out[0], img[0]) and torch.equal(tout[0], tgt[0])
    assert torch.equal(out[1], img[1].flip(-1)) and torch.equal(tout[1], tgt[1].flip(-1))
    assert torch.equal(out[2], img[2].flip(-2)) and torch.equal(tout[2], tgt[2].flip(-2))
    assert torch.equal(out[3], img[3].flip(-1).flip(-2))
    for b in (4, 5):
        ri, rt = O.augment(img[b].numpy(), tgt[b].numpy(), flags[b], angles[b], 0)
        mism = (out[b].numpy() != ri).any(0) | (tout[b].numpy() != rt)
        assert mism.sum() == 0        # index work: bit-exact (same fp32 op order as the oracle, round-half-even)
    # 30 more angles (uniform draws as sample_transforms makes them, plus the axis-aligned ones), odd and even sizes
    for (hh, ww) in ((40, 56), (37, 45)):
        im = torch.rand(len(many), 3, hh, ww, generator=g)
        tg = torch.randint(0, 3, (len(many), hh, ww), generator=g)
        fl = [4 + (i % 4) for i in range(len(many))]
        o, to = augment.apply(im.to(DEV), tg.to(DEV), fl, many, target_fill=2)
        torch.cuda.synchronize()
        for b in range(len(many)):
            ri, rt = O.augment(im[b].numpy(), tg[b].numpy(), fl[b], many[b], 2)
            assert (o[b].cpu().numpy() != ri).sum() == 0 and (to[b].cpu().numpy() != rt).sum() == 0, (hh, ww, many[b])
    # a 90 degree rotation of a square tile is an exact rot90
    sq = torch.rand(1, 2, 32, 32, generator=g)
    o2, _ = augment.apply(sq.to(DEV), None, [4], [90.0])
    assert torch.equal(o2.cpu()[0], torch.rot90(sq[0], 1, dims=(-2, -1)))


def test_gpu_tile_assembly_matches_reference_fixture():
    """fu_assemble_tiles (normalise + edge-crop buffer + multi-sensor concat on the GPU) against the fixture produced by the
    reference's own `normalize` / `_add_buffer_to_image` (oracle/make_assemble_golden.py).  None and 'global': the same two
    float32 operations as numpy -> exact; 'local': the statistics are accumulated in fp64 here and pairwise in float32 by
    numpy -> 1e-6 relative on mean / std, 3e-6 of the value range on the normalised image."""
    import json, os
    from conftest import GOLDEN
    from floodplanet_code_amd.datasets.assemble import assemble_tiles
    from oracle import unet_oracle as O
    z = np.load(os.path.join(GOLDEN, "assemble_golden.npz"))
    for case in json.loads(bytes(z["meta"]).decode()):
        srcs = [torch.from_numpy(x).to(DEV) for x in O.assemble_case_sources(case)]
        vh = torch.tensor([v[0] for v in case["valid"]], dtype=torch.int32)
        vw = torch.tensor([v[1] for v in case["valid"]], dtype=torch.int32)
        gp = None
        if case["norm_mode"] == "global":
            g = O.assemble_global_params(case)
            gp = (torch.from_numpy(np.concatenate([g[s[0]]["mean"] for s in case["sources"]])),
                  torch.from_numpy(np.concatenate([g[s[0]]["std"] for s in case["sources"]])))
        img, mean, std = assemble_tiles(srcs, case["norm_mode"], (vh, vw), gp)
        torch.cuda.synchronize()
        n = case["name"]
        ref = z[n + "_image"]
        B, ctot = ref.shape[:2]
        np.testing.assert_allclose(mean.cpu().numpy().reshape(B, ctot), z[n + "_mean"], rtol=1e-6, atol=0, err_msg=n)
        np.testing.assert_allclose(std.cpu().numpy().reshape(B, ctot), z[n + "_std"], rtol=1e-6, atol=0, err_msg=n)
        if case["norm_mode"] == "local":
            np.testing.assert_allclose(img.cpu().numpy(), ref, rtol=0, atol=3e-6 * max(1.0, np.abs(ref).max()), err_msg=n)
        else:
            np.testing.assert_array_equal(img.cpu().numpy(), ref, err_msg=n)
    with pytest.raises(NotImplementedError):
        assemble_tiles(srcs, "bogus")
