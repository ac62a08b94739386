"""GPU: the code that only runs in the BENCHED 16-bit dispatch, anchored to references that are not the HIP path itself.

bench.py's bf16 / fp16 step runs the row-stationary conv kernel (fu_conv_rs.hip; dispatched at >= 512 workgroups, i.e. never
at fixture sizes by default), whose dgrad launches also emit the next BatchNorm backward's two sums from their epilogue
(BnbFuse), the head-backward kernel with the same fused sums, and the 16-bit pooled BatchNorm backward.  Three layers of checks:

  (a) whole net: the reference fixtures (made from the real st_water_seg/models/unet.py:6-111) with the row-stationary kernel
      and its fused sums FORCED at fixture size (fu_test_conv_tile_mode(3)), in bf16 and fp16, at the stated 16-bit
      tolerances, plus per-tensor gradient norm and cosine;
  (b) op level, through the C ABI: the fused sums of rs<8>, rs<4> and of the head backward, and the 16-bit pooled BatchNorm
      backward, against torch (fp64 on the 16-bit-rounded operands / autograd of maxpool(relu(bn(y))));
  (c) per backward block at the bench shape: identical saved activations and identical incoming gradients into
      fu_backward_block(k) under the default dispatch and under the conservative one (square-tile fast kernel, lock-step
      wgrad, separate reduce passes) -- inside ONE block nothing decorrelates, so every parameter gradient of the block and
      every gradient map leaving it must agree to the element type's rounding;
  and a negative control: with the fused sums deliberately multiplied by 1.1 (fu_test_perturb_bnb_sums) (b) and (c) fail.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import case_inputs, is_dead_bias, load_golden
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
from floodplanet_code_amd.unet import HipUNet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BN_EPS = 1e-5

LOWP = {"bf16": dict(code=_lib.FU_BF16, dt=torch.bfloat16, eps=2.0 ** -8),
        "fp16": dict(code=_lib.FU_F16, dt=torch.float16, eps=2.0 ** -11)}
# whole-net tolerances against the fp32 reference fixtures: the stated ones of tests/test_gpu_unet.py (LOWP_TOL), plus per
# tensor: gradient norm within a factor and cosine (tensors of >= 64 elements whose reference norm is not noise)
NET_TOL = {"bf16": dict(lmax=0.25, lrms=0.05, loss=0.03, agree=0.93, cos_med=0.9, cos_min=0.75, norm=(0.5, 2.0)),
           "fp16": dict(lmax=0.06, lrms=0.008, loss=0.005, agree=0.99, cos_med=0.97, cos_min=0.95, norm=(0.7, 1.4))}
# ONE tolerance table for all fixtures.  (Round 3 had a second, wider one for f_full_c8_32: that fixture is B = 1, i.e. FOUR samples
# per channel at its 2 x 2 level -- a degenerate BatchNorm, median bf16 cosine 0.84 whatever kernel runs -- and proved nothing in
# 16-bit arithmetic; round 4 replaced it here by f_full_c8_32_b8, the same 32 x 32 full-width net with 32 samples per channel.)


def stream():
    return torch.cuda.current_stream().cuda_stream


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


def rnd(x, dt):
    return x.to(dt).float()


def to_nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to(dt).to(DEV)


def from_nhwc(x):
    return x.float().permute(0, 3, 1, 2).contiguous().cpu()


@pytest.fixture
def forced_rs():
    lib = _lib.load()
    lib.fu_test_conv_tile_mode(3)
    lib._forced_mode = 3
    yield lib
    lib.fu_test_conv_tile_mode(0)
    lib.fu_test_bnb_separate(0)
    lib.fu_test_perturb_bnb_sums(1.0)


@pytest.fixture
def forced_pp():
    """The persistent ping-pong kernel (fu_conv_pp.hip) wherever a launch is eligible (>= 8 tiles of 16 x 32 pixels, H % 32 ==
    0); by default it is dispatched from 256 tiles on, i.e. never at fixture sizes."""
    lib = _lib.load()
    lib.fu_test_conv_tile_mode(4)
    lib._forced_mode = 4
    yield lib
    lib.fu_test_conv_tile_mode(0)
    lib.fu_test_bnb_separate(0)
    lib.fu_test_perturb_bnb_sums(1.0)


# ------------------------------------------------------------------------------------------------------------------
# (a) whole net, row-stationary kernel + fused sums forced at fixture size, against the reference fixtures
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["f_full_c8_64_b8", "f_full_c8_64_b2", "f_full_c8_32_b8"])
def test_forced_row_stationary_step_against_reference_fixture(name, prec, forced_rs):
    _forced_step_against_reference_fixture(name, prec, forced_rs)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["f_full_c8_64_b8", "f_full_c8_64_b2"])
def test_forced_pingpong_step_against_reference_fixture(name, prec, forced_pp):
    """The same whole-net check with the persistent ping-pong kernel on every eligible layer (the 64 x 64 and 32 x 32 levels of
    these fixtures; the rest falls back to the default kernels)."""
    _forced_step_against_reference_fixture(name, prec, forced_pp)


def _forced_step_against_reference_fixture(name, prec, lib):
    meta, z = load_golden(name)
    batch, st = case_inputs(meta)
    ii = meta["resolved_ignore_index"]
    tol = dict(NET_TOL[prec])
    net = HipUNet(meta["n_in"], 3, base_channels=meta["base"], precision=prec)
    net.load_state_dict(st)
    net.to(DEV).train()
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    loss, logits = net.loss(x, t, ii, return_logits=True)
    loss.backward()
    torch.cuda.synchronize()
    g_fused = net.flat_grads().clone()
    d = logits.detach().cpu().numpy() - z["logits1"]
    assert np.abs(d).max() <= tol["lmax"], np.abs(d).max()
    assert np.sqrt((d ** 2).mean()) <= tol["lrms"]
    assert abs(loss.item() - z["loss1"].item()) <= tol["loss"]
    agree = (logits.detach().cpu().numpy().argmax(1) == z["logits1"].argmax(1)).mean()
    assert agree >= tol["agree"], agree
    assert torch.isfinite(g_fused).all() and g_fused.abs().max() > 0
    # (the full-width fixtures store gradient norms and 64-element slices; the full reference gradients for the cosines
    #  come from the oracle, which make_golden.py pinned bit-exactly to the reference and which the slices re-check here)
    from oracle import unet_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads()))
    _, _, g_ref = O.loss_and_grads({k: v.clone() for k, v in st.items()}, batch, ii)
    cos, worst = [], (1.0, None)
    for j, (k, p) in enumerate(net.named_parameters()):
        if is_dead_bias(k):
            continue
        ref_norm = z["grad_stats1"][j][2]
        if f"g1s_{j}" in z.files:
            assert np.allclose(g_ref[k].reshape(-1)[:64].numpy(), z[f"g1s_{j}"], rtol=1e-4, atol=1e-7 + 1e-4 * ref_norm), k
        if ref_norm >= 1e-5:
            r = p.grad.norm().item() / ref_norm
            assert tol["norm"][0] <= r <= tol["norm"][1], (k, r)
        if p.numel() >= 64 and ref_norm >= 1e-5:
            a, b = p.grad.cpu().double().reshape(-1), g_ref[k].double().reshape(-1)
            c = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
            cos.append(c)
            if c < worst[0]:
                worst = (c, k)
    print(f"{name} {prec} forced rs: logits max {np.abs(d).max():.4f} rms {np.sqrt((d ** 2).mean()):.4f} agree {agree:.4f} "
          f"cos median {np.median(cos):.4f} min {worst[0]:.4f} ({worst[1]})")
    assert np.median(cos) >= tol["cos_med"], np.median(cos)
    assert worst[0] >= tol["cos_min"], worst
    # ... and no worse than the default dispatch (fast / general kernels, separate reduce passes) on the same fixture
    lib.fu_test_conv_tile_mode(0)
    net.zero_grad(set_to_none=True)
    loss_d = net.loss(x, t, ii)
    loss_d.backward()
    torch.cuda.synchronize()
    lib.fu_test_conv_tile_mode(lib._forced_mode)
    cos_d = []
    for j, (k, p) in enumerate(net.named_parameters()):
        if not is_dead_bias(k) and p.numel() >= 64 and z["grad_stats1"][j][2] >= 1e-5:
            a, b = p.grad.cpu().double().reshape(-1), g_ref[k].double().reshape(-1)
            cos_d.append((a @ b / (a.norm() * b.norm() + 1e-30)).item())
    print(f"    default dispatch: cos median {np.median(cos_d):.4f} min {min(cos_d):.4f}")
    assert np.median(cos) >= np.median(cos_d) - 0.03 and worst[0] >= min(cos_d) - 0.1
    # the forced dispatch really took the fused-sum route: with the sums back in their own reduce pass the result differs
    # (another summation order, g after its rounding) -- but only by rounding: loss identical, last BatchNorm's gradients to 4e-4
    lib.fu_test_bnb_separate(1)
    loss2 = net.loss(x, t, ii)
    net.zero_grad(set_to_none=True)
    loss2.backward()
    torch.cuda.synchronize()
    g_sep = net.flat_grads().clone()
    lib.fu_test_bnb_separate(0)
    assert loss2.item() == loss.item()
    assert not torch.equal(g_sep, g_fused)
    for (k, p, off, n) in net._table:
        if k in ("up4.conv.double_conv.4.weight", "up4.conv.double_conv.4.bias"):
            # (measured 2e-5 ... 1.1e-4 over the three fixtures and two precisions; a wrong sum is off by O(1))
            assert rel(g_fused[off:off + n], g_sep[off:off + n]) <= 4e-4, k


# ------------------------------------------------------------------------------------------------------------------
# (b) op level
# ------------------------------------------------------------------------------------------------------------------
def bn_case(C, g, dt, y):
    """BatchNorm coefficients of a train-mode BN over y [B,C,H,W] (16-bit-rounded values) with random affine parameters."""
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.3
    yd = y.double()
    mean = yd.mean((0, 2, 3))
    var = yd.var((0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + BN_EPS)
    a = gamma.double() * invstd
    b = beta.double() - mean * a
    return gamma, beta, mean.float(), invstd.float(), a.float(), b.float()


RS_SUM_SHAPES = [
    # B, Cout (the dgrad's K), C0 (destination channels), H, W           kernel
    (4, 64, 64, 256, 256),     # rs<8>: 512 tall tiles
    (8, 128, 128, 128, 128),   # rs<8>: two channel tiles, 4 chunks
    (2, 128, 64, 32, 48),      # rs<4>
    (1, 256, 192, 32, 32),     # rs<4>, three channel tiles, 8 chunks
    (3, 64, 64, 16, 16),       # rs<4>, every tile a border tile
]


def _dgrad_bnsums(lib, lw, shape, seed=1):
    B, Cout, C0, H, W = shape
    g = torch.Generator().manual_seed(seed)
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.randn(Cout, C0, 3, 3, generator=g) / (3.0 * Cout ** 0.5)
    y = rnd(torch.randn(B, C0, H, W, generator=g), lw["dt"])
    gamma, beta, mean, invstd, a, b = bn_case(C0, g, lw["dt"], y)
    gref = torch.nn.grad.conv2d_input((B, C0, H, W), rnd(w, lw["dt"]), rnd(dy, lw["dt"]), padding=1)
    m = (a.double().view(1, -1, 1, 1) * y.double() + b.double().view(1, -1, 1, 1)) > 0
    gm = gref.double() * m
    xh = (y.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
    s1_ref, s2_ref = gm.sum((0, 2, 3)), (gm * xh).sum((0, 2, 3))
    dx = torch.full((B, H, W, C0), float("nan"), device=DEV, dtype=lw["dt"])
    s1 = torch.empty(C0, device=DEV)
    s2 = torch.empty(C0, device=DEV)
    ddy, dw_, dyv = to_nhwc(dy, lw["dt"]), w.to(DEV), to_nhwc(y, lw["dt"])
    da, db, dm, di = a.to(DEV), b.to(DEV), mean.to(DEV), invstd.to(DEV)
    check(lib.fu_op_conv3x3_dgrad_bnsums(lw["code"], ptr(ddy), Cout, ptr(dw_), ptr(dx), C0, ptr(dyv), ptr(da), ptr(db),
                                         ptr(dm), ptr(di), ptr(s1), ptr(s2), B, H, W, stream()))
    torch.cuda.synchronize()
    return from_nhwc(dx), gref, s1.cpu(), s2.cpu(), s1_ref, s2_ref


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", RS_SUM_SHAPES)
def test_row_stationary_dgrad_fused_bn_backward_sums(shape, prec, forced_rs):
    """sum g*m and sum g*m*xhat from the dgrad epilogue (g = the fp32 accumulators) against fp64 sums over torch's dgrad of
    the rounded operands: <= 1e-4 relative (vector norm over the channels); the data gradient itself to one rounding unit."""
    lw = LOWP[prec]
    dx, gref, s1, s2, s1_ref, s2_ref = _dgrad_bnsums(forced_rs, lw, shape)
    assert rel(dx, gref) < lw["eps"]
    assert rel(s1, s1_ref) <= 1e-4, rel(s1, s1_ref)
    assert rel(s2, s2_ref) <= 1e-4, rel(s2, s2_ref)


# the persistent ping-pong kernel accumulates every output element in the row-stationary kernel's order (chunk, column shift,
# kernel row) and sums its statistics in rs<8>'s order: same bits.  B, C0, C1, Cout, H, W, BatchNorm prologue
PP_FWD = [(2, 64, 0, 64, 64, 48, True), (2, 32, 32, 128, 32, 32, True), (16, 64, 64, 64, 128, 128, True),
          (16, 256, 256, 256, 32, 32, True), (8, 96, 0, 64, 32, 16, False), (16, 128, 0, 256, 64, 64, False),
          (4, 512, 0, 512, 32, 32, True)]
PP_DGRAD = [(2, 64, 64, 0, 64, 32), (16, 64, 64, 64, 128, 128), (4, 512, 512, 512, 32, 32), (16, 256, 128, 128, 64, 64)]
PP_SUMS = [(4, 64, 64, 128, 128), (8, 128, 128, 64, 64), (2, 256, 192, 32, 32), (2, 64, 64, 64, 32)]


def _bits(t):
    return t.view(torch.int32) if t.dtype == torch.float32 else t.view(torch.int16)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_pingpong_kernel_is_bit_identical_to_row_stationary(prec):
    """Forward (BatchNorm prologue, two sources, bias, statistics), dgrad (two destinations) and dgrad with the fused
    BatchNorm-backward sums: tile mode 4 against tile mode 3 on the same operands.  Outputs always bit for bit; the statistics /
    sums bit for bit where mode 3 runs rs<8> (>= 512 of its tall tiles: the same statistics tiles), to 1e-5 where it runs rs<4>."""
    lib = _lib.load()
    lw = LOWP[prec]
    dt, code = lw["dt"], lw["code"]

    def both(run):
        outs = []
        try:
            for m in (3, 4):
                lib.fu_test_conv_tile_mode(m)
                outs.append(run())
                torch.cuda.synchronize()
        finally:
            lib.fu_test_conv_tile_mode(0)
        return outs

    def compare(outs, tall, what):
        assert torch.isfinite(outs[1][0].float()).all(), what
        assert torch.equal(_bits(outs[0][0]), _bits(outs[1][0])), what
        for p, q in zip(outs[0][1:], outs[1][1:]):
            if p.dtype != torch.float32:
                assert torch.equal(_bits(p), _bits(q)), what
            elif tall:
                assert torch.equal(_bits(p), _bits(q)), what
            else:
                assert ((p - q).abs() <= 1e-5 * (p.abs().max() + 1)).all(), what

    for sh in PP_FWD:
        B, C0, C1, Cout, H, W, bn = sh
        g = torch.Generator().manual_seed(0)
        x0 = torch.randn(B, H, W, C0, generator=g).to(DEV).to(dt)
        x1 = torch.randn(B, H, W, C1, generator=g).to(DEV).to(dt) if C1 else None
        a = (torch.rand(C0, generator=g) + 0.5).to(DEV)
        b = (torch.randn(C0, generator=g) * 0.1).to(DEV)
        w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 10).to(DEV)
        bias = torch.randn(Cout, generator=g).to(DEV)

        def run():
            y = torch.full((B, H, W, Cout), float("nan"), device=DEV, dtype=dt)
            ssum, ssq = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
            check(lib.fu_op_conv3x3_fwd(code, ptr(x0), C0, ptr(a) if bn else None, ptr(b) if bn else None, ptr(x1) if C1 else None,
                                        C1, ptr(w), ptr(bias), ptr(y), Cout, B, H, W, ptr(ssum), ptr(ssq), stream()))
            return y, ssum, ssq
        compare(both(run), B * (H // 32) * (W // 16) * (Cout // 64) >= 512, ("fwd", sh))
    for sh in PP_DGRAD:
        B, Cout, C0, C1, H, W = sh
        g = torch.Generator().manual_seed(1)
        dy = torch.randn(B, H, W, Cout, generator=g).to(DEV).to(dt)
        w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 3).to(DEV)

        def run():
            dx0 = torch.full((B, H, W, C0), float("nan"), device=DEV, dtype=dt)
            dx1 = torch.full((B, H, W, C1), float("nan"), device=DEV, dtype=dt) if C1 else None
            check(lib.fu_op_conv3x3_dgrad(code, ptr(dy), Cout, ptr(w), ptr(dx0), C0, ptr(dx1) if C1 else None, C1, B, H, W,
                                          stream()))
            return (dx0,) + ((dx1,) if C1 else ())
        compare(both(run), True, ("dgrad", sh))
    for sh in PP_SUMS:
        B, Cout, C0, H, W = sh
        g = torch.Generator().manual_seed(2)
        dy = torch.randn(B, H, W, Cout, generator=g).to(DEV).to(dt)
        w = (torch.randn(Cout, C0, 3, 3, generator=g) / (3.0 * Cout ** 0.5)).to(DEV)
        yv = torch.randn(B, H, W, C0, generator=g).to(DEV).to(dt)
        a = (torch.rand(C0, generator=g) + 0.5).to(DEV)
        b = (torch.randn(C0, generator=g) * 0.3).to(DEV)
        mean = (torch.randn(C0, generator=g) * 0.1).to(DEV)
        invstd = (torch.rand(C0, generator=g) + 0.5).to(DEV)

        def run():
            dx = torch.full((B, H, W, C0), float("nan"), device=DEV, dtype=dt)
            s1, s2 = torch.empty(C0, device=DEV), torch.empty(C0, device=DEV)
            check(lib.fu_op_conv3x3_dgrad_bnsums(code, ptr(dy), Cout, ptr(w), ptr(dx), C0, ptr(yv), ptr(a), ptr(b), ptr(mean),
                                                 ptr(invstd), ptr(s1), ptr(s2), B, H, W, stream()))
            return dx, s1, s2
        compare(both(run), B * (H // 32) * (W // 16) * (C0 // 64) >= 512, ("dgrad + sums", sh))


def _head_bwd(lib, lw, seed=3):
    B, Cc, H, W, ncls = 2, 64, 64, 80, 3
    g = torch.Generator().manual_seed(seed)
    y = rnd(torch.randn(B, Cc, H, W, generator=g), lw["dt"])
    gamma, beta, mean, invstd, a, b = bn_case(Cc, g, lw["dt"], y)
    dl = torch.randn(B, H, W, ncls, generator=g) * 1e-3
    w = torch.randn(ncls, Cc, generator=g) * 0.2
    zpre = a.double().view(1, -1, 1, 1) * y.double() + b.double().view(1, -1, 1, 1)
    zact = torch.relu(zpre).permute(0, 2, 3, 1).reshape(-1, Cc)              # [npix, C]
    dl2 = dl.double().reshape(-1, ncls)
    g_ref = dl2 @ w.double()                                                   # [npix, C]
    dw_ref, db_ref = dl2.t() @ zact, dl2.sum(0)
    m = (zpre > 0).permute(0, 2, 3, 1).reshape(-1, Cc)
    xh = ((y.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1).reshape(-1, Cc)
    s1_ref, s2_ref = (g_ref * m).sum(0), (g_ref * m * xh).sum(0)
    npix = B * H * W
    gout = torch.full((npix, Cc), float("nan"), device=DEV, dtype=lw["dt"])
    dw = torch.empty(ncls, Cc, device=DEV)
    db = torch.empty(ncls, device=DEV)
    s1 = torch.empty(Cc, device=DEV)
    s2 = torch.empty(Cc, device=DEV)
    ddl, dyv, dwt = dl.to(DEV).contiguous(), to_nhwc(y, lw["dt"]), w.to(DEV)
    da, dbb, dm, di = a.to(DEV), b.to(DEV), mean.to(DEV), invstd.to(DEV)
    check(lib.fu_op_head_bwd(lw["code"], ptr(ddl), ptr(dyv), ptr(da), ptr(dbb), ptr(dwt), Cc, ncls, npix, ptr(gout),
                             ptr(dw), ptr(db), ptr(dm), ptr(di), ptr(s1), ptr(s2), stream()))
    torch.cuda.synchronize()
    return gout.float().cpu(), g_ref, dw.cpu(), dw_ref, db.cpu(), db_ref, s1.cpu(), s1_ref, s2.cpu(), s2_ref


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_head_backward_with_fused_bn_backward_sums(prec):
    """k_head_bwd<.., true>: data gradient, OutConv weight / bias gradient and the last BatchNorm's backward sums against fp64."""
    lw = LOWP[prec]
    gout, g_ref, dw, dw_ref, db, db_ref, s1, s1_ref, s2, s2_ref = _head_bwd(_lib.load(), lw)
    assert rel(gout, g_ref) < lw["eps"]
    assert rel(dw, dw_ref) <= 1e-4 and rel(db, db_ref) <= 1e-5
    assert rel(s1, s1_ref) <= 1e-4, rel(s1, s1_ref)
    assert rel(s2, s2_ref) <= 1e-4, rel(s2, s2_ref)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("shape,pooled", [((2, 64, 32, 48), True), ((1, 32, 37, 45), True), ((2, 128, 16, 16), True),
                                          ((2, 64, 32, 48), False)])
def test_16bit_bn_relu_backward_with_folded_pool_against_autograd(shape, pooled, prec):
    """k_bn_bwd_pool<T, reduce / apply> (and the plain reduce / apply pair) in the 16-bit modes against autograd (fp64) of
    relu(batch_norm(y)) consumed by a skip path and by max_pool2d: dL/dy to one rounding unit, dgamma / dbeta to 1e-4."""
    lw = LOWP[prec]
    lib = _lib.load()
    B, Cc, H, W = shape
    g = torch.Generator().manual_seed(11)
    y = rnd(torch.randn(B, Cc, H, W, generator=g), lw["dt"])
    gamma, beta, mean, invstd, a, b = bn_case(Cc, g, lw["dt"], y)
    g_skip = rnd(torch.randn(B, Cc, H, W, generator=g), lw["dt"])
    g_pool = rnd(torch.randn(B, Cc, H // 2, W // 2, generator=g), lw["dt"]) if pooled else None
    y64 = y.double().requires_grad_(True)
    ga64, be64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    out = torch.relu(F.batch_norm(y64, None, None, ga64, be64, True, 0.0, BN_EPS))
    obj = (out * g_skip.double()).sum()
    if pooled:
        obj = obj + (F.max_pool2d(out, 2) * g_pool.double()).sum()
    obj.backward()
    gbuf = to_nhwc(g_skip, lw["dt"])
    dyv = to_nhwc(y, lw["dt"])
    dgp = to_nhwc(g_pool, lw["dt"]) if pooled else None
    dgamma = torch.empty(Cc, device=DEV)
    dbeta = torch.empty(Cc, device=DEV)
    da, dbb, dm, di = a.to(DEV), b.to(DEV), mean.to(DEV), invstd.to(DEV)
    check(lib.fu_op_bn_bwd(lw["code"], ptr(gbuf), ptr(dyv), Cc, B, H, W, ptr(da), ptr(dbb), ptr(dm), ptr(di), ptr(dgp),
                           ptr(dgamma), ptr(dbeta), stream()))
    torch.cuda.synchronize()
    assert rel(from_nhwc(gbuf), y64.grad) < 1.5 * lw["eps"]
    assert rel(dgamma.cpu(), ga64.grad) <= 1e-4 and rel(dbeta.cpu(), be64.grad) <= 1e-4


@pytest.mark.parametrize("prec", ["bf16"])
def test_negative_control_perturbed_fused_sums_fail_the_op_checks(prec, forced_rs):
    """The checks above can fail: with every fused sum multiplied by 1.1 the op-level comparisons are off by ~10 %."""
    lw = LOWP[prec]
    lib = forced_rs
    lib.fu_test_perturb_bnb_sums(1.1)
    _, _, s1, s2, s1_ref, s2_ref = _dgrad_bnsums(lib, lw, RS_SUM_SHAPES[2])
    assert rel(s1, s1_ref) > 0.05 and rel(s2, s2_ref) > 0.05
    r = _head_bwd(lib, lw)
    assert rel(r[6], r[7]) > 0.05 and rel(r[8], r[9]) > 0.05
    lib.fu_test_perturb_bnb_sums(1.0)


# ------------------------------------------------------------------------------------------------------------------
# (c) per backward block at the bench shape: default dispatch == conservative dispatch to the element type's rounding
# ------------------------------------------------------------------------------------------------------------------
_hip = None


def _memcpy_dtod(dst, src, nbytes):
    """hipMemcpy device -> device through the process's HIP runtime (the one torch loaded)."""
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
    rc = _hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3)   # hipMemcpyDeviceToDevice
    assert rc == 0, rc


def _buffer(net, block, which, dt):
    """Snapshot of one context-owned 16-bit tensor (fu_test_get_buffer) as a torch tensor."""
    lib = _lib.load()
    p, n = C.c_void_p(), C.c_int64()
    check(lib.fu_test_get_buffer(net._ctx, block, which, C.byref(p), C.byref(n)))
    assert p.value and n.value > 0, (block, which)
    t = torch.empty(n.value, dtype=dt, device=DEV)
    torch.cuda.synchronize()
    _memcpy_dtod(t.data_ptr(), p.value, n.value * 2)
    return t


def _outgoing(net, plan_idx, dt):
    """The gradient maps a plan block hands to other blocks: (label, tensor) list."""
    nb = 9
    out = []
    if plan_idx >= 5:                                    # up block k = plan_idx - 5: skip level 3 - k, low input = previous block
        k = plan_idx - 5
        out.append((f"skip gradient (level {3 - k})", _buffer(net, 3 - k, 3, dt)))
        out.append(("gradient of the low-resolution input", _buffer(net, plan_idx - 1, 3, dt)))
    elif plan_idx >= 1:                                  # down block: dL/d(pooled input)
        out.append(("gradient of the pooled input", _buffer(net, plan_idx, 4, dt)))
    assert nb == 9
    return out


def _run_block(net, x, t, b, conservative, perturb=1.0):
    """forward + loss + backward blocks 0..b, block b under the chosen dispatch; returns the block's parameter gradients
    and outgoing gradient maps."""
    lib = _lib.load()
    dev = x.device
    dt = LOWP[net.precision]["dt"]
    net._forward_raw(x, True, want_logits=False)
    net._loss_raw(t, 0, dev)
    s = net._stream(dev)
    for blk in range(b):
        check(lib.fu_backward_block(net._ctx, blk, None, s))
    try:
        if conservative:
            lib.fu_test_conv_tile_mode(1)
            lib.fu_test_force_lockstep_wgrad(1)
            lib.fu_test_bnb_separate(1)
        lib.fu_test_perturb_bnb_sums(perturb)
        check(lib.fu_backward_block(net._ctx, b, None, s))
    finally:
        lib.fu_test_conv_tile_mode(0)
        lib.fu_test_force_lockstep_wgrad(0)
        lib.fu_test_bnb_separate(0)
        lib.fu_test_perturb_bnb_sums(1.0)
    torch.cuda.synchronize()
    o, n = C.c_int64(), C.c_int64()
    check(lib.fu_block_param_range(net._ctx, b, C.byref(o), C.byref(n)))
    grads = {k: net.flat_grads()[off:off + m].clone() for (k, p, off, m) in net._table
             if o.value <= off < o.value + n.value}
    plan_idx = 9 - b if b <= 4 else 9 - b          # b = 1..4 -> 8..5 (up4..up1); b = 5..9 -> 4..0 (down4..inc)
    return grads, _outgoing(net, plan_idx, dt)


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_every_backward_block_default_dispatch_equals_conservative_dispatch(prec):
    """BASELINE configs[1]'s shape (B=16, 8 bands, 256x256, full width).  For every backward block k = 1..9: the same forward,
    the same blocks 0..k-1 (default dispatch, deterministic), then block k once under the default dispatch (row-stationary /
    fast conv kernels, fused BatchNorm-backward sums, ping-pong wgrad, weight-gradient chain on the side stream) and once
    under the conservative one.  Inside one block the two routes see identical inputs; they differ in summation order and in
    whether the fused sums use g before or after its rounding to the element type, so every parameter gradient of the block
    and every gradient map leaving it must agree to the element type's rounding (2^-7 relative for bf16, 2^-10 for fp16)."""
    from oracle import unet_oracle as O
    torch.manual_seed(0)
    net = HipUNet(8, 3, precision=prec).to(DEV).train()
    batch = O.make_batch(16, 8, 256, 256, seed=11)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    tol = 2.0 ** -7 if prec == "bf16" else 2.0 ** -10
    worst = (0.0, None)
    for b in range(1, 10):
        g_def, out_def = _run_block(net, x, t, b, conservative=False)
        g_con, out_con = _run_block(net, x, t, b, conservative=True)
        assert g_def.keys() == g_con.keys() and len(g_def) == 8
        for k in g_def:
            if is_dead_bias(k) or g_con[k].norm().item() < 1e-9:
                continue
            e = rel(g_def[k], g_con[k])
            worst = max(worst, (e, k))
            assert e <= tol, (b, k, e)
        for (label, a), (_, c) in zip(out_def, out_con):
            e = rel(a.float(), c.float())
            worst = max(worst, (e, f"block {b}: {label}"))
            assert e <= tol, (b, label, e)
    print(f"{prec}: worst per-block default-vs-conservative deviation {worst[0]:.3e} ({worst[1]}); bound {tol:.3e}")
    assert worst[0] > 0.0            # the two dispatches really are different code


def test_backward_blocks_at_the_largest_benched_shape_default_equals_conservative():
    """BASELINE configs[3]'s shape at its batch (9 bands = image + DEM side by side, 512x512, B=16, fp16): the largest grids
    the bench launches (8192-workgroup conv launches, 4 M pixels per map).  Same per-block check as above on the first and
    the last decoder block, the deepest encoder block and the stem: an index that overflows or a tile decode that wraps at
    this size shows up as a gradient far outside the rounding bound."""
    from oracle import unet_oracle as O
    torch.manual_seed(0)
    net = HipUNet(9, 2, precision="fp16").to(DEV).train()
    batch = O.make_batch(16, 9, 512, 512, seed=5)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    tol = 2.0 ** -10
    worst = (0.0, None)
    for b in (1, 4, 5, 9):
        g_def, out_def = _run_block(net, x, t, b, conservative=False)
        g_con, out_con = _run_block(net, x, t, b, conservative=True)
        for k in g_def:
            if is_dead_bias(k) or g_con[k].norm().item() < 1e-9:
                continue
            assert torch.isfinite(g_def[k]).all(), k
            e = rel(g_def[k], g_con[k])
            worst = max(worst, (e, k))
            assert e <= tol, (b, k, e)
        for (label, a), (_, c) in zip(out_def, out_con):
            e = rel(a.float(), c.float())
            worst = max(worst, (e, f"block {b}: {label}"))
            assert e <= tol, (b, label, e)
    print(f"512x512 B=16 fp16: worst per-block default-vs-conservative deviation {worst[0]:.3e} ({worst[1]}); bound {tol:.3e}")
    assert worst[0] > 0.0


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_head_gradient_recomputed_in_the_apply_pass_equals_the_stored_one(prec):
    """Bench shape.  By default the head backward does not store g = dlogits . W: the BatchNorm-backward apply pass of the
    last conv recomputes it in fp32 (k_bn_bwd_apply_head).  With fu_test_head_store_g(1) the head backward stores g in the
    element type and the plain apply pass reads it.  The head's own gradients are bit-identical (same kernel arithmetic),
    block up4's parameter gradients and outgoing gradient maps agree to the element type's rounding."""
    from oracle import unet_oracle as O
    lib = _lib.load()
    torch.manual_seed(0)
    net = HipUNet(8, 3, precision=prec).to(DEV).train()
    batch = O.make_batch(16, 8, 256, 256, seed=11)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    dt = LOWP[prec]["dt"]
    res = []
    for store in (0, 1):
        net._forward_raw(x, True, want_logits=False)
        net._loss_raw(t, 0, DEV)
        s = net._stream(DEV)
        try:
            lib.fu_test_head_store_g(store)
            check(lib.fu_backward_block(net._ctx, 0, None, s))
            check(lib.fu_backward_block(net._ctx, 1, None, s))
        finally:
            lib.fu_test_head_store_g(0)
        torch.cuda.synchronize()
        grads = {}
        for b in (0, 1):
            o, n = C.c_int64(), C.c_int64()
            check(lib.fu_block_param_range(net._ctx, b, C.byref(o), C.byref(n)))
            grads.update({k: net.flat_grads()[off:off + m].clone() for (k, p, off, m) in net._table
                          if o.value <= off < o.value + n.value})
        res.append((grads, _outgoing(net, 8, dt)))
    (g_re, out_re), (g_st, out_st) = res
    tol = 2.0 ** -7 if prec == "bf16" else 2.0 ** -10
    worst = 0.0
    for k in g_re:
        if k.startswith("outc."):
            assert torch.equal(g_re[k], g_st[k]), k
            continue
        if is_dead_bias(k) or g_st[k].norm().item() < 1e-9:
            continue
        e = rel(g_re[k], g_st[k])
        worst = max(worst, e)
        assert e <= tol, (k, e)
    for (label, a), (_, c) in zip(out_re, out_st):
        e = rel(a.float(), c.float())
        worst = max(worst, e)
        assert e <= tol, (label, e)
    print(f"{prec}: recomputed vs stored head gradient, worst deviation {worst:.3e}; bound {tol:.3e}")
    assert worst > 0.0               # the two routes really are different code


def test_negative_control_perturbed_fused_sums_fail_the_block_check():
    """The per-block check can fail: fused sums x 1.1 inside block 2 (up3) move that block's first BatchNorm's gradients by
    10 % (bound of the real check: 2^-7)."""
    from oracle import unet_oracle as O
    torch.manual_seed(0)
    net = HipUNet(8, 3, precision="bf16").to(DEV).train()
    batch = O.make_batch(16, 8, 256, 256, seed=11)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    g_bad, _ = _run_block(net, x, t, 2, conservative=False, perturb=1.1)
    g_con, _ = _run_block(net, x, t, 2, conservative=True)
    e = {k: rel(g_bad[k], g_con[k]) for k in g_bad if not is_dead_bias(k)}
    assert e["up3.conv.double_conv.1.bias"] > 0.05 and e["up3.conv.double_conv.1.weight"] > 0.05, e
