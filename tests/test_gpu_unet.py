"""GPU: the whole HIP training step (through HipUNet -> C ABI) against
  (1) the golden fixtures generated from the real reference, and
  (2) the oracle run live on the host CPU on the same seeded inputs,
plus size-independent properties at the benchmark size.

Tolerances (fp32 mode): logits |d| <= 1e-4 (north-star bound; observed 1-2e-5), loss 1e-5.
Gradients: the reference's OWN fp32 gradients sit up to 1.7e-3 (relative L2 per tensor) away from an fp64 run of
the same graph on these cases, because tiny BatchNorm batches at the deep levels amplify rounding and ReLU /
max-pool near-ties flip (measured on the GPU box, see DESIGN.md "Parity": on some cases torch-fp32 takes the
flip and the HIP path does not, on others the reverse; on the 300x300 case torch-fp32 is 6e-3 off the fp64
truth while the HIP path is 1.5e-3 off; after a later change that only moved roundings by one ulp -- explicit fma in
the BN activation -- the HIP path drew a flip torch did not: worst tensor 2.2e-2 from fp64, median 2.2e-3 against
torch's 3.9e-3).  Which side takes a flip is luck, so against fp32 references each tensor is bounded loosely (3e-2:
a wrong kernel is off by O(1)) and the MEDIAN over the live tensors tightly (5e-3); test_matches_live_oracle
additionally compares both with an fp64 run of the oracle: the HIP
gradients must be within max(3x the torch-fp32 error, 1e-2) of the fp64 truth.  The 18 conv biases in front of a BatchNorm are
excluded from elementwise checks: their gradient is analytically zero (conftest.is_dead_bias).
"""
GRAD_TOL = 3e-2        # per tensor (flip-dominated, see above)
GRAD_MEDIAN_TOL = 8e-3  # median of the per-tensor relative errors of one backward pass (5e-3 until the two larger-batch
                        # full-width fixtures were added: on f_full_c4_96_b6 torch-fp32 ITSELF sits 6.2e-3 (median) from an
                        # fp64 run, the HIP path 2.2e-3, hence 6.6e-3 between the two -- tools/grad_truth_diag.py)
import numpy as np
import pytest
import torch

from conftest import case_inputs, golden_names, is_dead_bias, load_golden
from floodplanet_code_amd.models import build_model
from floodplanet_code_amd.unet import HipUNet
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOGIT_TOL = 1e-4

BILINEAR_CASES = golden_names()   # includes the two bilinear=False (ConvTranspose2d) fixtures


def build(meta, st):
    net = HipUNet(meta["n_in"], meta["n_classes"], bilinear=meta["bilinear"], base_channels=meta["base"])
    net.load_state_dict(st, strict=True)
    return net.to(DEV)


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.mark.parametrize("name", BILINEAR_CASES)
def test_training_step_matches_reference_fixture(name):
    meta, z = load_golden(name)
    batch, st = case_inputs(meta)
    ii, lr = meta["resolved_ignore_index"], meta["lr"]
    ef = len(meta.get("extras", ())) > 0
    x = O.assemble_input(batch, ef).to(DEV)
    tgt = batch["target"].to(DEV)
    net = build(meta, st)
    net.train()

    # ---- step 1: forward (+logits), loss, backward
    loss, logits = net.loss(x, tgt, ii, return_logits=True)
    loss.backward()
    torch.cuda.synchronize()
    assert np.abs(logits.detach().cpu().numpy() - z["logits1"]).max() <= LOGIT_TOL
    assert abs(loss.item() - z["loss1"].item()) <= 1e-5
    conf = net.pop_confusion().cpu().numpy()
    assert conf.sum() == meta["n_valid"]
    srt = np.sort(z["logits1"], axis=1)
    if (srt[:, -1] - srt[:, -2]).min() > 10 * LOGIT_TOL:   # no near-ties -> argmax must agree exactly
        np.testing.assert_array_equal(conf, z["confusion1"])
    names = meta["names"]
    grads = dict(zip([n for n, _ in net.named_parameters()], [p.grad.detach().cpu() for p in net.parameters()]))
    assert list(grads) == names
    errs = []
    for j, k in enumerate(names):
        if is_dead_bias(k):
            continue
        s = z["grad_stats1"][j]
        g = grads[k].double()
        # (ConvTranspose biases sit in front of conv+BN: nearly dead gradients of ~1e-4, absolute floor applies)
        assert abs(g.norm().item() - s[2]) <= GRAD_TOL * s[2] + 2e-5, (k, g.norm().item(), s[2])
        if f"g1_{j}" in z.files:
            e = rel(grads[k], torch.from_numpy(z[f"g1_{j}"]))
            assert e <= GRAD_TOL or s[2] < 1e-7, k
            if s[2] >= 1e-5:
                errs.append(e)
        else:
            ref = torch.from_numpy(z[f"g1s_{j}"]).double()
            scale = max(ref.norm().item(), s[2] * (64 / max(64, g.numel())) ** 0.5)
            d = (g.reshape(-1)[:64] - ref).norm().item()
            assert d <= GRAD_TOL * scale + 2e-5, k
            if s[2] >= 1e-5:
                errs.append(d / (scale + 1e-30))
    if errs:
        assert float(np.median(errs)) <= GRAD_MEDIAN_TOL, float(np.median(errs))
    # BN running statistics after the first training forward
    bn_keys = __import__("json").loads(bytes(z["bn_keys"]).decode())
    sd = net.state_dict()
    for j, k in enumerate(bn_keys):
        ref = z[f"bn1_{j}"]
        if k.endswith("running_mean"):
            continue  # contains the (noise-driven) conv bias only through init here: checked below with tolerance
        np.testing.assert_allclose(sd[k].cpu().numpy(), ref, rtol=2e-4, atol=1e-6, err_msg=k)
    for j, k in enumerate(bn_keys):
        if k.endswith("running_mean"):
            np.testing.assert_allclose(sd[k].cpu().numpy(), z[f"bn1_{j}"], rtol=1e-4, atol=2e-6, err_msg=k)

    # ---- Adam step 1 (native fused kernel), then step 2
    net.adam_step(lr, 1)
    net.zero_grad(set_to_none=True)
    loss2, logits2 = net.loss(x, tgt, ii, return_logits=True)
    loss2.backward()
    net.adam_step(lr, 2)
    torch.cuda.synchronize()
    assert abs(loss2.item() - z["loss2"].item()) <= 3e-4 * max(1.0, abs(z["loss2"].item()))
    if not meta.get("all_ignored"):
        # the first Adam step moves every weight by +-lr*sign(g): weights whose gradient is rounding noise
        # take opposite steps in two implementations, so step-2 logits agree only to ~lr * fan-in effects
        # (observed max over the fixtures: 0.03-0.05 on 270k logits of the 300x300 case; rms is 100x smaller)
        d2 = logits2.detach().cpu().numpy() - z["logits2"]
        # (rms: <= 5e-3 on the original fixtures; 9.4e-3 on f_full_c4_96_b6, whose three label values leave more weights
        #  with noise-level gradients)
        assert np.abs(d2).max() <= 1e-1 and np.sqrt((d2 ** 2).mean()) <= 1.5e-2, (np.abs(d2).max(), np.sqrt((d2 ** 2).mean()))
    sd = net.state_dict()
    for j, k in enumerate(names):
        if is_dead_bias(k):
            continue
        s = z["param_stats2"][j]
        p = sd[k].cpu()
        # +-lr sign steps on noise-level gradients: allow a third of the elements to sit 2*lr apart
        assert abs(p.double().norm().item() - s[2]) <= 1e-4 * s[2] + 0.7 * lr * p.numel() ** 0.5 + 1e-6, k
        if f"p2_{j}" in z.files:
            d = (p - torch.from_numpy(z[f"p2_{j}"])).abs().max().item()
            # Adam normalises tiny gradients to +-lr steps: elements whose gradient is rounding noise can
            # legitimately take opposite +-lr steps twice; bound by 4.2*lr and require the bulk to agree.
            assert d <= 4.2 * lr, (k, d)
            assert rel(p, torch.from_numpy(z[f"p2_{j}"])) <= 2e-2, k
    # ---- eval-mode forward with the updated running statistics
    net.eval()
    with torch.no_grad():
        ev = net(x)
    torch.cuda.synchronize()
    assert np.abs(ev.cpu().numpy() - z["eval_logits"]).max() <= 2e-2


@pytest.mark.parametrize("name", ["f_full_c8_64_b8", "f_full_c4_96_b6"])
def test_large_batch_full_width_gradients_against_fp64_truth(name):
    """The two larger-batch full-width fixtures (B=8 at 64x64, B=6 at 96x96: 128 / 216 samples per BatchNorm channel at the
    deepest level).  MEASURED (tools/grad_truth_diag.py): a larger batch does NOT make fp32 gradients of this net 1e-3-exact
    -- the reference's own torch-fp32 gradients are 5.2e-3 / 6.2e-3 (median rel-L2 per tensor) from an fp64 run of the same
    graph, the HIP path 6.1e-3 / 2.2e-3.  So the honest tight statement is against the fp64 truth: the HIP gradients must be
    as close to it as the reference arithmetic is (median <= 1.5x, worst tensor <= 2x), and logits / loss exact as ever."""
    meta, z = load_golden(name)
    batch, st = case_inputs(meta)
    ii = meta["resolved_ignore_index"]
    net = build(meta, st).train()
    x, tgt = O.assemble_input(batch, False).to(DEV), batch["target"].to(DEV)
    loss, logits = net.loss(x, tgt, ii, return_logits=True)
    loss.backward()
    torch.cuda.synchronize()
    assert np.abs(logits.detach().cpu().numpy() - z["logits1"]).max() <= LOGIT_TOL
    assert abs(loss.item() - z["loss1"].item()) <= 1e-5
    torch.set_num_threads(min(16, torch.get_num_threads()))
    _, _, g32 = O.loss_and_grads({k: v.clone() for k, v in st.items()}, batch, ii)
    st64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in st.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    _, _, g64 = O.loss_and_grads(st64, b64, ii)
    e_hip, e_ref = [], []
    for k, p in net.named_parameters():
        if is_dead_bias(k):
            continue
        n64 = g64[k].norm().item() + 1e-30
        e_hip.append((p.grad.cpu().double() - g64[k]).norm().item() / n64)
        e_ref.append((g32[k].double() - g64[k]).norm().item() / n64)
    print(f"{name}: HIP vs fp64 median {np.median(e_hip):.2e} max {max(e_hip):.2e}; torch-fp32 vs fp64 median "
          f"{np.median(e_ref):.2e} max {max(e_ref):.2e}")
    assert np.median(e_hip) <= max(1.5 * np.median(e_ref), 2e-3), (np.median(e_hip), np.median(e_ref))
    assert max(e_hip) <= max(2.0 * max(e_ref), 5e-3), (max(e_hip), max(e_ref))


@pytest.mark.parametrize("shape", [(2, 8, 64, 64, 16), (1, 5, 50, 38, 8), (2, 8, 64, 64, 64)])
def test_matches_live_oracle(shape):
    """Same seeded inputs through the oracle on the host CPU (fp32 AND fp64) and through the HIP path:
    the HIP result must be as close to the fp64 truth as the reference arithmetic (torch fp32) is."""
    B, Cc, H, W, base = shape
    st = O.make_state(Cc, 3, base, True, seed=3)
    batch = O.make_batch(B, Cc, H, W, seed=7, n_label_values=3)
    st_o = {k: v.clone() for k, v in st.items()}
    logits_o, loss_o, grads_o = O.loss_and_grads(st_o, batch, 0)
    st64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in st.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    logits64, loss64, grads64 = O.loss_and_grads(st64, b64, 0)
    net = HipUNet(Cc, 3, base_channels=base)
    net.load_state_dict(st)
    net.to(DEV).train()
    loss, logits = net.loss(batch["image"].to(DEV), batch["target"].to(DEV), 0, return_logits=True)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_o.item()) <= 1e-5
    assert (logits.detach().cpu() - logits_o).abs().max().item() <= LOGIT_TOL
    e_ref_logits = (logits_o.double() - logits64).abs().max().item()
    e_hip_logits = (logits.detach().cpu().double() - logits64).abs().max().item()
    assert e_hip_logits <= max(4 * e_ref_logits, 5e-5), (e_hip_logits, e_ref_logits)
    errs_hip, errs_ref = [], []
    for (k, p) in net.named_parameters():
        if is_dead_bias(k):
            continue
        n64 = grads64[k].norm().item() + 1e-30
        e_hip = (p.grad.cpu().double() - grads64[k]).norm().item() / n64
        e_ref = (grads_o[k].double() - grads64[k]).norm().item() / n64
        errs_hip.append(e_hip)
        errs_ref.append(e_ref)
        # single tensors are heavy-tailed (one flipped ReLU / pool decision in a 32-sample BN channel moves a
        # gradient by ~1e-2): bound each loosely, and the median over the 56 live tensors tightly
        assert e_hip <= max(3 * e_ref, 3e-2), (k, e_hip, e_ref)
    med_hip, med_ref = float(np.median(errs_hip)), float(np.median(errs_ref))
    assert med_hip <= max(3 * med_ref, 1e-2), (med_hip, med_ref)
    net.eval()
    with torch.no_grad():
        ev = net(batch["image"].to(DEV)).cpu()
    ev_o = O.eval_forward(st_o, batch)
    assert (ev - ev_o).abs().max().item() <= LOGIT_TOL


def test_autograd_path_equals_fused_path():
    """logits = model(x); torch CE; loss.backward()  ==  fused HIP loss path (same kernels underneath)."""
    st = O.make_state(4, 3, 8, True, seed=1)
    batch = O.make_batch(2, 4, 32, 32, seed=2)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    a = HipUNet(4, 3, base_channels=8); a.load_state_dict(st); a.to(DEV).train()
    b = HipUNet(4, 3, base_channels=8); b.load_state_dict(st); b.to(DEV).train()
    la = torch.nn.functional.cross_entropy(a(x), t, ignore_index=0)
    la.backward()
    lb = b.loss(x, t, 0)
    lb.backward()
    torch.cuda.synchronize()
    assert abs(la.item() - lb.item()) < 1e-6
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if not is_dead_bias(k):
            assert rel(pa.grad, pb.grad) < 1e-5, k


def test_plugin_training_and_validation_steps():
    torch.manual_seed(0)
    m = build_model("ef_model", {"ms_image": 8, "dem": 1}, 3, 1e-3, log_image_iter=50, to_rgb_fcn=None,
                    ignore_index=0, base_channels=8).to(DEV)
    opt = m.configure_optimizers()
    batch = O.make_batch(2, 8, 64, 64, seed=3, extra=("dem",))
    batch = {k: v.to(DEV) for k, v in batch.items()}
    losses = []
    for it in range(8):
        opt.zero_grad()
        loss = m.training_step(batch, it)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]
    assert "train_MulticlassJaccardIndex" in m.logged
    m.validation_step(batch, 0)
    m.validation_epoch_end([0])
    assert 0.0 <= m.logged["val_MulticlassJaccardIndex"].item() <= 1.0
    sd = m.state_dict()
    assert sd["model.inc.double_conv.1.num_batches_tracked"].item() == 8


def test_all_ignored_batch_gives_zero_loss_and_exact_zero_grads():
    meta, z = load_golden("s_all_ignored")
    batch, st = case_inputs(meta)
    net = build(meta, st).train()
    loss = net.train_step(batch["image"].to(DEV), batch["target"].to(DEV), meta["resolved_ignore_index"])
    torch.cuda.synchronize()
    assert loss.item() == 0.0
    assert float(net.flat_grads().abs().max()) == 0.0
    assert not bool(torch.isnan(net.flat_grads()).any())


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16"])
def test_full_size_properties(prec):
    """BASELINE configs[1] exactly as bench.py runs it (B=16, 8ch, 256x256, full width; bf16 = the benched dispatch: tall
    conv tile, ping-pong wgrad, weight-gradient chain on the side stream): properties that need no CPU reference.
    (a) determinism: two runs give bit-identical loss and gradients (fixed-order reductions, no atomics) -- with the side
        stream ON in bf16;
    (b) eval-mode logits of a sub-batch do not depend on the rest of the batch;
    (c) the gradient of an all-ignored batch is exactly zero;
    (d) bf16 only: the default dispatch equals the conservative one (square conv tiles, lock-step wgrad kernel, everything
        on one stream) to bf16 rounding -- a wrong kernel variant on one layer is off by O(1) on that layer's gradient."""
    torch.manual_seed(0)
    net = HipUNet(8, 3, precision=prec).to(DEV).train()
    batch = O.make_batch(16, 8, 256, 256, seed=11)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    l1 = net.train_step(x, t, 0).item()
    g1 = net.flat_grads().clone()
    rm1 = net.state_dict()["inc.double_conv.1.running_mean"].clone()
    l2 = net.train_step(x, t, 0).item()
    torch.cuda.synchronize()
    assert l1 == l2 and torch.equal(g1, net.flat_grads())
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert not torch.equal(rm1, net.state_dict()["inc.double_conv.1.running_mean"])  # momentum update happened
    if prec in ("bf16", "fp16"):
        from floodplanet_code_amd import _lib
        lib = _lib.load()
        try:
            lib.fu_test_conv_tile_mode(1)
            lib.fu_test_force_lockstep_wgrad(1)
            _lib.check(lib.fu_set_side_stream(net._ctx, 0))
            l3 = net.train_step(x, t, 0).item()
            g3 = net.flat_grads().clone()
            torch.cuda.synchronize()
        finally:
            lib.fu_test_conv_tile_mode(0)
            lib.fu_test_force_lockstep_wgrad(0)
            _lib.check(lib.fu_set_side_stream(net._ctx, 1))
        # Tolerances are MEASURED decorrelation, not slack (tools/dispatch_diag*.py): one layer run on
        # the two tile shapes differs in 1e-4 of its bf16 outputs by one ulp (fp32 summation order), but those flips
        # re-round everything downstream -- after ~4 layers the two runs are independent bf16 rounding realisations:
        # logits 1.5 % apart, and the backward adds ~1.7 % per layer (0.0001 outc, 0.017 up4.3, 0.04 up4.0 ... 0.20-0.25
        # at the encoder end), while side stream on/off and ping-pong/lock-step wgrad are bit-identical.  A wrong kernel
        # on any layer moves that layer's gradient (and everything upstream of a wrong dgrad) by O(1).
        assert abs(l3 - l1) <= 1e-4 * max(1.0, abs(l1)), (l1, l3)
        tol = {"outc.conv.weight": 2e-3, "up4.conv.double_conv.3.weight": 0.05, "up4.conv.double_conv.0.weight": 0.10}
        worst = 0.0
        for (k, p), gv1, gv3 in zip(net.named_parameters(), _views(net, g1), _views(net, g3)):
            if is_dead_bias(k) or gv1.norm().item() < 1e-7 or p.dim() != 4:
                continue
            worst = max(worst, rel(gv3, gv1))
            assert rel(gv3, gv1) <= tol.get(k, 0.35), (k, rel(gv3, gv1))
        assert 0.0 < worst
        # the two switches that must not change a bit: side stream off, lock-step wgrad kernel (default conv tiles)
        try:
            lib.fu_test_force_lockstep_wgrad(1)
            _lib.check(lib.fu_set_side_stream(net._ctx, 0))
            l4 = net.train_step(x, t, 0).item()
            torch.cuda.synchronize()
            assert l4 == l1
            # (one exception since round 4: the 8-band first conv's weight gradient has its own kernel, k_wgrad_bf16_c8, which
            #  the lock-step switch turns off -- the same exact products in another fp32 summation order)
            for (k, p), gv1, gv4 in zip(net.named_parameters(), _views(net, g1), _views(net, net.flat_grads())):
                if k == "inc.double_conv.0.weight":
                    assert rel(gv4, gv1) <= 1e-5, (k, rel(gv4, gv1))
                else:
                    assert torch.equal(gv4, gv1), k
        finally:
            lib.fu_test_force_lockstep_wgrad(0)
            _lib.check(lib.fu_set_side_stream(net._ctx, 1))
        # BatchNorm-backward sums from the producer of the gradient (row-stationary dgrad epilogue, head backward) against
        # the separate reduce pass.  The head's sums differ only in summation order and in using g before its rounding
        # (dgamma / dbeta of the last BatchNorm: 1e-4); one layer further the changed coefficients have re-rounded dy
        # (measured decorrelation as above), a wrong sum would be off by O(1) there and in everything upstream.
        try:
            lib.fu_test_bnb_separate(1)
            l5 = net.train_step(x, t, 0).item()
            g5 = net.flat_grads().clone()
            torch.cuda.synchronize()
        finally:
            lib.fu_test_bnb_separate(0)
        assert l5 == l1
        tight = {"up4.conv.double_conv.4.weight": 1e-4, "up4.conv.double_conv.4.bias": 1e-4,
                 "up4.conv.double_conv.1.weight": 0.03, "up4.conv.double_conv.1.bias": 0.03,
                 "outc.conv.weight": 0.0, "outc.conv.bias": 0.0, "up4.conv.double_conv.3.weight": 1e-3}
        seen = 0
        for (k, p), gv1, gv5 in zip(net.named_parameters(), _views(net, g1), _views(net, g5)):
            if is_dead_bias(k) or gv1.norm().item() < 1e-7:
                continue
            if k in tight:
                seen += 1
                assert rel(gv5, gv1) <= tight[k], (k, rel(gv5, gv1))
            elif p.dim() == 4:
                assert rel(gv5, gv1) <= 0.35, (k, rel(gv5, gv1))
        assert seen == len(tight)
        # the two routes are really different code -- unless the environment has forced the alternate route on BOTH runs
        # (FU_BNB_SEPARATE=1 / FU_HEAD_STORE_G=1 switch the library's dispatch globally: round 3's `r3_alt1` / `r3_alt2` runs
        # of the whole GPU suite under those two settings; then the comparison is of one route with itself)
        import os
        if os.environ.get("FU_BNB_SEPARATE") == "1" or os.environ.get("FU_HEAD_STORE_G") == "1":
            print("routes forced equal by the environment: 'routes differ' assertion skipped")
        else:
            assert not torch.equal(g5, g1)
    net.eval()
    with torch.no_grad():
        full = net(x)
        part = net(x[3:5])
    if prec == "fp32":
        assert torch.equal(full[3:5], part)
    else:
        # 16-bit modes: the conv tile heuristic depends on the batch (the tall tile, with its 16-channel K chunks, runs only
        # where it yields >= 2048 workgroups), and another K-summation order re-rounds everything downstream: close, not
        # identical.  With one tile family for both batches (32- and 64-channel tiles share the K order) it IS identical.
        assert (full[3:5] - part).abs().max().item() <= (0.1 if prec == "bf16" else 0.02)
        try:
            lib.fu_test_conv_tile_mode(1)
            with torch.no_grad():
                full1 = net(x)
                part1 = net(x[3:5])
        finally:
            lib.fu_test_conv_tile_mode(0)
        assert torch.equal(full1[3:5], part1)
    net.train()
    lz = net.train_step(x, torch.zeros_like(t), 0).item()
    assert lz == 0.0 and float(net.flat_grads().abs().max()) == 0.0


def _views(net, flat):
    return [flat[off:off + n].view(p.shape) for _, p, off, n in net._table]


def test_fused_adam_kernel_equals_torch_adam_on_fixture_gradients():
    """fu_adam_step alone: load the reference fixture's step-1 gradients into the flat gradient buffer, run the fused
    kernel twice, compare with the oracle's torch.optim.Adam restatement (pinned bit-exactly against torch.optim.Adam by
    make_golden.py) element by element.  Every operation of k_adam rounds like ATen's, so the bound is 1e-7 relative
    (observed: bit-identical); the 18 dead biases are INCLUDED here -- given the same gradient the update is the same."""
    meta, z = load_golden("s_b2c4_32")
    batch, st = case_inputs(meta)
    lr = 1e-3
    net = build(meta, st).train()
    x, t = O.assemble_input(batch, False).to(DEV), batch["target"].to(DEV)
    net.train_step(x, t, meta["resolved_ignore_index"])          # creates the context and binds the buffers
    names = meta["names"]
    grads = {}
    for j, k in enumerate(names):
        assert f"g1_{j}" in z.files
        grads[k] = torch.from_numpy(z[f"g1_{j}"]).clone()
    st_o = {k: v.clone() for k, v in st.items()}
    opt = O.new_adam_state(st_o)
    net.load_state_dict(st)                                       # undo nothing: parameters are still the initial ones
    for step in (1, 2, 3):
        for (k, p, off, n) in net._table:
            net.flat_grads()[off:off + n].copy_(grads[k].reshape(-1).to(DEV))
        net.adam_step(lr, step)
        O.adam_update(st_o, grads, opt, lr)
        grads = {k: g * 0.5 + 0.01 * st_o[k] for k, g in grads.items()}     # new gradients for the next step
    torch.cuda.synchronize()
    m, v = net.adam_state()
    for (k, p, off, n) in net._table:
        ref = st_o[k]
        d = (p.detach().cpu() - ref).abs().max().item()
        assert d <= 1e-7 * max(1.0, ref.abs().max().item()), (k, d)
        # (ATen's CPU lerp is an fmadd in its vector body and a plain expression in the scalar tail: one-ulp differences of
        #  the first moment on a few elements are the CPU's own inconsistency, hence the absolute floor)
        ms = opt["m"][k].abs().max().item()
        assert torch.allclose(m[off:off + n].cpu().view(p.shape), opt["m"][k], rtol=1e-6, atol=1e-6 * ms + 1e-30), k
        assert torch.allclose(v[off:off + n].cpu().view(p.shape), opt["v"][k], rtol=1e-6, atol=1e-30), k


def test_adam_state_survives_context_recreation():
    """The Adam moments are caller-owned flat buffers (ABI 3): a validation pass at another tile size, a larger batch or
    a second .to(device) re-creates the device context but must not reset the optimiser.  Interrupted run == plain run."""
    st = O.make_state(4, 3, 8, True, seed=2)
    b_small = O.make_batch(2, 4, 32, 32, seed=3)
    b_big = O.make_batch(4, 4, 32, 32, seed=4)
    b_val = O.make_batch(1, 4, 48, 48, seed=5)

    def run(interrupt):
        net = HipUNet(4, 3, base_channels=8)
        net.load_state_dict(st)
        net.to(DEV).train()
        step = 0
        for it in range(3):
            step += 1
            net.train_step(b_small["image"].to(DEV), b_small["target"].to(DEV), 0)
            net.adam_step(1e-3, step)
        if interrupt:
            net.eval()
            with torch.no_grad():
                net(b_val["image"].to(DEV))            # other H, W: the context is re-created
            net.train()
            net.to(DEV)                                 # no-op move: flat buffers are rebuilt
        for it in range(2):
            step += 1
            net.train_step(b_big["image"].to(DEV), b_big["target"].to(DEV), 0)   # batch 4 > max_batch 2: re-created
            net.adam_step(1e-3, step)
        torch.cuda.synchronize()
        return net.flat_parameters().clone(), net.adam_state()[0].clone()

    p_plain, m_plain = run(False)
    p_int, m_int = run(True)
    assert m_plain.abs().max() > 0
    assert torch.equal(m_plain, m_int)
    assert torch.equal(p_plain, p_int)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_captured_hipgraph_step_equals_eager_step(prec):
    """DataParallelTrainer(graph=True): forward + CE + backward (with its side-stream fork / join) + Adam captured once
    into a hipGraph and replayed, the Adam scalars refreshed per step from device memory.  Same kernels, same order, same
    arguments: parameters, Adam moments, BatchNorm buffers and the loss are bit-identical to the eager trainer's, also when
    the batch arrives in other tensors than the captured ones (copied into the captured buffers)."""
    from floodplanet_code_amd.distributed import DataParallelTrainer
    st = O.make_state(8, 3, 32, True, seed=4)
    batches = [O.make_batch(4, 8, 64, 64, seed=40 + i) for i in range(3)]
    outs = []
    for graph in (False, True):
        net = HipUNet(8, 3, base_channels=32, precision=prec)
        net.load_state_dict(st)
        net.to(DEV).train()
        tr = DataParallelTrainer(net, lr=1e-3, graph=graph)
        losses = []
        for it in range(6):
            b = batches[it % 3]
            losses.append(tr.step(b["image"].to(DEV), b["target"].to(DEV), 0).item())    # fresh tensors every step
        torch.cuda.synchronize()
        assert (tr._graph is not None) == graph
        outs.append((losses, net.flat_parameters().clone(), net.adam_state()[0].clone(), net._flat_rm.clone(),
                     net._flat_nbt.clone(), net.flat_grads().clone()))
    (la, pa, ma, ra, na, ga), (lb, pb, mb, rb, nb_, gb) = outs
    assert la == lb
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(ra, rb) and torch.equal(na, nb_) and torch.equal(ga, gb)
    assert int(na[0]) == 6


def test_captured_step_keeps_callers_batches_and_unsynchronised_scalars():
    """(1) the captured step owns private input buffers -- a retained list of device batches cycled through
    graph=True must stay unmodified and train exactly as the eager trainer does (the first version captured the caller's first
    batch in place: later steps overwrote it, and skipped the copy when that tensor came round again); (2) no host
    synchronisation between steps: the Adam scalars of step k must not be overwritten by later steps before their copy has
    executed (a ring of pinned slots guarded by events); (3) the returned losses are tensors of their own (collected and
    read at the end, as a logger does)."""
    from floodplanet_code_amd.distributed import DataParallelTrainer
    st = O.make_state(8, 3, 16, True, seed=7)
    host = [O.make_batch(2, 8, 64, 64, seed=70 + i) for i in range(3)]
    outs = []
    for graph in (False, True):
        net = HipUNet(8, 3, base_channels=16, precision="bf16")
        net.load_state_dict(st)
        net.to(DEV).train()
        tr = DataParallelTrainer(net, lr=1e-2, graph=graph)
        dev_batches = [(b["image"].to(DEV).contiguous().float(), b["target"].to(DEV).contiguous().long()) for b in host]
        keep = [(x.clone(), t.clone()) for x, t in dev_batches]
        losses = []
        for it in range(12):                       # 12 steps > the 8 slots of the scalar ring, no .item() in between
            x, t = dev_batches[it % 3]
            losses.append(tr.step(x, t, 0))
        torch.cuda.synchronize()
        assert (tr._graph is not None) == graph
        for (x, t), (kx, kt) in zip(dev_batches, keep):
            assert torch.equal(x, kx) and torch.equal(t, kt)          # the caller's batches are untouched
        outs.append(([l.item() for l in losses], net.flat_parameters().clone(), net.adam_state()[0].clone()))
    (la, pa, ma), (lb, pb, mb) = outs
    assert len(set(lb)) > 1                         # twelve different losses, not twelve views of the last one
    assert la == lb
    assert torch.equal(pa, pb) and torch.equal(ma, mb)


def test_backward_of_a_stale_forward_raises():
    """One forward in flight per module (the saved activations live in the single device context): backward() of a graph
    whose forward is not the latest training forward must raise instead of using the newer activations."""
    st = O.make_state(4, 3, 8, True, seed=1)
    batch = O.make_batch(2, 4, 32, 32, seed=2)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    net = HipUNet(4, 3, base_channels=8)
    net.load_state_dict(st)
    net.to(DEV).train()
    l1 = net.loss(x, t, 0)
    l2 = net.loss(x * 0.5, t, 0)
    with pytest.raises(RuntimeError, match="no longer the latest training forward"):
        l1.backward()
    l2.backward()
    assert all(p.grad is not None for p in net.parameters())


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16"])
def test_repeated_backward_of_one_loss_gives_the_same_gradients(prec):
    """fu_backward twice after one fu_loss_ce (retain_graph, or fu_backward_block(0) called again): the stored loss gradient is
    never modified, so the second backward reproduces the first bit for bit -- in fp16 too, where the first backward used to
    leave the loss scale inside the stored gradient and the second one then picked scale 1 -- and the upstream factor of
    fu_scale_loss_grad is applied once, not compounded."""
    from floodplanet_code_amd import _lib
    lib = _lib.load()
    st = O.make_state(8, 3, 16, True, seed=1)
    batch = O.make_batch(2, 8, 64, 64, seed=2)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    net = HipUNet(8, 3, base_channels=16, precision=prec)
    net.load_state_dict(st)
    net.to(DEV).train()
    net._forward_raw(x, True, want_logits=False)
    net._loss_raw(t, 0, x.device)
    net._backward_raw(None, x.device)
    g_plain = net.flat_grads().clone()
    four = torch.tensor(4.0, device=DEV)
    for _ in range(2):                                  # setting the factor twice replaces it
        _lib.check(lib.fu_scale_loss_grad(net._ctx, four.data_ptr(), net._stream(x.device)))
    net._backward_raw(None, x.device)
    g1 = net.flat_grads().clone()
    net._backward_raw(None, x.device)
    g2 = net.flat_grads().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert torch.equal(g1, g2)
    # a power-of-two factor commutes with every rounding of the (linear) backward, and fp16's device-side loss scale comes
    # out four times smaller for the four times larger gradient: the result is the plain gradient times four
    assert rel(g1, 4.0 * g_plain) <= 1e-6


def test_fp16_overflowing_step_is_skipped_and_the_loss_scale_backs_off():
    """fp16 mode: the device-side loss scale follows max|dL/dlogits| (scaled into [32, 64)); what the chain multiplies on top
    can still push a gradient map past 65504 -- here OutConv weights of 3e3, so that the very first fp16 gradient map
    (dlogits x W) overflows.  The inf / NaN then sits in the gradient buffer: that step's Adam launch must leave
    parameters and moments untouched, the next backward scales the loss down further, and after a few skipped steps the
    updates go through again with finite values everywhere (what a GradScaler does, on the device)."""
    st = O.make_state(8, 3, 16, True, seed=1)
    st["outc.conv.weight"][:] = st["outc.conv.weight"].sign() * 3.0e3
    batch = O.make_batch(2, 8, 64, 64, seed=2)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    net = HipUNet(8, 3, base_channels=16, precision="fp16")
    net.load_state_dict(st)
    net.to(DEV).train()
    net.train_step(x, t, 0)
    torch.cuda.synchronize()
    assert not torch.isfinite(net.flat_grads()).all()             # the scenario really overflows
    p0, m0 = net.flat_parameters().clone(), net.adam_state()[0].clone()
    net.adam_step(1e-3, 1)
    torch.cuda.synchronize()
    assert torch.equal(net.flat_parameters(), p0) and torch.equal(net.adam_state()[0], m0)      # skipped: nothing moved
    assert net.fp16_guard_state() == (1, 1)
    applied_at = None
    for step in range(2, 14):
        net.train_step(x, t, 0)
        finite = bool(torch.isfinite(net.flat_grads()).all())
        before = net.flat_parameters().clone()
        net.adam_step(1e-3, step)
        torch.cuda.synchronize()
        moved = not torch.equal(net.flat_parameters(), before)
        assert moved == finite                                    # a step is applied exactly when its gradients are finite
        if moved and applied_at is None:
            applied_at = step
    skipped, backoff = net.fp16_guard_state()
    assert applied_at is not None and 1 <= skipped < 12 and backoff >= 1, (applied_at, skipped, backoff)
    assert torch.isfinite(net.flat_parameters()).all() and torch.isfinite(net.adam_state()[0]).all()
    # bf16 / fp32 contexts: no guard, nothing to report
    n32 = HipUNet(8, 3, base_channels=16)
    n32.load_state_dict(st)
    n32.to(DEV).train()
    n32.train_step(x, t, 0)
    n32.adam_step(1e-3, 1)
    assert n32.fp16_guard_state() == (0, 0)


def test_plugin_path_uses_fused_adam_and_keeps_torch_semantics():
    """configure_optimizers() returns HipAdam (one fused launch); the Lightning loop zero_grad / training_step / backward /
    step gives the same parameters as torch.optim.Adam driven through the same loop (FU_TORCH_ADAM=1), the upstream
    gradient of loss.backward() is honoured (loss * 3), accumulation over two backward calls adds up, and the optimiser
    state round-trips through state_dict like torch.optim.Adam's."""
    import os
    batch = O.make_batch(2, 8, 64, 64, seed=3)
    batch = {k: v.to(DEV) for k, v in batch.items()}

    def make(torch_adam):
        torch.manual_seed(0)
        m = build_model("ms_model", {"ms_image": 8}, 3, 1e-3, log_image_iter=50, to_rgb_fcn=None, ignore_index=0,
                        base_channels=8).to(DEV)
        os.environ["FU_TORCH_ADAM"] = "1" if torch_adam else "0"
        try:
            opt = m.configure_optimizers()
        finally:
            os.environ.pop("FU_TORCH_ADAM")
        return m, opt

    ma, oa = make(False)
    mb, ob = make(True)
    from floodplanet_code_amd.unet import HipAdam
    assert isinstance(oa, HipAdam) and isinstance(ob, torch.optim.Adam)
    for it in range(4):
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad()
            loss = m.training_step(batch, it)
            loss.backward()
            o.step()
    torch.cuda.synchronize()
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        if not is_dead_bias(k.replace("model.", "", 1)):
            # same gradients, same update rule; torch's GPU (foreach) Adam groups the scalar factors differently, and Adam
            # turns a last-bit difference of a noise-level gradient into a visible fraction of its +-lr step: bound the
            # worst element by 2 % of the 4 steps' travel, and the mean deviation by 0.1 % of it
            assert (pa - pb).abs().max().item() <= 0.02 * 4 * 1e-3, k
            assert (pa - pb).abs().mean().item() <= 0.001 * 4 * 1e-3, k
    # upstream gradient and accumulation over two DIFFERENT micro-batches (p.grad aliases the flat gradient buffer after
    # the first backward; the second backward overwrites that buffer and must add the first gradient back)
    batch_b = {k: v.to(DEV) for k, v in O.make_batch(2, 8, 64, 64, seed=17).items()}
    oa.zero_grad()
    (ma.training_step(batch, 0) * 3.0).backward()
    g3 = [p.grad.clone() for p in ma.parameters()]
    oa.zero_grad()
    ma.training_step(batch, 0).backward()
    g1 = [p.grad.clone() for p in ma.parameters()]
    oa.zero_grad()
    ma.training_step(batch_b, 0).backward()
    gb = [p.grad.clone() for p in ma.parameters()]
    assert any(rel(b, a) > 1e-2 for a, b in zip(g1, gb))          # really different gradients
    oa.zero_grad()
    ma.training_step(batch, 0).backward()
    ma.training_step(batch_b, 0).backward()          # p.grad present (aliasing the flat buffer) -> accumulate
    g2 = [p.grad.clone() for p in ma.parameters()]
    oa.zero_grad(set_to_none=False)                  # zeroed in place: the next backward accumulates into zeros
    ma.training_step(batch_b, 0).backward()
    gz = [p.grad.clone() for p in ma.parameters()]
    ext = [torch.full_like(p, 0.25) for p in ma.parameters()]    # user-owned .grad tensors outside the flat buffer
    for p, e in zip(ma.parameters(), ext):
        p.grad = e
    ma.training_step(batch, 0).backward()
    ge = [p.grad.clone() for p in ma.parameters()]
    # (BatchNorm running statistics moved between the calls; the gradients do not depend on them in train mode)
    for (k, _), a, b, c, d, z, e in zip(ma.named_parameters(), g1, g2, g3, gb, gz, ge):
        if is_dead_bias(k.replace("model.", "", 1)) or a.norm().item() < 1e-7:
            continue
        assert rel(b, a + d) <= 1e-5, k
        assert rel(c, 3 * a) <= 1e-5, k
        assert rel(z, d) <= 1e-6, k
        assert rel(e, a + 0.25) <= 1e-5, k
    oa.step()                                         # gradients accumulated outside the flat buffer are copied home
    assert all(rel(v, e) <= 1e-6 for v, e in zip(ma.model.grad_views(), ge))
    sd = oa.state_dict()
    assert len(sd["state"]) == len(list(ma.parameters())) and sd["param_groups"][0]["lr"] == 1e-3
    m2, o2 = make(False)
    m2.load_state_dict(ma.state_dict())
    m2.training_step(batch, 0)                        # creates the flat buffers on the device
    o2.load_state_dict(sd)
    assert torch.equal(m2.model.adam_state()[0], ma.model.adam_state()[0])


def test_miou_vs_ref_full_width_training_heldout_tiles():
    """`mIoU vs ref` (BASELINE.json metric): the oracle -- the reference's arithmetic -- trained the FULL-WIDTH
    net for 100 Adam steps on 32 seeded 8-band 128x128 tiles of a task that has to be learnt and scored 16 held-out tiles
    (tests/golden/miou_golden.json, oracle/make_miou_golden.py).  HIP fp32, bf16 and fp16 train from the same state on the
    same tiles.  Two statements:
      * while two correct implementations still walk the same trajectory (the first 40 steps) the per-step training loss
        follows the reference curve: fp32 within 1 % (measured 0.35 %), the 16-bit modes within 8 % at the worst step and
        2 % on average (measured 5 % / 1 %) -- a systematic error anywhere in forward, loss, backward, BatchNorm statistics
        or Adam shows here within a few steps;
      * the micro Jaccard of the eval-mode predictions on the held-out tiles matches the reference path's: the end point
        of 100 steps is a chaotic function of rounding (tools/miou_diag.py: HIP fp32 tracks the reference to 0.3 % for 50
        steps of a constant-rate run and parts from it afterwards like any second run of the reference would; the
        workload therefore lowers the rate for its last 40 steps, after which correct runs agree to 0.003).  The held-out score sits well below 1: an implementation that trains
        worse -- a wrong gradient, a lossy optimiser, broken running statistics in eval mode -- shows up here."""
    import bench
    r = bench.miou_vs_ref(torch.device(DEV), "bf16", precisions=["fp32", "bf16", "fp16"])
    print(r["jaccard_heldout"], r["jaccard_train"], r["final_train_loss"], r["loss_curve_dev_first_40_steps"])
    ref = r["jaccard_heldout"]["oracle_fp32"]
    assert 0.6 < ref < 0.97                          # the task was learnt, and not saturated
    t = r["loss_curve_dev_first_40_steps"]
    assert t["hip_fp32"]["max_abs_dev"] <= 0.01, t
    for k in ("hip_bf16", "hip_fp16"):
        assert t[k]["max_abs_dev"] <= 0.08 and t[k]["mean_abs_dev"] <= 0.02, t
    assert abs(r["gap_vs_ref"]["hip_fp32"]) <= 0.005, r          # measured: -0.0026 (bf16 -0.0002, fp16 0.0000)
    assert abs(r["gap_vs_ref"]["hip_bf16"]) <= 0.02, r
    assert abs(r["gap_vs_ref"]["hip_fp16"]) <= 0.02, r
    for k in ("hip_fp32", "hip_bf16", "hip_fp16"):
        assert abs(r["final_train_loss"][k] - r["final_train_loss"]["oracle_fp32"]) <= 0.2 * r["final_train_loss"]["oracle_fp32"], r


# ---------------------------------------------------------------------------------------------------
# bf16 mode (bf16 activations / weights on the matrix cores, fp32 accumulation, statistics, loss, master weights).
# Stated tolerance against the fp32 reference: the reference's own bf16-autocast run deviates from its fp32 run
# by max 0.124 / rms 0.022 on the logits, so: logits max |d| <= 0.25, rms <= 0.05, loss |d| <= 0.03,
# >= 93 % argmax agreement; training behaviour is checked by trajectory, not elementwise.
# ---------------------------------------------------------------------------------------------------
# fp16 mode (BASELINE configs[3], FU_F16): 11 significant bits instead of 8 -> its own, TIGHTER stated tolerance, from the
# same measurement (tools/lowp_diag.py on the four fixtures: logits max 0.015-0.043 / rms 0.0029-0.0035 against bf16's
# 0.12-0.23 / 0.023-0.027; argmax agreement 0.997-1.0 against 0.972-0.987): logits max |d| <= 0.06, rms <= 0.008,
# loss |d| <= 0.005, >= 99 % argmax agreement, median gradient cosine >= 0.95 (bf16: 0.9).
LOWP_TOL = {"bf16": dict(lmax=0.25, lrms=0.05, loss=0.03, agree=0.93, cos=0.9),
            "fp16": dict(lmax=0.06, lrms=0.008, loss=0.005, agree=0.99, cos=0.95)}


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["m_base8_64", "f_full_c8_64_b2", "m_base16_300"])
def test_bf16_step_within_stated_tolerance_of_fp32_reference(name, prec):
    meta, z = load_golden(name)
    batch, st = case_inputs(meta)
    ii = meta["resolved_ignore_index"]
    tol = LOWP_TOL[prec]
    net = HipUNet(meta["n_in"], 3, base_channels=meta["base"], precision=prec)
    net.load_state_dict(st)
    net.to(DEV).train()
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    loss, logits = net.loss(x, t, ii, return_logits=True)
    loss.backward()
    torch.cuda.synchronize()
    d = logits.detach().cpu().numpy() - z["logits1"]
    assert np.abs(d).max() <= tol["lmax"], np.abs(d).max()
    assert np.sqrt((d ** 2).mean()) <= tol["lrms"]
    assert abs(loss.item() - z["loss1"].item()) <= tol["loss"]
    agree = (logits.detach().cpu().numpy().argmax(1) == z["logits1"].argmax(1)).mean()
    assert agree >= tol["agree"], agree
    g = net.flat_grads()
    assert torch.isfinite(g).all() and g.abs().max() > 0
    cos = []
    for j, (k, p) in enumerate(net.named_parameters()):
        if is_dead_bias(k):
            continue
        # the gradient buffer holds TRUE gradients in every mode (fp16's internal loss scale is removed where parameter
        # gradients are written): the norms must match the fp32 reference's, whatever the rounding noise does to directions
        ref_norm = z["grad_stats1"][j][2]
        if ref_norm >= 1e-5:
            assert 0.5 * ref_norm <= p.grad.norm().item() <= 2.0 * ref_norm, (k, p.grad.norm().item(), ref_norm)
        if f"g1_{j}" in z.files and p.numel() >= 64:
            a, b = p.grad.cpu().double().reshape(-1), torch.from_numpy(z[f"g1_{j}"]).double().reshape(-1)
            cos.append((a @ b / (a.norm() * b.norm() + 1e-30)).item())
    if cos:
        assert np.median(cos) >= tol["cos"], np.median(cos)


def test_bf16_training_trajectory_tracks_fp32():
    st = O.make_state(8, 3, 16, True, seed=5, nontrivial_bn=False)
    batch = O.make_batch(4, 8, 64, 64, seed=9)
    x, t = batch["image"].to(DEV), batch["target"].to(DEV)
    finals = {}
    for prec in ("fp32", "bf16", "fp16"):
        net = HipUNet(8, 3, base_channels=16, precision=prec)
        net.load_state_dict(st)
        net.to(DEV).train()
        losses = []
        for step in range(1, 41):
            losses.append(net.train_step(x, t, 0).item())
            net.adam_step(1e-3, step)
        net.eval()
        with torch.no_grad():
            pred = net(x).argmax(1)
        valid = t != 0
        finals[prec] = (losses, (pred[valid] == t[valid]).float().mean().item())
    l32, a32 = finals["fp32"]
    for prec in ("bf16", "fp16"):
        l16, a16 = finals[prec]
        assert l32[-1] < 0.5 * l32[0] and l16[-1] < 0.5 * l16[0]
        assert abs(l16[-1] - l32[-1]) <= 0.15 * l32[0], prec
        assert abs(a16 - a32) <= 0.05, prec


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("bf16", 3e-2)])
def test_bce_dice_extension_matches_its_oracle(prec, tol):
    """North-star extension (no reference counterpart): BCE + soft Dice, fp32 wave-shuffle reductions.
    Checked against oracle.bce_dice_loss through the whole net (loss value, gradients)."""
    st = O.make_state(8, 3, 16, True, seed=2)
    batch = O.make_batch(2, 8, 64, 64, seed=4, n_label_values=3)
    st_o = {k: v.clone() for k, v in st.items()}
    _, loss_o, grads_o = O.loss_and_grads(st_o, batch, 0, loss_kind="bce_dice", dice_weight=0.7)
    net = HipUNet(8, 3, base_channels=16, precision=prec)
    net.load_state_dict(st)
    net.to(DEV).train()
    loss = net.loss(batch["image"].to(DEV), batch["target"].to(DEV), 0, kind="bce_dice", dice_weight=0.7)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_o.item()) <= tol * 3
    if prec == "fp32":
        for k, p in net.named_parameters():
            if not is_dead_bias(k):
                assert rel(p.grad.cpu(), grads_o[k]) <= GRAD_TOL, k
    # all-ignored -> exactly zero loss and gradients
    lz = net.train_step(batch["image"].to(DEV), torch.zeros_like(batch["target"]).to(DEV), 0, kind="bce_dice")
    assert lz.item() == 0.0 and float(net.flat_grads().abs().max()) == 0.0


def test_minimal_trainer_runs_the_lightning_protocol(tmp_path):
    """BASELINE configs[0] plumbing (registry -> plugin -> training/validation hooks -> Adam -> top-k checkpoints)
    on synthetic 4-band tiles, batch 2, through floodplanet_code_amd.fit.fit_model (the loop fit.py hands to Lightning)."""
    from floodplanet_code_amd.fit import SyntheticTiles, fit_model
    from floodplanet_code_amd.models import WaterSegmentationModel
    ch = {"ms_image": 4}
    train = SyntheticTiles(6, 2, ch, 64, 64, DEV, seed=1)
    valid = SyntheticTiles(2, 2, ch, 64, 64, DEV, seed=2)
    cfg = dict(lr=2e-3, n_epochs=3, save_topk_models=2, ignore_index=0,
               model=dict(name="ms_model", model_kwargs=dict(optimizer_name="adam", base_channels=8)))
    best = fit_model(cfg, train, valid, ch, 3, exp_dir=str(tmp_path), device=DEV)
    import os
    files = sorted(os.listdir(os.path.join(tmp_path, "checkpoints")))
    assert len(files) == 2 and best.endswith(".ckpt") and "val_MulticlassJaccardIndex=" in best
    hist = fit_model.last_model.history
    assert hist[-1]["train_loss"] < hist[0]["train_loss"]
    # the checkpoint reloads through the reference-style classmethod and reproduces the eval logits
    m2 = WaterSegmentationModel.load_from_checkpoint(best, in_channels=ch, n_classes=3, lr=2e-3, ignore_index=0,
                                                     base_channels=8).to(DEV)
    m2._set_model_to_eval()
    batch = next(iter(valid))
    with torch.no_grad():
        out = m2(batch)
    assert out.shape == (2, 3, 64, 64) and torch.isfinite(out).all()


@pytest.mark.parametrize("name", ["stitch_overlap_96x112", "stitch_partial_120x100"])
def test_gpu_stitching_matches_reference_stitcher_fixture(name):
    """Eval forward + overlap-average stitching (fu_stitch_add / fu_stitch_finalize on the logits resident after the
    forward) against canvases produced by the reference's OWN ImageStitcher_v2 (utils_image.py:363-494, run by
    oracle/make_stitch_golden.py exactly as predict.py:296-334 drives it), incl. crops cut at the raster's edge.
    The HIP logits are within 1e-4 of the reference network's, so the averaged probabilities are within 1e-4 too."""
    import json, os
    from conftest import GOLDEN
    from floodplanet_code_amd.stitch import GpuImageStitcher
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    st = O.make_state(meta["C"], 3, meta["base"], True, seed=meta["param_seed"])
    net = HipUNet(meta["C"], 3, base_channels=meta["base"])
    net.load_state_dict(st)
    net.to(DEV).eval()
    H, W, S = meta["H"], meta["W"], meta["S"]
    big = torch.from_numpy(O.hash_uniform(meta["C"] * H * W, meta["data_seed"], 77).astype(np.float32)
                           .reshape(meta["C"], H, W))
    x = torch.zeros(len(meta["boxes"]), meta["C"], S, S)
    for i, (h0, w0, hE, wE) in enumerate(meta["boxes"]):
        x[i, :, :hE - h0, :wE - w0] = big[:, h0:hE, w0:wE]
    with torch.no_grad():
        logits = net(x.to(DEV))
    assert np.abs(logits.cpu().numpy() - z["logits"]).max() <= LOGIT_TOL
    stitch = GpuImageStitcher(net, DEV)
    for i, box in enumerate(meta["boxes"]):
        stitch.add_image(i, "img", tuple(box), H, W)
    np.testing.assert_array_equal(stitch.weight_canvas["img"].cpu().numpy(), z["weight"])
    got, am = stitch.combine("img")
    torch.cuda.synchronize()
    np.testing.assert_allclose(got.cpu().numpy(), z["canvas"], rtol=0, atol=1e-4)
    srt = np.sort(z["canvas"], axis=-1)
    decided = (srt[..., -1] - srt[..., -2]) > 5e-4          # away from near-ties the class map must be identical
    assert decided.mean() > 0.95
    np.testing.assert_array_equal(am.cpu().numpy()[decided], z["argmax"][decided])


@pytest.mark.parametrize("prec", ["fp16", "bf16"])
def test_baseline_config4_512_tiles_with_dem_channel_mixed_precision_dice(prec):
    """BASELINE configs[3] as stated: 512x512 tiles, 8 bands + DEM (9 channels through the early-fusion concat), mixed fp16
    (fp16 operands on the matrix cores, fp32 accumulation / BN statistics / master weights) with the BCE + soft-Dice
    extension whose spatial reductions run in fp32 (wave-shuffle partials -> fp64 finalize).  Checked against the same
    steps in fp32 on the same tiles: first loss within 2e-3 (fp16) / 2e-2 (bf16), the four-step trajectory within 5 %."""
    from floodplanet_code_amd.fit import SyntheticTiles
    batch = next(iter(SyntheticTiles(1, 2, {"ms_image": 8, "dem": 1}, 512, 512, DEV, seed=3)))
    runs = {}
    for p in ("fp32", prec):
        torch.manual_seed(0)
        m = build_model("ef_model", {"ms_image": 8, "dem": 1}, 3, 1e-3, log_image_iter=50, to_rgb_fcn=None,
                        ignore_index=-100, precision=p).to(DEV)
        assert m.model.n_channels == 9 and m.model.precision == p
        m._set_model_to_train()
        assert m._gather_input(batch).shape == (2, 9, 512, 512)
        x = m._gather_sources(batch)             # [image, dem]: gathered by the first layout conversion, no concat copy
        assert [tuple(v.shape) for v in x] == [(2, 8, 512, 512), (2, 1, 512, 512)]
        losses = []
        for step in range(1, 5):
            loss = m.model.train_step(x, batch["target"], -100, kind="bce_dice", dice_weight=1.0)
            assert torch.isfinite(m.model.flat_grads()).all()
            m.model.adam_step(1e-3, step)
            losses.append(loss.item())
        torch.cuda.synchronize()
        runs[p] = losses
    ref, got = runs["fp32"], runs[prec]
    assert all(np.isfinite(got)) and got[-1] < got[0]
    assert abs(got[0] - ref[0]) <= (2e-3 if prec == "fp16" else 2e-2), (got, ref)
    assert max(abs(a - b) for a, b in zip(got, ref)) <= 0.05 * ref[0], (got, ref)


def test_baseline_config5_12_channel_stack_with_gpu_augmentation():
    """BASELINE configs[4]: 12-channel stacked input, augmentation on the GPU before the HIP training step."""
    from floodplanet_code_amd import augment
    from floodplanet_code_amd.fit import SyntheticTiles
    torch.manual_seed(0)
    net = HipUNet(12, 3, precision="bf16").to(DEV).train()
    batch = next(iter(SyntheticTiles(1, 4, {"ms_image": 12}, 256, 256, DEV, seed=5)))
    rng = np.random.RandomState(0)
    losses = []
    for step in range(1, 5):
        flags, angles = augment.sample_transforms(4, rng=rng)
        x, t = augment.apply(batch["image"], batch["target"], flags, angles, target_fill=0)
        loss = net.train_step(x, t, 0)
        net.adam_step(1e-3, step)
        losses.append(loss.item())
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and min(losses[1:]) < losses[0]


def test_convtranspose_variant_bf16_tracks_fp32():
    """bilinear=False in bf16: same stated tolerance as the bilinear net against the reference fixture."""
    meta, z = load_golden("f_convT_c4_32")
    batch, st = case_inputs(meta)
    net = HipUNet(meta["n_in"], 3, bilinear=False, base_channels=64, precision="bf16")
    net.load_state_dict(st)
    net.to(DEV).train()
    loss, logits = net.loss(batch["image"].to(DEV), batch["target"].to(DEV), 0, return_logits=True)
    loss.backward()
    torch.cuda.synchronize()
    d = logits.detach().cpu().numpy() - z["logits1"]
    assert np.abs(d).max() <= 0.25 and np.sqrt((d ** 2).mean()) <= 0.05
    assert abs(loss.item() - z["loss1"].item()) <= 0.03
    assert torch.isfinite(net.flat_grads()).all()


@pytest.mark.parametrize("norm_mode", [None, "local"])
def test_device_assembled_batches_equal_the_reference_functions_fixture(tmp_path, norm_mode):
    """SURVEY 8(f) rank 1 + 4 on the data path: TIFF reader -> crop grid -> raw crops + valid sizes to the device ->
    fu_assemble_tiles (per-tile normalise, zero padding of edge crops) -> batch, against the item dicts produced by the
    reference's own get_crop_slices / _crop_image / normalize / _add_buffer_to_image on the same synthetic rasters
    (oracle/make_loader_golden.py; 18 of the 24 crops are cut at the raster's edge).  norm_mode None: two float32
    operations per element, exact; 'local': the statistics are accumulated in fp64 on the device and pairwise in float32
    by numpy -> 1e-6 relative on mean / std, 3e-6 of the value range on the image."""
    import json, os, sys
    sys.path.insert(0, os.path.dirname(__file__))
    from conftest import GOLDEN
    from tools.tiff_writer import make_floodplanet_tree
    from floodplanet_code_amd.datasets import FloodplanetTiles, TileLoader, generate_image_slice_object
    z = np.load(os.path.join(GOLDEN, "loader_golden.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    t = dict(meta["tree"])
    t["regions"] = tuple(t["regions"])
    root = str(tmp_path / "tree")
    make_floodplanet_tree(root, **t)
    sp = generate_image_slice_object(meta["crop"]["height"], meta["crop"]["width"], meta["crop"]["stride"])
    ds = FloodplanetTiles(root, "all", sp, eval_region=["RegA"], sensor="S1", ignore_index=meta["ignore_index"],
                          norm_mode=norm_mode, output_metadata=True)
    loader = TileLoader(ds, 4, DEV, shuffle=False, device_assembly=True)
    seen = 0
    for batch in loader:
        assert batch["image"].device.type == "cuda" and batch["image"].dtype == torch.float32
        for b, md in enumerate(batch["metadata"]):
            cp = md["crop_params"]
            k = f"{md['region_name']}/{os.path.splitext(os.path.basename(md['image_path']))[0]}/{cp.h0}_{cp.w0}"
            ref = z[f"{k}/{norm_mode}/image"]
            got = batch["image"][b].cpu().numpy()
            if norm_mode is None:
                np.testing.assert_array_equal(got, ref, err_msg=k)
            else:
                np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6 * max(1.0, np.abs(ref).max()), err_msg=k)
            np.testing.assert_allclose(batch["mean"][b].cpu().numpy().reshape(-1), z[f"{k}/{norm_mode}/mean"], rtol=1e-6, atol=0)
            np.testing.assert_allclose(batch["std"][b].cpu().numpy().reshape(-1), z[f"{k}/{norm_mode}/std"], rtol=1e-6, atol=0)
            np.testing.assert_array_equal(batch["target"][b].cpu().numpy(), z[f"{k}/target"], err_msg=k)
            seen += 1
    assert seen == len(meta["items"]) == 24


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_multi_source_input_equals_the_concatenated_input(prec):
    """ef_model.py:28-44 concatenates image and auxiliary maps; fu_forward_srcs gathers the same channels inside the NCHW ->
    NHWC conversion instead.  Same values in the same channel slots -> logits, loss and gradients bit-identical, for the
    plain UNet and for the late-fusion net (encoder windows straddling the source boundary)."""
    from floodplanet_code_amd.latefusion import HipLateFusion
    from collections import OrderedDict
    batch = O.make_batch(2, 8, 64, 64, seed=3, extra=("dem", "slope"))
    img, dem, slope, t = (batch[k].to(DEV) for k in ("image", "dem", "slope", "target"))
    cat = torch.cat([img, dem, slope], 1)
    torch.manual_seed(0)
    nets = [HipUNet(10, 3, base_channels=16, precision=prec).to(DEV),
            HipLateFusion(OrderedDict([("ms_image", 8), ("dem", 1), ("slope", 1)]), 3, base_channels=16, precision=prec).to(DEV)]
    for net in nets:
        net.train()
        la, lga = net.loss(cat, t, 0, return_logits=True)
        la.backward()
        ga = net.flat_grads().clone()
        net.zero_grad(set_to_none=True)
        for parts in ([img, dem, slope], [img[:, :5], torch.cat([img[:, 5:], dem], 1), slope]):
            lb, lgb = net.loss(parts, t, 0, return_logits=True)
            lb.backward()
            torch.cuda.synchronize()
            assert torch.equal(lga, lgb) and la.item() == lb.item()
            assert torch.equal(ga, net.flat_grads())
            net.zero_grad(set_to_none=True)
        net.eval()
        with torch.no_grad():
            assert torch.equal(net(cat), net([img, dem, slope]))
    with pytest.raises(ValueError):
        nets[0]([img, dem])                      # 9 channels for a 10-channel model


def test_device_lanczos_tiles_equal_the_host_resample_bit_for_bit(tmp_path):
    """fu_resize_lanczos4_tiles (round 4; reference: the per-item whole-raster cv2.INTER_LANCZOS4 resize of
    st_water_seg/datasets/floodplanet.py:338-340, utils/utils_image.py:11-54): (a) the kernel on random windows / tables against
    the numpy restatement of the same taps -- bit for bit, NaN pixels and the S1 / S2 / L8 / PS scalings included; (b) the
    loader's device-resampling path (window cut-out in the worker, resample + scaling + assembly in HBM) against its
    host-resampling path (whole-raster resample in the worker) on rasters whose images are smaller than their labels and whose
    tile grid cuts crops at the raster's edge: identical images for norm_mode None, 3e-6 for 'local' (fp64 statistics on the
    device), identical targets.  OpenCV itself is absent from the image: parity with cv2 stays unpinned (datasets/resize.py)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from tools.tiff_writer import make_floodplanet_tree
    from floodplanet_code_amd.datasets import FloodplanetTiles, TileLoader, generate_image_slice_object
    from floodplanet_code_amd.datasets.assemble import resize_lanczos4_tiles
    from floodplanet_code_amd.datasets.resize import lanczos4_axis_window, resize_lanczos4, resize_lanczos4_tile
    g = np.random.default_rng(3)
    src = (g.random((3, 45, 52), dtype=np.float32) * 70 - 50).astype(np.float32)
    src[1, 7, 9] = np.nan
    full = resize_lanczos4(src, 150, 131)
    tiles = [(0, 64, 0, 64), (64, 128, 64, 128), (128, 150, 67, 131), (100, 150, 120, 131)]
    wins, tabs = [], []
    for (h0, hE, w0, wE) in tiles:
        iy, wy, (y0, y1) = lanczos4_axis_window(45, 150, h0, hE, 64)
        ix, wx, (x0, x1) = lanczos4_axis_window(52, 131, w0, wE, 64)
        wins.append(src[:, y0:y1, x0:x1])
        tabs.append((iy, wy, ix, wx))
    wh, ww = max(w.shape[1] for w in wins), max(w.shape[2] for w in wins)
    win = np.zeros((len(tiles), 3, wh, ww), dtype=np.float32)
    for b, w in enumerate(wins):
        win[b, :, :w.shape[1], :w.shape[2]] = w
    stack = lambda k: torch.from_numpy(np.stack([t[k] for t in tabs]))
    scal = {0: lambda v: v, 1: lambda v: np.nan_to_num(np.clip((v + 50) / 100, 0, 1)), 2: lambda v: np.clip(v / 2 ** 12, 0, 1),
            3: lambda v: np.clip(v, 0, 18607.72) / 18607.72, 4: lambda v: v / 2 ** 16}
    for mode, fn in scal.items():
        got = resize_lanczos4_tiles(torch.from_numpy(win).to(DEV), stack(0), stack(1), stack(2), stack(3), mode).cpu().numpy()
        for b, (h0, hE, w0, wE) in enumerate(tiles):
            ref_t = resize_lanczos4_tile(wins[b], *tabs[b])
            np.testing.assert_array_equal(ref_t[:, :hE - h0, :wE - w0], full[:, h0:hE, w0:wE])      # tile == crop of the whole
            want = fn(ref_t).astype(np.float32)
            if mode == 0:
                np.testing.assert_array_equal(np.isnan(got[b]), np.isnan(want))
            np.testing.assert_array_equal(np.nan_to_num(got[b]), np.nan_to_num(want), err_msg=f"mode {mode} tile {b}")
    # (b) the loader
    root = str(tmp_path / "data")
    make_floodplanet_tree(root, label_size=150, s1_size=52)
    sp = generate_image_slice_object(64, 64, 64)
    for norm_mode in (None, "local"):
        ds = FloodplanetTiles(root, "all", sp, eval_region=["RegC"], sensor="S1", ignore_index=0, norm_mode=norm_mode)
        host = TileLoader(ds, 3, DEV, shuffle=False, device_assembly=True)
        devl = TileLoader(ds, 3, DEV, shuffle=False, device_assembly=True, device_resize=True)
        n = 0
        for bh, bd in zip(host, devl):
            assert bd["image"].device.type == "cuda" and bd["image"].shape == bh["image"].shape
            if norm_mode is None:
                assert torch.equal(bh["image"], bd["image"])
            else:
                assert (bh["image"] - bd["image"]).abs().max().item() <= 3e-6 * max(1.0, bh["image"].abs().max().item())
            assert torch.equal(bh["target"], bd["target"])
            n += bh["image"].shape[0]
        assert n == len(ds) and n >= 18


def test_tiff_tiles_train_through_the_registry(tmp_path):
    """BASELINE configs[0] on rasters in the bundled sample's format (S1: 2-band float32 planar strips, uint8 labels at a
    higher resolution): TIFF decode -> Lanczos resample -> tile grid -> GPU augmentation -> registry model -> fit loop."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from tools.tiff_writer import make_floodplanet_tree
    from floodplanet_code_amd.datasets import FloodplanetTiles, TileLoader, generate_image_slice_object
    from floodplanet_code_amd.fit import fit_model
    root = str(tmp_path / "data")
    make_floodplanet_tree(root, label_size=128, s1_size=48)
    sp = generate_image_slice_object(64, 64, 64)
    tr = FloodplanetTiles(root, "train", sp, eval_region=["RegC"], sensor="S1", ignore_index=0)
    va = FloodplanetTiles(root, "valid", sp, eval_region=["RegC"], sensor="S1", ignore_index=0)
    assert len(tr) == 16 and len(va) == 8 and tr.n_channels == {"ms_image": 2}
    train = TileLoader(tr, 2, DEV, shuffle=True, seed=0, transforms={}, ignore_index=0, device_assembly=True)
    valid = TileLoader(va, 2, DEV)
    b = next(iter(train))
    assert b["image"].shape == (2, 2, 64, 64) and b["image"].device.type == "cuda" and b["target"].dtype == torch.int64
    cfg = dict(lr=2e-3, n_epochs=2, save_topk_models=1, ignore_index=0,
               model=dict(name="ms_model", model_kwargs=dict(optimizer_name="adam", base_channels=8)))
    best = fit_model(cfg, train, valid, tr.n_channels, 3, exp_dir=str(tmp_path / "exp"), device=DEV)
    hist = fit_model.last_model.history
    assert best.endswith(".ckpt") and len(hist) == 2
    assert all(np.isfinite(h["train_loss"]) and 0.0 <= h["val_MulticlassJaccardIndex"] <= 1.0 for h in hist)
