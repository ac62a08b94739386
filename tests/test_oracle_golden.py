"""CPU: the oracle (oracle/unet_oracle.py) against the golden fixtures produced from the REAL reference by
oracle/make_golden.py.  In the generating container the match is bit-exact; here a tight tolerance is used
because another host CPU may pick other ATen/oneDNN code paths."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, case_inputs, golden_names, is_dead_bias, lf_case_inputs, load_golden
from oracle import unet_oracle as O

FAST = [n for n in golden_names() if n.startswith("s_")] + ["m_base8_64", "f_full_c8_32"]


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference_fixture(name):
    torch.set_num_threads(4)
    meta, z = load_golden(name)
    batch, st = case_inputs(meta)
    ii, lr = meta["resolved_ignore_index"], meta["lr"]
    ef = len(meta.get("extras", ())) > 0
    opt = O.new_adam_state(st)
    logits1, loss1, grads1 = O.train_step(st, opt, batch, ii, lr, meta["bilinear"], ef)
    np.testing.assert_allclose(logits1.numpy(), z["logits1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(loss1.item(), z["loss1"].item(), rtol=1e-5, atol=1e-6)
    names = meta["names"]
    for j, k in enumerate(names):
        if is_dead_bias(k):  # analytically zero: pure rounding noise, thread-count dependent
            continue
        g = grads1[k].double()
        s = z["grad_stats1"][j]
        assert abs(g.pow(2).sum().sqrt().item() - s[2]) <= 1e-4 * max(s[2], 1e-6) + 1e-7, k
        if f"g1_{j}" in z.files:
            ref = torch.from_numpy(z[f"g1_{j}"]).double()
            assert (g - ref).norm() <= 1e-4 * ref.norm() + 1e-9, k
    conf = O.confusion_counts(logits1.argmax(1), batch["target"], meta["n_classes"], ii)
    assert conf.sum() == meta["n_valid"]
    if np.abs(np.sort(z["logits1"], axis=1)[:, -1] - np.sort(z["logits1"], axis=1)[:, -2]).min() > 1e-4:
        np.testing.assert_array_equal(conf, z["confusion1"])
    logits2, loss2, _ = O.train_step(st, opt, batch, ii, lr, meta["bilinear"], ef)
    np.testing.assert_allclose(loss2.item(), z["loss2"].item(), rtol=2e-4, atol=1e-6)
    ev = O.eval_forward(st, batch, meta["bilinear"], ef)
    np.testing.assert_allclose(ev.numpy(), z["eval_logits"], rtol=0, atol=2e-3)


def test_all_ignored_gives_zero_loss_and_zero_grads():
    meta, z = load_golden("s_all_ignored")
    batch, st = case_inputs(meta)
    _, loss, grads = O.loss_and_grads(st, batch, meta["resolved_ignore_index"])
    assert loss.item() == 0.0 and z["loss1"].item() == 0.0
    assert all(float(g.abs().max()) == 0.0 for g in grads.values())


def test_algorithmic_flops_match_survey():
    fwd, train = O.conv_flops_per_tile(8, 256, 256)
    assert abs(fwd / 1e9 - 80.354) < 1e-3 and abs(train / 1e9 - 240.459) < 1e-3
    fwd, train = O.conv_flops_per_tile(9, 512, 512)
    assert abs(fwd / 1e9 - 321.720) < 1e-3 and abs(train / 1e9 - 962.442) < 1e-3


def test_param_spec_counts():
    spec = O.param_spec(8, 3)
    n = sum(int(np.prod(s)) if len(s) else 1 for _, (s, kind) in spec.items() if O.is_trainable(kind))
    assert n == 17270403 and len(spec) == 128
    assert sum(1 for _, (_, kind) in spec.items() if O.is_trainable(kind)) == 74


@pytest.mark.parametrize("name", [n for n in golden_names(late_fusion=True) if n != "lf_f_full_32"])
def test_late_fusion_oracle_matches_reference_fixture(name):
    """oracle lf_* (lf_model.py:29-92 restated) against the fixtures made from the reference's own UNetEncoder /
    UNetDecoder classes (oracle/make_golden_lf.py; bit-exact there)."""
    torch.set_num_threads(4)
    meta, z = load_golden(name)
    batch, in_ch, st = lf_case_inputs(meta)
    lr = meta["lr"]
    opt = O.new_adam_state(st)
    logits1, loss1, grads1 = O.lf_train_step(st, opt, batch, in_ch, 0, lr)
    np.testing.assert_allclose(logits1.numpy(), z["logits1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(loss1.item(), z["loss1"].item(), rtol=1e-5, atol=1e-6)
    for j, k in enumerate(meta["names"]):
        if is_dead_bias(k):
            continue
        g, s = grads1[k].double(), z["grad_stats1"][j]
        assert abs(g.pow(2).sum().sqrt().item() - s[2]) <= 1e-4 * max(s[2], 1e-6) + 1e-7, k
        if f"g1_{j}" in z.files:
            ref = torch.from_numpy(z[f"g1_{j}"]).double()
            assert (g - ref).norm() <= 1e-4 * ref.norm() + 1e-9, k
    _, loss2, _ = O.lf_train_step(st, opt, batch, in_ch, 0, lr)
    np.testing.assert_allclose(loss2.item(), z["loss2"].item(), rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(O.lf_eval_forward(st, batch, in_ch).numpy(), z["eval_logits"], rtol=0, atol=2e-3)


def test_late_fusion_param_spec():
    from collections import OrderedDict
    spec = O.lf_param_spec(OrderedDict([("ms_image", 4), ("dem", 1), ("slope", 1)]), 3)
    keys = list(spec)
    assert keys[0] == "encoders.ms_image.inc.double_conv.0.weight" and keys[-1] == "concat_convs.4.bias"
    assert spec["concat_convs.3.weight"][0] == (512, 1536, 1, 1) and spec["concat_convs.4.weight"][0] == (512, 1536, 1, 1)
    assert "decoder.outc.conv.weight" in spec and "decoder.inc.double_conv.0.weight" not in spec


@pytest.mark.parametrize("name", ["stitch_overlap_96x112", "stitch_partial_120x100"])
def test_oracle_stitching_matches_the_reference_class(name):
    """O.stitch_reference against canvases produced by the reference's own ImageStitcher_v2 (AST-extracted and run in the
    build container by oracle/make_stitch_golden.py), incl. crops cut at the raster's edge."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    canvas, weight = O.stitch_reference(z["logits"], meta["boxes"], meta["H"], meta["W"])
    np.testing.assert_array_equal(weight, z["weight"])
    np.testing.assert_allclose(canvas, z["canvas"], rtol=0, atol=3e-7)     # scipy softmax vs exp/sum: last-ulp float32
    assert canvas.dtype == z["canvas"].dtype == np.float64
    # and the oracle's eval forward still produces the stored logits from the seeds alone
    st = O.make_state(meta["C"], 3, meta["base"], True, seed=meta["param_seed"])
    big = torch.from_numpy(O.hash_uniform(meta["C"] * meta["H"] * meta["W"], meta["data_seed"], 77)
                           .astype(np.float32).reshape(meta["C"], meta["H"], meta["W"]))
    S = meta["S"]
    h0, w0, hE, wE = meta["boxes"][-1]
    x = torch.zeros(1, meta["C"], S, S)
    x[0, :, :hE - h0, :wE - w0] = big[:, h0:hE, w0:wE]
    lg = O.eval_forward(st, {"image": x})
    np.testing.assert_allclose(lg[0].numpy(), z["logits"][-1], rtol=0, atol=2e-5)


def test_oracle_tile_assembly_matches_the_reference_methods():
    """O.assemble_tiles against outputs of the reference's own `BaseDataset.normalize` / `_add_buffer_to_image`
    (AST-extracted and run in the build container by oracle/make_assemble_golden.py): None / local / global normalisation,
    several sensors, edge crops smaller than the nominal tile."""
    z = np.load(os.path.join(GOLDEN, "assemble_golden.npz"))
    for case in json.loads(bytes(z["meta"]).decode()):
        srcs = O.assemble_case_sources(case)
        names = [s[0] for s in case["sources"]]
        image, mean, std = O.assemble_tiles(srcs, [tuple(v) for v in case["valid"]], case["norm_mode"],
                                            O.assemble_global_params(case), names)
        n = case["name"]
        np.testing.assert_allclose(mean, z[n + "_mean"], rtol=1e-6, atol=0, err_msg=n)
        np.testing.assert_allclose(std, z[n + "_std"], rtol=1e-6, atol=0, err_msg=n)
        np.testing.assert_allclose(image, z[n + "_image"], rtol=0, atol=2e-6 * max(1.0, np.abs(z[n + "_image"]).max()), err_msg=n)
