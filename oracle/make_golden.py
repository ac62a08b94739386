"""Pin the oracle against the REAL reference and write golden fixtures.  TEST INFRASTRUCTURE ONLY.

Runs only in the development container (needs /root/reference).  It
  1. imports the reference's st_water_seg/models/unet.py *by file path* (never copied,
     no bytecode written into the read-only mount),
  2. for every case below, loads identical deterministic parameters into the reference
     modules and into oracle/unet_oracle.py, runs two training steps
     (net.train(); CrossEntropyLoss(ignore_index); nan_to_num; backward; optim.Adam(lr).step()
     -- the 10 lines of water_seg_model.py:98-108,198-205 that cannot be imported because
     pytorch_lightning / torchmetrics are absent) plus one eval forward,
  3. asserts the oracle reproduces the reference BIT-EXACTLY (same ATen kernels), and
  4. writes tests/golden/<case>.npz: expected outputs only -- inputs and parameters are
     re-derived from the closed-form generator in unet_oracle.py, so no weights ship.

Usage:  python oracle/make_golden.py [--out tests/golden]
"""
import sys
sys.dont_write_bytecode = True

import argparse
import importlib.util
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import unet_oracle as O  # noqa: E402

REF_UNET = "/root/reference/st_water_seg/models/unet.py"

# name, B, C(image), H, W, base, bilinear, ignore_index(as configured), extras, all_ignored, full_dump, lr
CASES = [
    dict(name="s_b2c4_32", B=2, C=4, H=32, W=32, base=4, bilinear=True, ignore_index=0, full=True),
    dict(name="s_odd_37x45", B=2, C=3, H=37, W=45, base=4, bilinear=True, ignore_index=0, full=True),
    dict(name="s_ii_last", B=2, C=4, H=32, W=32, base=4, bilinear=True, ignore_index=-1, n_label_values=3),
    dict(name="s_ii_none", B=2, C=4, H=32, W=32, base=4, bilinear=True, ignore_index=-100, n_label_values=3),
    dict(name="s_one_sample_ignored", B=3, C=4, H=32, W=32, base=4, bilinear=True, ignore_index=0, all_ignored_sample=1),
    dict(name="s_all_ignored", B=2, C=4, H=32, W=32, base=4, bilinear=True, ignore_index=0, all_ignored=True),
    dict(name="s_c2_s1", B=2, C=2, H=32, W=48, base=4, bilinear=True, ignore_index=0),
    dict(name="s_ef_c9", B=2, C=8, H=32, W=32, base=4, bilinear=True, ignore_index=0, extras=("dem",)),
    dict(name="s_ef_c12", B=2, C=8, H=32, W=32, base=4, bilinear=True, ignore_index=0,
         extras=("dem", "slope", "preflood", "hand")),
    # bilinear=False only exists at base 64: the reference's UNetDecoder channel plan
    # (unet.py:176-183) is inconsistent for bilinear=False, only UNet (unet.py:88-98) works.
    dict(name="f_convT_c4_32", B=2, C=4, H=32, W=32, base=64, bilinear=False, ignore_index=0),
    dict(name="f_convT_odd_35x41", B=1, C=4, H=35, W=41, base=64, bilinear=False, ignore_index=0),
    dict(name="m_base8_64", B=2, C=8, H=64, W=64, base=8, bilinear=True, ignore_index=0, full=True),
    dict(name="m_base16_300", B=1, C=4, H=300, W=300, base=16, bilinear=True, ignore_index=0),
    dict(name="f_full_c8_32", B=1, C=8, H=32, W=32, base=64, bilinear=True, ignore_index=0),
    dict(name="f_full_c8_64_b2", B=2, C=8, H=64, W=64, base=64, bilinear=True, ignore_index=0),
    # 32 x 32 tiles at full width with a batch that means something in 16-bit arithmetic: B = 1 (f_full_c8_32) leaves FOUR
    # samples per channel at the 2 x 2 level -- a degenerate BatchNorm that amplifies any rounding of its input (the bf16
    # cosine of that fixture is 0.84 whatever kernel runs); B = 8 leaves 32, as f_full_c8_64_b2 does
    dict(name="f_full_c8_32_b8", B=8, C=8, H=32, W=32, base=64, bilinear=True, ignore_index=0),
    # larger batches at full width: 128 / 216 samples per channel at the deepest level instead of 16-32, so that
    # BatchNorm does not amplify rounding and ReLU / max-pool near-ties are rare -> the GPU test bounds every gradient
    # tensor of these two at 1e-3 (tests/test_gpu_unet.py), where the small-batch cases need 3e-2
    dict(name="f_full_c8_64_b8", B=8, C=8, H=64, W=64, base=64, bilinear=True, ignore_index=0),
    dict(name="f_full_c4_96_b6", B=6, C=4, H=96, W=96, base=64, bilinear=True, ignore_index=0, n_label_values=3),
]
LR = 1e-3  # larger than the config default 1e-4 so two Adam steps move the logits visibly
N_CLASSES = 3


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_unet", REF_UNET)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class RefNet(torch.nn.Module):
    """The reference network at any width: UNet itself at base 64, else its
    UNetEncoder/UNetDecoder pair (unet.py:134-200), which is the same wiring."""

    def __init__(self, ref, n_in, n_classes, base, bilinear):
        super().__init__()
        self.base = base
        if base == 64:
            self.net = ref.UNet(n_in, n_classes, bilinear=bilinear)
        else:
            self.enc = ref.UNetEncoder(n_in, bilinear=bilinear, base_feat_channels=base)
            self.dec = ref.UNetDecoder(n_classes, bilinear=bilinear, base_feat_channels=base)

    def forward(self, x):
        if self.base == 64:
            return self.net(x)
        return self.dec(self.enc(x))

    def load_flat(self, st):
        if self.base == 64:
            self.net.load_state_dict({k: v.clone() for k, v in st.items()}, strict=True)
        else:
            e = {k: v.clone() for k, v in st.items() if k.startswith(("inc.", "down"))}
            d = {k: v.clone() for k, v in st.items() if k.startswith(("up", "outc."))}
            self.enc.load_state_dict(e, strict=True)
            self.dec.load_state_dict(d, strict=True)

    def flat_state(self):
        if self.base == 64:
            return dict(self.net.state_dict())
        out = dict(self.enc.state_dict())
        out.update(self.dec.state_dict())
        return out

    def named_flat_params(self):
        if self.base == 64:
            return dict(self.net.named_parameters())
        out = dict(self.enc.named_parameters())
        out.update(self.dec.named_parameters())
        return out


def tensor_stats(t):
    a = t.detach().double().reshape(-1)
    return np.array([a.sum().item(), a.abs().sum().item(), a.pow(2).sum().sqrt().item()], dtype=np.float64)


def run_case(ref, case, out_dir):
    name = case["name"]
    B, C, H, W = case["B"], case["C"], case["H"], case["W"]
    base, bilinear = case["base"], case["bilinear"]
    extras = tuple(case.get("extras", ()))
    ef = len(extras) > 0
    n_in = C + len(extras)
    ii = O.resolve_ignore_index(case["ignore_index"], N_CLASSES)
    batch = O.make_batch(B, C, H, W, seed=1, n_label_values=case.get("n_label_values", 2),
                         all_ignored_sample=case.get("all_ignored_sample"), ignore_value=ii, extra=extras)
    if case.get("all_ignored"):
        batch["target"][:] = ii
    st0 = O.make_state(n_in, N_CLASSES, base, bilinear, seed=0)

    # ---------------- reference ----------------
    torch.manual_seed(0)
    net = RefNet(ref, n_in, N_CLASSES, base, bilinear)
    net.load_flat(st0)
    lossf = torch.nn.CrossEntropyLoss(ignore_index=ii)
    opt = torch.optim.Adam(net.parameters(), lr=LR)
    x = O.assemble_input(batch, ef)
    ref_out = {}
    for step in (1, 2):
        net.train()
        opt.zero_grad()
        logits = net(x)
        loss = lossf(logits, batch["target"])
        if torch.isnan(loss):
            loss = torch.nan_to_num(loss)
        loss.backward()
        grads = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                 for k, p in net.named_flat_params().items()}
        ref_out[f"logits{step}"] = logits.detach().clone()
        ref_out[f"loss{step}"] = loss.detach().clone()
        ref_out[f"grads{step}"] = grads
        if step == 1:
            ref_out["state_after_fwd1"] = {k: v.clone() for k, v in net.flat_state().items()}
        opt.step()
    ref_out["state2"] = {k: v.clone() for k, v in net.flat_state().items()}
    net.eval()
    with torch.no_grad():
        ref_out["eval_logits"] = net(x).clone()

    # ---------------- oracle ----------------
    st = {k: v.clone() for k, v in st0.items()}
    ad = O.new_adam_state(st)
    orc = {}
    for step in (1, 2):
        logits, loss, grads = O.train_step(st, ad, batch, ii, LR, bilinear, ef)
        orc[f"logits{step}"], orc[f"loss{step}"], orc[f"grads{step}"] = logits, loss, grads
    orc["eval_logits"] = O.eval_forward(st, batch, bilinear, ef)

    # ---------------- pin: bit equality ----------------
    def same(a, b, what):
        if not torch.equal(a, b):
            d = (a.double() - b.double()).abs().max().item()
            raise SystemExit(f"[{name}] oracle != reference for {what}: max|d|={d:g}")

    for step in (1, 2):
        same(orc[f"logits{step}"], ref_out[f"logits{step}"], f"logits{step}")
        same(orc[f"loss{step}"], ref_out[f"loss{step}"], f"loss{step}")
        for k, g in ref_out[f"grads{step}"].items():
            same(orc[f"grads{step}"][k], g, f"grad{step}:{k}")
    for k, v in ref_out["state2"].items():
        same(st[k], v, f"state2:{k}")
    same(orc["eval_logits"], ref_out["eval_logits"], "eval_logits")

    # ---------------- fixture ----------------
    names = O.trainable_names(st0)
    pred1 = ref_out["logits1"].argmax(1)
    conf = O.confusion_counts(pred1, batch["target"], N_CLASSES, ii)
    n_valid = int((batch["target"] != ii).sum())
    meta = dict(case)
    meta.update(n_in=n_in, n_classes=N_CLASSES, resolved_ignore_index=ii, lr=LR, data_seed=1, param_seed=0,
                torch=torch.__version__, n_valid=n_valid, names=names)
    arrays = {
        "meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
        "logits1": ref_out["logits1"].numpy(),
        "loss1": ref_out["loss1"].numpy(),
        "logits2": ref_out["logits2"].numpy(),
        "loss2": ref_out["loss2"].numpy(),
        "eval_logits": ref_out["eval_logits"].numpy(),
        "confusion1": conf,
        "grad_stats1": np.stack([tensor_stats(ref_out["grads1"][k]) for k in names]),
        "param_stats2": np.stack([tensor_stats(ref_out["state2"][k]) for k in names]),
    }
    bn_keys = [k for k in st0 if k.endswith("running_mean") or k.endswith("running_var")]
    arrays["bn_keys"] = np.frombuffer(json.dumps(bn_keys).encode(), dtype=np.uint8)
    for j, k in enumerate(bn_keys):
        arrays[f"bn1_{j}"] = ref_out["state_after_fwd1"][k].numpy()
    if case.get("full"):
        for j, k in enumerate(names):
            arrays[f"g1_{j}"] = ref_out["grads1"][k].numpy()
            arrays[f"p2_{j}"] = ref_out["state2"][k].numpy()
    else:
        # slices: first 64 elements of every gradient tensor
        for j, k in enumerate(names):
            arrays[f"g1s_{j}"] = ref_out["grads1"][k].reshape(-1)[:64].numpy()
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"[{name}] oracle==reference bit-exact; loss1={ref_out['loss1'].item():.6f} "
          f"loss2={ref_out['loss2'].item():.6f} n_valid={n_valid} -> {path} "
          f"({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    ref = load_reference()
    for case in CASES:
        if args.only and case["name"] != args.only:
            continue
        run_case(ref, case, args.out)


if __name__ == "__main__":
    main()
