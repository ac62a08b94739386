"""Pin the late-fusion oracle (oracle/unet_oracle.py lf_*) against the REAL reference and write golden fixtures.
TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).

lf_model.py itself cannot be imported (torchmetrics / pytorch_lightning are absent), so the reference side is built
from the reference's own UNetEncoder / UNetDecoder classes (imported by file path from st_water_seg/models/unet.py) and
torch.nn.Conv2d, wired by the 40 lines of lf_model.py:29-92 read as text: ModuleDict of encoders, decoder,
ModuleList of Conv2d(fs*n, fs, 1, 1); forward = encode, concat per level in the fixed order, fuse, decode.
Asserts bit equality with the oracle over two Adam steps + one eval forward, then writes tests/golden/lf_<case>.npz.

Usage: PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_lf.py
"""
import sys
sys.dont_write_bytecode = True

import json
import os
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import unet_oracle as O            # noqa: E402
from oracle.make_golden import load_reference, tensor_stats   # noqa: E402

LR = 1e-3
N_CLASSES = 3
CASES = [
    dict(name="lf_s_img4_dem", B=2, H=32, W=32, base=4, in_channels=[("ms_image", 4), ("dem", 1)], full=True),
    dict(name="lf_s_three_odd", B=2, H=37, W=45, base=4, in_channels=[("dem", 1), ("ms_image", 3), ("slope", 1)],
         full=True),
    dict(name="lf_s_single", B=2, H=32, W=32, base=4, in_channels=[("ms_image", 4)]),
    dict(name="lf_m_base8_64", B=2, H=64, W=64, base=8, in_channels=[("ms_image", 8), ("dem", 1)], full=True),
    dict(name="lf_f_full_32", B=1, H=32, W=32, base=64, in_channels=[("ms_image", 8), ("dem", 1)]),
]


class RefLateFusion(torch.nn.Module):
    def __init__(self, ref, in_channels, n_classes, base):
        super().__init__()
        self.in_channels = in_channels
        self.encoders = torch.nn.ModuleDict()
        for name, ch in in_channels.items():                                   # lf_model.py:33-36
            self.encoders[name] = ref.UNetEncoder(ch, base_feat_channels=base)
        self.decoder = ref.UNetDecoder(n_classes, base_feat_channels=base)      # lf_model.py:38
        sizes = [base, base * 2, base * 4, base * 8, base * 8]                  # lf_model.py:41 at base 64
        self.concat_convs = torch.nn.ModuleList(
            [torch.nn.Conv2d(fs * len(in_channels), fs, 1, 1) for fs in sizes])  # lf_model.py:42-45

    def forward(self, batch):                                                   # lf_model.py:54-92
        image_feats = self.encoders["ms_image"](batch["image"])
        extra = []
        for key in ("dem", "slope", "preflood", "pre_post_difference", "hand"):
            if key in batch:
                extra.append(self.encoders[key](batch[key]))
        for ef in extra:
            for i, (a, b) in enumerate(zip(image_feats, ef)):
                image_feats[i] = torch.concat([a, b], dim=1)
        fused = [cc(f) for f, cc in zip(image_feats, self.concat_convs)]
        return self.decoder(fused)


def run_case(ref, case, out_dir):
    name, B, H, W, base = case["name"], case["B"], case["H"], case["W"], case["base"]
    in_ch = OrderedDict(case["in_channels"])
    extras = tuple(k for k in O.LF_FORWARD_ORDER if k in in_ch and k != "ms_image")
    batch = O.make_batch(B, in_ch["ms_image"], H, W, seed=1, extra=extras)
    st0 = O.lf_make_state(in_ch, N_CLASSES, base, seed=0)

    net = RefLateFusion(ref, in_ch, N_CLASSES, base)
    net.load_state_dict({k: v.clone() for k, v in st0.items()}, strict=True)
    lossf = torch.nn.CrossEntropyLoss(ignore_index=0)
    opt = torch.optim.Adam(net.parameters(), lr=LR)
    ref_out = {}
    for step in (1, 2):
        net.train()
        opt.zero_grad()
        logits = net(batch)
        loss = lossf(logits, batch["target"])
        if torch.isnan(loss):
            loss = torch.nan_to_num(loss)
        loss.backward()
        ref_out[f"logits{step}"], ref_out[f"loss{step}"] = logits.detach().clone(), loss.detach().clone()
        ref_out[f"grads{step}"] = {k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                                   for k, p in net.named_parameters()}
        if step == 1:
            ref_out["state_after_fwd1"] = {k: v.clone() for k, v in net.state_dict().items()}
        opt.step()
    ref_out["state2"] = {k: v.clone() for k, v in net.state_dict().items()}
    net.eval()
    with torch.no_grad():
        ref_out["eval_logits"] = net(batch).clone()

    st = {k: v.clone() for k, v in st0.items()}
    ad = O.new_adam_state(st)
    orc = {}
    for step in (1, 2):
        orc[f"logits{step}"], orc[f"loss{step}"], orc[f"grads{step}"] = O.lf_train_step(st, ad, batch, in_ch, 0, LR)
    orc["eval_logits"] = O.lf_eval_forward(st, batch, in_ch)

    def same(a, b, what):
        if not torch.equal(a, b):
            raise SystemExit(f"[{name}] oracle != reference for {what}: max|d|="
                             f"{(a.double() - b.double()).abs().max().item():g}")
    for step in (1, 2):
        same(orc[f"logits{step}"], ref_out[f"logits{step}"], f"logits{step}")
        same(orc[f"loss{step}"], ref_out[f"loss{step}"], f"loss{step}")
        for k, g in ref_out[f"grads{step}"].items():
            same(orc[f"grads{step}"][k], g, f"grad{step}:{k}")
    for k, v in ref_out["state2"].items():
        same(st[k], v, f"state2:{k}")
    same(orc["eval_logits"], ref_out["eval_logits"], "eval_logits")

    names = O.trainable_names(st0)
    meta = dict(case)
    meta.update(n_classes=N_CLASSES, ignore_index=0, lr=LR, data_seed=1, param_seed=0, torch=torch.__version__,
                names=names)
    arrays = {"meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
              "logits1": ref_out["logits1"].numpy(), "loss1": ref_out["loss1"].numpy(),
              "logits2": ref_out["logits2"].numpy(), "loss2": ref_out["loss2"].numpy(),
              "eval_logits": ref_out["eval_logits"].numpy(),
              "grad_stats1": np.stack([tensor_stats(ref_out["grads1"][k]) for k in names]),
              "param_stats2": np.stack([tensor_stats(ref_out["state2"][k]) for k in names])}
    bn_keys = [k for k in st0 if k.endswith("running_mean") or k.endswith("running_var")]
    arrays["bn_keys"] = np.frombuffer(json.dumps(bn_keys).encode(), dtype=np.uint8)
    for j, k in enumerate(bn_keys):
        arrays[f"bn1_{j}"] = ref_out["state_after_fwd1"][k].numpy()
    for j, k in enumerate(names):
        if case.get("full"):
            arrays[f"g1_{j}"] = ref_out["grads1"][k].numpy()
        else:
            arrays[f"g1s_{j}"] = ref_out["grads1"][k].reshape(-1)[:64].numpy()
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"[{name}] oracle==reference bit-exact; loss1={ref_out['loss1'].item():.6f} "
          f"loss2={ref_out['loss2'].item():.6f} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(HERE), "tests", "golden")
    torch.set_num_threads(8)
    ref = load_reference()
    for case in CASES:
        run_case(ref, case, out)
