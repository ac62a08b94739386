"""CPU oracle for the FloodPlanet UNet training step.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, in functional torch-CPU fp32, of the arithmetic the
reference executes on its hot path.  It is the *checker* for the HIP path: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``floodplanet_code_amd/`` imports it and the
product path never falls back to it.

Parity status: PINNED.  ``oracle/make_golden.py`` (run in the development
container, where ``/root/reference`` is mounted) imports the reference's own
``st_water_seg/models/unet.py`` by file path, loads the same parameters into it
and into this restatement, and checks logits / loss / every gradient / BN
buffers / Adam-updated parameters for bit equality before writing the fixtures
under ``tests/golden/``.  ``tests/test_oracle_golden.py`` re-checks this file
against those fixtures on every run (CPU).

What is restated (reference file:line, relative to /root/reference):
  * DoubleConv         st_water_seg/models/unet.py:6-20
  * Down               st_water_seg/models/unet.py:23-32
  * Up (both variants) st_water_seg/models/unet.py:35-67
  * OutConv            st_water_seg/models/unet.py:70-77
  * UNet wiring        st_water_seg/models/unet.py:80-111 (and the
                       base_feat_channels generalisation of :134-200)
  * forward of the plugin + early-fusion concat
                       st_water_seg/models/water_seg_model.py:87-90,
                       st_water_seg/models/ef_model.py:24-47
  * training_step loss st_water_seg/models/water_seg_model.py:98-108
                       (CrossEntropyLoss(ignore_index) :40, NaN guard :104-106,
                       argmax :107)
  * optimiser          st_water_seg/models/water_seg_model.py:198-205
                       (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8,
                       no weight decay, bias correction)
  * metric counters    st_water_seg/models/water_seg_model.py:46-63 (micro
                       F1 / Jaccard / Accuracy with ignore_index; torchmetrics
                       itself is third-party and absent, so these three are
                       "parity unpinned" -- see metrics_from_counts()).

Parameters live in a flat ``dict`` whose keys and shapes are exactly the
reference ``UNet.state_dict()`` keys (OIHW fp32), so a state dict moves between
the reference, this oracle and the HIP module unchanged.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default (unet.py:15,17)
BN_MOMENTUM = 0.1      # nn.BatchNorm2d default
ADAM_BETA1, ADAM_BETA2, ADAM_EPS = 0.9, 0.999, 1e-8
EF_EXTRA_KEYS = ("dem", "slope", "preflood", "pre_post_difference", "hand")  # ef_model.py:28-41


# --------------------------------------------------------------------------- #
# architecture description
# --------------------------------------------------------------------------- #
def channel_plan(n_channels: int, base: int = 64, bilinear: bool = True):
    """Channel plan of unet.py:88-98 / :143-149 / :176-183 (channel_factor=1).

    Returns (enc, dec) where enc[i] = (prefix, cin, cmid, cout) for inc/down1..4 and
    dec[i] = (prefix, c_low, c_skip, cmid, cout) for up1..4.
    """
    factor = 2 if bilinear else 1
    e = [base, base * 2, base * 4, base * 8, (base * 16) // factor]
    enc = [("inc.double_conv", n_channels, e[0], e[0])]
    for i in range(1, 5):
        enc.append((f"down{i}.maxpool_conv.1.double_conv", e[i - 1], e[i], e[i]))
    dec = []
    low = e[4]
    outs = [(base * 8) // factor, (base * 4) // factor, (base * 2) // factor, base]
    for k in range(4):
        skip = e[3 - k]
        if bilinear:
            cin = low + skip
            mid = cin // 2
        else:
            # ConvTranspose2d(in, in//2) where in = 2*skip
            cin = (low // 2) + skip
            mid = outs[k]
        dec.append((f"up{k + 1}", low, skip, mid, outs[k]))
        low = outs[k]
    return enc, dec


def param_spec(n_channels: int, n_classes: int, base: int = 64,
               bilinear: bool = True) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """state_dict key -> (shape, kind) in the reference's registration order.

    kind in {conv_w, conv_b, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, convT_w, convT_b}.
    """
    enc, dec = channel_plan(n_channels, base, bilinear)
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()

    def double_conv(prefix, cin, cmid, cout):
        for idx, (ci, co) in ((0, (cin, cmid)), (3, (cmid, cout))):
            spec[f"{prefix}.{idx}.weight"] = ((co, ci, 3, 3), "conv_w")
            spec[f"{prefix}.{idx}.bias"] = ((co,), "conv_b")
            b = idx + 1
            spec[f"{prefix}.{b}.weight"] = ((co,), "bn_w")
            spec[f"{prefix}.{b}.bias"] = ((co,), "bn_b")
            spec[f"{prefix}.{b}.running_mean"] = ((co,), "bn_rm")
            spec[f"{prefix}.{b}.running_var"] = ((co,), "bn_rv")
            spec[f"{prefix}.{b}.num_batches_tracked"] = ((), "bn_nbt")

    for prefix, cin, cmid, cout in enc:
        double_conv(prefix, cin, cmid, cout)
    for prefix, low, skip, mid, cout in dec:
        if bilinear:
            double_conv(f"{prefix}.conv.double_conv", low + skip, mid, cout)
        else:
            spec[f"{prefix}.up.weight"] = ((low, low // 2, 2, 2), "convT_w")
            spec[f"{prefix}.up.bias"] = ((low // 2,), "convT_b")
            double_conv(f"{prefix}.conv.double_conv", low // 2 + skip, mid, cout)
    spec["outc.conv.weight"] = ((n_classes, base, 1, 1), "conv_w")
    spec["outc.conv.bias"] = ((n_classes,), "conv_b")
    return spec


def is_trainable(kind: str) -> bool:
    return kind in ("conv_w", "conv_b", "bn_w", "bn_b", "convT_w", "convT_b")


# --------------------------------------------------------------------------- #
# deterministic, framework-independent generator (fixtures never ship weights)
# --------------------------------------------------------------------------- #
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def hash_uniform(n: int, seed: int, stream: int) -> np.ndarray:
    """n float64 values in [0,1): u[i] = splitmix64(i + 2^32*stream + 2^48*seed) >> 11 / 2^53."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = idx + (np.uint64(stream) << np.uint64(32)) + (np.uint64(seed) << np.uint64(48))
        z = _splitmix64(key)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_state(n_channels: int, n_classes: int, base: int = 64, bilinear: bool = True,
               seed: int = 0, nontrivial_bn: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic parameters.  conv: U(+-1/sqrt(fan_in)) like torch's default init
    (kaiming_uniform(a=sqrt(5))), BN gamma U(0.5,1.5) / beta U(-0.2,0.2) when
    nontrivial_bn (so affine paths are exercised), else ones/zeros."""
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for stream, (name, (shape, kind)) in enumerate(param_spec(n_channels, n_classes, base, bilinear).items()):
        n = int(np.prod(shape)) if len(shape) else 1
        if kind in ("conv_w", "convT_w"):
            fan_in = shape[1] * shape[2] * shape[3] if kind == "conv_w" else shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)
            v = (hash_uniform(n, seed, stream) * 2.0 - 1.0) * bound
        elif kind in ("conv_b", "convT_b"):
            v = (hash_uniform(n, seed, stream) * 2.0 - 1.0) * 0.1
        elif kind == "bn_w":
            v = 0.5 + hash_uniform(n, seed, stream) if nontrivial_bn else np.ones(n)
        elif kind == "bn_b":
            v = (hash_uniform(n, seed, stream) - 0.5) * 0.4 if nontrivial_bn else np.zeros(n)
        elif kind == "bn_rm":
            v = np.zeros(n)
        elif kind == "bn_rv":
            v = np.ones(n)
        elif kind == "bn_nbt":
            st[name] = torch.zeros((), dtype=torch.int64)
            continue
        else:
            raise AssertionError(kind)
        st[name] = torch.from_numpy(v.astype(np.float32)).reshape(shape).clone()
    return st


def make_batch(B: int, C: int, H: int, W: int, seed: int = 1, n_label_values: int = 2,
               all_ignored_sample: Optional[int] = None, ignore_value: int = 0,
               extra: Tuple[str, ...] = ()) -> Dict[str, torch.Tensor]:
    """Synthetic tile batch shaped like Floodplanet_Dataset.__getitem__ output
    (floodplanet.py:644-648): image f32 [B,C,H,W] in [0,1), target i64 [B,H,W] made of
    smooth blobs over {0..n_label_values-1}."""
    img = hash_uniform(B * C * H * W, seed, 1).astype(np.float32).reshape(B, C, H, W)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    tgt = np.zeros((B, H, W), dtype=np.int64)
    ph = hash_uniform(B * 6, seed, 2).reshape(B, 6)
    for b in range(B):
        f = (np.sin(yy * (0.11 + 0.2 * ph[b, 0]) + 6.28 * ph[b, 1])
             + np.cos(xx * (0.09 + 0.2 * ph[b, 2]) + 6.28 * ph[b, 3])
             + np.sin((xx + yy) * (0.05 + 0.1 * ph[b, 4]) + 6.28 * ph[b, 5]))
        q = (f - f.min()) / (f.max() - f.min() + 1e-9)
        tgt[b] = np.minimum((q * n_label_values).astype(np.int64), n_label_values - 1)
    if all_ignored_sample is not None:
        tgt[all_ignored_sample] = ignore_value
    batch = {"image": torch.from_numpy(img), "target": torch.from_numpy(tgt)}
    for j, key in enumerate(extra):
        e = hash_uniform(B * H * W, seed, 10 + j).astype(np.float32).reshape(B, 1, H, W)
        batch[key] = torch.from_numpy(e)
    return batch


def make_task_tiles(n: int, C: int, S: int, seed: int, signal: float = 0.07) -> Dict[str, torch.Tensor]:
    """A segmentation task that has to be LEARNT (the `mIoU vs ref` workload, oracle/make_miou_golden.py): n tiles
    [C, S, S].  Target: smooth blobs over {0, 1, 2}; 0 is the ignore class (conf/config.yaml:26), 1 and 2 have to be told
    apart.  Image: U[0,1) noise + `signal` * (a fixed +-1 band signature of the pixel's class), the whole tile multiplied by
    a smooth illumination field in [0.7, 1.3] -- per pixel and band the two classes differ by 2*signal against a noise
    std of 0.29, so a pixel-wise linear rule reaches ~88 % and spatial context is needed for more.  Closed form in
    (n, C, S, seed): both sides of a comparison build identical tiles from the seed alone."""
    b = make_batch(n, C, S, S, seed=seed, n_label_values=3)
    img, tgt = b["image"].numpy().copy(), b["target"].numpy()
    sig = np.where(hash_uniform(3 * C, 977, 5).reshape(3, C) < 0.5, -1.0, 1.0).astype(np.float32)
    sig[0] = 0.0
    img += np.float32(signal) * np.transpose(sig[tgt], (0, 3, 1, 2))
    yy, xx = np.meshgrid(np.arange(S, dtype=np.float32), np.arange(S, dtype=np.float32), indexing="ij")
    ph = hash_uniform(n * 4, seed, 6).reshape(n, 4).astype(np.float32)
    for k in range(n):
        gain = 1.0 + 0.3 * np.sin(yy * (0.02 + 0.05 * ph[k, 0]) + 6.28 * ph[k, 1]) * np.cos(xx * (0.02 + 0.05 * ph[k, 2]) + 6.28 * ph[k, 3])
        img[k] *= gain.astype(np.float32)[None]
    return {"image": torch.from_numpy(img.astype(np.float32)), "target": torch.from_numpy(tgt)}


# --------------------------------------------------------------------------- #
# forward
# --------------------------------------------------------------------------- #
def _conv_bn_relu(x, st, prefix, idx, training):
    w, b = st[f"{prefix}.{idx}.weight"], st[f"{prefix}.{idx}.bias"]
    y = F.conv2d(x, w, b, stride=1, padding=1)                      # unet.py:14,16
    bn = f"{prefix}.{idx + 1}"
    if training:
        # nn.BatchNorm2d.forward: bump the counter, then batch statistics
        st[f"{bn}.num_batches_tracked"] += 1
    y = F.batch_norm(y, st[f"{bn}.running_mean"], st[f"{bn}.running_var"],
                     st[f"{bn}.weight"], st[f"{bn}.bias"], training, BN_MOMENTUM, BN_EPS)
    return F.relu(y)                                                # unet.py:15,17


def _double_conv(x, st, prefix, training):
    x = _conv_bn_relu(x, st, prefix, 0, training)
    return _conv_bn_relu(x, st, prefix, 3, training)


def assemble_input(batch: Dict[str, torch.Tensor], early_fusion: bool) -> torch.Tensor:
    """water_seg_model.py:87-90 (image only) or ef_model.py:24-47 (concat extras in fixed order)."""
    x = batch["image"]
    if early_fusion:
        for key in EF_EXTRA_KEYS:
            if key in batch:
                x = torch.concat([x, batch[key]], dim=1)
    return x


def unet_forward(st: Dict[str, torch.Tensor], x: torch.Tensor, training: bool,
                 bilinear: bool = True) -> torch.Tensor:
    """unet.py:100-111.  Mutates BN running buffers in `st` when training (as nn.BatchNorm2d does)."""
    feats = [_double_conv(x, st, "inc.double_conv", training)]
    for i in range(1, 5):
        p = F.max_pool2d(feats[-1], 2)                              # unet.py:29
        feats.append(_double_conv(p, st, f"down{i}.maxpool_conv.1.double_conv", training))
    cur = feats[4]
    for k in range(4):
        skip = feats[3 - k]
        if bilinear:
            up = F.interpolate(cur, scale_factor=2, mode="bilinear", align_corners=True)  # unet.py:43-45
        else:
            up = F.conv_transpose2d(cur, st[f"up{k + 1}.up.weight"], st[f"up{k + 1}.up.bias"], stride=2)
        dy = skip.shape[2] - up.shape[2]
        dx = skip.shape[3] - up.shape[3]
        up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])                   # unet.py:57-62
        cur = _double_conv(torch.cat([skip, up], dim=1), st, f"up{k + 1}.conv.double_conv", training)  # :66-67
    return F.conv2d(cur, st["outc.conv.weight"], st["outc.conv.bias"])                   # unet.py:74-77


def ce_loss(logits: torch.Tensor, target: torch.Tensor, ignore_index: int) -> torch.Tensor:
    """water_seg_model.py:103-106: CrossEntropyLoss(ignore_index) (mean over non-ignored pixels),
    NaN (all pixels ignored) replaced by 0 through nan_to_num."""
    loss = F.cross_entropy(logits, target, ignore_index=ignore_index)
    if torch.isnan(loss):
        loss = torch.nan_to_num(loss)
    return loss


def bce_dice_loss(logits: torch.Tensor, target: torch.Tensor, ignore_index: int, dice_weight: float = 1.0):
    """North-star EXTENSION, PARITY UNPINNED BY THE REFERENCE (the reference has no Dice/BCE anywhere,
    SURVEY.md section 0.1): per-pixel binary cross entropy + soft Dice on p = softmax(logits)[:, 1] (flood class)
    against t = [target == 1], over pixels whose target != ignore_index:
        BCE  = -(1/N) sum [t log p + (1-t) log(1-p)]
        Dice = 1 - (2 sum p t + 1) / (sum p + sum t + 1)
    all-ignored batch -> 0.  This definition IS the specification the HIP kernels are tested against."""
    logp_all = F.log_softmax(logits, dim=1)
    p = logp_all[:, 1].exp()
    logp = logp_all[:, 1]
    others = torch.cat([logits[:, :1], logits[:, 2:]], dim=1)
    log1mp = torch.logsumexp(others, dim=1) - torch.logsumexp(logits, dim=1)
    valid = (target != ignore_index) & (target >= 0) & (target < logits.shape[1])
    n = valid.sum()
    if int(n) == 0:
        return logits.sum() * 0.0
    t = (target == 1).to(logits.dtype)
    v = valid.to(logits.dtype)
    bce = -((t * logp + (1 - t) * log1mp) * v).sum() / n
    inter = (p * t * v).sum()
    dice = 1 - (2 * inter + 1) / ((p * v).sum() + (t * v).sum() + 1)
    return bce + dice_weight * dice


def resolve_ignore_index(ignore_index: int, n_classes: int) -> int:
    """water_seg_model.py:35-36."""
    return n_classes - 1 if ignore_index == -1 else ignore_index


# --------------------------------------------------------------------------- #
# metrics (argmax confusion counters; formulas restate torchmetrics' micro average)
# --------------------------------------------------------------------------- #
def confusion_counts(pred: torch.Tensor, target: torch.Tensor, n_classes: int,
                     ignore_index: Optional[int]) -> np.ndarray:
    """[n_classes, n_classes] int64 matrix M[t, p] over pixels whose target != ignore_index."""
    p = pred.reshape(-1).numpy().astype(np.int64)
    t = target.reshape(-1).numpy().astype(np.int64)
    keep = np.ones_like(t, dtype=bool) if ignore_index is None else (t != ignore_index)
    keep &= (t >= 0) & (t < n_classes)
    m = np.bincount(t[keep] * n_classes + p[keep], minlength=n_classes * n_classes)
    return m.reshape(n_classes, n_classes)


def metrics_from_counts(m: np.ndarray, ignore_index: Optional[int] = None) -> Dict[str, float]:
    """Micro-averaged multiclass F1 / Jaccard / Accuracy from a confusion matrix M[target, pred] whose
    ignored-target pixels were already dropped (water_seg_model.py:46-63: MetricCollection of
    F1Score / JaccardIndex / Accuracy, task="multiclass", average="micro", ignore_index).
    PARITY UNPINNED: torchmetrics (third party, pinned 0.10.0 in environment.yml:188, code needs >=0.11) is
    absent here; the formulas restate its reductions:
      * stat-scores micro (F1, Accuracy): tp = trace, fp = fn = total - tp;
      * `_jaccard_index_reduce(confmat, average="micro", ignore_index)`: num = sum(diag),
        denom = sum(union) - union[ignore_index] when 0 <= ignore_index < n_classes, union = rows + cols - diag.
        (Without predictions in the ignore class this equals tp / (tp + fp + fn).)
    floodplanet_code_amd/metrics.py is the product's statement of the same rule; tests/test_models_api.py checks
    that the two agree, including on a matrix with predictions in the ignore class."""
    m = np.asarray(m, dtype=np.float64)
    n = m.shape[0]
    tp = float(np.trace(m))
    tot = float(m.sum())
    fp = fn = tot - tp
    f1 = 2 * tp / (2 * tp + fp + fn) if tot > 0 else 0.0
    acc = tp / tot if tot > 0 else 0.0
    union = m.sum(0) + m.sum(1) - np.diag(m)
    denom = float(union.sum())
    if ignore_index is not None and 0 <= ignore_index < n:
        denom -= float(union[ignore_index])
    jac = tp / denom if denom > 0 else 0.0
    return {"MulticlassF1Score": f1, "MulticlassJaccardIndex": jac, "MulticlassAccuracy": acc}


# --------------------------------------------------------------------------- #
# tile assembly: normalise, edge-crop buffer, channel concat (base_dataset.py:77-113, :271-325; ef_model.py:28-44)
# --------------------------------------------------------------------------- #
def assemble_case_sources(case):
    """The raw tiles of a case of oracle/make_assemble_golden.py from the closed-form generator (seeds only)."""
    out = []
    for k, (name, ch, scale, shift) in enumerate(case["sources"]):
        n = case["B"] * ch * case["H"] * case["W"]
        x = hash_uniform(n, 1000 + k, sum(map(ord, case["name"]))).astype(np.float32)
        x = x.reshape(case["B"], ch, case["H"], case["W"])
        out.append((x * np.float32(scale) + np.float32(shift)).astype(np.float32))
    return out


def assemble_global_params(case):
    return {name: {"mean": (np.arange(ch, dtype=np.float32) * np.float32(0.1 * scale) + np.float32(shift)),
                   "std": (np.float32(scale) * (np.float32(0.5) + np.arange(ch, dtype=np.float32) * np.float32(0.05)))}
            for (name, ch, scale, shift) in case["sources"]}


def assemble_tiles(sources, valid, norm_mode, global_params=None, names=None, pad_value=0.0):
    """sources: list of float32 [B, C_k, H, W]; valid: [(h_b, w_b)] -> (image [B, sum C, H, W], mean, std [B, sum C]).
    Per sample and source: crop -> `normalize` (None: mean 0 / std 1; 'local': per-channel mean and population std of the
    crop, `flat.mean(axis=1)` / `flat.std(axis=1)`; 'global': the dataset's parameters; then `image -= mean; image /= std`,
    base_dataset.py:95-111) -> `_add_buffer_to_image` (constant 0 canvas of the nominal size, crop in the top-left corner,
    :306-320) -> concatenation along the channel axis.  PINNED: tests/golden/assemble_golden.npz was produced by the
    reference's own two methods (oracle/make_assemble_golden.py)."""
    B, _, H, W = sources[0].shape
    ctot = sum(s.shape[1] for s in sources)
    image = np.full((B, ctot, H, W), pad_value, dtype=np.float32)
    mean = np.zeros((B, ctot), dtype=np.float32)
    std = np.ones((B, ctot), dtype=np.float32)
    for b in range(B):
        vh, vw = valid[b]
        c0 = 0
        for k, x in enumerate(sources):
            ch = x.shape[1]
            crop = x[b, :, :vh, :vw].copy()
            if norm_mode == "local":
                flat = crop.reshape(ch, vh * vw)
                m, s = flat.mean(axis=1), flat.std(axis=1)
            elif norm_mode == "global":
                m, s = global_params[names[k]]["mean"], global_params[names[k]]["std"]
            elif norm_mode is None:
                m, s = np.zeros(ch, dtype=crop.dtype), np.ones(ch, dtype=crop.dtype)
            else:
                raise NotImplementedError(f'Normalization mode "{norm_mode}" not implemented.')
            crop -= m[:, None, None]
            crop /= s[:, None, None]
            image[b, c0:c0 + ch, :vh, :vw] = crop
            mean[b, c0:c0 + ch] = m
            std[b, c0:c0 + ch] = s
            c0 += ch
    return image, mean, std


# --------------------------------------------------------------------------- #
# overlap-average stitching of tile predictions (utils/utils_image.py:363-494 as predict.py:296-334 drives it)
# --------------------------------------------------------------------------- #
def stitch_reference(logits: np.ndarray, boxes, H: int, W: int):
    """ImageStitcher_v2.add_image / _combine_images restated: per crop i with box (h0, w0, hE, wE), softmax of the
    [c, h, w] logits over c (scipy.special.softmax on the float32 'h w c' array, predict.py:298-301) is added into a
    float32 canvas [H, W, c] at [h0:hE, w0:wE] (cut to dh x dw, utils_image.py:455-461), a float64 weight canvas counts
    contributions, the result is canvas / (weight[:, :, None] + 1e-5) (float64) and nan_to_num (:472-489).
    PINNED: tests/golden/stitch_*.npz were produced by the reference's own class (oracle/make_stitch_golden.py)."""
    n, c = logits.shape[0], logits.shape[1]
    canvas = np.zeros([H, W, c], dtype=np.float32)
    weight = np.zeros([H, W], dtype="float")
    for i, (h0, w0, hE, wE) in enumerate(boxes):
        x = logits[i].transpose(1, 2, 0).astype(np.float32)
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        pred = e / e.sum(axis=-1, keepdims=True)
        dh, dw = hE - h0, wE - w0
        canvas[h0:hE, w0:wE, :] += pred[:dh, :dw, :]
        weight[h0:hE, w0:wE] += np.ones([dh, dw], dtype="float")
    out = np.nan_to_num(canvas / (weight[:, :, None] + 1e-5))
    return out, weight


# --------------------------------------------------------------------------- #
# augmentation (datasets/base_dataset.py:494-555 -> torchvision F.hflip / F.vflip / F.rotate)
# --------------------------------------------------------------------------- #
def augment(image: np.ndarray, target: np.ndarray, flag: int, angle_deg: float, target_fill: int = 0):
    """hflip (flag & 1) -> vflip (& 2) -> rotate (& 4) of ONE sample, image [C,H,W], target [H,W].
    PARITY UNPINNED: torchvision is absent; the rotate restates its tensor path (v0.13, the version range of
    environment.yml): _get_inverse_affine_matrix(centre 0, -angle) -> affine_grid with half-pixel centres ->
    grid_sample(nearest, align_corners=False, zeros): source = R(angle) * (dst + 0.5 - size/2) + size/2 - 0.5,
    rounded half-to-even, zero fill outside."""
    img, tgt = image, target
    if flag & 1:
        img, tgt = img[:, :, ::-1], tgt[:, ::-1]
    if flag & 2:
        img, tgt = img[:, ::-1, :], tgt[::-1, :]
    if flag & 4:
        C, H, W = img.shape
        th = np.float32(angle_deg) * np.float32(0.017453292519943295)
        # cos / sin in double, rounded once to float32 (the HIP kernel does the same: both sides then hold the correctly
        # rounded float32 value, where float32 cos implementations differ in the last ulp); every later operation is
        # one float32 rounding, left to right -- the kernel replays exactly this sequence (test: zero index mismatches)
        cs, sn = np.float32(np.cos(np.float64(th))), np.float32(np.sin(np.float64(th)))
        oy, ox = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
        xc = ox + np.float32(0.5) - np.float32(0.5 * W)
        yc = oy + np.float32(0.5) - np.float32(0.5 * H)
        xs = cs * xc - sn * yc + np.float32(0.5 * W) - np.float32(0.5)
        ys = sn * xc + cs * yc + np.float32(0.5 * H) - np.float32(0.5)
        sx, sy = np.rint(xs).astype(np.int64), np.rint(ys).astype(np.int64)
        inside = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
        sxc, syc = np.clip(sx, 0, W - 1), np.clip(sy, 0, H - 1)
        img = np.where(inside[None], np.ascontiguousarray(img)[:, syc, sxc], 0).astype(image.dtype)
        tgt = np.where(inside, np.ascontiguousarray(tgt)[syc, sxc], target_fill).astype(target.dtype)
    return np.ascontiguousarray(img), np.ascontiguousarray(tgt)


# --------------------------------------------------------------------------- #
# training step (what Lightning's automatic optimisation does around training_step)
# --------------------------------------------------------------------------- #
def trainable_names(st: Dict[str, torch.Tensor], n_channels=None) -> List[str]:
    return [k for k in st if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def new_adam_state(st: Dict[str, torch.Tensor]) -> Dict[str, object]:
    names = trainable_names(st)
    return {"step": 0,
            "m": {k: torch.zeros_like(st[k]) for k in names},
            "v": {k: torch.zeros_like(st[k]) for k in names}}


def adam_update(st, grads, opt, lr):
    """torch.optim.Adam single-tensor update order (water_seg_model.py:200)."""
    opt["step"] += 1
    t = opt["step"]
    bc1 = 1.0 - ADAM_BETA1 ** t
    bc2 = 1.0 - ADAM_BETA2 ** t
    step_size = lr / bc1
    bc2_sqrt = bc2 ** 0.5
    for k, g in grads.items():
        m, v = opt["m"][k], opt["v"][k]
        m.lerp_(g, 1.0 - ADAM_BETA1)
        v.mul_(ADAM_BETA2).addcmul_(g, g, value=1.0 - ADAM_BETA2)
        denom = (v.sqrt() / bc2_sqrt).add_(ADAM_EPS)
        st[k].addcdiv_(m, denom, value=-step_size)


def loss_and_grads(st, batch, ignore_index, bilinear=True, early_fusion=False,
                   training=True, loss_kind="ce", dice_weight=1.0):
    """forward + loss + autograd backward.  Returns (logits, loss, grads dict)."""
    names = trainable_names(st)
    leaves = {}
    work = dict(st)
    for k in names:
        leaves[k] = st[k].detach().clone().requires_grad_(True)
        work[k] = leaves[k]
    x = assemble_input(batch, early_fusion)
    logits = unet_forward(work, x, training, bilinear)
    # running buffers were updated in `work` (same tensor objects as st) -> nothing to copy back
    if loss_kind == "ce":
        loss = ce_loss(logits, batch["target"], ignore_index)
    else:
        loss = bce_dice_loss(logits, batch["target"], ignore_index, dice_weight)
    loss.backward()
    grads = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k]))
             for k in names}
    return logits.detach(), loss.detach(), grads


def train_step(st, opt, batch, ignore_index, lr, bilinear=True, early_fusion=False):
    """One optimiser step: zero_grad -> training_step -> backward -> Adam.step (fit.py:95-97)."""
    logits, loss, grads = loss_and_grads(st, batch, ignore_index, bilinear, early_fusion, True)
    with torch.no_grad():
        adam_update(st, grads, opt, lr)
    return logits, loss, grads


def eval_forward(st, batch, bilinear=True, early_fusion=False):
    with torch.no_grad():
        return unet_forward(dict(st), assemble_input(batch, early_fusion), False, bilinear)


# --------------------------------------------------------------------------- #
# algorithmic work: conv MACs x 2 only
# --------------------------------------------------------------------------- #
def conv_flops_per_tile(n_channels, H, W, base=64, bilinear=True, n_classes=3):
    """(fwd_flops, train_flops) per tile; train = 3*fwd - dgrad(first conv)."""
    enc, dec = channel_plan(n_channels, base, bilinear)
    fwd = 0.0
    h, w = H, W
    sizes = [(H, W)]
    for i in range(1, 5):
        h, w = h // 2, w // 2
        sizes.append((h, w))
    first = None
    for i, (_, cin, cmid, cout) in enumerate(enc):
        h, w = sizes[i]
        f1 = 2.0 * 9 * cin * cmid * h * w
        f2 = 2.0 * 9 * cmid * cout * h * w
        if first is None:
            first = f1
        fwd += f1 + f2
    for k, (_, low, skip, mid, cout) in enumerate(dec):
        h, w = sizes[3 - k]
        if bilinear:
            cin = low + skip
        else:
            cin = low // 2 + skip
            hl, wl = sizes[4 - k]
            fwd += 2.0 * 4 * low * (low // 2) * hl * wl
        fwd += 2.0 * 9 * cin * mid * h * w + 2.0 * 9 * mid * cout * h * w
    fwd += 2.0 * base * n_classes * H * W
    return fwd, 3.0 * fwd - first


# --------------------------------------------------------------------------- #
# Late fusion (st_water_seg/models/lf_model.py:29-92, feat_fusion='concat_conv')
# --------------------------------------------------------------------------- #
LF_FORWARD_ORDER = ("ms_image", "dem", "slope", "preflood", "pre_post_difference", "hand")   # lf_model.py:56-76
LF_BATCH_KEY = {"ms_image": "image"}                                                          # encoder name -> batch key


def lf_param_spec(in_channels: "OrderedDict[str, int]", n_classes: int, base: int = 64):
    """state_dict key -> (shape, kind) in the reference's registration order (lf_model.py:31-45): encoders in
    in_channels order, decoder, concat_convs."""
    spec: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    for name, ch in in_channels.items():
        for k, v in param_spec(ch, n_classes, base, True).items():
            if k.startswith(("inc.", "down")):
                spec[f"encoders.{name}.{k}"] = v
    for k, v in param_spec(1, n_classes, base, True).items():
        if k.startswith(("up", "outc.")):
            spec[f"decoder.{k}"] = v
    n = len(in_channels)
    for l, fs in enumerate([base, base * 2, base * 4, base * 8, base * 8]):        # [64,128,256,512,512] at base 64
        spec[f"concat_convs.{l}.weight"] = ((fs, fs * n, 1, 1), "conv_w")
        spec[f"concat_convs.{l}.bias"] = ((fs,), "conv_b")
    return spec


def lf_make_state(in_channels, n_classes: int, base: int = 64, seed: int = 0):
    """Deterministic parameters for the late-fusion net (same generator and value ranges as make_state)."""
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for stream, (name, (shape, kind)) in enumerate(lf_param_spec(in_channels, n_classes, base).items()):
        n = int(np.prod(shape)) if len(shape) else 1
        u = hash_uniform(n, seed + 1000, stream)
        if kind == "conv_w":
            v = (u * 2.0 - 1.0) / math.sqrt(shape[1] * shape[2] * shape[3])
        elif kind == "conv_b":
            v = (u * 2.0 - 1.0) * 0.1
        elif kind == "bn_w":
            v = 0.5 + u
        elif kind == "bn_b":
            v = (u - 0.5) * 0.4
        elif kind == "bn_rm":
            v = np.zeros(n)
        elif kind == "bn_rv":
            v = np.ones(n)
        elif kind == "bn_nbt":
            st[name] = torch.zeros((), dtype=torch.int64)
            continue
        else:
            raise AssertionError(kind)
        st[name] = torch.from_numpy(v.astype(np.float32)).reshape(shape).clone()
    return st


def _encoder_forward(st, x, prefix, training):
    """unet.py:151-159: [x1..x5]."""
    feats = [_double_conv(x, st, f"{prefix}inc.double_conv", training)]
    for i in range(1, 5):
        feats.append(_double_conv(F.max_pool2d(feats[-1], 2), st, f"{prefix}down{i}.maxpool_conv.1.double_conv",
                                  training))
    return feats


def _decoder_forward(st, feats, prefix, training):
    """unet.py:186-192 (bilinear)."""
    cur = feats[4]
    for k in range(4):
        skip = feats[3 - k]
        up = F.interpolate(cur, scale_factor=2, mode="bilinear", align_corners=True)
        dy, dx = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]
        up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        cur = _double_conv(torch.cat([skip, up], dim=1), st, f"{prefix}up{k + 1}.conv.double_conv", training)
    return F.conv2d(cur, st[f"{prefix}outc.conv.weight"], st[f"{prefix}outc.conv.bias"])


def lf_forward(st, batch, in_channels, training: bool):
    """lf_model.py:54-92: encode every input, concatenate the features level by level in the fixed order, fuse each
    level with its 1x1 conv, decode."""
    feats = None
    for name in LF_FORWARD_ORDER:
        if name not in in_channels:
            continue
        f = _encoder_forward(st, batch[LF_BATCH_KEY.get(name, name)], f"encoders.{name}.", training)
        feats = f if feats is None else [torch.concat([a, b], dim=1) for a, b in zip(feats, f)]
    fused = [F.conv2d(x, st[f"concat_convs.{l}.weight"], st[f"concat_convs.{l}.bias"]) for l, x in enumerate(feats)]
    return _decoder_forward(st, fused, "decoder.", training)


def lf_loss_and_grads(st, batch, in_channels, ignore_index, training=True):
    names = trainable_names(st)
    leaves, work = {}, dict(st)
    for k in names:
        leaves[k] = st[k].detach().clone().requires_grad_(True)
        work[k] = leaves[k]
    logits = lf_forward(work, batch, in_channels, training)
    loss = ce_loss(logits, batch["target"], ignore_index)
    loss.backward()
    grads = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])) for k in names}
    return logits.detach(), loss.detach(), grads


def lf_train_step(st, opt, batch, in_channels, ignore_index, lr):
    logits, loss, grads = lf_loss_and_grads(st, batch, in_channels, ignore_index, True)
    with torch.no_grad():
        adam_update(st, grads, opt, lr)
    return logits, loss, grads


def lf_eval_forward(st, batch, in_channels):
    with torch.no_grad():
        return lf_forward(dict(st), batch, in_channels, False)
