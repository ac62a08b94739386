"""On-GPU tile augmentation: the hflip / vflip / rotate transforms of the reference's BaseDataset
(st_water_seg/datasets/base_dataset.py:494-555, driven by conf/config.yaml:41-52) applied to a whole batch that is
already resident in HBM, instead of per item in DataLoader workers.

``sample_transforms`` mirrors the reference's draws (one uniform coin per transform, uniform angle); ``apply``
runs the single HIP gather kernel behind ``fu_augment`` on image and target together."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

HFLIP, VFLIP, ROTATE = 1, 2, 4


def sample_transforms(batch_size: int, cfg: Optional[dict] = None,
                      rng: Optional[np.random.RandomState] = None) -> Tuple[np.ndarray, np.ndarray]:
    """flags int32 [B], angles float32 [B] drawn like BaseDataset.sample_transforms (defaults: conf/config.yaml:41-52)."""
    rng = rng or np.random
    cfg = cfg or {}
    h = cfg.get("hflip", {"active": True, "likelihood": 0.5})
    v = cfg.get("vflip", {"active": True, "likelihood": 0.5})
    r = cfg.get("rotate", {"active": True, "likelihood": 0.5, "min_rot_angle": 0, "max_rot_angle": 360})
    flags = np.zeros(batch_size, dtype=np.int32)
    angles = np.zeros(batch_size, dtype=np.float32)
    for b in range(batch_size):
        if h.get("active") and rng.rand() < h["likelihood"]:
            flags[b] |= HFLIP
        if v.get("active") and rng.rand() < v["likelihood"]:
            flags[b] |= VFLIP
        if r.get("active") and rng.rand() < r["likelihood"]:
            flags[b] |= ROTATE
            angles[b] = rng.uniform(r["min_rot_angle"], r["max_rot_angle"], size=1)[0]
    return flags, angles


def apply(image: torch.Tensor, target: Optional[torch.Tensor], flags, angles, target_fill: int = 0):
    """image f32 [B,C,H,W] and target i64 [B,H,W] on a ROCm device -> augmented copies."""
    if image.device.type != "cuda":
        raise RuntimeError("augment.apply runs only on a ROCm GPU; there is no CPU fallback")
    image = image.contiguous().float()
    B, C, H, W = image.shape
    dev = image.device
    f = torch.as_tensor(np.asarray(flags, dtype=np.int32), device=dev)
    a = torch.as_tensor(np.asarray(angles, dtype=np.float32), device=dev)
    out = torch.empty_like(image)
    tgt = target.contiguous().long() if target is not None else None
    tout = torch.empty_like(tgt) if tgt is not None else None
    check(_lib.load().fu_augment(ptr(image), ptr(tgt), ptr(out), ptr(tout), ptr(f), ptr(a), B, C, H, W,
                                 int(target_fill), torch.cuda.current_stream(dev).cuda_stream))
    return out, tout
