"""Batches of FloodplanetTiles on the training device: collate on the host (optionally in DataLoader workers, as
fit.py:56-63 configures), one host->device copy per batch, then -- for the training split -- the reference's
hflip / vflip / rotate transforms (base_dataset.py:494-555, conf/config.yaml:41-52) on the whole batch in HBM through
`floodplanet_code_amd.augment` instead of per item on the CPU."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .floodplanet import collate_tiles

__all__ = ["TileLoader"]


class TileLoader:
    def __init__(self, dataset, batch_size: int, device, shuffle: bool = False, seed: int = 0, drop_last: bool = False,
                 num_workers: int = 0, transforms: Optional[dict] = None, ignore_index: int = 0):
        """transforms: None, or the reference's `transforms` config dict ({} = its defaults) -> GPU augmentation."""
        self.dataset, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        self.transforms, self.ignore_index = transforms, ignore_index
        self._rng = np.random.RandomState(seed)
        g = torch.Generator().manual_seed(seed)
        self._dl = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, drop_last=drop_last,
                                               num_workers=num_workers, collate_fn=collate_tiles, generator=g,
                                               pin_memory=self.device.type == "cuda")

    def __len__(self):
        return len(self._dl)

    def __iter__(self):
        for batch in self._dl:
            out = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            if self.transforms is not None:
                from .. import augment
                flags, angles = augment.sample_transforms(out["image"].shape[0], self.transforms, self._rng)
                out["image"], out["target"] = augment.apply(out["image"], out["target"], flags, angles,
                                                            target_fill=self.ignore_index)
            yield out
