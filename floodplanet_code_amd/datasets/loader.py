"""Batches of FloodplanetTiles on the training device.

Two paths:
  * host assembly (default): collate on the host (optionally in DataLoader workers, as fit.py:56-63 configures), one
    host->device copy per batch of finished tiles;
  * device assembly (`device_assembly=True`): the workers ship the RAW crops (scaled, un-normalised, un-padded) and their
    valid sizes; the per-tile normalisation (base_dataset.py:77-113), the zero padding of edge crops to the nominal tile
    (:271-325) and -- with `extra_sources` -- the channel concatenation of further inputs run on the whole batch in HBM
    through `datasets.assemble.assemble_tiles` (C ABI fu_assemble_tiles), instead of per item on the CPU.
Either way the training split then gets the reference's hflip / vflip / rotate transforms (base_dataset.py:494-555,
conf/config.yaml:41-52) on the whole batch in HBM through `floodplanet_code_amd.augment`."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .floodplanet import RawTileView, collate_raw_tiles, collate_tiles

__all__ = ["TileLoader"]


class TileLoader:
    def __init__(self, dataset, batch_size: int, device, shuffle: bool = False, seed: int = 0, drop_last: bool = False,
                 num_workers: int = 0, transforms: Optional[dict] = None, ignore_index: int = 0,
                 device_assembly: bool = False):
        """transforms: None, or the reference's `transforms` config dict ({} = its defaults) -> GPU augmentation."""
        self.dataset, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        self.transforms, self.ignore_index = transforms, ignore_index
        self.device_assembly = bool(device_assembly)
        if self.device_assembly and self.device.type != "cuda":
            raise RuntimeError("device_assembly=True runs fu_assemble_tiles on a ROCm GPU; there is no CPU fallback")
        self._rng = np.random.RandomState(seed)
        g = torch.Generator().manual_seed(seed)
        src = RawTileView(dataset) if self.device_assembly else dataset
        self._dl = torch.utils.data.DataLoader(src, batch_size=batch_size, shuffle=shuffle, drop_last=drop_last,
                                               num_workers=num_workers, generator=g,
                                               collate_fn=collate_raw_tiles if self.device_assembly else collate_tiles,
                                               pin_memory=self.device.type == "cuda")

    def __len__(self):
        return len(self._dl)

    def _assemble(self, batch):
        from .assemble import assemble_tiles
        raw = batch["raw"].to(self.device, non_blocking=True)
        vh = batch["valid_h"].to(self.device, non_blocking=True)
        vw = batch["valid_w"].to(self.device, non_blocking=True)
        image, mean, std = assemble_tiles([raw], self.dataset.norm_mode, (vh, vw))
        out = {"image": image, "mean": mean, "std": std, "target": batch["target"].to(self.device, non_blocking=True)}
        if "metadata" in batch:
            out["metadata"] = batch["metadata"]
        return out

    def __iter__(self):
        for batch in self._dl:
            if self.device_assembly:
                out = self._assemble(batch)
            else:
                out = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            if self.transforms is not None:
                from .. import augment
                flags, angles = augment.sample_transforms(out["image"].shape[0], self.transforms, self._rng)
                out["image"], out["target"] = augment.apply(out["image"], out["target"], flags, angles,
                                                            target_fill=self.ignore_index)
            yield out
