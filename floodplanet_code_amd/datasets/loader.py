"""Batches of FloodplanetTiles on the training device.

Two paths:
  * host assembly (default): collate on the host (optionally in DataLoader workers, as fit.py:56-63 configures), one
    host->device copy per batch of finished tiles;
  * device assembly (`device_assembly=True`): the workers ship the RAW crops (scaled, un-normalised, un-padded) and their
    valid sizes; the per-tile normalisation (base_dataset.py:77-113), the zero padding of edge crops to the nominal tile
    (:271-325) and -- with `extra_sources` -- the channel concatenation of further inputs run on the whole batch in HBM
    through `datasets.assemble.assemble_tiles` (C ABI fu_assemble_tiles), instead of per item on the CPU.
  * device resampling (`device_resize=True`, with device assembly): the workers do not even resample -- the reference's per-item
    whole-raster Lanczos-4 resize (floodplanet.py:338-340, the CPU hot spot of its loader) becomes a per-tile one on the device
    (C ABI fu_resize_lanczos4_tiles): the worker cuts the window of the source raster the tile touches and makes the tap tables.
Either way the training split then gets the reference's hflip / vflip / rotate transforms (base_dataset.py:494-555,
conf/config.yaml:41-52) on the whole batch in HBM through `floodplanet_code_amd.augment`."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .floodplanet import RawTileView, WindowTileView, collate_raw_tiles, collate_tiles, collate_window_tiles

__all__ = ["TileLoader"]


class TileLoader:
    def __init__(self, dataset, batch_size: int, device, shuffle: bool = False, seed: int = 0, drop_last: bool = False,
                 num_workers: int = 0, transforms: Optional[dict] = None, ignore_index: int = 0,
                 device_assembly: bool = False, device_resize: bool = False):
        """transforms: None, or the reference's `transforms` config dict ({} = its defaults) -> GPU augmentation."""
        self.dataset, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        self.transforms, self.ignore_index = transforms, ignore_index
        self.device_assembly = bool(device_assembly)
        self.device_resize = bool(device_resize)
        if self.device_resize and not self.device_assembly:
            raise ValueError("device_resize=True needs device_assembly=True (the resampled tiles are assembled in HBM)")
        if self.device_assembly and self.device.type != "cuda":
            raise RuntimeError("device_assembly=True runs fu_assemble_tiles on a ROCm GPU; there is no CPU fallback")
        self._rng = np.random.RandomState(seed)
        g = torch.Generator().manual_seed(seed)
        if self.device_resize:
            src, coll = WindowTileView(dataset), collate_window_tiles
        elif self.device_assembly:
            src, coll = RawTileView(dataset), collate_raw_tiles
        else:
            src, coll = dataset, collate_tiles
        self._dl = torch.utils.data.DataLoader(src, batch_size=batch_size, shuffle=shuffle, drop_last=drop_last,
                                               num_workers=num_workers, generator=g, collate_fn=coll,
                                               pin_memory=self.device.type == "cuda",
                                               persistent_workers=num_workers > 0)

    def __len__(self):
        return len(self._dl)

    def _assemble(self, batch):
        from .assemble import assemble_tiles, resize_lanczos4_tiles
        if self.device_resize:
            dev = self.device
            raw = resize_lanczos4_tiles(batch["window"].to(dev, non_blocking=True), batch["iy"].to(dev, non_blocking=True),
                                        batch["wy"].to(dev, non_blocking=True), batch["ix"].to(dev, non_blocking=True),
                                        batch["wx"].to(dev, non_blocking=True), batch["scale_mode"])
        else:
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
