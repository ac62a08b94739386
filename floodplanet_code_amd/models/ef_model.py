"""EarlyFusionModel -- drop-in for st_water_seg/models/ef_model.py:6-47: the same UNet, whose input is the
image with the optional auxiliary maps concatenated along C in the fixed order
dem, slope, preflood, pre_post_difference, hand (ef_model.py:28-44)."""
from __future__ import annotations

import torch

from .water_seg_model import WaterSegmentationModel

EXTRA_KEYS = ('dem', 'slope', 'preflood', 'pre_post_difference', 'hand')


class EarlyFusionModel(WaterSegmentationModel):

    def _gather_input(self, batch):
        images = batch['image']
        extra = [batch[k] for k in EXTRA_KEYS if k in batch]
        if extra:
            images = torch.concat([images] + extra, dim=1)
        return images
