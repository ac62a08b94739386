"""EarlyFusionModel -- drop-in for st_water_seg/models/ef_model.py:6-47: the same UNet, whose input is the
image with the optional auxiliary maps concatenated along C in the fixed order
dem, slope, preflood, pre_post_difference, hand (ef_model.py:28-44).

The reference materialises `torch.concat([images, dem, ...], dim=1)`; here the network takes the tensors as they are
(`HipUNet` accepts a list, C ABI `fu_forward_srcs`) and the channels are gathered by the NCHW -> NHWC conversion in front of
the first conv -- one pass instead of two, no concatenated copy.  `_gather_input` still returns the concatenated tensor for
callers that want it."""
from __future__ import annotations

import torch

from .water_seg_model import WaterSegmentationModel

EXTRA_KEYS = ('dem', 'slope', 'preflood', 'pre_post_difference', 'hand')


class EarlyFusionModel(WaterSegmentationModel):

    def _gather_sources(self, batch):
        return [batch['image']] + [batch[k] for k in EXTRA_KEYS if k in batch]

    def _gather_input(self, batch):
        srcs = self._gather_sources(batch)
        return srcs[0] if len(srcs) == 1 else torch.concat(srcs, dim=1)
