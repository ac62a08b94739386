"""GPU overlap-average stitching of tile predictions -- the canvas arithmetic of the reference's ImageStitcher_v2
(st_water_seg/utils/utils_image.py:364-494) as predict.py:329-347 uses it: every crop's softmax is added into an
[H, W, n_classes] canvas at [h0:hE, w0:wE], a weight canvas counts the contributions, the result is
canvas / (weight + 1e-5).  Here the softmax + accumulate runs on the logits that are still resident in the HIP
context after an eval forward, so predictions never leave HBM until the final map is read."""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import _lib
from ._lib import check, ptr


class GpuImageStitcher:
    def __init__(self, net, device):
        self.net = net                      # HipUNet whose last eval forward produced the crops
        self.device = torch.device(device)
        self.image_canvas: Dict[str, torch.Tensor] = {}
        self.weight_canvas: Dict[str, torch.Tensor] = {}

    def add_image(self, sample: int, image_name: str, crop_info, og_height: int, og_width: int) -> None:
        """crop_info: object or tuple with h0, w0, hE, wE (datasets/utils.py CropParams)."""
        h0, w0, hE, wE = (crop_info if isinstance(crop_info, (tuple, list))
                          else (crop_info.h0, crop_info.w0, crop_info.hE, crop_info.wE))
        k = self.net.n_classes
        if image_name not in self.image_canvas:
            self.image_canvas[image_name] = torch.zeros(og_height, og_width, k, device=self.device)
            self.weight_canvas[image_name] = torch.zeros(og_height, og_width, device=self.device)
        cv, wt = self.image_canvas[image_name], self.weight_canvas[image_name]
        check(_lib.load().fu_stitch_add(self.net._ctx, int(sample), ptr(cv), ptr(wt), og_height, og_width, int(h0),
                                        int(w0), int(hE), int(wE), torch.cuda.current_stream(self.device).cuda_stream))

    def combine(self, image_name: str) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (probabilities [H, W, n_classes], argmax [H, W]); like _combine_images + the argmax of predict.py."""
        cv, wt = self.image_canvas[image_name], self.weight_canvas[image_name]
        am = torch.empty(cv.shape[:2], dtype=torch.int64, device=self.device)
        check(_lib.load().fu_stitch_finalize(ptr(cv), ptr(wt), cv.shape[2], cv.shape[0], cv.shape[1], ptr(am),
                                             torch.cuda.current_stream(self.device).cuda_stream))
        return cv, am
