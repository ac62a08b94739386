"""Tile grid over a raster (host side).  Mirrors st_water_seg/datasets/utils.py:22-52 (`CropParams`), :55-83
(`generate_image_slice_object`) and :86-209 (`get_crop_slices`): same arguments, same order of the returned crops, same
exceptions -- including the reference's quirk that the remainder crops along the bottom edge are `[h0, w0, rem_h,
crop_height]` (utils.py:199-200 writes crop_height where the width is meant; identical for square crops, which is all
the configs use)."""
from __future__ import annotations

import collections
from typing import List, Optional, Tuple, Union

__all__ = ["CropParams", "generate_image_slice_object", "get_crop_slices"]


class CropParams:
    """utils.py:22-52.  [h0, hE) x [w0, wE) inside an og_height x og_width raster; max_crop_* = nominal tile size."""

    def __init__(self, h0, w0, height, width, og_height, og_width, max_crop_height, max_crop_width):
        self.h0, self.w0 = h0, w0
        self.height, self.width = height, width
        self.hE, self.wE = h0 + height, w0 + width
        self.og_height, self.og_width = og_height, og_width
        self.max_crop_height, self.max_crop_width = max_crop_height, max_crop_width

    def __str__(self) -> str:
        return f"H0: {self.h0} | W0:{self.w0} \nHE: {self.hE} | WE: {self.wE}"


def generate_image_slice_object(height, width=None, stride=None, scale=1):
    """utils.py:55-83: (height, width, scale, stride) with width / stride defaulting to height."""
    ImageSlice = collections.namedtuple("ImageSlice", ["height", "width", "scale", "stride"])
    return ImageSlice(height, height if width is None else width, scale, height if stride is None else stride)


def _count(extent: int, step: int, crop: int) -> int:
    # number of k >= 0 with k*step + crop <= extent
    return 0 if crop > extent else (extent - crop) // step + 1


def get_crop_slices(height: int, width: int, crop_height: int, crop_width: int,
                    step: Optional[Union[int, Tuple[int, int]]] = None, mode: str = "exact") -> List[List[int]]:
    """All crops [h0, w0, h, w] of a height x width raster (utils.py:86-209).
    exact: full crops plus remainder crops at the right / bottom edge (never past the raster);
    over:  fixed-size crops covering the raster (the last row / column reaches past it);
    under: fixed-size crops that fit (may not cover the raster)."""
    if step is not None:
        if type(step) is tuple:
            h_step, w_step = step[0], step[1]
        elif type(step) is int:
            h_step, w_step = step, step
        else:
            raise TypeError(f"Invalid step type: {type(step)}")
        if h_step <= 0:
            raise ValueError(f"Step of size {h_step} is too small.")
        if w_step <= 0:
            raise ValueError(f"Step of size {w_step} is too small.")
        if h_step > height:
            raise ValueError(f"Step of size {h_step} is too large for height {height}")
        if w_step > width:
            raise ValueError(f"Step of size {w_step} is too large for width {width}")
    else:
        h_step, w_step = crop_height, crop_width
    if mode not in ("over", "under", "exact"):
        raise NotImplementedError(f"Invalid mode: {mode}")
    nh, nw = _count(height, h_step, crop_height), _count(width, w_step, crop_width)
    if mode == "over":
        nh, nw = nh + 1, nw + 1
    out = [[i * h_step, j * w_step, crop_height, crop_width] for i in range(nh) for j in range(nw)]
    if mode == "exact":
        rem_h, rem_w = height - nh * h_step, width - nw * w_step
        if rem_w != 0:
            out += [[i * h_step, nw * w_step, crop_height, rem_w] for i in range(nh)]
        if rem_h != 0:
            out += [[nh * h_step, j * w_step, rem_h, crop_height] for j in range(nw)]   # (sic: utils.py:199-200)
        if rem_h != 0 and rem_w != 0:
            out.append([nh * h_step, nw * w_step, rem_h, rem_w])
    return out
