"""GPU tile assembly: per-tile normalisation, edge-crop buffer and multi-sensor channel concatenation of a whole batch in HBM
(C ABI `fu_assemble_tiles`), replacing the per-item CPU work of the reference's `BaseDataset.normalize`
(st_water_seg/datasets/base_dataset.py:77-113), `_add_buffer_to_image` (:271-325) and the channel concatenation of the
fused inputs (ef_model.py:28-44; Planet + Sentinel-1 stacks of BASELINE configs[4]).  The output feeds `fu_augment` /
`fu_forward` directly."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from .. import _lib
from .._lib import check, ptr

NORM_MODES = {None: 0, "local": 1, "global": 2}


def assemble_tiles(sources: Sequence[torch.Tensor], norm_mode: Optional[str] = None,
                   valid_hw: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                   global_params: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, pad_value: float = 0.0):
    """sources: fp32 NCHW [B, C_k, H, W] tiles on a ROCm device (raw crops in the top-left corner of the nominal tile).
    -> (image [B, sum C, H, W], mean [B, sum C, 1, 1], std [B, sum C, 1, 1]) as the item dict of the reference carries them."""
    if norm_mode not in NORM_MODES:
        raise NotImplementedError(f'Normalization mode "{norm_mode}" not implemented.')      # base_dataset.py:106-108
    if not sources or sources[0].device.type != "cuda":
        raise RuntimeError("assemble_tiles runs only on a ROCm GPU; there is no CPU fallback")
    srcs = [s.contiguous().float() for s in sources]
    B, _, H, W = srcs[0].shape
    for s in srcs:
        if s.shape[0] != B or s.shape[2:] != (H, W):
            raise ValueError("all sources must share batch and tile size")
    dev = srcs[0].device
    ctot = sum(s.shape[1] for s in srcs)
    out = torch.empty(B, ctot, H, W, dtype=torch.float32, device=dev)
    mode = NORM_MODES[norm_mode]
    mean = torch.zeros(B, ctot, dtype=torch.float32, device=dev)
    std = torch.ones(B, ctot, dtype=torch.float32, device=dev)
    gm = gs = None
    if mode == 2:
        if global_params is None:
            raise ValueError("norm_mode 'global' needs (mean, std) per channel")
        gm, gs = (t.to(dev).float().contiguous() for t in global_params)
        mean[:] = gm
        std[:] = gs
    vh = vw = None
    if valid_hw is not None:
        vh, vw = (t.to(dev).to(torch.int32).contiguous() for t in valid_hw)
    arr = (C.c_void_p * len(srcs))(*[s.data_ptr() for s in srcs])
    chs = (C.c_int32 * len(srcs))(*[s.shape[1] for s in srcs])
    check(_lib.load().fu_assemble_tiles(arr, chs, len(srcs), B, H, W, ptr(vh), ptr(vw), mode, ptr(gm), ptr(gs),
                                        float(pad_value), ptr(out), ptr(mean) if mode == 1 else None,
                                        ptr(std) if mode == 1 else None, torch.cuda.current_stream(dev).cuda_stream))
    return out, mean.view(B, ctot, 1, 1), std.view(B, ctot, 1, 1)


def resize_lanczos4_tiles(windows: torch.Tensor, iy: torch.Tensor, wy: torch.Tensor, ix: torch.Tensor, wx: torch.Tensor,
                          scale_mode: int = 0) -> torch.Tensor:
    """C ABI fu_resize_lanczos4_tiles: windows fp32 [B, C, wh, ww] + the tap tables of `resize.lanczos4_axis_window`
    ([B, TH, 8] / [B, TW, 8]) on a ROCm device -> the resampled (and sensor-scaled) tiles [B, C, TH, TW]."""
    if windows.device.type != "cuda":
        raise RuntimeError("resize_lanczos4_tiles runs only on a ROCm GPU; the host restatement is datasets.resize")
    dev = windows.device
    win = windows.contiguous().float()
    iy, ix = (t.to(dev).to(torch.int32).contiguous() for t in (iy, ix))
    wy, wx = (t.to(dev).float().contiguous() for t in (wy, wx))
    B, Cc, wh, ww = win.shape
    TH, TW = iy.shape[1], ix.shape[1]
    if iy.shape != (B, TH, 8) or wy.shape != (B, TH, 8) or ix.shape != (B, TW, 8) or wx.shape != (B, TW, 8):
        raise ValueError("tap tables must be [B, tile_h, 8] / [B, tile_w, 8]")
    out = torch.empty(B, Cc, TH, TW, dtype=torch.float32, device=dev)
    check(_lib.load().fu_resize_lanczos4_tiles(ptr(win), B, Cc, wh, ww, ptr(iy), ptr(wy), ptr(ix), ptr(wx), TH, TW,
                                               int(scale_mode), ptr(out), torch.cuda.current_stream(dev).cuda_stream))
    return out
