"""Raster resampling used when an image and its label raster differ in size (st_water_seg/utils/utils_image.py:11-60
`resize_image`: `cv2.resize(..., interpolation=cv2.INTER_LANCZOS4)` per band for images, `cv2.INTER_NEAREST` for
labels, floodplanet.py:573-577).  OpenCV is absent from this image, so this is a restatement of its published
conventions -- **parity unpinned** (SURVEY 8c iv):
  * pixel centres: src = (dst + 0.5) * (src_size / dst_size) - 0.5; taps at floor(src) - 3 .. floor(src) + 4;
  * Lanczos a = 4 weights sin(pi x) sin(pi x / 4) / (pi^2 x^2 / 4), normalised to sum 1; borders replicate;
  * nearest: src = min(floor(dst * src_size / dst_size), src_size - 1).
Separable, float32 accumulation in numpy; host side only."""
from __future__ import annotations

import numpy as np

__all__ = ["resize_lanczos4", "resize_nearest", "resize_image", "lanczos4_axis_window", "resize_lanczos4_tile"]


def _lanczos4_axis(n_src: int, n_dst: int):
    scale = n_src / n_dst
    x = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
    x0 = np.floor(x)
    frac = x - x0
    taps = np.arange(-3, 5, dtype=np.float64)                       # 8 taps
    d = frac[:, None] - taps[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(np.abs(d) < 1e-12, 1.0, np.sin(np.pi * d) * np.sin(np.pi * d / 4) / (np.pi * np.pi * d * d / 4))
    w = np.where(np.abs(d) < 4, w, 0.0)
    w /= w.sum(axis=1, keepdims=True)
    idx = np.clip(x0[:, None].astype(np.int64) + taps[None, :].astype(np.int64), 0, n_src - 1)
    return idx, w.astype(np.float32)


def lanczos4_axis_window(n_src: int, n_dst: int, d0: int, d1: int, n_out: int):
    """Rows [d0, d1) of the axis resampled from n_src to n_dst, for a tile of n_out >= d1 - d0 rows: -> (idx int32 [n_out, 8]
    relative to the source window, weights fp32 [n_out, 8], (s0, s1) = the window of the source axis the rows touch).  Rows
    beyond d1 - d0 (a tile at the raster's edge) get zero weights.  n_src == n_dst is the reference's no-op (resize_image
    returns its input): identity tables (weight 1 on the tap at offset 0), not Lanczos weights of offset 0 (sin(pi k) is not
    exactly 0).  The device kernel (C ABI fu_resize_lanczos4_tiles) and `resize_lanczos4_tile` consume these tables."""
    n = d1 - d0
    if not (0 <= d0 <= d1 <= n_dst and n <= n_out):
        raise ValueError(f"rows [{d0}, {d1}) of {n_dst} do not fit a tile of {n_out}")
    idx = np.zeros((n_out, 8), dtype=np.int64)
    w = np.zeros((n_out, 8), dtype=np.float32)
    if n_src == n_dst:
        idx[:n] = np.arange(d0, d1)[:, None]
        w[:n, 3] = 1.0
    else:
        ia, wa = _lanczos4_axis(n_src, n_dst)
        idx[:n], w[:n] = ia[d0:d1], wa[d0:d1]
    s0 = int(idx[:n].min()) if n else 0
    s1 = int(idx[:n].max()) + 1 if n else 1
    idx[:n] -= s0
    return idx.astype(np.int32), w, (s0, s1)


def resize_lanczos4_tile(window: np.ndarray, iy, wy, ix, wx) -> np.ndarray:
    """Host restatement of the device kernel: window [C, wh, ww] + the tables of `lanczos4_axis_window` -> [C, TH, TW] =
    the tile cut out of resize_lanczos4(whole raster) (same taps, same order, fp32: bit for bit)."""
    a = np.asarray(window, dtype=np.float32)
    t = np.zeros((a.shape[0], iy.shape[0], a.shape[2]), dtype=np.float32)
    for k in range(8):
        t += wy[None, :, k, None] * a[:, iy[:, k], :]
    out = np.zeros((a.shape[0], iy.shape[0], ix.shape[0]), dtype=np.float32)
    for k in range(8):
        out += wx[None, None, :, k] * t[:, :, ix[:, k]]
    return out


def resize_lanczos4(image: np.ndarray, height: int, width: int) -> np.ndarray:
    """image [H, W] or [C, H, W] -> same rank at height x width, float32."""
    a = np.asarray(image, dtype=np.float32)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[None]
    if a.ndim != 3:
        raise NotImplementedError(f'Cannot resize image with "{np.ndim(image)}" dimensions.')
    iy, wy = _lanczos4_axis(a.shape[1], height)
    ix, wx = _lanczos4_axis(a.shape[2], width)
    t = np.zeros((a.shape[0], height, a.shape[2]), dtype=np.float32)    # rows: 8 weighted gathers
    for k in range(8):
        t += wy[None, :, k, None] * a[:, iy[:, k], :]
    out = np.zeros((a.shape[0], height, width), dtype=np.float32)       # columns
    for k in range(8):
        out += wx[None, None, :, k] * t[:, :, ix[:, k]]
    return out[0] if squeeze else out


def resize_nearest(image: np.ndarray, height: int, width: int) -> np.ndarray:
    a = np.asarray(image)
    h, w = a.shape[-2], a.shape[-1]
    iy = np.minimum((np.arange(height) * (h / height)).astype(np.int64), h - 1)
    ix = np.minimum((np.arange(width) * (w / width)).astype(np.int64), w - 1)
    return a[..., iy[:, None], ix[None, :]]


def resize_image(image: np.ndarray, desired_height: int, desired_width: int, resize_mode: str = "lanczos4"):
    """utils_image.py:11-60: no-op when the size already matches."""
    if image.shape[-2] == desired_height and image.shape[-1] == desired_width:
        return image
    if image.ndim not in (2, 3):
        raise NotImplementedError(f'Cannot resize image with "{image.ndim}" dimensions.')
    if resize_mode == "lanczos4":
        return resize_lanczos4(image, desired_height, desired_width)
    if resize_mode == "nearest":
        return resize_nearest(image, desired_height, desired_width)
    raise NotImplementedError(f"resize mode {resize_mode}")
