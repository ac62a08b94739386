"""Minimal TIFF reader for the rasters bundled with the reference (`CSDAP_complete/<region>/{S1,L8,PS,labels}/*.tif`):
classic little- or big-endian TIFF, strips, no compression, chunky or planar-separate samples, 8/16/32/64-bit unsigned /
signed / IEEE samples.  It replaces the two third-party calls on the reference's data path that are absent here --
`tifffile.imread(path)` (st_water_seg/datasets/floodplanet.py:313,495,568: array of shape [bands, H, W] for
planar-separate files, [H, W, bands] for chunky multi-band files, [H, W] for one band) and
`rasterio.open(path).height / .width` (floodplanet.py:103-104) -- with the same return conventions.  Anything else
(tiles, compression, BigTIFF, sub-byte samples) raises `TiffError`: the bundled files do not use it.

Host-side I/O, numpy only; not part of the GPU hot path.
"""
from __future__ import annotations

import struct
from typing import Dict, Tuple

import numpy as np

__all__ = ["TiffError", "read_tiff", "tiff_size", "tiff_info"]


class TiffError(ValueError):
    pass


_TYPE_FMT = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d",
             16: "Q", 17: "q"}
_TAG_NAMES = {256: "width", 257: "height", 258: "bits", 259: "compression", 262: "photometric", 273: "strip_offsets",
              277: "samples", 278: "rows_per_strip", 279: "strip_bytes", 284: "planar", 339: "sample_format",
              322: "tile_width", 324: "tile_offsets"}


def _parse_ifd(buf: memoryview) -> Tuple[str, Dict[str, tuple]]:
    if len(buf) < 8:
        raise TiffError("file too short for a TIFF header")
    head = bytes(buf[:2])
    if head == b"II":
        bo = "<"
    elif head == b"MM":
        bo = ">"
    else:
        raise TiffError("not a TIFF file (bad byte-order mark)")
    magic, ifd = struct.unpack(bo + "HI", buf[2:8])
    if magic == 43:
        raise TiffError("BigTIFF is not supported")
    if magic != 42:
        raise TiffError(f"not a TIFF file (magic {magic})")
    if ifd + 2 > len(buf):
        raise TiffError("IFD offset beyond the end of the file")
    (n,) = struct.unpack(bo + "H", buf[ifd:ifd + 2])
    if ifd + 2 + 12 * n > len(buf):
        raise TiffError("truncated IFD")
    tags: Dict[str, tuple] = {}
    for i in range(n):
        e = buf[ifd + 2 + 12 * i: ifd + 14 + 12 * i]
        tag, typ, cnt = struct.unpack(bo + "HHI", e[:8])
        name = _TAG_NAMES.get(tag)
        if name is None:
            continue                      # geo keys, nodata, ... : not needed to decode the samples
        fmt = _TYPE_FMT.get(typ)
        if fmt is None:
            raise TiffError(f"tag {tag}: unknown field type {typ}")
        per = struct.calcsize(bo + fmt)
        nbytes = per * cnt
        if nbytes <= 4:
            raw = bytes(e[8:8 + nbytes])
        else:
            (off,) = struct.unpack(bo + "I", e[8:12])
            if off + nbytes > len(buf):
                raise TiffError(f"tag {tag}: value beyond the end of the file")
            raw = bytes(buf[off:off + nbytes])
        tags[name] = struct.unpack(bo + fmt[0] * (cnt * len(fmt)), raw)
    return bo, tags


def _layout(tags: Dict[str, tuple]):
    for req in ("width", "height", "strip_offsets", "strip_bytes"):
        if req not in tags:
            if req.startswith("strip") and "tile_offsets" in tags:
                raise TiffError("tiled TIFF is not supported")
            raise TiffError(f"missing required tag: {req}")
    width, height = int(tags["width"][0]), int(tags["height"][0])
    spp = int(tags.get("samples", (1,))[0])
    bits = tags.get("bits", (1,) * spp)
    if len(set(bits)) != 1 or len(bits) not in (1, spp):
        raise TiffError(f"mixed bits per sample {bits} are not supported")
    bits = int(bits[0])
    if int(tags.get("compression", (1,))[0]) != 1:
        raise TiffError(f"compressed TIFF (scheme {tags['compression'][0]}) is not supported")
    fmt = tags.get("sample_format", (1,))
    if len(set(fmt)) != 1:
        raise TiffError(f"mixed sample formats {fmt} are not supported")
    kind = {1: "u", 2: "i", 3: "f"}.get(int(fmt[0]))
    if kind is None or bits not in (8, 16, 32, 64) or (kind == "f" and bits < 32):
        raise TiffError(f"unsupported sample type: format {fmt[0]}, {bits} bits")
    planar = int(tags.get("planar", (1,))[0])
    if planar not in (1, 2):
        raise TiffError(f"bad PlanarConfiguration {planar}")
    rps = int(tags.get("rows_per_strip", (height,))[0])
    rps = min(rps, height) if rps > 0 else height
    return width, height, spp, bits, kind, planar, rps


def tiff_info(path: str) -> dict:
    """Header fields only (first IFD): height, width, samples, dtype, planar configuration."""
    with open(path, "rb") as fh:
        buf = memoryview(fh.read())
    bo, tags = _parse_ifd(buf)
    width, height, spp, bits, kind, planar, rps = _layout(tags)
    return {"height": height, "width": width, "samples": spp, "dtype": np.dtype(f"{kind}{bits // 8}").name,
            "planar": planar, "rows_per_strip": rps, "byteorder": bo}


def tiff_size(path: str) -> Tuple[int, int]:
    """(height, width) -- what `rasterio.open(path).height, .width` give (floodplanet.py:103-104)."""
    info = tiff_info(path)
    return info["height"], info["width"]


def read_tiff(path: str) -> np.ndarray:
    """Decode the first image of the file.  Shapes follow `tifffile.imread`: one sample per pixel -> [H, W]; planar-
    separate -> [bands, H, W]; chunky with several samples -> [H, W, bands].  Native-endian, C-contiguous array."""
    with open(path, "rb") as fh:
        buf = memoryview(fh.read())
    bo, tags = _parse_ifd(buf)
    width, height, spp, bits, kind, planar, rps = _layout(tags)
    dt = np.dtype(f"{bo}{kind}{bits // 8}")
    offs, cnts = tags["strip_offsets"], tags["strip_bytes"]
    strips_per_plane = (height + rps - 1) // rps
    planes = spp if planar == 2 else 1
    if len(offs) != strips_per_plane * planes or len(cnts) != len(offs):
        raise TiffError(f"{len(offs)} strips, expected {strips_per_plane * planes}")
    row_elems = width * (1 if planar == 2 else spp)
    out = np.empty((planes, height, row_elems), dtype=dt.newbyteorder("="))
    for p in range(planes):
        for s in range(strips_per_plane):
            r0 = s * rps
            rows = min(rps, height - r0)
            need = rows * row_elems * dt.itemsize
            o, n = int(offs[p * strips_per_plane + s]), int(cnts[p * strips_per_plane + s])
            if n < need or o + need > len(buf):
                raise TiffError(f"strip {s} of plane {p}: {n} bytes at {o}, need {need}")
            out[p, r0:r0 + rows] = np.frombuffer(buf, dtype=dt, count=rows * row_elems, offset=o).reshape(rows, row_elems)
    if planar == 2:
        return out[0] if spp == 1 else out
    if spp == 1:
        return out[0]
    return out[0].reshape(height, width, spp)
