"""A synthetic CSDAP_complete tree in the bundled sample's format (Sentinel-1 = 2-band float32 rasters of
334-386 pixels in dB, planar-separate uncompressed strips; labels = uint8 1024 x 1024 rasters with values {1, 2}) -- input for
`bench.py --path loader` and for tests: the real rasters are not on the GPU box, and the loader's cost (TIFF decode, the
Lanczos-4 resample of every image raster to its label raster's size, crops, collation) depends on the format and the sizes,
not on the pixel values.  The writer emits classic little-endian TIFFs with the tags datasets/tiff.py reads."""
from __future__ import annotations

import os
import struct

import numpy as np

__all__ = ["write_strip_tiff", "make_s1_tree"]


def write_strip_tiff(path: str, array: np.ndarray, rows_per_strip: int = 16) -> None:
    """[H, W] or [bands, H, W] -> uncompressed, planar-separate (PlanarConfiguration 2) strip TIFF, little endian."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[None]
    bands, H, W = a.shape
    fmt_code = {"u": 1, "i": 2, "f": 3}[a.dtype.kind]
    le = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<")))
    strips = [le[b, r:r + rows_per_strip].tobytes() for b in range(bands) for r in range(0, H, rows_per_strip)]
    n = len(strips)
    # (tag, type, values): 3 = SHORT, 4 = LONG; StripOffsets are patched once the directory's size is known
    tags = [(256, 3, [W]), (257, 3, [H]), (258, 3, [a.dtype.itemsize * 8] * bands), (259, 3, [1]), (262, 3, [1]),
            (273, 4, [0] * n), (277, 3, [bands]), (278, 3, [rows_per_strip]), (279, 4, [len(s) for s in strips]),
            (284, 3, [2 if bands > 1 else 1]), (339, 3, [fmt_code] * bands)]
    size = lambda typ, cnt: (2 if typ == 3 else 4) * cnt
    cursor = 8 + 2 + 12 * len(tags) + 4
    where = {}
    for tag, typ, vals in tags:
        if size(typ, len(vals)) > 4:
            where[tag] = cursor
            cursor += size(typ, len(vals))
    offs, c = [], cursor
    for st in strips:
        offs.append(c)
        c += len(st)
    entries, blobs = [], []
    for tag, typ, vals in tags:
        if tag == 273:
            vals = offs
        raw = struct.pack("<" + ("H" if typ == 3 else "I") * len(vals), *vals)
        if len(raw) <= 4:
            entries.append(struct.pack("<HHI", tag, typ, len(vals)) + raw.ljust(4, b"\0"))
        else:
            entries.append(struct.pack("<HHII", tag, typ, len(vals), where[tag]))
            blobs.append(raw)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(b"II" + struct.pack("<HI", 42, 8) + struct.pack("<H", len(tags)) + b"".join(entries) + struct.pack("<I", 0))
        fh.write(b"".join(blobs))
        fh.write(b"".join(strips))


def make_s1_tree(root: str, regions=("RegA", "RegB", "RegC"), images_per_region: int = 4, label_size: int = 1024,
                 s1_size: int = 360, seed: int = 0) -> int:
    """<root>/CSDAP_complete/<region>/{S1,labels}/<name>.tif; returns the number of labelled images."""
    g = np.random.default_rng(seed)
    n = 0
    for r in regions:
        for k in range(images_per_region):
            name = f"{r[:3].upper()}_{k}_{k + 7}"
            lab = g.integers(1, 3, size=(label_size, label_size), dtype=np.uint8)
            s1 = (g.random((2, s1_size, s1_size), dtype=np.float32) * 70 - 48).astype(np.float32)
            write_strip_tiff(os.path.join(root, "CSDAP_complete", r, "labels", name + ".tif"), lab)
            write_strip_tiff(os.path.join(root, "CSDAP_complete", r, "S1", name + ".tif"), s1)
            n += 1
    return n
