"""Test-side writer of uncompressed strip TIFFs (classic, II or MM, chunky or planar-separate) -- only used to make
inputs for the reader tests and the synthetic FloodPlanet tree; independent of floodplanet_code_amd.datasets.tiff."""
import os
import struct

import numpy as np


def write_tiff(path, array, planar=2, rows_per_strip=5, byteorder="<", extra_tags=(), compression=1, magic=42):
    """array: [H, W] or [bands, H, W] (bands-first regardless of `planar`)."""
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[None]
    bands, H, W = a.shape
    kind = {"u": 1, "i": 2, "f": 3}[a.dtype.kind]
    bits = a.dtype.itemsize * 8
    be = a.astype(a.dtype.newbyteorder(byteorder))
    strips = []
    if planar == 2 or bands == 1:
        for b in range(bands):
            for r0 in range(0, H, rows_per_strip):
                strips.append(be[b, r0:r0 + rows_per_strip].tobytes())
    else:
        chunky = np.ascontiguousarray(np.transpose(be, (1, 2, 0)))
        for r0 in range(0, H, rows_per_strip):
            strips.append(chunky[r0:r0 + rows_per_strip].tobytes())
    bo = byteorder
    tags = [(256, 3, [W]), (257, 3, [H]), (258, 3, [bits] * bands), (259, 3, [compression]), (262, 3, [1]),
            (273, 4, None), (277, 3, [bands]), (278, 3, [rows_per_strip]), (279, 4, [len(s) for s in strips]),
            (284, 3, [planar if bands > 1 else 1]), (339, 3, [kind] * bands)] + list(extra_tags)
    tags.sort(key=lambda t: t[0])
    header = (b"II" if bo == "<" else b"MM") + struct.pack(bo + "HI", magic, 8)
    ifd_size = 2 + 12 * len(tags) + 4
    cursor = 8 + ifd_size
    blobs, entries = [], []
    data_start_patch = None
    for tag, typ, vals in tags:
        fmt = {3: "H", 4: "I"}[typ]
        cnt = len(strips) if vals is None else len(vals)
        size = struct.calcsize(bo + fmt) * cnt
        if vals is None:
            data_start_patch = (len(entries), fmt, cnt, size)
            entries.append(None)
            continue
        raw = struct.pack(bo + fmt * cnt, *vals)
        if size <= 4:
            entries.append(struct.pack(bo + "HHI", tag, typ, cnt) + raw.ljust(4, b"\0"))
        else:
            entries.append(struct.pack(bo + "HHII", tag, typ, cnt, cursor))
            blobs.append(raw)
            cursor += size
    i, fmt, cnt, size = data_start_patch
    off_field_pos = cursor if size > 4 else None
    if size > 4:
        cursor += size
    offs, c = [], cursor
    for s in strips:
        offs.append(c)
        c += len(s)
    raw = struct.pack(bo + fmt * cnt, *offs)
    if size <= 4:
        entries[i] = struct.pack(bo + "HHI", 273, 4, cnt) + raw.ljust(4, b"\0")
    else:
        entries[i] = struct.pack(bo + "HHII", 273, 4, cnt, off_field_pos)
        blobs.append(raw)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as fh:
        fh.write(header + struct.pack(bo + "H", len(tags)) + b"".join(entries) + struct.pack(bo + "I", 0))
        fh.write(b"".join(blobs))
        fh.write(b"".join(strips))


def make_floodplanet_tree(root, regions=("RegA", "RegB", "RegC"), images_per_region=2, label_size=96, s1_size=40,
                          l8_size=24, seed=0):
    """A small CSDAP_complete tree shaped like the bundled sample: S1 2-band f32 in dB, L8 7-band f32, labels u8 {0,1,2}
    at a higher resolution than the images; one S1 image per region has no label."""
    g = np.random.default_rng(seed)
    made = {}
    for r in regions:
        for k in range(images_per_region):
            name = f"{r[:3].upper()}_{k}_{k + 7}"
            lab = g.integers(0, 3, size=(label_size, label_size), dtype=np.uint8)
            s1 = (g.random((2, s1_size, s1_size), dtype=np.float32) * 70 - 50).astype(np.float32)
            s1[0, 0, 0] = np.nan
            l8 = (g.random((7, l8_size, l8_size), dtype=np.float32) * 25000).astype(np.float32)
            write_tiff(os.path.join(root, "CSDAP_complete", r, "labels", name + ".tif"), lab, rows_per_strip=8)
            write_tiff(os.path.join(root, "CSDAP_complete", r, "S1", name + ".tif"), s1, planar=2, rows_per_strip=5)
            write_tiff(os.path.join(root, "CSDAP_complete", r, "L8", name + ".tif"), l8, planar=2, rows_per_strip=15)
            made[(r, name)] = {"label": lab, "S1": s1, "L8": l8}
        orphan = (g.random((2, s1_size, s1_size), dtype=np.float32) * 70 - 50).astype(np.float32)
        write_tiff(os.path.join(root, "CSDAP_complete", r, "S1", f"{r[:3].upper()}_orphan.tif"), orphan)
    return made
