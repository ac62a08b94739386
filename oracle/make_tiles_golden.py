"""Golden vectors for the tile grid and the bundled rasters (TEST INFRASTRUCTURE; runs only in the build container,
where /root/reference is mounted):

  * `get_crop_slices` of the reference (st_water_seg/datasets/utils.py:86-209) is executed on a list of cases.  Its
    module imports hydra, which is absent, so the one function is compiled from the file's syntax tree and run by
    itself (it depends on nothing else); inputs, outputs and raised exception types go to tests/golden/tiles_golden.json.
  * every bundled raster (CSDAP_complete/<region>/<sensor|labels>/*.tif) is summarised -- shape, dtype, sha256 of the
    decoded little-endian samples in [band, row, col] order -- by a decoder written independently of
    floodplanet_code_amd.datasets.tiff (PIL for the single-band uint8 label rasters; a direct strip walk for the
    planar float rasters, which PIL cannot open) -> tests/golden/rasters_golden.json.

usage: PYTHONDONTWRITEBYTECODE=1 python oracle/make_tiles_golden.py
"""
import ast, glob, hashlib, json, os, struct, sys

sys.dont_write_bytecode = True
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_reference_function(path, name):
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fn) == 1
    mod = ast.Module(body=fn, type_ignores=[])
    ns = {}
    exec(compile(mod, path, "exec"), ns)
    return ns[name]


def tiles_cases():
    cases = []
    for (h, w) in [(1024, 1024), (386, 386), (129, 111), (300, 300), (256, 256), (100, 40), (5, 7)]:
        for (ch, cw) in [(256, 256), (128, 128), (64, 96), (300, 300), (7, 5)]:
            for step in [None, 128, 64, (32, 48), 1000, 0, -3, 2.5]:
                for mode in ["exact", "over", "under", "bogus"]:
                    cases.append({"height": h, "width": w, "crop_height": ch, "crop_width": cw, "step": step,
                                  "mode": mode})
    return cases


def run_tiles():
    f = load_reference_function(os.path.join(REF, "st_water_seg/datasets/utils.py"), "get_crop_slices")
    out = []
    for c in tiles_cases():
        step = tuple(c["step"]) if isinstance(c["step"], (list, tuple)) else c["step"]
        try:
            res = {"slices": f(c["height"], c["width"], c["crop_height"], c["crop_width"], step, c["mode"])}
        except Exception as e:          # noqa: BLE001 -- the exception type is part of the contract
            res = {"raises": type(e).__name__}
        if "slices" in res and len(res["slices"]) > 400:      # keep the fixture small: count + checksum + a sample
            s = res["slices"]
            res = {"n": len(s), "sha": hashlib.sha256(json.dumps(s).encode()).hexdigest(), "head": s[:5],
                   "tail": s[-5:]}
        out.append({"args": c, **res})
    return out


def walk_strips(path):
    """Independent decode of an uncompressed little-endian strip TIFF -> [bands, H, W]."""
    b = open(path, "rb").read()
    assert b[:4] == b"II*\x00"
    (ifd,) = struct.unpack("<I", b[4:8])
    (n,) = struct.unpack("<H", b[ifd:ifd + 2])
    T = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack("<HHII", b[ifd + 2 + 12 * i: ifd + 14 + 12 * i])
        size = {1: 1, 2: 1, 3: 2, 4: 4, 12: 8}[typ] * cnt
        raw = b[ifd + 10 + 12 * i: ifd + 10 + 12 * i + size] if size <= 4 else b[val: val + size]
        if typ in (3, 4):
            T[tag] = struct.unpack("<" + ("H" if typ == 3 else "I") * cnt, raw)
    W, H, spp = T[256][0], T[257][0], T.get(277, (1,))[0]
    bits, fmt, planar, rps = T[258][0], T.get(339, (1,))[0], T.get(284, (1,))[0], T[278][0]
    assert T[259][0] == 1
    dt = {(8, 1): "<u1", (16, 1): "<u2", (32, 3): "<f4", (16, 2): "<i2", (32, 1): "<u4", (64, 3): "<f8"}[(bits, fmt)]
    data = b"".join(b[o:o + c] for o, c in zip(T[273], T[279]))
    a = np.frombuffer(data, dtype=dt)
    if planar == 2 or spp == 1:
        return a.reshape(spp, H, W)
    return np.transpose(a.reshape(H, W, spp), (2, 0, 1))


def run_rasters():
    from PIL import Image
    out = {}
    for p in sorted(glob.glob(os.path.join(REF, "CSDAP_complete/*/*/*.tif"))):
        rel = os.path.relpath(p, REF)
        a = walk_strips(p)
        if a.shape[0] == 1 and a.dtype == np.uint8:          # second opinion from PIL where it can read the file
            pil = np.asarray(Image.open(p))
            assert pil.shape == a.shape[1:] and (pil == a[0]).all(), rel
        out[rel] = {"shape": list(a.shape), "dtype": a.dtype.name,
                    "sha256": hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()}
    return out


if __name__ == "__main__":
    t = run_tiles()
    json.dump(t, open(os.path.join(OUT, "tiles_golden.json"), "w"), separators=(",", ":"))
    r = run_rasters()
    json.dump(r, open(os.path.join(OUT, "rasters_golden.json"), "w"), indent=0)
    print(len(t), "tile-grid cases;", len(r), "rasters")
