"""Golden vectors for the tile assembly (TEST INFRASTRUCTURE; runs only in the build container, where /root/reference is
mounted).  The reference's `BaseDataset` (st_water_seg/datasets/base_dataset.py) lives in a module that imports
torchvision / omegaconf / tifffile, absent here, so its two methods `normalize` (:77-113) and `_add_buffer_to_image`
(:271-325) are compiled out of the file's syntax tree and called on a stand-in `self` that carries the attributes they read
(norm_mode, global_norm_params).  Driven as `Floodplanet_Dataset.__getitem__` drives them (floodplanet.py:613-625): crop ->
normalize(image, sensor) -> _add_buffer_to_image(image, max_crop_h, max_crop_w); several sensors are then concatenated
along the channel axis (ef_model.py:28-44).  Inputs come from the closed-form generator (seeds only);
expected outputs -> tests/golden/assemble_golden.npz.

usage: PYTHONDONTWRITEBYTECODE=1 python oracle/make_assemble_golden.py
"""
import ast
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402

REF = "/root/reference/st_water_seg/datasets/base_dataset.py"
OUT = os.path.join(ROOT, "tests", "golden", "assemble_golden.npz")


def load_methods(path, cls, names):
    tree = ast.parse(open(path).read())
    c = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
    fns = [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names)
    ns = {"np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


CASES = [   # name, norm_mode, sources (name, channels, scale, shift), B, nominal H x W, valid sizes per sample
    dict(name="none_single", norm_mode=None, sources=[("PS", 4, 1.0, 0.0)], B=2, H=32, W=32, valid=[(32, 32), (32, 32)]),
    dict(name="local_two_sensors_edge", norm_mode="local", sources=[("PS", 4, 1.0, 0.0), ("S1", 2, 0.7, 0.1)], B=3, H=32, W=40,
         valid=[(32, 40), (20, 40), (17, 9)]),
    dict(name="global_three", norm_mode="global", sources=[("S2", 10, 1.0, 0.0), ("S1", 2, 70.0, -50.0), ("dem", 1, 900.0, 10.0)],
         B=2, H=24, W=24, valid=[(24, 24), (11, 24)]),
    dict(name="local_large_mean", norm_mode="local", sources=[("L8", 7, 2.0e4, 5.0e3)], B=2, H=48, W=48, valid=[(48, 48), (48, 30)]),
]


def case_sources(c):
    """raw tiles [B, C_k, H, W] per source (valid crop in the top-left corner, the rest is never read)"""
    out = []
    for k, (name, ch, scale, shift) in enumerate(c["sources"]):
        n = c["B"] * ch * c["H"] * c["W"]
        x = O.hash_uniform(n, 1000 + k, sum(map(ord, c["name"]))).astype(np.float32).reshape(c["B"], ch, c["H"], c["W"])
        out.append((x * np.float32(scale) + np.float32(shift)).astype(np.float32))
    return out


def global_params(c):
    return {name: {"mean": (np.arange(ch, dtype=np.float32) * np.float32(0.1 * scale) + np.float32(shift)),
                   "std": (np.float32(scale) * (np.float32(0.5) + np.arange(ch, dtype=np.float32) * np.float32(0.05)))}
            for (name, ch, scale, shift) in c["sources"]}


def main():
    normalize, add_buffer = load_methods(REF, "BaseDataset", ["normalize", "_add_buffer_to_image"])
    arrays = {}
    metas = []
    for c in CASES:
        srcs = case_sources(c)
        gp = global_params(c)
        me = types.SimpleNamespace(norm_mode=c["norm_mode"], global_norm_params=gp)
        B = c["B"]
        ctot = sum(ch for _, ch, _, _ in c["sources"])
        image = np.zeros((B, ctot, c["H"], c["W"]), dtype=np.float32)
        mean = np.zeros((B, ctot), dtype=np.float32)
        std = np.zeros((B, ctot), dtype=np.float32)
        for b in range(B):
            vh, vw = c["valid"][b]
            parts, ms, ss = [], [], []
            for (name, ch, _, _), x in zip(c["sources"], srcs):
                crop = x[b, :, :vh, :vw].copy()                              # _crop_image
                img, m, s = normalize(me, crop, name)                        # base_dataset.py:77-113 (in place on the crop)
                img = add_buffer(me, img, c["H"], c["W"])                    # :271-325, constant_value 0
                parts.append(np.asarray(img, dtype=np.float32))
                ms.append(np.asarray(m, dtype=np.float32).reshape(-1))
                ss.append(np.asarray(s, dtype=np.float32).reshape(-1))
            image[b] = np.concatenate(parts, axis=0)                         # channel concat of the fused inputs
            mean[b] = np.concatenate(ms)
            std[b] = np.concatenate(ss)
        arrays[c["name"] + "_image"] = image
        arrays[c["name"] + "_mean"] = mean
        arrays[c["name"] + "_std"] = std
        metas.append(c)
        print(c["name"], image.shape, "mean", float(mean.mean()), "std", float(std.mean()))
    np.savez_compressed(OUT, meta=np.frombuffer(json.dumps(metas).encode(), dtype=np.uint8), **arrays)


if __name__ == "__main__":
    main()
