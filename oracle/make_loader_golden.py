"""Golden item dicts for the tile data path (TEST INFRASTRUCTURE; runs only in the build container, where /root/reference
is mounted).

What `Floodplanet_Dataset.__getitem__` (st_water_seg/datasets/floodplanet.py:600-658) hands to the training step for
every crop of a small synthetic CSDAP_complete tree, produced by the reference's OWN functions wherever they are plain
numpy:
  * the crop grid: `get_crop_slices` (st_water_seg/datasets/utils.py:86-209, mode 'exact'), compiled out of the file's
    syntax tree (the module imports hydra, absent here);
  * `BaseDataset._crop_image`, `normalize` (base_dataset.py:77-113), `_add_buffer_to_image` (:271-325), compiled out of
    base_dataset.py the same way and called on a stand-in `self` (norm_mode, ignore_index), in the order __getitem__ calls
    them: crop -> normalize(image, sensor) -> buffer(image, 0) / buffer(target, ignore_index).
Restated here because their bodies sit behind tifffile / cv2 calls (absent): the S1 scaling of
`_load_crop_norm_S1_image` (floodplanet.py:345-348: clip((x + 50) / 100, 0, 1), nan_to_num) and the label mapping of
`_load_label_image` (:586-596: 2 -> 1, 0 -> ignore_index, else 0).  The rasters have the label raster's size, so no
resampling takes part (OpenCV's Lanczos-4 stays unpinned, SURVEY 8c iv).

The rasters come from tests/tools/tiff_writer.make_floodplanet_tree(seed) -- the test re-creates the same tree, reads it
back through the TIFF reader, the dataset and the loader (host assembly on the CPU; device assembly through
fu_assemble_tiles on the GPU) and compares every item with the arrays stored here, keyed by raster name and crop origin.

usage: PYTHONDONTWRITEBYTECODE=1 python oracle/make_loader_golden.py
"""
import ast
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tools.tiff_writer import make_floodplanet_tree  # noqa: E402

REF = "/root/reference/st_water_seg/datasets"
OUT = os.path.join(ROOT, "tests", "golden", "loader_golden.npz")

# the parameters the test re-creates the tree and the datasets with
TREE = dict(regions=("RegA", "RegB", "RegC"), images_per_region=2, label_size=96, s1_size=96, l8_size=24, seed=3)
CROP = dict(height=64, width=64, stride=64)
IGNORE_INDEX = 0
NORM_MODES = [None, "local"]


def load_function(path, name):
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fn) == 1
    ns = {}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns[name]


def load_methods(path, cls, names):
    tree = ast.parse(open(path).read())
    c = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0]
    fns = [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names)
    ns = {"np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def main():
    get_crop_slices = load_function(os.path.join(REF, "utils.py"), "get_crop_slices")
    crop_image, normalize, add_buffer = load_methods(os.path.join(REF, "base_dataset.py"), "BaseDataset",
                                                     ["_crop_image", "normalize", "_add_buffer_to_image"])
    with tempfile.TemporaryDirectory() as tmp:
        made = make_floodplanet_tree(tmp, **TREE)
    arrays, index = {}, []
    for (region, name), rasters in sorted(made.items()):
        label, s1 = rasters["label"], rasters["S1"]
        H, W = label.shape
        assert s1.shape[1:] == (H, W)
        for (h0, w0, dh, dw) in get_crop_slices(H, W, CROP["height"], CROP["width"], CROP["stride"], mode="exact"):
            cp = types.SimpleNamespace(h0=h0, w0=w0, hE=h0 + dh, wE=w0 + dw)
            # _load_crop_norm_S1_image (floodplanet.py:338-348), sizes equal: crop, scale to [0, 1], NaN -> 0
            img = crop_image(None, s1[:2], cp)
            img = np.nan_to_num(np.clip((img + 50) / 100, 0, 1))
            # _load_label_image (floodplanet.py:583-596)
            lab = crop_image(None, label, cp)
            tgt = np.zeros(lab.shape, dtype="uint8")
            tgt[lab == 2] = 1
            tgt[lab == 0] = IGNORE_INDEX
            for mode in NORM_MODES:
                me = types.SimpleNamespace(norm_mode=mode, ignore_index=IGNORE_INDEX)
                im, mean, std = normalize(me, img.copy(), "S1")
                im = add_buffer(me, im, CROP["height"], CROP["width"])
                tg = add_buffer(me, tgt, CROP["height"], CROP["width"], constant_value=IGNORE_INDEX)
                key = f"{region}/{name}/{h0}_{w0}/{mode}"
                arrays[key + "/image"] = np.asarray(im, dtype=np.float32)
                arrays[key + "/mean"] = np.asarray(mean, dtype=np.float32).reshape(-1)
                arrays[key + "/std"] = np.asarray(std, dtype=np.float32).reshape(-1)
                if mode is None:
                    arrays[f"{region}/{name}/{h0}_{w0}/target"] = np.asarray(tg).astype(np.int64)
            index.append({"region": region, "name": name, "h0": h0, "w0": w0, "valid": [dh, dw]})
    meta = {"tree": {k: (list(v) if isinstance(v, tuple) else v) for k, v in TREE.items()}, "crop": CROP,
            "ignore_index": IGNORE_INDEX, "norm_modes": NORM_MODES, "items": index}
    np.savez_compressed(OUT, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    partial = sum(1 for i in index if i["valid"] != [CROP["height"], CROP["width"]])
    print(f"{len(index)} crops ({partial} cut at the raster's edge) x {len(NORM_MODES)} norm modes -> {OUT} "
          f"({os.path.getsize(OUT) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
