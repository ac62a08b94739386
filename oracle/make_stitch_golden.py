"""Golden vectors for the overlap-average stitching of tile predictions (TEST INFRASTRUCTURE; runs only in the build
container, where /root/reference is mounted).

The reference's `ImageStitcher_v2` (st_water_seg/utils/utils_image.py:363-494) lives in a module that imports
tifffile / kwimage / cv2-style packages absent here, so the one class is compiled out of the file's syntax tree and
executed by itself -- its canvas arithmetic (`add_image`, `_combine_images`) needs only numpy, os and tqdm, all present.
It is driven exactly as predict.py:296-334 drives it: logits [c, h, w] -> 'c h w -> h w c' -> scipy softmax over the
last axis -> `add_image(pred, image_name, crop_params, og_height, og_width)` per crop -> `_combine_images()`.

Inputs come from the closed-form generator of oracle/unet_oracle.py (nothing is stored but the seeds): a seeded small
UNet state, a seeded "raster", crops of it on a grid; the logits are the oracle's eval-mode forward of every crop (the
oracle is pinned bit-exactly against the reference network by make_golden.py).  Expected outputs -> tests/golden/
stitch_*.npz: the combined canvas (float64, as numpy promotes float32 / (float64 + 1e-5)), the weight canvas, the
argmax.  Partial crops (dh < crop height at the bottom / right edge, `image[:dh, :dw]`) are part of every case.

usage: PYTHONDONTWRITEBYTECODE=1 python oracle/make_stitch_golden.py
"""
import ast
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np
import torch
from scipy.special import softmax

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402

REF = "/root/reference/st_water_seg/utils/utils_image.py"
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference_class(path, name):
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name]
    assert len(cls) == 1
    from tqdm import tqdm
    ns = {"np": np, "os": os, "tqdm": tqdm}
    exec(compile(ast.Module(body=cls, type_ignores=[]), path, "exec"), ns)
    return ns[name]


class Crop:   # the four attributes add_image reads off datasets/utils.py:CropParams
    def __init__(self, h0, w0, hE, wE):
        self.h0, self.w0, self.hE, self.wE = h0, w0, hE, wE


CASES = {
    # name: (canvas H, W, crop size, [(h0, w0)], n_channels, base, seeds)
    "stitch_overlap_96x112": dict(H=96, W=112, S=64, origins=[(0, 0), (0, 48), (32, 0), (32, 48)], C=4, base=8,
                                  param_seed=1, data_seed=0),
    # stride 48 on 120x100: the last row / column of crops reaches past the raster -> add_image cuts them
    "stitch_partial_120x100": dict(H=120, W=100, S=64, origins=[(h, w) for h in (0, 48, 96) for w in (0, 48, 96)][:8]
                                   + [(96, 96)], C=5, base=8, param_seed=2, data_seed=3),
}


def raster(C, H, W, seed):
    return torch.from_numpy(O.hash_uniform(C * H * W, seed, 77).astype(np.float32).reshape(C, H, W))


def crops_of(big, origins, S):
    """crop tensors [n, C, S, S] (zero-padded past the raster, like the dataset's padding) + the clipped crop boxes"""
    C, H, W = big.shape
    xs, boxes = [], []
    for (h0, w0) in origins:
        t = torch.zeros(C, S, S)
        hE, wE = min(h0 + S, H), min(w0 + S, W)
        t[:, :hE - h0, :wE - w0] = big[:, h0:hE, w0:wE]
        xs.append(t)
        boxes.append((h0, w0, hE, wE))
    return torch.stack(xs), boxes


def main():
    Stitcher = load_reference_class(REF, "ImageStitcher_v2")
    for name, c in CASES.items():
        st = O.make_state(c["C"], 3, c["base"], True, seed=c["param_seed"])
        big = raster(c["C"], c["H"], c["W"], c["data_seed"])
        x, boxes = crops_of(big, c["origins"], c["S"])
        logits = O.eval_forward(st, {"image": x})                      # [n, 3, S, S] fp32
        with tempfile.TemporaryDirectory() as tmp:
            sref = Stitcher(tmp, image_type_name="pred_softmax", save_backend="PIL", save_ext=".png")
            for i, (h0, w0, hE, wE) in enumerate(boxes):
                pred = logits[i].numpy().transpose(1, 2, 0)            # rearrange 'c h w -> h w c' (predict.py:298)
                pred = softmax(pred, axis=-1)                          # predict.py:301
                sref.add_image(pred, "img", Crop(h0, w0, hE, wE), c["H"], c["W"])
            weight = sref.weight_canvas["img"].copy()
            sref._combine_images()
            canvas = sref.image_canvas["img"]
        meta = dict(c, boxes=boxes, n_classes=3)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8),
                            canvas=canvas, weight=weight, argmax=canvas.argmax(-1).astype(np.int64),
                            logits=logits.numpy())
        print(name, canvas.shape, canvas.dtype, "weight max", weight.max(), "uncovered", int((weight == 0).sum()))


if __name__ == "__main__":
    main()
