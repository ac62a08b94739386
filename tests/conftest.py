import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, z


def golden_names(late_fusion=False):
    """UNet training-step fixtures (oracle/make_golden.py) or the late-fusion ones (oracle/make_golden_lf.py: lf_*)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith("lf_") == late_fusion
                  and not f.startswith(("stitch_", "assemble_", "loader_")))


def lf_case_inputs(meta):
    """Batch, ordered in_channels and initial state of a late-fusion golden case from the closed-form generator."""
    from collections import OrderedDict
    from oracle import unet_oracle as O
    in_ch = OrderedDict((k, v) for k, v in meta["in_channels"])
    extras = tuple(k for k in O.LF_FORWARD_ORDER if k in in_ch and k != "ms_image")
    batch = O.make_batch(meta["B"], in_ch["ms_image"], meta["H"], meta["W"], seed=meta["data_seed"], extra=extras)
    st = O.lf_make_state(in_ch, meta["n_classes"], meta["base"], seed=meta["param_seed"])
    return batch, in_ch, st


def case_inputs(meta):
    """Re-derive the batch and the initial state of a golden case from the closed-form generator."""
    from oracle import unet_oracle as O
    extras = tuple(meta.get("extras", ()))
    ii = meta["resolved_ignore_index"]
    batch = O.make_batch(meta["B"], meta["C"], meta["H"], meta["W"], seed=meta["data_seed"],
                         n_label_values=meta.get("n_label_values", 2),
                         all_ignored_sample=meta.get("all_ignored_sample"), ignore_value=ii, extra=extras)
    if meta.get("all_ignored"):
        batch["target"][:] = ii
    st = O.make_state(meta["n_in"], meta["n_classes"], meta["base"], meta["bilinear"], seed=meta["param_seed"])
    return batch, st


# conv biases that feed a BatchNorm have an analytically zero gradient: the reference
# produces rounding noise there which Adam then normalises, so they are excluded from elementwise checks.
def is_dead_bias(name):
    return name.endswith(".bias") and (".double_conv.0." in name or ".double_conv.3." in name)
