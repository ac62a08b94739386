"""HIP fp32 gradients vs the reference fixture (torch fp32) vs an fp64 run of the oracle (diagnostic)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import case_inputs, load_golden, is_dead_bias
from floodplanet_code_amd.unet import HipUNet
from oracle import unet_oracle as O
DEV = "cuda:0"
torch.set_num_threads(16)
for name in sys.argv[1:]:
    meta, z = load_golden(name)
    batch, st = case_inputs(meta)
    ii = meta["resolved_ignore_index"]
    net = HipUNet(meta["n_in"], 3, base_channels=meta["base"]); net.load_state_dict(st); net.to(DEV).train()
    loss, logits = net.loss(batch["image"].to(DEV), batch["target"].to(DEV), ii, return_logits=True)
    loss.backward(); torch.cuda.synchronize()
    st32 = {k: v.clone() for k, v in st.items()}
    _, _, g32 = O.loss_and_grads(st32, batch, ii)
    st64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in st.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    _, _, g64 = O.loss_and_grads(st64, b64, ii)
    eh, er, ehr = [], [], []
    for k, p in net.named_parameters():
        if is_dead_bias(k): continue
        n64 = g64[k].norm().item() + 1e-30
        eh.append((p.grad.cpu().double() - g64[k]).norm().item() / n64)
        er.append((g32[k].double() - g64[k]).norm().item() / n64)
        ehr.append((p.grad.cpu().double() - g32[k].double()).norm().item() / n64)
    print(f"{name}: HIP-vs-fp64 median {np.median(eh):.2e} max {max(eh):.2e} | torch32-vs-fp64 median {np.median(er):.2e} max {max(er):.2e} | HIP-vs-torch32 median {np.median(ehr):.2e} max {max(ehr):.2e}")
