import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from conftest import case_inputs, load_golden, is_dead_bias
from floodplanet_code_amd.unet import HipUNet
from oracle import unet_oracle as O
torch.set_num_threads(16)
for name in sys.argv[1:]:
    meta, z = load_golden(name)
    batch, st = case_inputs(meta)
    ii = meta["resolved_ignore_index"]
    ef = len(meta.get("extras", ())) > 0
    st32 = {k: v.clone() for k, v in st.items()}
    lo32, l32, g32 = O.loss_and_grads(st32, batch, ii, True, ef)
    st64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in st.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    lo64, l64, g64 = O.loss_and_grads(st64, b64, ii, True, ef)
    net = HipUNet(meta["n_in"], 3, base_channels=meta["base"]); net.load_state_dict(st); net.to("cuda:0").train()
    x = O.assemble_input(batch, ef).to("cuda:0")
    loss, logits = net.loss(x, batch["target"].to("cuda:0"), ii, return_logits=True)
    loss.backward(); torch.cuda.synchronize()
    print(name, "logits err vs f64: hip %.2e  torch32 %.2e | hip vs torch32 %.2e" % (
        (logits.detach().cpu().double() - lo64).abs().max(), (lo32.double() - lo64).abs().max(), (logits.detach().cpu() - lo32).abs().max()))
    print(" loss: hip %.8f t32 %.8f f64 %.8f" % (loss.item(), l32.item(), l64.item()))
    worst = []
    for k, p in net.named_parameters():
        if is_dead_bias(k): continue
        gh = p.grad.cpu().double(); r64 = g64[k]; r32 = g32[k].double()
        n = r64.norm() + 1e-30
        worst.append((((gh - r64).norm() / n).item(), ((r32 - r64).norm() / n).item(), ((gh - r32).norm() / n).item(), k))
    worst.sort(reverse=True)
    for w in worst[:6]:
        print("  rel-L2 vs f64: hip %.2e torch32 %.2e | hip vs torch32 %.2e  %s" % w)
    import statistics
    print("  median hip %.2e torch32 %.2e" % (statistics.median(w[0] for w in worst), statistics.median(w[1] for w in worst)))
