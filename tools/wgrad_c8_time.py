"""Weight gradient of the 8-band first conv at the bench shape: k_wgrad_bf16_c8 against the general 64-row kernel
(fu_test_force_lockstep_wgrad), events around 20 launches each (kernel + slab reduce + transpose).
    python3 tools/wgrad_c8_time.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
lib = _lib.load()
DEV = 'cuda:0'
B, H, W, Cout = 16, 256, 256, 64
g = torch.Generator().manual_seed(0)
x = torch.rand(B, H, W, 8, generator=g).to(DEV).to(torch.bfloat16)
dy = torch.randn(B, H, W, Cout, generator=g).to(DEV).to(torch.bfloat16)
s = lambda: torch.cuda.current_stream().cuda_stream
res = []
for lock in (1, 0, 1, 0):
    lib.fu_test_force_lockstep_wgrad(lock)
    dw = torch.empty(Cout, 8, 3, 3, device=DEV)
    f = lambda: check(lib.fu_op_conv3x3_wgrad(1, ptr(x), 8, None, None, None, 0, ptr(dy), Cout, ptr(dw), B, H, W, s()))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{'general' if lock else 'c8     '}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (wgrad + reduce + transpose)", flush=True)
    res.append(dw.clone())
lib.fu_test_force_lockstep_wgrad(0)
print("max |d| / max |ref|:", ((res[0] - res[1]).abs().max() / res[0].abs().max()).item())
