"""Per-phase s_memtime breakdown of k_conv3x3_bf16_pp (library built with EXTRA=-DFU_CONV_STAMPS)."""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, '.')
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH); raw.fu_debug_set_conv_stamps.argtypes = [C.c_void_p]
DEV = 'cuda:0'
def run(B, C0, Cout, H, W):
    x = torch.randn(B, H, W, C0, device=DEV).to(torch.bfloat16); a = torch.rand(C0, device=DEV) + 0.5; b = torch.randn(C0, device=DEV) * 0.1
    w = torch.randn(Cout, C0, 3, 3, device=DEV) / 10; bias = torch.zeros(Cout, device=DEV)
    y = torch.empty(B, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    dbg = torch.zeros(256 * 2 * 8 + 64, dtype=torch.int64, device=DEV)
    for it in range(3):
        raw.fu_debug_set_conv_stamps(dbg.data_ptr() if it == 2 else None)
        check(lib.fu_op_conv3x3_fwd(1, ptr(x), C0, ptr(a), ptr(b), None, 0, ptr(w), ptr(bias), ptr(y), Cout, B, H, W, None, None, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); raw.fu_debug_set_conv_stamps(None)
    d = dbg[:256 * 2 * 8].view(512, 8).cpu().numpy().astype(np.float64)
    for g in (0, 1):
        r = d[g::2]; r = r[r[:, 4] > 0]
        n = r[:, 4]
        print(f"{C0}->{Cout}@{H} group {g}: per iteration: staging {np.median(r[:,0]/n):.0f} | barrier {np.median(r[:,1]/n):.0f} | mfma part {np.median(r[:,2]/n):.0f} (of which before the MFMA block {np.median(r[:,6]/n):.0f}) | barrier {np.median(r[:,3]/n):.0f} | lifetime {np.median(r[:,5]):.0f}")
run(16, 512, 512, 32, 32)
run(16, 128, 128, 128, 128)
run(16, 64, 64, 256, 256)
