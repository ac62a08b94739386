"""Per-wave, per-phase s_memtime breakdown of k_wgrad_bf16_pp (library built with `make EXTRA=-DFU_CONV_STAMPS`)."""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, '.')
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH); raw.fu_debug_set_conv_stamps.argtypes = [C.c_void_p]
DEV = 'cuda:0'
def run(B, C0, Cout, H, W):
    x = torch.randn(B, H, W, C0, device=DEV).to(torch.bfloat16); a = torch.rand(C0, device=DEV) + 0.5; b = torch.randn(C0, device=DEV) * 0.1
    dy = torch.randn(B, H, W, Cout, device=DEV).to(torch.bfloat16)
    dw = torch.empty(Cout, C0, 3, 3, device=DEV)
    dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=DEV)
    for it in range(3):
        raw.fu_debug_set_conv_stamps(dbg.data_ptr() if it == 2 else None)
        check(lib.fu_op_conv3x3_wgrad(1, ptr(x), C0, ptr(a), ptr(b), None, 0, ptr(dy), Cout, ptr(dw), B, H, W, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); raw.fu_debug_set_conv_stamps(None)
    d = dbg.view(256, 8, 8).cpu().numpy().astype(np.float64)
    print(f"{C0}->{Cout}@{H}")
    for w in range(8):
        v = d[:, w]; v = v[v[:, 3] > 0]; n = v[:, 3]
        m = lambda k: np.median(v[:, k] / n)
        print(f"   wave {w}: mfma {m(0):.0f} | barrier1 {m(1):.0f} | bn+store {m(2):.0f} | load issue + prefetch {m(4):.0f} | barrier2 {m(5):.0f} | sum {m(0)+m(1)+m(2)+m(4)+m(5):.0f}   | loop gap {m(6):.0f} | period {m(0)+m(1)+m(2)+m(4)+m(5)+m(6):.0f}")
for shp in [(16, 128, 128, 128, 128), (16, 512, 512, 32, 32), (16, 256, 256, 64, 64)]:
    run(*shp)
