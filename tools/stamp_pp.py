"""s_memtime phase sums of the persistent ping-pong conv kernel (diagnostic build: tools/build_variant.sh stamps -DFU_CONV_STAMPS,
FU_LIB_PATH=tools/dbglibs/stamps.so).  Per wave: cycles in the MFMA phases, at the barrier behind them, in the staging phases,
in combine / epilogue / prefetch, at the barrier behind those; per step = divided by the workgroup's steps."""
import sys, ctypes as C, torch, numpy as np
sys.path.insert(0, '.')
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
raw.fu_debug_set_conv_stamps.argtypes = [C.c_void_p]
DEV = 'cuda:0'
def run(B, C0, Cout, H, W, bn=True):
    x = torch.randn(B, H, W, C0, device=DEV).to(torch.bfloat16); a = torch.rand(C0, device=DEV) + 0.5; b = torch.randn(C0, device=DEV) * 0.1
    w = torch.randn(Cout, C0, 3, 3, device=DEV) / 10; bias = torch.zeros(Cout, device=DEV)
    y = torch.empty(B, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    ssum = torch.empty(Cout, device=DEV); ssq = torch.empty(Cout, device=DEV)
    dbg = torch.zeros(256 * 8 * 16 + 64, dtype=torch.int64, device=DEV)
    lib.fu_test_conv_tile_mode(4)
    for it in range(3):
        raw.fu_debug_set_conv_stamps(dbg.data_ptr() if it == 2 else None)
        check(lib.fu_op_conv3x3_fwd(1, ptr(x), C0, ptr(a) if bn else None, ptr(b) if bn else None, None, 0, ptr(w), ptr(bias), ptr(y), Cout, B, H, W, ptr(ssum), ptr(ssq), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); raw.fu_debug_set_conv_stamps(None); lib.fu_test_conv_tile_mode(0)
    d = dbg[:256 * 8 * 16].view(256, 8, 16).cpu().numpy().astype(np.float64)
    d = d[d[:, 0, 8] > 0]
    T = d[:, :, 8]
    med = lambda v: float(np.median(v))
    print(f"{C0}->{Cout} @{H} B={B} bn={bn}: wgs={len(d)} steps/wg={med(T):.0f} chunks/tile={C0 // 32}")
    for g in (0, 1):
        x = d[:, 4 * g:4 * g + 4, :]
        t = x[:, :, 8]
        print(f"  group {g}: per step: mfma+convert {med(x[:,:,0]/t):.0f} | barrier {med(x[:,:,1]/t):.0f} | DMA+refill+wait+combine {med(x[:,:,2]/t):.0f} | prefetch {med(x[:,:,3]/t):.0f} | barrier {med(x[:,:,4]/t):.0f} || loop {med(x[:,:,5]/t):.0f} per step, prologue+tail {med(x[:,:,6]-x[:,:,5]):.0f} | lifetime {med(x[:,:,6]):.0f} cycles = {med(x[:,:,7])/100:.1f} us -> {med(x[:,:,6])/med(x[:,:,7])*100:.0f} MHz")
        ne = np.maximum(x[:,:,12],1)
        print(f"           per tile: epilogue {med(x[:,:,11]/ne):.0f} cycles (rows: convert + store {med(x[:,:,10]/ne):.0f}, sums {med(x[:,:,14]/ne):.0f}), barrier between the two tile-end phases {med(x[:,:,13]/ne):.0f} ({med(x[:,:,12]):.0f} tiles per workgroup) | prologue {med(x[:,:,15]):.0f}, tail {med(x[:,:,6]-x[:,:,5]-x[:,:,15]):.0f}")
    st = d[:, :, 9]; end = st + d[:, :, 6]
    print(f"  span {end.max() - st.min():.0f} cycles; start skew p95 {np.percentile(st - st.min(), 95):.0f}")
for a in [(16, 64, 64, 256, 256), (16, 128, 128, 128, 128), (16, 256, 256, 64, 64), (16, 512, 512, 32, 32), (16, 64, 64, 256, 256, False)]:
    run(*a)
