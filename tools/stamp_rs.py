"""s_memtime phase sums of the row-stationary conv kernel (diagnostic build: make EXTRA=-DFU_CONV_STAMPS, FU_LIB_PATH)."""
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
    nwg = B * (H // 16) * (W // 16) * (Cout // 64) + 64
    dbg = torch.zeros(nwg * 11 + 64, dtype=torch.int64, device=DEV)
    lib.fu_test_conv_tile_mode(3)
    for it in range(3):
        raw.fu_debug_set_conv_stamps(dbg.data_ptr() if it == 2 else None)
        check(lib.fu_op_conv3x3_fwd(1, ptr(x), C0, ptr(a) if bn else None, ptr(b) if bn else None, None, 0, ptr(w), ptr(bias), ptr(y), Cout, B, H, W, ptr(ssum), ptr(ssq), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); raw.fu_debug_set_conv_stamps(None); lib.fu_test_conv_tile_mode(0)
    nreal = int((dbg[:nwg * 10].view(nwg, 10)[:, 0] > 10 ** 9).sum().item())    # workgroups launched (T0 is a raw timestamp)
    rt = dbg[nreal * 10: nreal * 11].cpu().numpy().astype(np.float64)            # s_memrealtime ticks (100 MHz) per workgroup
    di = dbg[:nwg * 10].view(nwg, 10).cpu().numpy()
    dma = (di[:, 9] >> 32).astype(np.float64)
    di[:, 9] &= 0xffffffff
    d = di.astype(np.float64)
    keep = d[:, 0] > 1e9
    d, dma = d[keep], dma[keep]
    nch = C0 // 32
    med = lambda v: float(np.median(v))
    print(f"{C0}->{Cout} @{H} B={B}: wgs={len(d)} chunks={nch} | per chunk: barrier1 wait {med(d[:,4])/nch:.0f} | DMA issue + A convert/store {med(d[:,5])/nch:.0f} (DMA issue {med(dma)/nch:.0f}, waiting for the A loads {med(d[:,9])/nch:.0f}) | vmcnt0+barrier2 {med(d[:,6])/nch:.0f} | mfma block {med(d[:,7])/nch:.0f} "
          f"|| prologue {med(d[:,1]-d[:,0]):.0f} | epilogue {med(d[:,8]-d[:,2]):.0f} | drain {med(d[:,3]-d[:,8]):.0f} | lifetime {med(d[:,3]-d[:,0]):.0f} cycles = {np.median(rt)/100:.1f} us -> clock {med(d[:,3]-d[:,0])/np.median(rt)*100:.0f} MHz | lifetime p5/p50/p95/max {np.percentile(d[:,3]-d[:,0],5):.0f}/{med(d[:,3]-d[:,0]):.0f}/{np.percentile(d[:,3]-d[:,0],95):.0f}/{(d[:,3]-d[:,0]).max():.0f} | start skew p95 {np.percentile(d[:,0]-d[:,0].min(),95):.0f} max {(d[:,0]-d[:,0].min()).max():.0f} | span {d[:,3].max()-d[:,0].min():.0f} cycles")
run(16, 256, 256, 64, 64)
run(16, 128, 128, 128, 128)
run(16, 64, 64, 256, 256)
run(16, 512, 512, 32, 32)
