"""Persistent ping-pong conv kernel (tile mode 4, fu_conv_pp.hip) against the row-stationary kernel (tile mode 3) on the same
operands: outputs, BatchNorm statistics and fused BatchNorm-backward sums must be BIT-IDENTICAL (same accumulation and
summation order by construction).  Also prints a wall-clock per launch for both (events around 10 launches).
    python3 tools/pp_check.py [--time]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
lib = _lib.load()
DEV = 'cuda:0'
TIME = '--time' in sys.argv
# B, C0, C1, Cout, H, W, bn
FWD = [(2, 64, 0, 64, 64, 48, True), (2, 32, 32, 128, 32, 32, True), (16, 64, 0, 64, 256, 256, True),
       (16, 64, 64, 64, 256, 256, True), (16, 128, 0, 128, 128, 128, True), (16, 256, 256, 256, 64, 64, True),
       (16, 512, 512, 512, 32, 32, True), (16, 256, 0, 512, 32, 32, False), (8, 96, 0, 64, 32, 16, False),
       (16, 64, 0, 128, 128, 128, False), (16, 128, 0, 256, 64, 64, False), (16, 512, 0, 512, 32, 32, True)]
# B, Cout (K of the dgrad), C0, C1 (destinations), H, W
DGR = [(2, 64, 64, 0, 64, 32), (16, 64, 64, 0, 256, 256), (16, 64, 64, 64, 256, 256), (16, 512, 512, 512, 32, 32),
       (16, 128, 128, 0, 128, 128), (16, 256, 128, 128, 128, 128)]
# B, Cout, C0, H, W
BNS = [(4, 64, 64, 256, 256), (8, 128, 128, 128, 128), (2, 256, 192, 32, 32), (16, 64, 64, 256, 256), (2, 64, 64, 64, 64)]
s = lambda: torch.cuda.current_stream().cuda_stream
bad = 0


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for code, dt, nm in ((1, torch.bfloat16, 'bf16'), (2, torch.float16, 'fp16')):
    for sh in FWD:
        B, C0, C1, Cout, H, W, bn = sh
        g = torch.Generator().manual_seed(0)
        x0 = torch.randn(B, H, W, C0, generator=g).to(DEV).to(dt)
        x1 = torch.randn(B, H, W, C1, generator=g).to(DEV).to(dt) if C1 else None
        a = (torch.rand(C0, generator=g) + 0.5).to(DEV); b = (torch.randn(C0, generator=g) * 0.1).to(DEV)
        w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 10).to(DEV); bias = torch.randn(Cout, generator=g).to(DEV)
        outs, ts = [], []
        for m in (3, 4):
            y = torch.full((B, H, W, Cout), float('nan'), device=DEV, dtype=dt)
            ssum = torch.empty(Cout, device=DEV); ssq = torch.empty(Cout, device=DEV)
            lib.fu_test_conv_tile_mode(m)
            f = lambda: check(lib.fu_op_conv3x3_fwd(code, ptr(x0), C0, ptr(a) if bn else None, ptr(b) if bn else None,
                                                    ptr(x1) if C1 else None, C1, ptr(w), ptr(bias), ptr(y), Cout, B, H, W,
                                                    ptr(ssum), ptr(ssq), s()))
            f(); torch.cuda.synchronize()
            if TIME and code == 1: ts.append(timed(f))
            outs.append((y.clone(), ssum.clone(), ssq.clone()))
        lib.fu_test_conv_tile_mode(0)
        tall = B * (H // 32) * (W // 16) * (Cout // 64) >= 512          # mode 3 = rs<8>: same statistics tiles; else rs<4>
        eq = lambda p, q: torch.equal(p.view(torch.int32) if p.dtype == torch.float32 else p.view(torch.int16),
                                      q.view(torch.int32) if q.dtype == torch.float32 else q.view(torch.int16))
        close = lambda p, q: bool(((p - q).abs() <= 1e-5 * (p.abs().max() + 1)).all())
        ok = eq(outs[0][0], outs[1][0]) and all((eq if tall else close)(p, q) for p, q in zip(outs[0][1:], outs[1][1:]))
        fin = bool(torch.isfinite(outs[1][0].float()).all())
        d = (outs[0][0].float() - outs[1][0].float()).abs().max().item()
        print(f"{nm} fwd {sh}: identical={ok} finite={fin} max|d|={d:.3g}" + (f"  rs {ts[0]:.1f} us  pp {ts[1]:.1f} us" if ts else ""), flush=True)
        bad += (not ok) or (not fin)
    for sh in DGR:
        B, Cout, C0, C1, H, W = sh
        g = torch.Generator().manual_seed(1)
        dy = torch.randn(B, H, W, Cout, generator=g).to(DEV).to(dt)
        w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 3).to(DEV)
        outs, ts = [], []
        for m in (3, 4):
            dx0 = torch.full((B, H, W, C0), float('nan'), device=DEV, dtype=dt)
            dx1 = torch.full((B, H, W, C1), float('nan'), device=DEV, dtype=dt) if C1 else None
            lib.fu_test_conv_tile_mode(m)
            f = lambda: check(lib.fu_op_conv3x3_dgrad(code, ptr(dy), Cout, ptr(w), ptr(dx0), C0, ptr(dx1) if C1 else None, C1,
                                                      B, H, W, s()))
            f(); torch.cuda.synchronize()
            if TIME and code == 1: ts.append(timed(f))
            outs.append((dx0.clone(),) + ((dx1.clone(),) if C1 else ()))
        lib.fu_test_conv_tile_mode(0)
        ok = all(torch.equal(p.view(torch.int16), q.view(torch.int16)) for p, q in zip(*outs))
        fin = all(bool(torch.isfinite(p.float()).all()) for p in outs[1])
        print(f"{nm} dgrad {sh}: identical={ok} finite={fin}" + (f"  rs {ts[0]:.1f} us  pp {ts[1]:.1f} us" if ts else ""), flush=True)
        bad += (not ok) or (not fin)
    for sh in BNS:
        B, Cout, C0, H, W = sh
        g = torch.Generator().manual_seed(2)
        dy = torch.randn(B, H, W, Cout, generator=g).to(DEV).to(dt)
        w = (torch.randn(Cout, C0, 3, 3, generator=g) / (3.0 * Cout ** 0.5)).to(DEV)
        yv = torch.randn(B, H, W, C0, generator=g).to(DEV).to(dt)
        a = (torch.rand(C0, generator=g) + 0.5).to(DEV); b = (torch.randn(C0, generator=g) * 0.3).to(DEV)
        mean = (torch.randn(C0, generator=g) * 0.1).to(DEV); invstd = (torch.rand(C0, generator=g) + 0.5).to(DEV)
        outs, ts = [], []
        for m in (3, 4):
            dx = torch.full((B, H, W, C0), float('nan'), device=DEV, dtype=dt)
            s1 = torch.empty(C0, device=DEV); s2 = torch.empty(C0, device=DEV)
            lib.fu_test_conv_tile_mode(m)
            f = lambda: check(lib.fu_op_conv3x3_dgrad_bnsums(code, ptr(dy), Cout, ptr(w), ptr(dx), C0, ptr(yv), ptr(a), ptr(b),
                                                             ptr(mean), ptr(invstd), ptr(s1), ptr(s2), B, H, W, s()))
            f(); torch.cuda.synchronize()
            if TIME and code == 1: ts.append(timed(f))
            outs.append((dx.clone(), s1.clone(), s2.clone()))
        lib.fu_test_conv_tile_mode(0)
        tall = B * (H // 32) * (W // 16) * (C0 // 64) >= 512
        eq = lambda p, q: torch.equal(p.view(torch.int32) if p.dtype == torch.float32 else p.view(torch.int16),
                                      q.view(torch.int32) if q.dtype == torch.float32 else q.view(torch.int16))
        close = lambda p, q: bool(((p - q).abs() <= 1e-5 * (p.abs().max() + 1)).all())
        ok = eq(outs[0][0], outs[1][0]) and all((eq if tall else close)(p, q) for p, q in zip(outs[0][1:], outs[1][1:]))
        fin = all(bool(torch.isfinite(p.float()).all()) for p in outs[1])
        print(f"{nm} dgrad+bnsums {sh}: identical={ok} finite={fin}" + (f"  rs {ts[0]:.1f} us  pp {ts[1]:.1f} us" if ts else ""), flush=True)
        bad += (not ok) or (not fin)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
