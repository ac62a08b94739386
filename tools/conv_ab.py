"""A/B timing of the bf16 3x3 forward kernels (general / fast) on UNet layer shapes.
Run under the profiler and parse its kernel trace (the op itself also packs weights and synchronises):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab -- python3 tools/conv_ab.py
    python3 tools/conv_ab.py --parse gpurun_out/ab
Every (shape, path) makes exactly 7 conv dispatches, in the order of SHAPES x (general, fast)."""
import sys, torch
sys.path.insert(0, '.')
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
lib = _lib.load()
DEV = 'cuda:0'

def run(B, C0, C1, Cout, H, W, general=0, path=0):
    g = torch.Generator(device='cpu').manual_seed(0)
    x0 = torch.randn(B, H, W, C0, generator=g).to(DEV).to(torch.bfloat16)
    x1 = torch.randn(B, H, W, C1, generator=g).to(DEV).to(torch.bfloat16) if C1 else None
    a = (torch.rand(C0, generator=g) + 0.5).to(DEV); b = (torch.randn(C0, generator=g) * 0.1).to(DEV)
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 10).to(DEV); bias = torch.zeros(Cout, device=DEV)
    y = torch.empty(B, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    lib.fu_test_force_general_conv(general)
    s = torch.cuda.current_stream().cuda_stream
    call = lambda: check(lib.fu_op_conv3x3_fwd(1, ptr(x0), C0, ptr(a), ptr(b), ptr(x1) if C1 else None, C1, ptr(w), ptr(bias), ptr(y), Cout, B, H, W, None, None, s))
    for _ in range(2): call()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    lib.fu_test_force_general_conv(0)
    return min(ts) * 1e3, y

SHAPES = [(16, 64, 0, 64, 256, 256), (16, 128, 0, 128, 128, 128), (16, 256, 0, 256, 64, 64), (16, 512, 0, 512, 32, 32),
          (16, 512, 0, 512, 16, 16), (16, 512, 512, 512, 32, 32), (16, 64, 64, 64, 256, 256), (16, 8, 0, 64, 256, 256)]
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    import csv, glob
    rows = []
    for fn in glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'k_conv3x3_bf16' in r['Kernel_Name']:
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']))
    rows.sort()
    assert len(rows) == 7 * 2 * len(SHAPES), len(rows)
    for i, sh in enumerate(SHAPES):
        B, C0, C1, Co, H, W = sh
        gf = 2 * 9 * (C0 + C1) * Co * B * H * W / 1e9
        out = []
        for j, nm in enumerate(('general', 'fast')):
            grp = rows[(i * 2 + j) * 7:(i * 2 + j + 1) * 7]
            t = min(d for _, d, _ in grp[2:]) / 1e3
            out.append(f"{nm} {t:.1f}us {gf / t:.0f}TF")
        print(sh, ' | '.join(out))
    sys.exit(0)
for sh in SHAPES:
    B, C0, C1, Co, H, W = sh
    gf = 2 * 9 * (C0 + C1) * Co * B * H * W / 1e9
    tg, yg = run(*sh, general=1); tf, yf = run(*sh)
    print(f"{sh}: general {tg:.0f}us | fast {tf:.0f}us (event times incl. weight packing; use --parse on a kernel trace)", flush=True)
