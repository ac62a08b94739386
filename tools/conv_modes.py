"""Per-layer timing of the 16-bit conv forward kernels by tile mode (0 = heuristic, 1 = square, 2 = tall, 3 = row-stationary).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cm -- python3 tools/conv_modes.py
    python3 tools/conv_modes.py --parse gpurun_out/cm
Every (shape, mode) makes exactly 6 conv dispatches in the order SHAPES x MODES; the minimum of the last 4 is reported."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODES = [int(m) for m in os.environ.get('FU_CM_MODES', '0,3').split(',')]
N = 6
# B, C0, C1, Cout, H, W  (UNet layers of the bench step, forward and dgrad shapes)
SHAPES = [(16, 64, 0, 64, 256, 256), (16, 64, 64, 64, 256, 256), (16, 64, 0, 128, 256, 256), (16, 64, 0, 128, 128, 128),
          (16, 128, 0, 128, 128, 128), (16, 128, 128, 128, 128, 128), (16, 128, 0, 256, 64, 64), (16, 256, 0, 256, 64, 64),
          (16, 256, 256, 256, 64, 64), (16, 256, 0, 512, 32, 32), (16, 512, 0, 512, 32, 32), (16, 512, 512, 512, 32, 32),
          (16, 512, 0, 256, 32, 32), (16, 512, 0, 512, 16, 16)]
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    import csv, glob
    rows = []
    for fn in glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'k_conv3x3' in r['Kernel_Name']:
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']))
    rows.sort()
    assert len(rows) == N * len(MODES) * len(SHAPES), len(rows)
    for i, sh in enumerate(SHAPES):
        B, C0, C1, Co, H, W = sh
        gf = 2 * 9 * (C0 + C1) * Co * B * H * W / 1e9
        out = []
        for j, m in enumerate(MODES):
            grp = rows[(i * len(MODES) + j) * N:(i * len(MODES) + j + 1) * N]
            t = min(d for _, d, _ in grp[2:]) / 1e3
            nm = grp[-1][2].split('(')[0].replace('void fu::', '')
            out.append(f"mode{m} {nm[:34]:34s} {t:7.1f}us {gf / t:6.0f}TF")
        print(f"{str(sh):34s}", ' | '.join(out))
    sys.exit(0)
import torch
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
lib = _lib.load()
DEV = 'cuda:0'
for sh in SHAPES:
    B, C0, C1, Cout, H, W = sh
    g = torch.Generator(device='cpu').manual_seed(0)
    x0 = torch.randn(B, H, W, C0, generator=g).to(DEV).to(torch.bfloat16)
    x1 = torch.randn(B, H, W, C1, generator=g).to(DEV).to(torch.bfloat16) if C1 else None
    a = (torch.rand(C0, generator=g) + 0.5).to(DEV); b = (torch.randn(C0, generator=g) * 0.1).to(DEV)
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / 10).to(DEV); bias = torch.zeros(Cout, device=DEV)
    y = torch.empty(B, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    ssum = torch.empty(Cout, device=DEV); ssq = torch.empty(Cout, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    outs = []
    for m in MODES:
        lib.fu_test_conv_tile_mode(m)
        for _ in range(N):
            check(lib.fu_op_conv3x3_fwd(1, ptr(x0), C0, ptr(a), ptr(b), ptr(x1) if C1 else None, C1, ptr(w), ptr(bias), ptr(y), Cout, B, H, W, ptr(ssum), ptr(ssq), s))
        torch.cuda.synchronize()
        outs.append(y.float().clone())
    lib.fu_test_conv_tile_mode(0)
    d = (outs[0] - outs[-1]).abs().max().item()
    print(sh, "max |first - second mode|", d, flush=True)
