#!/bin/bash
# usage: tools/ab_serial.sh lib0.so lib1.so ...  (under tools/dbglibs/): one-stream rocprofv3 kernel statistics of bench.py per
# variant, printing the per-step totals of the kernel classes -- the comparison that does not depend on how the two backward
# streams share the GPU.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FU_NO_SIDE_STREAM=1
for l in "$@"; do
  export FU_LIB_PATH="$GRAFT_REPO_ROOT/tools/dbglibs/$l"
  rm -rf gpurun_out/abs_$l
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abs_$l -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-loader > gpurun_out/abs_$l.log 2>&1 || exit 1
  f=$(find gpurun_out/abs_$l -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$l" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
cls = {'conv': ('k_conv3x3_',), 'wgrad': ('k_wgrad_bf16',), 'wgrad_red': ('k_wgrad_reduce', 'k_wgrad_transpose'), 'bn_bwd': ('k_bn_bwd',)}
tot = {k: 0.0 for k in cls}; allt = 0.0
for r in rows:
    c = int(r['Calls'])
    if c % 13 or 'at::' in r['Name']: continue
    t = int(r['TotalDurationNs']) / 13 / 1000; allt += t
    for k, pats in cls.items():
        if any(p in r['Name'] for p in pats): tot[k] += t
print(sys.argv[2], ' '.join(f"{k} {v:.1f}" for k, v in tot.items()), f"all {allt:.1f} us/step")
PY
done
