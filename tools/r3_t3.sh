#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -m gpu -x -q > gpurun_out/r3_t3.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" gpurun_out/r3_t3.log | tail -4
export FU_NO_SIDE_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_ks_t3s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-eval > gpurun_out/r3_ks_t3s.log 2>&1
unset FU_NO_SIDE_STREAM
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_ks_t3 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-eval > gpurun_out/r3_ks_t3.log 2>&1
python3 - <<'PY'
import csv,glob
for d in ('r3_ks_t3s','r3_ks_t3'):
    f=glob.glob('gpurun_out/%s/*/*_kernel_stats.csv'%d)[0]
    for r in csv.DictReader(open(f)):
        if 'upsample' in r['Name'] or 'head' in r['Name']:
            print(d, r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3)
PY
