#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_benched_dispatch.py -m gpu -x -q > gpurun_out/r3_t1.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r3_t1.log
timeout -k 10 200 python3 tools/phase_time.py
for r in 1 2; do timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['achieved'])"; done
