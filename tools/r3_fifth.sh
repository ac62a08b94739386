#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_unet.py -q -m gpu -s -x -k "captured or miou or fixture" > gpurun_out/r3_t5.log 2>&1; echo "tests rc=$?"
grep -E "oracle_fp32|passed|failed|Error|rror:" gpurun_out/r3_t5.log | head -20
for g in 0 1 0 1; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-eval --graph $g 2> gpurun_out/r3_bench_g$g.err | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph=$g', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['step_launch_mode'], d['step_other_mode'], d['roofline']['achieved'])"
done
tail -3 gpurun_out/r3_bench_g1.err
