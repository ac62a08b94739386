#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu -x > gpurun_out/r3_t4.log 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r3_t4.log
python3 -m pytest tests/test_gpu_unet.py -q -m gpu -s -k "miou" 2>&1 | grep -E "oracle_fp32|passed|failed" | head
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"
cat gpurun_out/r3_bench1.json; tail -3 gpurun_out/r3_bench1.err
