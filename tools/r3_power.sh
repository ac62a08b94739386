#!/bin/bash
# average socket power / clocks while bench.py runs (is the step power-limited?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40
( timeout -k 10 120 python3 bench.py --steps 3000 --warmup 5 --no-cpu-baseline --no-miou --no-eval --no-serial-pass > gpurun_out/r3_power_bench.json 2>/dev/null ) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $BP
tail -c 300 gpurun_out/r3_power_bench.json
