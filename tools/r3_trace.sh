#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_ks2 -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-eval > gpurun_out/r3_ks2.log 2>&1; echo "trace rc=$?"
export FU_NO_SIDE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_ks2s -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-eval > gpurun_out/r3_ks2s.log 2>&1; echo "serial trace rc=$?"
