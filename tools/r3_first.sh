#!/bin/bash
# round 3, first GPU call: the new parity anchors, the whole GPU suite, one bench line and a kernel trace with timestamps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_benched_dispatch.py -x -q -m gpu -s > gpurun_out/r3_t_bd.log 2>&1; echo "benched-dispatch rc=$?"
tail -25 gpurun_out/r3_t_bd.log
python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_benched_dispatch.py > gpurun_out/r3_t_all.log 2>&1; echo "all rc=$?"
tail -15 gpurun_out/r3_t_all.log
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err; echo "bench rc=$?"
cat gpurun_out/r3_bench0.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_ks0 -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou > gpurun_out/r3_ks0.log 2>&1; echo "trace rc=$?"
ls gpurun_out/r3_ks0/*/ | head
