#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_benched_dispatch.py -q -m gpu -s > gpurun_out/r3_t_bd.log 2>&1; echo "benched-dispatch rc=$?"
grep -E "forced rs|worst per-block|passed|failed|Error|assert" gpurun_out/r3_t_bd.log | head -40
export FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/exp.so
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-serial-pass 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['achieved'])"
}
for rep in 1 2; do
  run base FU_DUMMY=1
  run wgrad256thr_512wg FU_WGRAD_MODE=1
  run wgrad256thr_256wg FU_WGRAD_MODE=2
  run reserve1of8 FU_SIDE_CU_RESERVE=1
  run reserve2of8 FU_SIDE_CU_RESERVE=2
  run noside FU_NO_SIDE_STREAM=1
done
