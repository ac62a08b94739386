#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/exp.so
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-serial-pass --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['achieved'])"
}
for rep in 1 2; do
  run base FU_DUMMY=0
  run wg224 FU_WGRAD_TARGET=224
  run wg192 FU_WGRAD_TARGET=192
  run wg128 FU_WGRAD_TARGET=128
  run prio_default FU_SIDE_PRIO_DEFAULT=1
done
