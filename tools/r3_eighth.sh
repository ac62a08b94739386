#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu > gpurun_out/r3_t8.log 2>&1; echo "all tests rc=$?"; tail -4 gpurun_out/r3_t8.log
export FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/exp.so
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-serial-pass --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['achieved'])"
}
for rep in 1 2; do
  run base176 FU_DUMMY=0
  run c64_384 FU_WGRAD_TARGET64=384
  run c64_320 FU_WGRAD_TARGET64=320
  run c64_256 FU_WGRAD_TARGET64=256
done
