#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/exp.so
FU_RS_MODE=2 timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_benched_dispatch.py -q -m gpu -x -k "bf16_forward or bf16_dgrad or fused_bn or forced_row" > gpurun_out/r3_ws_t.log 2>&1; echo "ws tests rc=$?"; tail -5 gpurun_out/r3_ws_t.log
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], d['ms_per_step_median'], r['achieved'], r.get('achieved_serial'), r.get('achieved_serial_conv_only'))"
}
for rep in 1 2; do
  run base FU_RS_MODE=0
  run ws_small FU_RS_MODE=1
  run ws_all FU_RS_MODE=2
done
