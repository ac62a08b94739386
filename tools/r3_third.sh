#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_benched_dispatch.py tests/test_gpu_unet.py tests/test_gpu_distributed.py tests/test_gpu_ops.py -q -m gpu -s -x > gpurun_out/r3_t3.log 2>&1; echo "tests rc=$?"
grep -E "forced rs|default dispatch|worst per-block|passed|failed|Error|assert" gpurun_out/r3_t3.log | head -40
export FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/exp.so
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-serial-pass 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['achieved'])"
}
for rep in 1 2; do
  run base FU_EXP_SKIP=0
  run skip_bnfwd_fin FU_EXP_SKIP=1
  run skip_bnbwd_fin FU_EXP_SKIP=2
  run skip_wgrad_red FU_EXP_SKIP=4
  run skip_pack FU_EXP_SKIP=8
  run skip_bn_apply FU_EXP_SKIP=16
  run skip_ups_bwd FU_EXP_SKIP=32
  run skip_pool_bnb FU_EXP_SKIP=64
  run skip_all_fin FU_EXP_SKIP=3
done
