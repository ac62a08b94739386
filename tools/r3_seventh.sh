#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "bf16_forward" > gpurun_out/r3_t7.log 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/r3_t7.log
python3 -m pytest tests/test_gpu_unet.py tests/test_gpu_benched_dispatch.py -q -m gpu -x -k "full_size or fixture or block" > gpurun_out/r3_t7b.log 2>&1; echo "unet rc=$?"; tail -3 gpurun_out/r3_t7b.log
export FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/exp.so
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-serial-pass --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline']['achieved'])"
}
for rep in 1 2; do
  run base FU_DUMMY=0
  run wg208 FU_WGRAD_TARGET=208
  run wg192 FU_WGRAD_TARGET=192
  run wg176 FU_WGRAD_TARGET=176
  run wg160 FU_WGRAD_TARGET=160
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_ks1 -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-eval > gpurun_out/r3_ks1.log 2>&1; echo "trace rc=$?"
