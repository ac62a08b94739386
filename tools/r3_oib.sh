#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_unet.py -q -m gpu -x -k "inside_backward or captured or full_size or fixture" > gpurun_out/r3_oib_t.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3_oib_t.log
run() {  # label, env assignments...
  label=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-eval --no-serial-pass 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', d['value'], d['ms_per_step'], d['ms_per_step_median'], r['achieved'])"
}
for rep in 1 2 3; do
  run separate FU_NO_FUSED_OPTIMIZER=1
  run fused FU_DUMMY=1
done
