#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), r['achieved'], r.get('achieved_serial'))" || exit 1; }
for r in 1 2 3; do
FU_HEAD_STORE_G=1 run stored
run recompute
done
timeout -k 10 200 python3 tools/phase_time.py
export FU_NO_SIDE_STREAM=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_ks_t2s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-eval > gpurun_out/r3_ks_t2s.log 2>&1
python3 tools/per_layer.py gpurun_out/r3_ks_t2s > gpurun_out/r3_t2s_per_layer.txt 2>&1; tail -3 gpurun_out/r3_t2s_per_layer.txt
