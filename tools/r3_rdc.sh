#!/bin/bash
# A/B: product library (9 device code objects) against the -fgpu-rdc build (one code object); kernel trace of the latter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2; do
for l in floodplanet_code_amd/libfloodunet.so tools/dbglibs/rdc.so; do
  FU_LIB_PATH="$GRAFT_REPO_ROOT/$l" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['achieved'])" || exit 1
done
done
export FU_LIB_PATH="$GRAFT_REPO_ROOT/tools/dbglibs/rdc.so"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_ks_rdc -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-eval > gpurun_out/r3_ks_rdc.log 2>&1
python3 tools/timeline.py gpurun_out/r3_ks_rdc | head -12
