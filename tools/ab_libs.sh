#!/bin/bash
# usage: tools/ab_libs.sh lib0.so lib1.so ...   (variants under tools/dbglibs/): bench.py with each, same box
cd $GRAFT_REPO_ROOT
for l in "$@"; do
  cp tools/dbglibs/$l floodplanet_code_amd/libfloodunet.so
  for r in 1 2; do
    timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d['roofline']['achieved'])" || exit 1
  done
done
