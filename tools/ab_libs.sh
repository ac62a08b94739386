#!/bin/bash
# usage: tools/ab_libs.sh lib0.so lib1.so ...   (variants under tools/dbglibs/ or absolute paths): bench.py with
# each on the same box.  The variant is selected through FU_LIB_PATH (floodplanet_code_amd/_lib.py); the product
# library in the tree is never overwritten.
cd $GRAFT_REPO_ROOT
for l in "$@"; do
  case "$l" in /*) lp="$l" ;; *) lp="$GRAFT_REPO_ROOT/tools/dbglibs/$l" ;; esac
  for r in 1 2; do
    FU_LIB_PATH="$lp" timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-loader 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d['roofline']['achieved'])" || exit 1
  done
done
