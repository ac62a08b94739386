#!/bin/bash
# usage (GPU box): tools/cm_ab.sh MODES lib0.so lib1.so ...  -> per-layer kernel times (tools/conv_modes.py under rocprofv3) for each
# variant library under tools/dbglibs/ ("product" = the library in the tree)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FU_CM_MODES=$1; shift
for l in "$@"; do
  if [ "$l" = product ]; then unset FU_LIB_PATH; else export FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/$l; fi
  rm -rf gpurun_out/cm_$l
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cm_$l -- python3 tools/conv_modes.py > gpurun_out/cm_$l.log 2>&1 || { echo "FAILED $l"; tail -5 gpurun_out/cm_$l.log; exit 1; }
  echo "== $l"; python3 tools/conv_modes.py --parse gpurun_out/cm_$l
done
