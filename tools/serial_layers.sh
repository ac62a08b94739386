#!/bin/bash
# usage (GPU box): tools/serial_layers.sh TAG [ENV=VAL ...]  -> one-stream kernel trace of bench.py, per-layer table (tools/per_layer.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; shift
for e in "$@"; do export "$e"; done
export FU_NO_SIDE_STREAM=1
rm -rf gpurun_out/sl_$T
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sl_$T -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-miou --no-loader --no-eval > gpurun_out/sl_$T.log 2>&1 || { tail -5 gpurun_out/sl_$T.log; exit 1; }
python3 tools/per_layer.py gpurun_out/sl_$T > gpurun_out/sl_$T.txt; cat gpurun_out/sl_$T.txt
