#!/bin/bash
# the block-wise backward of the N > 1 path against the plain one, one process: what the path itself costs
# (FU_DP_FORCE_BLOCKS=1: block-wise backward without a collective; FU_DP_JOIN_AT_BUCKETS=1: the compute stream joins the
# weight-gradient stream at every bucket end, as before round 3, instead of fencing the launch stream), then one nccl rank
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_distributed.py -m gpu -x -q > gpurun_out/r3_dist.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" gpurun_out/r3_dist.log | tail -3
run() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-eval --no-serial-pass 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))" || exit 1; }
for r in 1 2; do
run plain
FU_DP_FORCE_BLOCKS=1 run blocks_fenced
FU_DP_FORCE_BLOCKS=1 FU_DP_JOIN_AT_BUCKETS=1 run blocks_joined
done
FU_DIST_BACKEND=nccl timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tools/dp_diag.py 2>&1 | grep "^mode2"
FU_DP_JOIN_AT_BUCKETS=1 FU_DIST_BACKEND=nccl timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 tools/dp_diag.py 2>&1 | grep "^mode2"
