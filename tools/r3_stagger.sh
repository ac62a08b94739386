#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), r['achieved'], r.get('achieved_serial'))" || exit 1; }
for r in 1 2; do
FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/head.so run head
run new
export FU_LIB_PATH=$GRAFT_REPO_ROOT/tools/dbglibs/exp.so
run exp0
FU_RS_STAGGER=20000,2 run st20k_2
FU_RS_STAGGER=40000,2 run st40k_2
FU_RS_STAGGER=40000,4 run st40k_4
FU_RS_STAGGER=30000,8 run st30k_8
unset FU_LIB_PATH
done
