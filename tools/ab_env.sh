#!/bin/bash
# usage: tools/ab_env.sh VAR  -> bench.py alternately without / with VAR=1 (same box)
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export $1=1; else unset $1; fi
  timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
