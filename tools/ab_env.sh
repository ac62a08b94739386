#!/bin/bash
# usage (GPU box): tools/ab_env.sh "VAR=a" "VAR=b" ...   bench.py twice under each environment setting (same box, same library)
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  for r in 1 2; do
    env $e timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-miou --no-loader 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d.get('ms_per_step_median'), d['roofline']['achieved'], d['roofline'].get('achieved_serial'))" || exit 1
  done
done
