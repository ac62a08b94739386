cd $GRAFT_REPO_ROOT
for e in "FU_X=0" "FU_NO_SIDE_STREAM=1"; do
  env $e timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --graph 1 --no-cpu-baseline --no-miou --no-loader --no-eval 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e graph', d['value'], d['ms_per_step'], d.get('step_other_mode'))"
done
