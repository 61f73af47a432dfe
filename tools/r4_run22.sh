#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in tiny mobile; do
for st in 2 3 4; do
  [ $wl == mobile ] && [ $st == 3 ] && continue
  [ $wl == tiny ] && [ $st == 3 ] && continue
  python bench.py --workload $wl --streams $st --steps 200 --warmup 20 --no-api --no-cpu-baseline --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl streams $st', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
done
python bench.py --workload tiny --steps 200 --warmup 20 --no-api --no-cpu-baseline --no-sustained --materialize-io 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiny materialize-io', d['value'], d['ms_per_step'])"
