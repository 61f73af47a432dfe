#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for m in auto off; do
    YOLO_CU_PARTITION=$m python bench.py --steps 100 --warmup 10 --no-api --no-cpu-baseline --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('partition $m', d['value'], d['ms_per_step'], d['config']['cu_partition'][:20])"
  done
done
