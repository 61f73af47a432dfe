#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for dbg in 0 131072; do
    YOLO_CONV_DEBUG=$dbg python bench.py --steps 60 --warmup 10 --no-api --no-cpu-baseline --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug $dbg', d['value'], d['ms_per_step'], d['config']['mean_detections_per_image'])"
  done
done
