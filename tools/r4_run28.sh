#!/bin/bash
cd $GRAFT_REPO_ROOT
for ns in 1 per-pipeline; do
  echo "== YOLO_NMS_STREAMS=$ns"
  YOLO_NMS_STREAMS=$ns python tools/host_profile.py tiny stream 400 2>&1 | grep "images/s"
  YOLO_NMS_STREAMS=$ns python tools/host_profile.py mobile stream 200 2>&1 | grep "images/s"
  YOLO_NMS_STREAMS=$ns python bench.py --workload tiny --steps 200 --warmup 20 --no-api --no-cpu-baseline --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiny value', d['value'], d['ms_per_step'])"
done
