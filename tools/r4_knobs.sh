#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="python bench.py --steps 100 --warmup 10 --no-api --no-cpu-baseline --no-sustained"
for r in 1 2; do
$B 2>/dev/null | line "base"
$B --streams 4 2>/dev/null | line "streams4"
YOLO_NMS_PRIORITY=0 $B 2>/dev/null | line "nms-prio-0"
YOLO_FUSE_RESUNIT=64 $B 2>/dev/null | line "fuse64"
YOLO_FUSE_RESUNIT=448 $B 2>/dev/null | line "fuse64+128+256"
YOLO_NMS_STREAM=0 $B 2>/dev/null | line "nms-on-pipeline-stream"
done
