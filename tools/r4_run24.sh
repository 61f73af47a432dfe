#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', d['value'], d['ms_per_step'], c.get('detect_api_images_per_s'), c.get('detect_stream_api_images_per_s'), c.get('sustained_images_per_s'))"; }
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-api 2>/dev/null | line "no-api"
python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line "api"
YOLO_NMS_PRIORITY=0 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line "api nms-prio-0"
YOLO_BENCH_SKIP_STREAM_API=1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line "api no-stream-api"
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-api 2>/dev/null | line "no-api again"
