#!/bin/bash
cd $GRAFT_REPO_ROOT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$1', d['value'], d['ms_per_step'], c.get('detect_api_images_per_s'), c.get('detect_stream_api_images_per_s'), c.get('sustained_images_per_s'))"; }
GPU_MAX_HW_QUEUES=12 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line "api hwq12"
GPU_MAX_HW_QUEUES=16 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line "api hwq16"
GPU_MAX_HW_QUEUES=16 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-api 2>/dev/null | line "no-api hwq16"
python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line "api hwq8"
