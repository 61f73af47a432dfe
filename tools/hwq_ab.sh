#!/bin/bash
for i in 1 2; do for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('hwq $q', d['value'], d['ms_per_step'], d['config'].get('detect_api_images_per_s'))"
done; done
