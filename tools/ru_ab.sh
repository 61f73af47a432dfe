#!/bin/bash
# the 64-channel residual unit: 20x20-tile kernel (default) vs the persistent 16x16-tile kernel (YOLO_RESUNIT_DEBUG=128), micro + bench
for v in 0 128; do echo "YOLO_RESUNIT_DEBUG=$v"; YOLO_RESUNIT_DEBUG=$v PYTHONPATH=. python tools/resunit_micro.py 32 2>&1 | grep -v amdgpu | head -2; done
for i in 1 2 3; do for v in 128 0; do
  YOLO_RESUNIT_DEBUG=$v python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('YOLO_RESUNIT_DEBUG $v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['mean_detections_per_image'])"
done; done
