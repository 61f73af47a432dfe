#!/bin/bash
for i in 1 2 3; do for v in 0 1; do
  YOLO_NMS_STREAM=$v python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-api 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nms_stream $v', d['value'], d['ms_per_step'], d['config']['mean_detections_per_image'])"
done; done
