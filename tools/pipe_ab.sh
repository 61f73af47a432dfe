#!/bin/bash
# tools/pipe_ab.sh ROUNDS [bench args]: whole-batch pipelines vs sub-batch pipelines, interleaved
rounds=$1; shift
for i in $(seq 1 $rounds); do for v in halves batches; do
  python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-api --pipeline $v "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pipeline $v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['mean_detections_per_image'], d['config'].get('cu_partition'))"
done; done
