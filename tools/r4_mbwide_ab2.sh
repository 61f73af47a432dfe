#!/bin/bash
# wide blocks vs three-launch blocks, one and two pipelines
mkdir -p gpurun_out
for st in 1 2; do for mode in narrow 1; do
  echo "== streams $st fuse $mode"
  YOLO_FUSE_MBCONV=$mode timeout -k 10 300 python bench.py --workload mobile --no-cpu-baseline --no-sustained --no-api --streams $st 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'
done; done > gpurun_out/mbw_ab2.txt 2>&1
cat gpurun_out/mbw_ab2.txt
