#!/bin/bash
# YOLOv3-tiny: tile forms of the 13x13 / 26x26 3x3 layers (YOLO_CONV_DEBUG bits), interleaved
mkdir -p gpurun_out
for r in 1 2; do for dbg in 0 128 256 16384; do
  v=$(YOLO_CONV_DEBUG=$dbg timeout -k 10 200 python bench.py --workload tiny --no-cpu-baseline --no-sustained --no-api --steps 200 --warmup 20 2>/dev/null | grep -o '"value": [0-9.]*')
  echo "round $r debug $dbg $v"
done; done > gpurun_out/tiny_variants.txt 2>&1
for dbg in 0 128; do echo "== debug $dbg"; YOLO_CONV_DEBUG=$dbg timeout -k 10 200 python tools/layer_profile.py --workload tiny --compact 2>&1 | grep -v amdgpu; done >> gpurun_out/tiny_variants.txt 2>&1
cat gpurun_out/tiny_variants.txt
