#!/bin/bash
# phase ablation of the narrow inverted-residual kernel (timing only)
mkdir -p gpurun_out
for dbg in 0 2 4 8 6 14; do
  echo "== debug $dbg"
  YOLO_MBCONV_DEBUG=$dbg timeout -k 10 200 python tools/layer_profile.py --workload mobile 2>&1 | grep mbconv | head -7 | awk '{printf "%s/%s/s%s@%s:%s ", $5,$4,$7,$3,$8} END {print ""}'
done > gpurun_out/mbn_dbg.txt 2>&1
cat gpurun_out/mbn_dbg.txt
