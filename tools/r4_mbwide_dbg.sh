#!/bin/bash
# phase ablation of the wide inverted-residual kernel (timing only; the committed profiles/r04_mbconv_phase_ablation.txt also holds the
# removed 1024-thread build, which an earlier version of this script selected with YOLO_MBWIDE_NT)
mkdir -p gpurun_out
for dbg in 0 2 4 8 16 14 30; do
  echo "== debug $dbg"
  YOLO_MBWIDE_DEBUG=$dbg timeout -k 10 200 python tools/layer_profile.py --workload mobile 2>&1 | grep mbconv | tail -10 | awk '{printf "%s/%s/%s:%s ", $5,$4,$7,$8} END {print ""}'
done > gpurun_out/mbw_dbg.txt 2>&1
cat gpurun_out/mbw_dbg.txt
