#!/bin/bash
# phase ablation of the wide inverted-residual kernel (timing only)
mkdir -p gpurun_out
for nt in 512 1024; do
for dbg in 0 2 4 8 16 14 30; do
  echo "== NT $nt debug $dbg"
  YOLO_MBWIDE_NT=$nt YOLO_MBWIDE_DEBUG=$dbg timeout -k 10 200 python tools/layer_profile.py --workload mobile 2>&1 | grep mbconv | tail -10 | awk '{printf "%s/%s/%s:%s ", $5,$4,$7,$8} END {print ""}'
done; done > gpurun_out/mbw_dbg.txt 2>&1
cat gpurun_out/mbw_dbg.txt
