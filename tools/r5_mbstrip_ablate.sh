#!/bin/bash
# Role ablation of the row-strip inverted-residual kernel (timing only: results are wrong with the bits set).  -> gpurun_out/r5h_ablate.txt
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out/r5h_ablate.txt
: > $O
for D in 128 130 132 136 142; do
  echo "YOLO_MBCONV_DEBUG=$D (128 strip; +2 no expand, +4 no depthwise, +8 no projection)" >> $O
  YOLO_MBCONV_DEBUG=$D python tools/layer_profile.py --workload mobile --compact 2>/dev/null | grep -E "^ *[1-6] mbconv" >> $O
done
cat $O
