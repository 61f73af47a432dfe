#!/bin/bash
# Cache policy of the fused residual units' y stores (buffer_store aux field: 0 default, 1 sc0, 2 nt, 3 sc0 + nt): the residual re-read of a
# unit misses L2 because the tile's own stores evict x; do streaming stores leave x in L2?  Libraries: tools/dbg/ab/libyolo_hip_aux{1,2,3}.so
# (conv_resunit_t20.hip built with -DYOLO_RU_STORE_AUX=n, linked with the shipped objects).
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
for A in 0 2 1 3; do
  if [ $A = 0 ]; then unset YOLO_HIP_LIB; else export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/ab/libyolo_hip_aux$A.so; fi
  echo "aux $A" >> $O/r5L_micro.txt
  timeout -k 10 120 python tools/resunit_micro.py 32 2>&1 | grep -v amdgpu | head -2 >> $O/r5L_micro.txt
done
cat $O/r5L_micro.txt
for i in 1 2 3; do
  for A in 0 2 3; do
    if [ $A = 0 ]; then unset YOLO_HIP_LIB; else export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/ab/libyolo_hip_aux$A.so; fi
    timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-api --no-sustained > $O/r5L_b.json 2> $O/r5L_b.err
    python - "$i" "$A" <<'PY' | tee -a gpurun_out/r5L_ab.txt
import json, sys
j = json.loads(open("gpurun_out/r5L_b.json").read().strip().splitlines()[-1])
print("round", sys.argv[1], "store aux", sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"]["mean_detections_per_image"])
PY
  done
done
