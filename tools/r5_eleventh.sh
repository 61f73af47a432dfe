#!/bin/bash
# Round 5, call 11: conv3x3_small (YOLOv3-tiny layers 1-2) with the pooled tile stored one tile late + LDS-only barriers
# (YOLO_SMALL_DEBUG=1: the old order); new residual cases of the inverted-residual test.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inverted_residual or tiny or Tiny or small or pool or lite or Lite" > $O/r5u_tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/r5u_tests.log
[ $rc -ne 0 ] && exit 1
python tools/layer_profile.py --workload tiny --compact > $O/r5u_layers_tiny.txt 2>&1
YOLO_SMALL_DEBUG=1 python tools/layer_profile.py --workload tiny --compact > $O/r5u_layers_tiny_old.txt 2>&1
paste <(sed -n 3,6p $O/r5u_layers_tiny_old.txt | awk '{print $1, $4, $5, $8}') <(sed -n 3,6p $O/r5u_layers_tiny.txt | awk '{print $8}')
grep total $O/r5u_layers_tiny*.txt
for i in 1 2 3; do
  for L in 1 0; do
    YOLO_SMALL_DEBUG=$L timeout -k 10 200 python bench.py --workload tiny --no-cpu-baseline --no-api --no-sustained > $O/r5u_t.json 2> $O/r5u_t.err
    python - "$i" "$L" <<'PY' | tee -a gpurun_out/r5u_ab.txt
import json, sys
j = json.loads(open("gpurun_out/r5u_t.json").read().strip().splitlines()[-1])
print("round", sys.argv[1], "YOLO_SMALL_DEBUG", sys.argv[2], j["value"], j["ms_per_step"])
PY
  done
done
