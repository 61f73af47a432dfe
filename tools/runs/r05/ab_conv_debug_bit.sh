#!/bin/bash
# A/B of one YOLO_CONV_DEBUG bit on a workload:  bash tools/runs/r05/ab_conv_debug_bit.sh <workload> <bit> <tag> "<pytest -k expression>"
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
WL=$1; BIT=$2; TAG=$3; KEXPR=$4
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" > $O/${TAG}_tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/${TAG}_tests.log
[ $rc -ne 0 ] && exit 1
python tools/layer_profile.py --workload $WL --compact > $O/${TAG}_layers.txt 2>&1
YOLO_CONV_DEBUG=$BIT python tools/layer_profile.py --workload $WL --compact > $O/${TAG}_layers_bit.txt 2>&1
grep total $O/${TAG}_layers*.txt
for i in 1 2 3; do
  for D in $BIT 0; do
    YOLO_CONV_DEBUG=$D timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-api --no-sustained > $O/${TAG}_b.json 2> $O/${TAG}_b.err
    python - "$i" "$D" "$TAG" <<'PY' | tee -a gpurun_out/${TAG}_ab.txt
import json, sys
j = json.loads(open(f"gpurun_out/{sys.argv[3]}_b.json").read().strip().splitlines()[-1])
print("round", sys.argv[1], "YOLO_CONV_DEBUG", sys.argv[2], j["value"], j["ms_per_step"])
PY
  done
done
