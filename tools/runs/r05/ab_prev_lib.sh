#!/bin/bash
# Generic A/B of the library at HEAD against tools/dbg/ab/libyolo_hip_prev.so (one source file of the previous commit linked with the current
# objects):  bash tools/runs/r05/ab_prev_lib.sh <workload> <tag> "<pytest -k expression>"
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
WL=$1; TAG=$2; KEXPR=$3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" > $O/${TAG}_tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/${TAG}_tests.log
[ $rc -ne 0 ] && exit 1
python tools/layer_profile.py --workload $WL --compact > $O/${TAG}_layers.txt 2>&1
YOLO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/ab/libyolo_hip_prev.so python tools/layer_profile.py --workload $WL --compact > $O/${TAG}_layers_prev.txt 2>&1
paste <(grep -E "^ +[0-9]+ " $O/${TAG}_layers_prev.txt | awk '{print $1, $2, $8}') <(grep -E "^ +[0-9]+ " $O/${TAG}_layers.txt | awk '{print $8}') | head -30
grep total $O/${TAG}_layers*.txt
for i in 1 2 3; do
  for L in prev new; do
    if [ $L = prev ]; then export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/ab/libyolo_hip_prev.so; else unset YOLO_HIP_LIB; fi
    timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-api --no-sustained > $O/${TAG}_b.json 2> $O/${TAG}_b.err
    python - "$i" "$L" "$TAG" <<'PY' | tee -a gpurun_out/${TAG}_ab.txt
import json, sys
j = json.loads(open(f"gpurun_out/{sys.argv[3]}_b.json").read().strip().splitlines()[-1])
print("round", sys.argv[1], sys.argv[2], j["value"], j["ms_per_step"])
PY
  done
done
unset YOLO_HIP_LIB
