#!/bin/bash
# Round 5, call 9: inverted-residual tile kernel with the x tile stashed between the depthwise phase and the projection's stores
# (no vmcnt(0) behind the y stores at the top of a tile, none in the depthwise phase): parity tests, per-layer table and A/B against
# the previous library (tools/dbg/ab/libyolo_hip_prev.so = HEAD's conv_mbconv.hip with today's other objects).
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inverted_residual or mobile or Mobile or mbconv or small_models or secondary" > $O/r5y_tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/r5y_tests.log
[ $rc -ne 0 ] && exit 1
python tools/layer_profile.py --workload mobile --compact > $O/r5y_layers_mobile.txt 2>&1
YOLO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/ab/libyolo_hip_prev.so python tools/layer_profile.py --workload mobile --compact > $O/r5y_layers_mobile_prev.txt 2>&1
paste <(grep mbconv $O/r5y_layers_mobile_prev.txt | awk '{print $1, $9}') <(grep mbconv $O/r5y_layers_mobile.txt | awk '{print $9}')
for i in 1 2 3; do
  for L in prev new; do
    if [ $L = prev ]; then export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/ab/libyolo_hip_prev.so; else unset YOLO_HIP_LIB; fi
    timeout -k 10 200 python bench.py --workload mobile --no-cpu-baseline --no-api --no-sustained > $O/r5y_m.json 2> $O/r5y_m.err
    python - "$i" "$L" <<'PY' | tee -a gpurun_out/r5y_ab.txt
import json, sys
j = json.loads(open("gpurun_out/r5y_m.json").read().strip().splitlines()[-1])
print("round", sys.argv[1], sys.argv[2], j["value"], j["ms_per_step"])
PY
  done
done
unset YOLO_HIP_LIB
for S in 1; do
  timeout -k 10 200 python bench.py --workload mobile --streams 1 --no-cpu-baseline --no-api --no-sustained > $O/r5y_m1.json 2> $O/r5y_m1.err
  YOLO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/ab/libyolo_hip_prev.so timeout -k 10 200 python bench.py --workload mobile --streams 1 --no-cpu-baseline --no-api --no-sustained > $O/r5y_m1p.json 2> $O/r5y_m1p.err
done
python - <<'PY' | tee -a gpurun_out/r5y_ab.txt
import json
for f, n in (("r5y_m1p", "one pipeline, prev"), ("r5y_m1", "one pipeline, new")):
    j = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(n, j["value"], j["ms_per_step"])
PY
