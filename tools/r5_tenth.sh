#!/bin/bash
# Round 5, call 10: wide inverted-residual kernel with LDS-only barriers where no DMA'd data is needed (YOLO_MBWIDE_DEBUG=32: the old barriers)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inverted_residual or mobile or Mobile or mbconv or small_models or secondary" > $O/r5t_tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/r5t_tests.log
[ $rc -ne 0 ] && exit 1
python tools/layer_profile.py --workload mobile --compact > $O/r5t_layers_mobile.txt 2>&1
YOLO_MBWIDE_DEBUG=32 python tools/layer_profile.py --workload mobile --compact > $O/r5t_layers_mobile_old.txt 2>&1
paste <(grep mbconv $O/r5t_layers_mobile_old.txt | awk '{print $1, $4, $5, $8}') <(grep mbconv $O/r5t_layers_mobile.txt | awk '{print $8}')
grep total $O/r5t_layers_mobile*.txt
for i in 1 2 3; do
  for L in 32 0; do
    YOLO_MBWIDE_DEBUG=$L timeout -k 10 200 python bench.py --workload mobile --no-cpu-baseline --no-api --no-sustained > $O/r5t_m.json 2> $O/r5t_m.err
    python - "$i" "$L" <<'PY' | tee -a gpurun_out/r5t_ab.txt
import json, sys
j = json.loads(open("gpurun_out/r5t_m.json").read().strip().splitlines()[-1])
print("round", sys.argv[1], "YOLO_MBWIDE_DEBUG", sys.argv[2], j["value"], j["ms_per_step"])
PY
  done
done
for L in 32 0; do
  YOLO_MBWIDE_DEBUG=$L timeout -k 10 200 python bench.py --workload mobile --streams 1 --no-cpu-baseline --no-api --no-sustained > $O/r5t_m1.json 2> $O/r5t_m1.err
  python - "$L" <<'PY' | tee -a gpurun_out/r5t_ab.txt
import json, sys
j = json.loads(open("gpurun_out/r5t_m1.json").read().strip().splitlines()[-1])
print("one pipeline, YOLO_MBWIDE_DEBUG", sys.argv[1], j["value"], j["ms_per_step"])
PY
done
