#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "conv1 or first_layer or tiny or secondary or full_width or model_small or model_vs or lite or yolov3 or stay_inside or graph" > gpurun_out/r4_tests10.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r4_tests10.log
for k80 in 1 ""; do
  echo "YOLO_CONV1_K80=$k80"
  YOLO_CONV1_K80=$k80 python tools/layer_profile.py --workload tiny --compact 2>&1 | awk 'NR==3 || /total/'
done
for r in 1 2; do for k80 in 1 ""; do
  YOLO_CONV1_K80=$k80 python bench.py --workload tiny --steps 300 --warmup 20 --no-api --no-cpu-baseline --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiny K80=[$k80]', d['value'], d['ms_per_step'])"
done; done
