#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "conv1 or fused_first or tiny or secondary" > gpurun_out/r4_tests7.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4_tests7.log
for lib in "" pytorch_yolo_amd/csrc/alt/libyolo_w3.so pytorch_yolo_amd/csrc/alt/libyolo_w4.so; do
  echo "== lib ${lib:-default}"
  YOLO_HIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/layer_profile.py --workload tiny --no-p 2>&1 | awk 'NR>2 && NR<7 || /total/'
done
python bench.py --workload tiny --steps 200 --warmup 20 --no-cpu-baseline --no-api 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiny', d['value'], d['ms_per_step'], d['roofline']['frac'])"
