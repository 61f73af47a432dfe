#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "t20_stride2 or t20_epilogue" > gpurun_out/r4_tests13.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r4_tests13.log
[ $rc -eq 0 ] || exit 1
for d in 0 16777216; do echo "YOLO_CONV_DEBUG=$d"; YOLO_CONV_DEBUG=$d python tools/conv_micro.py --reps 30 32,320,320,64,128,3,2 32,160,160,128,256,3,2 32,80,80,256,512,3,2 2>&1 | grep -v amdgpu; done
for d in 0 16777216; do echo "layer table YOLO_CONV_DEBUG=$d: $(YOLO_CONV_DEBUG=$d python tools/layer_profile.py --workload spp --bs 32 --compact 2>&1 | awk 'NR==5||NR==8||NR==25{printf "%s ", $8}')"; done
for r in 1 2 3; do for d in 0 16777216; do
  YOLO_CONV_DEBUG=$d python bench.py --steps 100 --warmup 10 --no-api --no-cpu-baseline --no-sustained 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug $d', d['value'], d['ms_per_step'])"
done; done
