#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "t20_stride2" > gpurun_out/r4_tests12.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r4_tests12.log
[ $rc -eq 0 ] || exit 1
for d in 0 16777216; do echo "YOLO_CONV_DEBUG=$d"; YOLO_CONV_DEBUG=$d python tools/conv_micro.py --reps 30 32,320,320,64,128,3,2 2>&1 | grep -v amdgpu; done
for d in 0 16777216; do echo "layer table YOLO_CONV_DEBUG=$d: $(YOLO_CONV_DEBUG=$d python tools/layer_profile.py --workload spp --bs 32 --compact 2>&1 | awk 'NR==5{print $8}')"; done
