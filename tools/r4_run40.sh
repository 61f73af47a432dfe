#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "t20_stride2 or t20_epilogue or t20_conv" > gpurun_out/r4_tests11.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r4_tests11.log
[ $rc -eq 0 ] || exit 1
for d in 0 16777216; do echo "YOLO_CONV_DEBUG=$d"; YOLO_CONV_DEBUG=$d python tools/conv_micro.py --reps 30 32,320,320,64,128,3,2 32,160,160,128,256,3,2 2>&1 | grep -v amdgpu; done
YOLO_CONV_DEBUG=8388608 python tools/conv_micro.py --reps 30 32,160,160,128,256,3,2 32,80,80,256,512,3,2 2>&1 | grep -v amdgpu
