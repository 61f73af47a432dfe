#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "detect_stream or pipelined_detect" > gpurun_out/r4_tests6.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4_tests6.log
python tools/host_profile.py tiny stream 400 2>&1 | head -3
python tools/host_profile.py mobile stream 200 2>&1 | head -3
python tools/detect_modes.py spp 30 2>&1 | tail -6
python tools/detect_modes.py tiny 200 2>&1 | tail -6
