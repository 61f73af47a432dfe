#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "detect_stream or pipelined_detect or bench_line" > gpurun_out/r4_tests5.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r4_tests5.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python tools/host_profile.py tiny stream 400 > gpurun_out/r4_host_tiny_stream2.txt 2>&1; head -30 gpurun_out/r4_host_tiny_stream2.txt
python tools/host_profile.py spp stream 40 2>&1 | head -3
python tools/host_profile.py mobile stream 200 2>&1 | head -3
