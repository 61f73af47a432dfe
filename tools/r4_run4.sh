#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "diverges or stay_inside" -s > gpurun_out/r4_tests3.log 2>&1; echo "tests rc $?"; grep -v "^\s*$" gpurun_out/r4_tests3.log | tail -8
python tools/host_profile.py spp detect 30 > gpurun_out/r4_host_spp_detect.txt 2>&1; head -40 gpurun_out/r4_host_spp_detect.txt
python tools/host_profile.py tiny stream 400 > gpurun_out/r4_host_tiny_stream.txt 2>&1; head -40 gpurun_out/r4_host_tiny_stream.txt
python bench.py --workload tiny --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r4_bench_tiny0.json 2>gpurun_out/r4_bench_tiny0.err; cut -c1-1500 gpurun_out/r4_bench_tiny0.json
