#!/bin/bash
# round 4, GPU call 1: new parity tests, whole GPU suite, bench line, one rocprofv3 --kernel-trace --stats pass (must exit 0 now)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "benched_launch_list or wide_logits or threshold_straddlers" -s > gpurun_out/r4_new_tests.log 2>&1; echo "new tests rc $?"
tail -5 gpurun_out/r4_new_tests.log
python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_suite.log 2>&1; echo "gpu suite rc $?"
tail -3 gpurun_out/r4_gpu_suite.log
python bench.py --steps 100 --warmup 10 > gpurun_out/r4_bench0.json 2> gpurun_out/r4_bench0.err; echo "bench rc $?"
cat gpurun_out/r4_bench0.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r4_kstats_spp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_kstats_spp -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-api --no-sustained > gpurun_out/r4_kstats_spp_bench.json 2> gpurun_out/r4_kstats_spp.err; echo "rocprofv3 rc $?"
tail -3 gpurun_out/r4_kstats_spp.err
