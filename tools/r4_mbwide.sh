#!/bin/bash
# wide inverted-residual blocks: kernel tests, whole-model test, layer table, bench A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_inverted_residual or mobilenet" > gpurun_out/mbw_tests.log 2>&1
rc=$?; tail -5 gpurun_out/mbw_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/layer_profile.py --workload mobile > gpurun_out/mbw_layers.txt 2>&1 || exit 1
for i in 1 2; do
  timeout -k 10 300 python bench.py --workload mobile --no-cpu-baseline --no-sustained > gpurun_out/mbw_bench_wide_$i.json 2>gpurun_out/mbw_bench_wide_$i.err || exit 1
  YOLO_FUSE_MBCONV=narrow timeout -k 10 300 python bench.py --workload mobile --no-cpu-baseline --no-sustained > gpurun_out/mbw_bench_narrow_$i.json 2>gpurun_out/mbw_bench_narrow_$i.err || exit 1
done
grep -h -o '"value": [0-9.]*' gpurun_out/mbw_bench_*.json
