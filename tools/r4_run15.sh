#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "head_decode_filter" -s > gpurun_out/r4_tests8.log 2>&1; rc=$?; echo "filter tests rc $rc"; grep "compact NMS\|passed\|failed\|Error" gpurun_out/r4_tests8.log | tail -12
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_suite3.log 2>&1; rc=$?; echo "suite rc $rc"; tail -5 gpurun_out/r4_gpu_suite3.log
[ $rc -eq 0 ] || exit 1
for r in 1 2 3; do
  for m in "--materialize-io" ""; do
    python bench.py --steps 60 --warmup 10 --no-api --no-cpu-baseline --no-sustained $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode [$m]', d['value'], d['ms_per_step'], d['config']['mean_detections_per_image'])"
  done
done
