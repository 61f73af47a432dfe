#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "head_decode_filter or benched_launch or detect_stream or headline" > gpurun_out/r4_tests9.log 2>&1; rc=$?; echo "tests rc $rc"; tail -4 gpurun_out/r4_tests9.log
[ $rc -eq 0 ] || exit 1
python tools/layer_profile.py --workload spp --bs 32 --compact 2>&1 | grep "head+decode"
for r in 1 2 3; do
  for m in "--materialize-io" ""; do
    python bench.py --steps 60 --warmup 10 --no-api --no-cpu-baseline --no-sustained $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mode [$m]', d['value'], d['ms_per_step'], d['config']['mean_detections_per_image'])"
  done
done
