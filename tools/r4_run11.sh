#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_suite2.log 2>&1; echo "suite rc $?"; tail -4 gpurun_out/r4_gpu_suite2.log
python tools/host_profile.py tiny stream 400 2>&1 | head -3
python tools/host_profile.py mobile stream 200 2>&1 | head -3
python tools/host_profile.py spp stream 40 2>&1 | head -3
python tools/host_profile.py tiny detect 200 2>&1 | head -3
python tools/host_profile.py spp detect 30 2>&1 | head -3
python bench.py --workload tiny --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r4_bench_tiny1.json 2>gpurun_out/r4_bench_tiny1.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_tiny1.json').read().strip().splitlines()[-1])
print("tiny", d["value"], d["ms_per_step"], {k:v for k,v in d["config"].items() if "images_per_s" in k}, d["roofline"]["frac"])
PY
