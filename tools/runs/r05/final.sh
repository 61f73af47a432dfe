#!/bin/bash
# Round 5, final call: the whole GPU suite at HEAD, smoke(), then the bench lines of the workloads whose kernels changed late in the
# round (YOLOv3-tiny, MobileNetV2-tiny) and the headline once more.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/r5z_gpu_suite.log 2>&1; rc=$?; echo "suite rc $rc"; tail -4 $O/r5z_gpu_suite.log
[ $rc -ne 0 ] && exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for WL in tiny mobile; do
  timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline > $O/r5z_bench_$WL.json 2> $O/r5z_bench_$WL.err; echo "bench $WL rc $?"
  python tools/layer_profile.py --workload $WL --compact > $O/r5z_layers_$WL.txt 2>&1
done
timeout -k 10 400 python bench.py > $O/r5z_bench_spp.json 2> $O/r5z_bench_spp.err; echo "bench spp rc $?"
python - <<'PY'
import json
for f in ("tiny", "mobile", "spp"):
    j = json.loads(open(f"gpurun_out/r5z_bench_{f}.json").read().strip().splitlines()[-1])
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"].get("detect_api_images_per_s"), j["config"].get("detect_stream_api_images_per_s"), j["config"].get("sustained_images_per_s"))
PY
grep total $O/r5z_layers_tiny.txt $O/r5z_layers_mobile.txt
