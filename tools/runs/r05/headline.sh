#!/bin/bash
# The committed headline line: `python bench.py` (defaults) twice and with the driver's flags twice, on one fresh box.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
for i in 1 2; do
  timeout -k 10 400 python bench.py > $O/r5q_bench_spp_$i.json 2> $O/r5q_bench_spp_$i.err; echo "bench spp $i rc $?"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r5q_bench_spp_driver_$i.json 2> $O/r5q_bench_spp_driver_$i.err; echo "bench spp (driver flags) $i rc $?"
done
python - <<'PY'
import json
for f in ("spp_1", "spp_driver_1", "spp_2", "spp_driver_2"):
    try:
        j = json.loads(open(f"gpurun_out/r5q_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"].get("detect_api_images_per_s"), j["config"].get("detect_stream_api_images_per_s"), j["config"].get("sustained_images_per_s"))
    except Exception as e:
        print(f, "unreadable", e)
PY
