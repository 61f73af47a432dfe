#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -x -q -m gpu > $O/r4f_gpu_suite.log 2>&1; echo "suite rc $?"; tail -2 $O/r4f_gpu_suite.log
python bench.py --steps 100 --warmup 10 > $O/r4f_bench_spp.json 2> $O/r4f_bench_spp.err; echo "bench spp rc $?"
python bench.py --steps 20 --warmup 5 > $O/r4f_bench_spp_driver.json 2> $O/r4f_bench_spp_driver.err; echo "bench spp (driver flags) rc $?"
for wl in tiny mobile efficient; do
  python bench.py --workload $wl --steps 200 --warmup 20 > $O/r4f_bench_$wl.json 2> $O/r4f_bench_$wl.err; echo "bench $wl rc $?"
done
python - <<'PY'
import json
for wl in ("spp","spp_driver","tiny","mobile","efficient"):
    d=json.loads(open(f'gpurun_out/r4f_bench_{wl}.json').read().strip().splitlines()[-1])
    c=d["config"]; r=d["roofline"]
    print(wl, d["value"], d["ms_per_step"], {k:v for k,v in c.items() if "images_per_s" in k}, "frac", r["frac"], (r.get("power") or {}).get("socket_w"))
PY
