#!/bin/bash
# Round 5, call 8: (a) per-role timeline of the row-strip inverted-residual kernel (diagnostic build, tools/mbstrip_timeline.py) with
# the role-ablation bits; (b) three pipelines for the small models (two and four were measured in round 4).
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
export YOLO_HIP_LIB=$GRAFT_REPO_ROOT/tools/dbg/mbstrip_stamps/libyolo_hip_mbstamps.so
for DBG in 0 2 4 8 14; do
  YOLO_MBCONV_DEBUG=$DBG timeout -k 10 200 python tools/mbstrip_timeline.py >> $O/r5r_mbstrip_timeline.txt 2>&1; echo "timeline dbg $DBG rc $?"
done
unset YOLO_HIP_LIB
for WL in tiny mobile; do
  for S in 2 3; do
    timeout -k 10 300 python bench.py --workload $WL --streams $S --no-cpu-baseline --no-api > $O/r5r_bench_${WL}_s$S.json 2> $O/r5r_bench_${WL}_s$S.err; echo "bench $WL streams $S rc $?"
  done
done
python - <<'PY'
import json
for wl in ("tiny", "mobile"):
    for s in (2, 3):
        try:
            j = json.loads(open(f"gpurun_out/r5r_bench_{wl}_s{s}.json").read().strip().splitlines()[-1])
            print(wl, "streams", s, j["value"], j["ms_per_step"], j["config"].get("sustained_images_per_s"))
        except Exception as e:
            print(wl, s, "unreadable", e)
PY
cat $O/r5r_mbstrip_timeline.txt
