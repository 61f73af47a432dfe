#!/bin/bash
# A/B of the depth-first sub-batch orders of the first launches (engine.Plan._depth_first), interleaved rounds on one box:
#   bash tools/r5_depth_ab.sh [rounds]   -> gpurun_out/r5_depth_ab.txt
cd $GRAFT_REPO_ROOT
R=${1:-3}
O=gpurun_out/r5_depth_ab.txt
: > $O
for r in $(seq 1 $R); do
  for SPEC in "" "0-5:2" "0-5:4" "0-2:4" "0-2:4,2-5:2" "0-2:2" "0-2:8,2-5:2"; do
    YOLO_DEPTH_FIRST="$SPEC" timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-api --no-sustained 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r spec [$SPEC]', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['ms_one_list_start_to_end'])" >> $O
  done
done
cat $O
