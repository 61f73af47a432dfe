#!/bin/bash
# Round 5, call 14: the persistent streaming 1x1 with 256 / 384 instead of 512 workgroups (leaves half of a CU's LDS to the other pipeline's launch)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
for i in 1 2 3; do
  for D in 0 67108864 134217728; do
    YOLO_CONV_DEBUG=$D timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-api --no-sustained > $O/r5x_b.json 2> $O/r5x_b.err
    python - "$i" "$D" <<'PY' | tee -a gpurun_out/r5x_ab.txt
import json, sys
j = json.loads(open("gpurun_out/r5x_b.json").read().strip().splitlines()[-1])
print("round", sys.argv[1], "YOLO_CONV_DEBUG", sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
  done
done
