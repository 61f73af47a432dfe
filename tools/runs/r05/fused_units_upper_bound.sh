#!/bin/bash
# Round 5, call 12: upper bound of what a persistent form of the fused units can gain - YOLO_RESUNIT_DEBUG bit 2048 (timing only,
# results wrong) does not wait for a tile's first x chunk (128-channel unit) / its halo (64-channel unit), as if they had been
# requested during the previous tile's epilogue; bit 8: no epilogue.  Kernel alone (tools/resunit_micro.py) and the whole step.
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
for D in 0 2048 8 2056; do
  echo "YOLO_RESUNIT_DEBUG=$D" >> $O/r5v_ru_micro.txt
  YOLO_RESUNIT_DEBUG=$D timeout -k 10 120 python tools/resunit_micro.py 32 2>&1 | grep -v amdgpu >> $O/r5v_ru_micro.txt
done
cat $O/r5v_ru_micro.txt
for i in 1 2 3; do
  for D in 0 2048; do
    YOLO_RESUNIT_DEBUG=$D timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-api --no-sustained > $O/r5v_b.json 2> $O/r5v_b.err
    python - "$i" "$D" <<'PY' | tee -a gpurun_out/r5v_ab.txt
import json, sys
j = json.loads(open("gpurun_out/r5v_b.json").read().strip().splitlines()[-1])
print("round", sys.argv[1], "YOLO_RESUNIT_DEBUG", sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
  done
done
