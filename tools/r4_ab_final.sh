#!/bin/bash
# Interleaved end-to-end A/Bs behind the round-4 defaults, one box:  bash tools/r4_ab_final.sh > profiles/r04_ab_defaults.txt
cd $GRAFT_REPO_ROOT
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'images/s', d['ms_per_step'], 'ms/step', 'frac', d['roofline']['frac'])"; }
B="python bench.py --steps 100 --warmup 10 --no-api --no-cpu-baseline --no-sustained"
echo "# YOLOv3-SPP 640x640 x 32, python bench.py --steps 100 --warmup 10 --no-api --no-cpu-baseline --no-sustained [flags], five interleaved rounds on one MI355X"
for r in 1 2 3 4 5; do
  $B 2>/dev/null | line "default (shared chip, compact NMS form)      "
  $B --cu-partition 2>/dev/null | line "--cu-partition (half of every XCD per pipe) "
  $B --materialize-io 2>/dev/null | line "--materialize-io (io stored, plain NMS)     "
  $B --streams 1 2>/dev/null | line "--streams 1 (one pipeline)                  "
done
echo "# YOLOv3-tiny 416x416 x 32, --steps 300"
for r in 1 2 3; do
  python bench.py --workload tiny --steps 300 --warmup 20 --no-api --no-cpu-baseline --no-sustained 2>/dev/null | line "tiny default                                "
  python bench.py --workload tiny --steps 300 --warmup 20 --no-api --no-cpu-baseline --no-sustained --materialize-io 2>/dev/null | line "tiny --materialize-io                       "
done
