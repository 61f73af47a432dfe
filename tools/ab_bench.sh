#!/bin/bash
# interleaved end-to-end A/B of YOLO_CONV_PP values on one box:  tools/ab_bench.sh ROUNDS PP_A PP_B ...
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for pp in "$@"; do
    YOLO_CONV_PP=$pp python bench.py --steps 30 --warmup 5 --no-api --no-cpu-baseline --no-sustained 2>/dev/null > /tmp/ab_line.json
    python - "$pp" <<'PY'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
print("YOLO_CONV_PP", sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], flush=True)
PY
  done
done
