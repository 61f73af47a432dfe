#!/bin/bash
# interleaved end-to-end A/B of one environment knob on one box:  tools/ab_env.sh ROUNDS VAR VALUE_A VALUE_B ... [-- bench args]
rounds=$1; var=$2; shift 2
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq 1 $rounds); do
  for v in "${vals[@]}"; do
    env "$var=$v" python bench.py --steps 30 --warmup 5 --no-api --no-cpu-baseline --no-sustained "$@" 2>/tmp/ab_err.txt > /tmp/ab_line.json || { tail -5 /tmp/ab_err.txt; continue; }
    python - "$var" "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], flush=True)
PY
  done
done
