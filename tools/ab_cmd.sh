#!/bin/bash
# tools/ab_cmd.sh ROUNDS -- bench args : repeated default-ish bench runs, prints value / ms
rounds=$1; shift; [ "$1" == "--" ] && shift
for r in $(seq 1 $rounds); do
  python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-api "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
