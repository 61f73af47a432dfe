#!/bin/bash
# sample socket power and clocks while the bench runs:  tools/power_sample.sh [bench args]
( python bench.py --steps 1500 --warmup 20 --no-api --no-cpu-baseline "$@" > /tmp/ps_line.json 2>/tmp/ps_err.txt ) &
bpid=$!
sleep 12
while kill -0 $bpid 2>/dev/null; do
  rocm-smi -d 0 --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use" | tr '\n' ' '
  echo
  sleep 0.7
done
wait $bpid
cat /tmp/ps_line.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
