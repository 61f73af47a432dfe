#!/bin/bash
# Round-5 call 7: the row-strip form of the narrow inverted-residual blocks (conv_mbconv.hip mbstrip_kernel).  -> gpurun_out/r5g_*
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "inverted_residual or mobilenet" > $O/r5g_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/r5g_tests.log
python tools/layer_profile.py --workload mobile --compact > $O/r5g_layers_mobile.txt 2>&1
YOLO_MBCONV_DEBUG=128 python tools/layer_profile.py --workload mobile --compact > $O/r5g_layers_mobile_strip.txt 2>&1
grep -E "mbconv|total" $O/r5g_layers_mobile.txt | head -9; grep -E "mbconv|total" $O/r5g_layers_mobile_strip.txt | head -9
run() { timeout -k 10 200 python bench.py $2 --steps 60 --warmup 10 --no-cpu-baseline --no-api --no-sustained 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', j['value'], j['ms_per_step'], j['roofline']['frac'])"; }
: > $O/r5g_ab.txt
for r in 1 2; do
  run "round $r mobile tile form" "--workload mobile" >> $O/r5g_ab.txt
  YOLO_MBCONV_DEBUG=128 run "round $r mobile strip form" "--workload mobile" >> $O/r5g_ab.txt
done
cat $O/r5g_ab.txt
