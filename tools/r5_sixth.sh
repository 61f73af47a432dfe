#!/bin/bash
# Round-5 call 6: the 16-lane-group filter epilogue (head_epilogue.h) under the tiled head kernels.  -> gpurun_out/r5f_*
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "head_decode or detect or benched or pipelined or nms" > $O/r5f_head_tests.log 2>&1; echo "head tests rc $?"; tail -3 $O/r5f_head_tests.log
python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r5f_layers_spp.txt 2>&1
grep -E "head|total" $O/r5f_layers_spp.txt
python tools/layer_profile.py --workload tiny --compact > $O/r5f_layers_tiny.txt 2>&1; grep -E "head|total" $O/r5f_layers_tiny.txt
python tools/layer_profile.py --workload mobile --compact > $O/r5f_layers_mobile.txt 2>&1; grep -E "head|total" $O/r5f_layers_mobile.txt
run() { timeout -k 10 200 python bench.py $2 --steps 60 --warmup 10 --no-cpu-baseline --no-api --no-sustained 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', j['value'], j['ms_per_step'], j['roofline']['frac'])"; }
: > $O/r5f_ab.txt
for r in 1 2 3; do run "round $r spp" >> $O/r5f_ab.txt; done
run "tiny" "--workload tiny" >> $O/r5f_ab.txt
run "mobile" "--workload mobile" >> $O/r5f_ab.txt
cat $O/r5f_ab.txt
