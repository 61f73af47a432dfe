#!/bin/bash
# Round-5 call 5: the candidate-row filter epilogue (head_epilogue.h) under the tiled and the pipelined head kernels.  -> gpurun_out/r5e_*
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "head_decode or detect or benched or pipelined or nms" > $O/r5e_head_tests.log 2>&1; echo "head tests rc $?"; tail -3 $O/r5e_head_tests.log
python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r5e_layers_spp.txt 2>&1
YOLO_CONV_DEBUG=67108864 python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r5e_layers_spp_tiledhead.txt 2>&1
grep -E "head|total" $O/r5e_layers_spp.txt $O/r5e_layers_spp_tiledhead.txt
run() { timeout -k 10 200 python bench.py $2 --steps 60 --warmup 10 --no-cpu-baseline --no-api --no-sustained 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['ms_one_list_start_to_end'])"; }
: > $O/r5e_ab.txt
for r in 1 2 3; do
  run "round $r pipelined heads" >> $O/r5e_ab.txt
  YOLO_CONV_DEBUG=67108864 run "round $r tiled heads" >> $O/r5e_ab.txt
done
run "tiny" "--workload tiny" >> $O/r5e_ab.txt
run "mobile" "--workload mobile" >> $O/r5e_ab.txt
cat $O/r5e_ab.txt
python tools/layer_profile.py --workload tiny --compact > $O/r5e_layers_tiny.txt 2>&1; grep -E "head|total" $O/r5e_layers_tiny.txt
