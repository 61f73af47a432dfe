#!/bin/bash
# Round-5 call 4: the pipelined head kernel (tests, per-layer table, end-to-end A/B against the tiled DECODE instances) + the whole GPU suite
# (the decode / filter epilogue moved into head_epilogue.h).   -> gpurun_out/r5d_*
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "head_decode" > $O/r5d_head_tests.log 2>&1; echo "head tests rc $?"; tail -3 $O/r5d_head_tests.log
python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r5d_layers_spp.txt 2>&1
YOLO_CONV_DEBUG=67108864 python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r5d_layers_spp_oldhead.txt 2>&1
grep -E "head|total" $O/r5d_layers_spp.txt $O/r5d_layers_spp_oldhead.txt
run() { timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-api --no-sustained 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['ms_one_list_start_to_end'])"; }
: > $O/r5d_ab.txt
for r in 1 2 3; do
  run "round $r pipelined heads" >> $O/r5d_ab.txt
  YOLO_CONV_DEBUG=67108864 run "round $r tiled heads" >> $O/r5d_ab.txt
done
cat $O/r5d_ab.txt
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r5d_gpu_suite.log 2>&1; echo "suite rc $?"; tail -3 $O/r5d_gpu_suite.log
grep -h "pairing rate\|benched list\]" $O/r5d_gpu_suite.log
