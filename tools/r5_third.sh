#!/bin/bash
# Round-5 call 3: pipelined streaming 1x1 (tests, per-layer table, end-to-end A/B against the first form), the tests call 2 did not reach,
# the 20x8-tile form of the 128-channel unit in isolation.   -> gpurun_out/r5c_*
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "conv1x1_stream or depth_first or small_models or ring_of_four or pairing_rate" > $O/r5c_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/r5c_tests.log
grep -h "pairing rate\|benched list\]" $O/r5c_tests.log
python tools/resunit_micro.py 32 > $O/r5c_ru_micro.txt 2>&1
YOLO_RESUNIT_DEBUG=2048 python tools/resunit_micro.py 32 >> $O/r5c_ru_micro.txt 2>&1
grep "C=128" $O/r5c_ru_micro.txt
python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r5c_layers_spp.txt 2>&1
YOLO_CONV_DEBUG=33554432 python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r5c_layers_spp_oldstream.txt 2>&1
grep -E "^ *(6|8|10|66|68) conv|total" $O/r5c_layers_spp.txt $O/r5c_layers_spp_oldstream.txt
run() { timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-api --no-sustained 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['ms_one_list_start_to_end'])"; }
: > $O/r5c_ab.txt
for r in 1 2 3; do
  run "round $r pipelined stream 1x1" >> $O/r5c_ab.txt
  YOLO_CONV_DEBUG=33554432 run "round $r first form" >> $O/r5c_ab.txt
done
cat $O/r5c_ab.txt
