#!/bin/bash
# Round-5 call 2: the new parity tests, the 20x8-tile form of the 128-channel unit (micro + end to end), depth-first sub-batch orders,
# the sharded API after routing it through detect_step.   -> gpurun_out/r5b_*
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "depth_first or small_models or ring_of_four or pairing_rate or fused_residual_unit" > $O/r5b_tests.log 2>&1; echo "tests rc $?"; tail -3 $O/r5b_tests.log
grep -h "pairing rate\|benched list\]" $O/r5b_tests.log
python tools/resunit_micro.py 32 > $O/r5b_ru_micro.txt 2>&1
YOLO_RESUNIT_DEBUG=2048 python tools/resunit_micro.py 32 >> $O/r5b_ru_micro.txt 2>&1
cat $O/r5b_ru_micro.txt
run() { timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-api --no-sustained 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['ms_one_list_start_to_end'])"; }
: > $O/r5b_ab.txt
for r in 1 2; do
  for SPEC in "" "0-5:2" "0-5:4" "0-2:4" "0-2:4,2-5:2"; do
    YOLO_DEPTH_FIRST="$SPEC" run "round $r depth [$SPEC]" >> $O/r5b_ab.txt
  done
  YOLO_RESUNIT_DEBUG=2048 run "round $r ru128 on 20x8 tiles" >> $O/r5b_ab.txt
done
cat $O/r5b_ab.txt
YOLO_BENCH_SHARDED_AT_1=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/r5b_bench_sharded1.json 2> $O/r5b_bench_sharded1.err; echo "sharded rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5b_bench_sharded1.json").read().strip().splitlines()[-1])
print("sharded@1", j["value"], j["ms_per_step"], j["config"].get("detect_api_images_per_s"))
PY
