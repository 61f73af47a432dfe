#!/bin/bash
# one-rank RCCL rehearsal of the sharded bench path (all-gather on the side stream): tools/shard_ab.sh "ENV=VAL ..." ...
for envs in "$@"; do
  env $envs YOLO_BENCH_SHARDED_AT_1=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$envs', d['value'], d['ms_per_step'], d['roofline']['ms_per_step_conv'], d['config'].get('detect_api_images_per_s'), d['config'].get('cu_partition'))"
done
