#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
YOLO_BENCH_SHARDED_AT_1=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4_sharded1.err | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=d['config']; print('sharded@1 (RCCL all-gather in the step):', d['value'], d['ms_per_step'], c.get('sharding'), c.get('detect_api_images_per_s'))"
tail -3 gpurun_out/r4_sharded1.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_driver_cmd.json ) 2>&1 | grep real
