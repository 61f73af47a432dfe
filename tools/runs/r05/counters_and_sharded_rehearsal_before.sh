#!/bin/bash
# Round-5 opening call on one box: GPU suite at HEAD, the headline bench line, the LDS / MFMA counter passes VERDICT r4 item 7 asks for
# (program directly after --), and the sharded code path rehearsed with one rank (item 5).   bash tools/runs/r05/<this file> -> gpurun_out/r5a_*
cd $GRAFT_REPO_ROOT
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r5a_gpu_suite.log 2>&1; echo "suite rc $?"; tail -2 $O/r5a_gpu_suite.log
timeout -k 10 400 python bench.py > $O/r5a_bench_spp.json 2> $O/r5a_bench_spp.err; echo "bench rc $?"
rm -rf $O/r5a_pmc_lds $O/r5a_pmc_mfma
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/r5a_pmc_lds -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/r5a_pmc_lds.json 2> $O/r5a_pmc_lds.err; echo "lds pass rc $?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r5a_pmc_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/r5a_pmc_mfma.json 2> $O/r5a_pmc_mfma.err; echo "mfma pass rc $?"
YOLO_BENCH_SHARDED_AT_1=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu-baseline > $O/r5a_bench_sharded1.json 2> $O/r5a_bench_sharded1.err; echo "sharded rc $?"
python - <<'PY'
import json
for f in ("r5a_bench_spp", "r5a_bench_sharded1"):
    try:
        j = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"].get("detect_api_images_per_s"), j["config"].get("detect_stream_api_images_per_s"))
    except Exception as e:
        print(f, "unreadable", e)
PY
