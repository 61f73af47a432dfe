#!/bin/bash
# Per-launch marginal cost inside the pipelined step for YOLOv3-tiny and MobileNetV2-tiny (tools/marginal_cost.py --each)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 700 python tools/marginal_cost.py profiles/r05_layers_tiny.txt --workload tiny --each > gpurun_out/r5M_marginal_tiny.txt 2> gpurun_out/r5M_marginal_tiny.err; echo "tiny rc $?"
tail -24 gpurun_out/r5M_marginal_tiny.txt
timeout -k 10 900 python tools/marginal_cost.py profiles/r05_layers_mobile.txt --workload mobile --each > gpurun_out/r5M_marginal_mobile.txt 2> gpurun_out/r5M_marginal_mobile.err; echo "mobile rc $?"
tail -32 gpurun_out/r5M_marginal_mobile.txt
