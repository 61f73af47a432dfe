#!/bin/bash
# Round 5, call 13: marginal cost of every launch family inside the pipelined step (tools/marginal_cost.py)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 800 python tools/marginal_cost.py profiles/r05_layers_spp.txt --workload spp > gpurun_out/r5w_marginal_cost.txt 2> gpurun_out/r5w_marginal_cost.err; echo "rc $?"
cat gpurun_out/r5w_marginal_cost.txt; tail -3 gpurun_out/r5w_marginal_cost.err
