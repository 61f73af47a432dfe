#!/bin/bash
# SQ counters of the MobileNetV2 kernels (what the waves of the fused blocks wait for): three separate --pmc passes, kernel trace only beside them
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf $O/mbpmc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/mbpmc_$i -- python3 bench.py --workload mobile --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/mbpmc_$i.json 2> $O/mbpmc_$i.err; echo "pass $i rc $?"
done
