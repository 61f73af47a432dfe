#!/bin/bash
# SQ counters of the stem kernel alone (tools/stem_micro.py): where its waves wait.   bash tools/dbg/stem_pmc.sh  (GPU box)
set +e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC" \
            "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/stem_pmc_$tag
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d gpurun_out/stem_pmc_$tag -- python3 tools/stem_micro.py 32 > gpurun_out/stem_pmc_$tag.log 2>&1
  echo "pass $tag done"
done
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/stem_pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stem' in r['Kernel_Name'] and 'stem2' in r['Kernel_Name'] or 'stem_kernel' in r['Kernel_Name']:
            t = tot[(r['Kernel_Name'][:40], r['Counter_Name'])]
            t[0] += float(r['Counter_Value']); t[1] += 1
for (k, c), (v, n) in sorted(tot.items()):
    print(f"{k:42s} {c:28s} {v / n:16.0f} per launch ({n} launches)")
PY
