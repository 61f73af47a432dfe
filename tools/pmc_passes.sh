#!/bin/bash
# HBM traffic (two separate --pmc passes: FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2) and the kernel-trace statistics of
# bench.py on the GPU box:   bash tools/pmc_passes.sh [workload]      -> gpurun_out/pmc_<workload>_{f,w}/, gpurun_out/kstats_<workload>/
# (rocprofv3 may crash in its own teardown after the CSVs are written: the exit codes are ignored)
set +e
WL=${1:-spp}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_${WL}_f gpurun_out/pmc_${WL}_w gpurun_out/kstats_${WL}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${WL}_f -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > gpurun_out/pmc_${WL}_f.json 2> gpurun_out/pmc_${WL}_f.err
echo fetch pass done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${WL}_w -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > gpurun_out/pmc_${WL}_w.json 2> gpurun_out/pmc_${WL}_w.err
echo write pass done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_${WL} -- python3 bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-api --no-sustained > gpurun_out/kstats_${WL}_bench.json 2> gpurun_out/kstats_${WL}.err
echo stats pass done
find gpurun_out/pmc_${WL}_f gpurun_out/pmc_${WL}_w gpurun_out/kstats_${WL} -name "*.csv" | head -3
if [ "$2" == "mfma" ]; then
  rm -rf gpurun_out/pmc_${WL}_mfma
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${WL}_mfma -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > gpurun_out/pmc_${WL}_mfma.json 2> gpurun_out/pmc_${WL}_mfma.err
  echo mfma pass done
  find gpurun_out/pmc_${WL}_mfma -name "*.csv"
fi
