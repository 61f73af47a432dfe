#!/bin/bash
# Round-5 evidence at HEAD on one box: bench lines (headline with the CPU baseline, secondary workloads), kernel statistics, PMC traffic
# (two separate passes: FETCH_SIZE takes 3 of the 4 TCC slots) and the MFMA-busy pass of the headline command, per-layer tables.
#   bash tools/r5_profiles.sh [part]    part = bench | pmc | all   -> gpurun_out/r5p_*   (program directly after --)
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
O=gpurun_out
PART=${1:-all}
if [ "$PART" != "pmc" ]; then
  timeout -k 10 400 python bench.py > $O/r5p_bench_spp.json 2> $O/r5p_bench_spp.err; echo "bench spp rc $?"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r5p_bench_spp_driver.json 2> $O/r5p_bench_spp_driver.err; echo "bench spp (driver flags) rc $?"
  for WL in tiny mobile efficient; do
    timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline > $O/r5p_bench_$WL.json 2> $O/r5p_bench_$WL.err; echo "bench $WL rc $?"
  done
  python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r5p_layers_spp.txt 2>&1
  python tools/layer_profile.py --workload tiny --compact > $O/r5p_layers_tiny.txt 2>&1
  python tools/layer_profile.py --workload mobile --compact > $O/r5p_layers_mobile.txt 2>&1
  grep total $O/r5p_layers_spp.txt $O/r5p_layers_tiny.txt $O/r5p_layers_mobile.txt
  python - <<'PY'
import json
for f in ("spp", "spp_driver", "tiny", "mobile", "efficient"):
    try:
        j = json.loads(open(f"gpurun_out/r5p_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["config"].get("detect_api_images_per_s"), j["config"].get("detect_stream_api_images_per_s"))
    except Exception as e:
        print(f, "unreadable", e)
PY
fi
if [ "$PART" != "bench" ]; then
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for WL in spp tiny mobile; do
    rm -rf $O/r5p_pmc_${WL}_f $O/r5p_pmc_${WL}_w $O/r5p_kstats_${WL}
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r5p_pmc_${WL}_f -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/r5p_pmc_${WL}_f.json 2> $O/r5p_pmc_${WL}_f.err; echo "$WL fetch pass rc $?"
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r5p_pmc_${WL}_w -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/r5p_pmc_${WL}_w.json 2> $O/r5p_pmc_${WL}_w.err; echo "$WL write pass rc $?"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5p_kstats_${WL} -- python3 bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-api --no-sustained > $O/r5p_kstats_${WL}_bench.json 2> $O/r5p_kstats_${WL}.err; echo "$WL stats pass rc $?"
  done
  rm -rf $O/r5p_pmc_spp_mfma
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r5p_pmc_spp_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/r5p_pmc_spp_mfma.json 2> $O/r5p_pmc_spp_mfma.err; echo "mfma pass rc $?"
  find $O/r5p_kstats_spp -name "*kernel_stats.csv" | head -2
fi
