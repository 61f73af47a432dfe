#!/bin/bash
# Round-4 evidence at HEAD on one box: kernel statistics, PMC traffic (two separate passes: FETCH_SIZE takes 3 of the 4 TCC slots) and the
# MFMA-busy pass of the headline command, per-layer tables.   bash tools/r4_profiles.sh   -> gpurun_out/r4p_*   (program directly after --)
cd $GRAFT_REPO_ROOT
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for WL in spp tiny mobile; do
  rm -rf $O/r4p_pmc_${WL}_f $O/r4p_pmc_${WL}_w $O/r4p_kstats_${WL}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r4p_pmc_${WL}_f -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/r4p_pmc_${WL}_f.json 2> $O/r4p_pmc_${WL}_f.err; echo "$WL fetch pass rc $?"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r4p_pmc_${WL}_w -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/r4p_pmc_${WL}_w.json 2> $O/r4p_pmc_${WL}_w.err; echo "$WL write pass rc $?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4p_kstats_${WL} -- python3 bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-api --no-sustained > $O/r4p_kstats_${WL}_bench.json 2> $O/r4p_kstats_${WL}.err; echo "$WL stats pass rc $?"
done
rm -rf $O/r4p_pmc_spp_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r4p_pmc_spp_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > $O/r4p_pmc_spp_mfma.json 2> $O/r4p_pmc_spp_mfma.err; echo "mfma pass rc $?"
python tools/layer_profile.py --workload spp --bs 32 --compact > $O/r4p_layers_spp.txt 2>&1
python tools/layer_profile.py --workload tiny --compact > $O/r4p_layers_tiny.txt 2>&1
python tools/layer_profile.py --workload mobile --compact > $O/r4p_layers_mobile.txt 2>&1
grep total $O/r4p_layers_spp.txt $O/r4p_layers_tiny.txt $O/r4p_layers_mobile.txt
grep -c . $O/r4p_pmc_spp_f/*/*counter_collection.csv
