set +e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_spp_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_spp_mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-api --no-sustained > gpurun_out/pmc_spp_mfma.json 2> gpurun_out/pmc_spp_mfma.err
echo done
find gpurun_out/pmc_spp_mfma -name "*.csv"
