cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh pmc_r02b_c3 c3 8 8 > gpurun_out/pmc_r02b_c3.log 2>&1
bash tools/pmc_passes.sh pmc_r02b_big big 8 8 > gpurun_out/pmc_r02b_big.log 2>&1
tail -3 gpurun_out/pmc_r02b_c3.log gpurun_out/pmc_r02b_big.log
