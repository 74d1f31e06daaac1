cd $GRAFT_REPO_ROOT
bash tools/prof_passes.sh r04p > gpurun_out/r04p.log 2>&1
tail -20 gpurun_out/r04p.log
cat gpurun_out/r04p/C4_pmc_wave_states.txt gpurun_out/r04p/C5_pmc_wave_states.txt
