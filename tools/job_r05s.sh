cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05s && timeout 120 tools/fold_small_lab | tee gpurun_out/r05s/fold_small_lab.txt
