cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05q && timeout 120 tools/clock_small_probe | tee gpurun_out/r05q/clock_small_probe.txt
