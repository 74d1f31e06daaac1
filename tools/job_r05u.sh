cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05u
for S in 601 602 603; do STRESS_SEED=$S STRESS_TRIALS=150 timeout 900 python tools/small_stress.py 2>&1 | grep -v WARNING | tail -4; done | tee gpurun_out/r05u/small_stress.txt
