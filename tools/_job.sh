cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O; rm -f $O/iter_stress.log
for sd in 1 2 3 4 5 6 7 8 9 10; do STRESS_SEED=$sd STRESS_TRIALS=60 STRESS_NLO_HALF=25 STRESS_NAUX=72 STRESS_ORACLE_MFLOP=4e5 timeout 900 python3 tools/iteration_stress.py >> $O/iter_stress.log 2>&1; echo "seed $sd rc $?" >> $O/iter_stress.log; done
grep -v "WARNING\|^larger\|degenerate" $O/iter_stress.log | cut -c1-900 | tail -30
