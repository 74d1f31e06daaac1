cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O; rm -f $O/hot_stress.log
for sd in 1 2 3 4 5 6 7 8; do STRESS_SEED=$sd STRESS_TRIALS=60 timeout 1200 python3 tools/hot_stress.py >> $O/hot_stress.log 2>&1; echo "seed $sd rc $?" >> $O/hot_stress.log; done
grep "stress ok\|^seed\|Error\|assert" $O/hot_stress.log | cut -c1-400 | tail -24
