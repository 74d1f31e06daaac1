cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O; rm -f $O/gso_stress.log
for sd in 1 2 3 4 5 6 7 8 9 10; do STRESS_SEED=$sd STRESS_TRIALS=100 timeout 900 python3 tools/gso_stress.py >> $O/gso_stress.log 2>&1; echo "seed $sd rc $?" >> $O/gso_stress.log; done
grep "stress ok\|^seed\|Error" $O/gso_stress.log | cut -c1-300 | tail -22
