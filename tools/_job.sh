cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O; rm -f $O/fit_stress2.log
for sd in 1 2 3 4 5 6 7 8 9 10; do STRESS_SEED=$sd STRESS_TRIALS=80 timeout 900 python3 tools/fit_stress.py >> $O/fit_stress2.log 2>&1; echo "seed $sd rc $?" >> $O/fit_stress2.log; done
grep "stress ok\|^seed\|Error" $O/fit_stress2.log | cut -c1-300 | tail -32
