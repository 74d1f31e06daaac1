cd $GRAFT_REPO_ROOT
O=gpurun_out/r04z; mkdir -p $O; rm -f $O/big_stress.log
STRESS_BIG=1 STRESS_SEED=77 STRESS_TRIALS=40 timeout 500 python3 tools/hot_stress.py >> $O/big_stress.log 2>&1; echo "hot big rc $?" >> $O/big_stress.log
STRESS_SEED=21 STRESS_TRIALS=60 STRESS_NMAX=700 STRESS_BMAX=6 timeout 400 python3 tools/eigh_stress.py >> $O/big_stress.log 2>&1; echo "eigh big rc $?" >> $O/big_stress.log
grep "stress ok\|rc \|Error\|assert" $O/big_stress.log | cut -c1-300 | tail -8
grep "^   mesh" $O/big_stress.log | sort -t'b' -k3 | tail -3
