cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
B="python3 bench.py --workload C4 --steps 3 --warmup 1 --fit-iters 0 --no-cpu-baseline --no-shard-pass"
$B > $O/c4_new.json 2> $O/c4_new.err
DMK_ERI_TAB_SUB=1 $B > $O/c4_sub1.json 2>> $O/c4_new.err
DMK_ERI_TAB_SUB=2 $B > $O/c4_sub2.json 2>> $O/c4_new.err
DMK_ERI_H1_BN=64 $B > $O/c4_bn64.json 2>> $O/c4_new.err
DMK_ERI_GROUP=8 $B > $O/c4_g8.json 2>> $O/c4_new.err
python3 bench.py --scaling weak --kl-per-gpu 2 --no-full-config --steps 2 --warmup 1 --fit-iters 0 --no-cpu-baseline --parity-budget-s 60 > $O/c5_quick.json 2> $O/c5_quick.err
tail -c 300 $O/c5_quick.err
