cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
B4="python3 bench.py --workload C4 --steps 5 --warmup 1 --fit-iters 0 --no-cpu-baseline --no-shard-pass"
$B4 > $O/c4_gen.json 2> $O/c4.err
DMK_ERI_GEN_STREAM=0 $B4 > $O/c4_nogen.json 2>> $O/c4.err
B5="python3 bench.py --scaling weak --kl-per-gpu 4 --no-full-config --steps 2 --warmup 1 --fit-iters 0 --no-cpu-baseline --parity-budget-s 100"
$B5 > $O/c5_gen.json 2> $O/c5.err
DMK_ERI_GEN_STREAM=0 $B5 > $O/c5_nogen.json 2>> $O/c5.err
