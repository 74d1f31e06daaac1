cd $GRAFT_REPO_ROOT
bash tools/prof_passes.sh r04p4 > gpurun_out/r04p4.log 2>&1
tail -3 gpurun_out/r04p4.log
O=gpurun_out/r04q4; mkdir -p $O
cp gpurun_out/r04p4/traffic_latest.json profiles/traffic_latest.json
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --workload C4 --steps 10 --warmup 2 > $O/bench_C4.json 2> $O/bench_C4.err
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
