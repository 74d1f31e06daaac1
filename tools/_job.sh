cd $GRAFT_REPO_ROOT
bash tools/prof_passes.sh r04p5 > gpurun_out/r04p5.log 2>&1
tail -3 gpurun_out/r04p5.log
O=gpurun_out/r04q5; mkdir -p $O
cp gpurun_out/r04p5/traffic_latest.json profiles/traffic_latest.json
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --workload C4 --steps 10 --warmup 2 > $O/bench_C4.json 2> $O/bench_C4.err
python3 bench.py --workload C3 --steps 10 --warmup 2 > $O/bench_C3.json 2> $O/bench_C3.err
python3 bench.py --workload C2 --steps 50 --warmup 5 > $O/bench_C2.json 2> $O/bench_C2.err
python3 bench.py --workload C1 --steps 50 --warmup 5 > $O/bench_C1.json 2> $O/bench_C1.err
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
