set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05o
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_gso.py -m gpu -x -q 2>&1 | tail -15
( time timeout 1700 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
tail -c 400 $O/bench_default.json
