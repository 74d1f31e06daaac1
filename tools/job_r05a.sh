set -x
mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "baseline_partition or local_unless or rank_sharded" 2>&1 | tail -15
export DMK_BENCH_BACKEND=gloo DMK_BENCH_ONE_GPU=1
timeout 900 python bench.py --workload C4 --gpus 4 --steps 3 --warmup 1 --fit-iters 0 > gpurun_out/r05a/bench_C4_x4_one_gpu.json 2> gpurun_out/r05a/bench_C4_x4.err
echo rc=$?; tail -c 1500 gpurun_out/r05a/bench_C4_x4_one_gpu.json; tail -5 gpurun_out/r05a/bench_C4_x4.err
