set -x
mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_fit.py -m gpu -x -q -k c5_shape 2>&1 | grep -v "^\[refine\]" | tail -30
python tools/fit_bench.py C5 300 > gpurun_out/r05c/fit_bench_fused.json 2> gpurun_out/r05c/fit_bench_fused.err; tail -c 1800 gpurun_out/r05c/fit_bench_fused.json
DMK_FIT_FUSED=0 python tools/fit_bench.py C5 300 > gpurun_out/r05c/fit_bench_chain.json 2>/dev/null; tail -c 900 gpurun_out/r05c/fit_bench_chain.json
DMK_FIT_POLL=0 python tools/fit_bench.py C5 300 > gpurun_out/r05c/fit_bench_fused_nopoll.json 2>/dev/null; tail -c 900 gpurun_out/r05c/fit_bench_fused_nopoll.json
