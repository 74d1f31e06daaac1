set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_fit.py tests/test_gpu_occ.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -5
python -m pytest tests/test_gpu_dist.py -m gpu -x -q -k "fit" 2>&1 | tail -5
FIT_BENCH_PROFILE=0 python tools/fit_bench.py C5 300 > $O/fit_bench_fused.json 2>/dev/null; python -c "import json;d=json.load(open('$O/fit_bench_fused.json'));print('fused',d['seconds_total'],d['ms_per_objective'],d['fused_objective_fallbacks'],d['table_passes_saved'],d['refinement_settle_pass_histogram'],d['err_end'],d['param_err_end'],d['objective_evals'],d['gradient_evals'])"
