set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_small.py -m gpu -x -q 2>&1 | tail -30
python -m pytest tests/test_gpu_fit.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
FIT_BENCH_PROFILE=0 python tools/fit_bench.py C5 300 > $O/fit_bench.json 2>/dev/null; python -c "import json;d=json.load(open('$O/fit_bench.json'));print('fit',d['seconds_total'],d['ms_per_objective'],d['fused_objective_fallbacks'],d['table_passes_saved'],d['on_ray_hits'],d['err_end'],d['param_err_end'],d['objective_evals'],d['gradient_evals'])"
for W in C1 C2; do python bench.py --workload $W --steps 200 --warmup 20 > $O/bench_$W.json 2>$O/bench_$W.err; python -c "import json;d=json.load(open('$O/bench_$W.json'));print('$W',d['value'],d['ms_per_step'],d.get('cpu_baseline',{}).get('value'),d['stage_seconds_per_step'],d.get('parity_stages_ok'))"; done
python -m pytest tests/test_gpu_bcs.py tests/test_gpu_gso.py -m gpu -x -q 2>&1 | tail -8
