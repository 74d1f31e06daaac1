set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_fit.py -m gpu -x -q 2>&1 | tail -5
FIT_BENCH_PROFILE=0 python tools/fit_bench.py C5 300 > $O/fit_bench_fused.json 2>/dev/null; python -c "import json;d=json.load(open('$O/fit_bench_fused.json'));print('fusedupd',d['seconds_total'],d['ms_per_objective'],d['fused_objective_fallbacks'],d['table_passes_saved'],d['on_ray_hits'],d['refinement_settle_pass_histogram'],d['err_end'],d['param_err_end'],d['objective_evals'],d['gradient_evals'])"
FIT_BENCH_PROFILE=0 DMK_EIGH_FUSED_UPDATE=0 python tools/fit_bench.py C5 300 > $O/fit_bench_nofusedupd.json 2>/dev/null; python -c "import json;d=json.load(open('$O/fit_bench_nofusedupd.json'));print('separate',d['seconds_total'],d['ms_per_objective'],d['fused_objective_fallbacks'],d['refinement_settle_pass_histogram'],d['err_end'])"
cd /tmp && export TMPDIR=/tmp
FIT_BENCH_PROFILE=0 rocprofv3 --kernel-trace --stats -d $O/fit_trace -- python3 $R/tools/fit_bench.py C5 300 > /dev/null 2> $O/trace.err
cd $R
T=$(ls $O/fit_trace/*/*.db | head -1)
python3 tools/rocprof_summary.py $T > $O/fit_fused_kernel_trace_summary.txt
python3 tools/rocprof_seq.py $T --tail 300 > $O/fit_fused_kernel_seq.txt
rm -rf $O/fit_trace
head -18 $O/fit_fused_kernel_trace_summary.txt | cut -c1-150
