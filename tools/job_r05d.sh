set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
FIT_BENCH_PROFILE=0 python tools/fit_bench.py C5 300 > $O/fit_bench_fused.json 2>/dev/null; python -c "import json;d=json.load(open('$O/fit_bench_fused.json'));print('fused',d['seconds_total'],d['ms_per_objective'],d['fused_objective_fallbacks'])"
FIT_BENCH_PROFILE=0 DMK_FIT_FUSED=0 python tools/fit_bench.py C5 300 > $O/fit_bench_chain.json 2>/dev/null; python -c "import json;d=json.load(open('$O/fit_bench_chain.json'));print('chain',d['seconds_total'],d['ms_per_objective'])"
cd /tmp && export TMPDIR=/tmp
FIT_BENCH_PROFILE=0 rocprofv3 --kernel-trace --stats -d $O/fit_trace -- python3 $R/tools/fit_bench.py C5 300 > /dev/null 2> $O/trace.err
cd $R
T=$(ls $O/fit_trace/*/*.db | head -1)
python3 tools/rocprof_summary.py $T > $O/fit_fused_kernel_trace_summary.txt
python3 tools/rocprof_seq.py $T --tail 400 > $O/fit_fused_kernel_seq.txt
rm -rf $O/fit_trace
head -30 $O/fit_fused_kernel_trace_summary.txt | cut -c1-150
