set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
mkdir -p $O
export DMK_EIGH_REFINE_DEBUG_X=0
rocprofv3 --kernel-trace --stats -d $O/fit_trace -- python3 $R/tools/fit_profile.py C5 300 > $O/fit_profile.txt 2> $O/fit_profile.err
cd $R
T=$(ls $O/fit_trace/*/*.db | head -1)
python3 tools/rocprof_summary.py $T > $O/fit_kernel_trace_summary.txt
python3 tools/rocprof_seq.py $T --tail 600 > $O/fit_kernel_seq.txt
rm -rf $O/fit_trace
tail -5 $O/fit_profile.txt | cut -c1-1500
head -60 $O/fit_kernel_trace_summary.txt
