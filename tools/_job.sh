cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
python3 tools/fit_profile.py C5 300 > $O/fit_cprofile.txt 2>&1
head -60 $O/fit_cprofile.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o fit -- python3 $GRAFT_REPO_ROOT/tools/fit_profile.py C5 300 > $GRAFT_REPO_ROOT/$O/fit_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
ls $O/prof | head; f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); head -25 $f
