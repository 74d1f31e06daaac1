set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
for S in 3 4 5 6 8; do timeout 300 tools/ozaki_lab 8192 1600 $S; done 2>&1 | tee $O/ozaki_lab.txt
timeout 300 tools/ozaki_lab 16384 1600 4 2>&1 | tee -a $O/ozaki_lab.txt
