cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout 300 tools/zhot_lab > $O/zhot_lab.txt 2>&1
tail -12 $O/zhot_lab.txt
