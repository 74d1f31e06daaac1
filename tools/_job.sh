cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout 300 tools/zhot_lab > $O/zhot_lab2.txt 2>&1
grep -B8 "nontemporal" $O/zhot_lab2.txt | head -12
