cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
./tools/zhot_lab > $O/zhot_lab.txt 2>&1
( time timeout 900 python3 -m pytest tests/test_gpu_fold.py tests/test_gpu_bcs.py tests/test_gpu_chain.py -m gpu -x -q ) > $O/pytest.log 2>&1
tail -8 $O/pytest.log
