cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_dist.py -q -m gpu -x -k "ranks_on_one_gpu" 2>&1 | tail -5
