cd $GRAFT_REPO_ROOT && DMK_SMALL_TIMING=1 python tools/phase_probe.py 2>&1 | tail -4
python -m pytest tests/test_gpu_small.py -m gpu -x -q 2>&1 | tail -5
