cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_production.py -q -m gpu -k "randomised" 2>&1 | tail -5
