cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_bcs.py tests/test_gpu_gso.py -q -m gpu -k "bath or bcs or gso or emb_basis" 2>&1 | tail -6
STRESS_SEED=3 STRESS_TRIALS=100 python3 tools/meanfield_stress.py 2>&1 | grep "stress ok"
STRESS_SEED=3 STRESS_TRIALS=100 python3 tools/bcs_stress.py 2>&1 | grep "stress ok"
