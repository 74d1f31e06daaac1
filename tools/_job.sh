cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
DMK_EIGH_DEBUG=1 python3 tools/eigh_bench.py > $O/eigh_bench_dbg.log 2>&1
python3 tools/eigh_bench.py > $O/eigh_bench.log 2>&1
grep -A1 "n=200" $O/eigh_bench_dbg.log | tail -4; cat $O/eigh_bench.log
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fit.py tests/test_gpu_bcs.py -m gpu -x -q -k "eigh or Diag or HF or fit or bath or bcs" ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python3 tools/eigh_stress.py > $O/eigh_stress.log 2>&1; tail -5 $O/eigh_stress.log
