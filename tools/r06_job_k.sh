cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_k
timeout 1900 python bench.py --steps 20 --warmup 5 --max-total-s 1800 > gpurun_out/r06_k/bench_default_20_plus_5_steps_honoured.json 2> gpurun_out/r06_k/bench_default_20_plus_5.err
STRESS_SEED=911 STRESS_TRIALS=200 timeout 900 python tools/iteration_stress.py 2>&1 | grep -v WARNING | tail -2 > gpurun_out/r06_k/iteration_stress_200.txt
STRESS_SEED=912 STRESS_TRIALS=300 timeout 1200 python tools/eri_stress.py 2>&1 | grep -v WARNING | tail -2 > gpurun_out/r06_k/eri_stress_300.txt
STRESS_SEED=913 STRESS_TRIALS=150 STRESS_BIG=1 timeout 1500 python tools/hot_stress.py 2>&1 | grep -v WARNING | tail -3 > gpurun_out/r06_k/hot_stress_big_150.txt
STRESS_SEED=914 STRESS_TRIALS=300 timeout 900 python tools/fit_stress.py 2>&1 | grep -v WARNING | tail -3 > gpurun_out/r06_k/fit_stress_300.txt
tail -c 400 gpurun_out/r06_k/bench_default_20_plus_5_steps_honoured.json; cat gpurun_out/r06_k/*_stress_*.txt
