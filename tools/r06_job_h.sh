cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_h
bash tools/stress_all.sh 906 r06_h > gpurun_out/r06_h/stress_all.log 2>&1
STRESS_SEED=907 STRESS_TRIALS=120 timeout 1500 python tools/hot_stress.py 2>&1 | grep -v WARNING > gpurun_out/r06_h/hot_stress_120.txt
DMK_BENCH_OVERRIDE='{"nlo":203,"naux":811,"nval":47}' timeout 900 python bench.py --scaling weak --kl-per-gpu 14 --steps 2 --warmup 1 --no-full-config --fit-iters 0 --no-cpu-baseline > gpurun_out/r06_h/bench_C5_off_tile_nao203_naux811_nemb250.json 2> gpurun_out/r06_h/bench_C5_off_tile.err
tail -3 gpurun_out/r06_h/stress_campaigns.txt; tail -2 gpurun_out/r06_h/hot_stress_120.txt; tail -c 1500 gpurun_out/r06_h/bench_C5_off_tile_nao203_naux811_nemb250.json
