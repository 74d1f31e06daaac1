cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
for sd in 11 12 13; do STRESS_SEED=$sd STRESS_TRIALS=150 STRESS_NMAX=300 STRESS_BMAX=40 timeout 600 python3 tools/eigh_stress.py >> $O/stress.log 2>&1; done
tail -4 $O/stress.log
python3 bench.py --parity-seed 101 > $O/bench_seed101.json 2> $O/bench_seed101.err
python3 bench.py --workload C4 --steps 10 --warmup 2 --parity-seed 202 > $O/bench_C4_seed202.json 2> $O/bench_C4_seed202.err
python3 - <<'PY'
import json
for n in ('bench_seed101','bench_C4_seed202'):
    try:
        d=json.loads(open('gpurun_out/r04t/%s.json'%n).read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['parity'])
    except Exception as e: print(n, 'failed', e)
PY
