cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05v
python -m pytest tests/test_gpu_df_object.py -m gpu -x -q 2>&1 | tail -8
for M in auto regenerate; do timeout 900 python bench.py --workload C4 --df $M --steps 10 --warmup 2 > gpurun_out/r05v/bench_C4_$M.json 2> gpurun_out/r05v/bench_C4_$M.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r05v/bench_C4_$M.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$M', d['value'], d['ms_per_step'], r['frac'], {k:(round(v['ms_total']/d['steps'],2)) for k,v in r['families'].items()}, d['parity_ok'], d['parity_maxabs'], d['input'][:60])"; tail -2 gpurun_out/r05v/bench_C4_$M.err; done
