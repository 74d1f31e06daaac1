cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r05t
for WL in C1 C2; do python bench.py --workload $WL --no-cpu-baseline > gpurun_out/r05t/bench_$WL.json 2> gpurun_out/r05t/bench_$WL.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r05t/bench_$WL.json').read().strip().splitlines()[-1]); print('$WL', d['ms_per_step'], d['steps'], {k:v['ms_per_step'] for k,v in d['roofline']['stages'].items()}, d.get('parity_stages_ok'))"; done
