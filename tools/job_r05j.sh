set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_small.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
for W in C2 C1; do python bench.py --workload $W --steps 300 --warmup 30 > $O/bench_$W.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$W.json'));print('$W',d['ms_per_step'],d.get('cpu_baseline',{}).get('value'),d.get('parity_stages_ok'),d['roofline']['stages'])"; done
python bench.py --workload C4 --steps 10 --warmup 2 > $O/bench_C4.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_C4.json'));print('C4',d['value'],d['ms_per_step'],d['value_executed_frac_of_peak'],d['parity_ok'],{k:(v['ms_total'],v['launches']) for k,v in d['roofline']['families'].items()})"
DMK_ERI_GEN_BATCH=0 python bench.py --workload C4 --steps 10 --warmup 2 --no-parity --no-cpu-baseline > $O/bench_C4_nobatch.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_C4_nobatch.json'));print('C4 nobatch',d['value'],d['ms_per_step'],d['value_executed_frac_of_peak'],{k:(v['ms_total'],v['launches']) for k,v in d['roofline']['families'].items() if k=='philox'})"
