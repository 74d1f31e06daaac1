set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for W in C2 C1; do
(cd $R && python bench.py --workload $W --steps 200 --warmup 20 > $O/plain_$W.json 2>/dev/null; python -c "import json;d=json.load(open('$O/plain_$W.json'));print('$W',d['ms_per_step'],d.get('cpu_baseline',{}).get('value'),d.get('parity_stages_ok'))")
rocprofv3 --kernel-trace --stats -d $O/tr_$W -- python3 $R/bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_$W.json 2> $O/bench_$W.err
T=$(ls $O/tr_$W/*/*.db | head -1)
python3 $R/tools/rocprof_summary.py $T | head -12 | cut -c1-150
python3 $R/tools/rocprof_seq.py $T --tail 12
rm -rf $O/tr_$W
done
cd $R
python -m pytest tests/test_gpu_small.py -m gpu -x -q 2>&1 | tail -5
