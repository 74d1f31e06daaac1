cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x; mkdir -p $O
( time timeout 1800 python3 -m pytest tests -m gpu -x -q ) > $O/pytest4.log 2>&1
tail -5 $O/pytest4.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py > $O/bench_default2.json 2> $O/bench_default2.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04x/bench_default2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['steps'], d['roofline']['frac'], d['roofline']['traffic'], d['parity_ok'], d['parity_stages_ok'], d['vcor_fit']['seconds_total'])
PY
