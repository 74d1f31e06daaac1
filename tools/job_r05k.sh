set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_small.py tests/test_gpu_occ.py -m gpu -x -q 2>&1 | tail -4
python - <<'PY'
import numpy as np, time
from libdmet_preview_amd import _lib, pipeline
ctx=_lib.get_ctx()
for w in ("C1","C2"):
    s=pipeline.SyntheticSystem.from_workload(ctx,w)
    o=pipeline.iteration(ctx,s,emb_ham=False); print(w,"sweeps",o["jacobi_sweeps"],"phase us",o["small_phase_us"])
# occupations at C5 size
from libdmet_preview_amd.routine import mfd
rng=np.random.default_rng(0); e=rng.standard_normal(86400)
d=ctx.to_device(e)
for rep in range(3):
    ctx.sync(); t=time.perf_counter(); occ,mu,nerr=mfd.assignocc_dev(ctx,d,43200,np.inf); ctx.sync(); print("occ C5 size ms", 1e3*(time.perf_counter()-t), mu, float(0.5*(np.sort(e)[43199]+np.sort(e)[43200])))
PY
for W in C2 C1; do python bench.py --workload $W --steps 300 --warmup 30 > $O/bench_$W.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_$W.json'));print('$W',d['ms_per_step'],d.get('cpu_baseline',{}).get('value'),d.get('parity_stages_ok'))"; done
