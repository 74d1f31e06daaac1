"""vcor-fit evaluation timings at a BASELINE size: python tools/fit_bench.py [workload=C5] [MaxIter=3]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdmet_preview_amd import _lib, pipeline

wl = sys.argv[1] if len(sys.argv) > 1 else "C5"
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _lib.get_ctx()
sysm = pipeline.SyntheticSystem.from_workload(ctx, wl)
timers = {}
d_rhoR, mf = pipeline.mean_field_stage(ctx, sysm, timers)
d_basis, nemb, sig = pipeline.bath_stage(ctx, sysm, d_rhoR, timers)
npair = nemb * (nemb + 1) // 2
sp = sysm.spin * (sysm.spin + 1) // 2
eri = ctx.zeros((sp, npair, npair), np.float64)
ham = pipeline.emb_ham_stage(ctx, sysm, d_basis, nemb, d_rhoR, eri, timers)
prof_on = os.environ.get("FIT_BENCH_PROFILE", "1") != "0"           # HIP events around every launch cost ~10 % of a converged fit
ctx.profile(prof_on)
out = pipeline.vcor_fit_stage(ctx, sysm, d_basis, nemb, ham["rdm1_emb"], MaxIter=mi)
prof = ctx.profile_read() if prof_on else {}
out.pop("vcor")
out["families_ms"] = {k: [round(v[0], 3), v[1]] for k, v in prof.items() if v[1]}
print(json.dumps(out))
