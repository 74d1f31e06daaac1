"""Host-side profile (cProfile) of FitVcorEmb at a BASELINE size: python tools/fit_profile.py [workload=C5] [MaxIter=3]"""
import sys, os, cProfile, pstats, io
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
pipeline.vcor_fit_stage(ctx, sysm, d_basis, nemb, ham["rdm1_emb"], MaxIter=1)      # warm up
pr = cProfile.Profile()
pr.enable()
out = pipeline.vcor_fit_stage(ctx, sysm, d_basis, nemb, ham["rdm1_emb"], MaxIter=mi)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
print({k: v for k, v in out.items() if k != "vcor"})
