import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from libdmet_preview_amd import _lib, pipeline
ctx = _lib.get_ctx()
for wl in ("C1", "C2"):
    sysm = pipeline.SyntheticSystem.from_workload(ctx, wl)
    for i in range(6):
        out = pipeline.iteration(ctx, sysm, emb_ham=False)
    print(wl, "phase_us (eig, occ, density, fold)", out["small_phase_us"], "sweeps", out["jacobi_sweeps"])
