"""Randomised differential campaign of the correlation-potential fit: errfunc / gradfunc of routine.slater.EmbFitDevice (dV_dparam
table, streaming contraction, eigh(nemb), occupations, analytic gradient; reference routine/slater.py:1040-1154, ftsystem.py:151-213)
against oracle/restate_fit.py on random embedding problems -- random meshes, 3 .. 40 orbitals per cell, random valence counts,
T = 0 and T > 0, impurity-only / impurity + diagonal index sets, remove_diag_grad.
A second loop does the same for the LATTICE fit (FullFitDevice against FullFit: objective at T = 0 and T > 0, analytic finite-T gradient).
    STRESS_SEED=1 STRESS_TRIALS=40 python tools/fit_stress.py          (test infrastructure: imports the oracle)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from oracle import restate_fit as F
from libdmet_preview_amd import _lib, pipeline
from libdmet_preview_amd.routine import slater
from libdmet_preview_amd.dmet import Hubbard
from libdmet_preview_amd.system.lattice import Lattice

ctx = _lib.get_ctx()
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "40"))
worst = {"dV": 0.0, "err": 0.0, "grad": 0.0}
t0, done, skipped = time.time(), 0, 0
for trial in range(trials):
    mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.5, 0.3, 0.1, 0.1]))
    nk = mesh[0] * mesh[1] * mesh[2]
    if nk < 2:
        mesh, nk = (2, 1, 1), 2
    nlo = int(rng.integers(3, 41))
    if nk * nlo > 1200:
        nlo = max(3, 1200 // nk)
    nval = int(rng.integers(1, nlo + 1))
    spin = 2
    beta = np.inf if rng.random() < 0.6 else float(rng.uniform(5.0, 40.0))
    sysm = pipeline.SyntheticSystem(ctx, mesh, nlo, 0, nval, spin, seed=int(rng.integers(1, 1 << 30)), name="fit_stress")
    d_rhoR, mf = pipeline.mean_field_stage(ctx, sysm)
    d_basis, nemb, sig = pipeline.bath_stage(ctx, sysm, d_rhoR)
    basis = d_basis.get().reshape(spin, nk, nlo, nemb)
    Fk = sysm.d_Fock_k.get().reshape(spin, nk, nlo, nlo)
    Sk = np.asarray([np.eye(nlo)] * nk)
    L = Lattice(nlo, mesh)
    L.val_idx, L.virt_idx, L.core_idx = list(range(nval)), list(range(nval, nlo)), []
    L.fock_lo_k = L.hcore_lo_k = Fk
    v = Hubbard.VcorLocal(False, False, nlo, idx_range=list(range(nval)))
    ov = F.VcorLocal(False, False, nlo, idx_range=list(range(nval)))
    bk = R.R2k(basis, mesh)
    rho_k = R.R2k(d_rhoR.get().reshape(spin, nk, nlo, nlo), mesh)
    rdm1_emb = np.asarray([np.einsum("kpa,kpq,kqb->ab", bk[s].conj(), rho_k[s], bk[s]).real / nk for s in range(spin)])
    ne = [int(round(np.trace(rdm1_emb[s]))) for s in range(spin)]
    if min(ne) < 1 or max(ne) >= nemb:
        skipped += 1
        continue
    noise = 0.02 * rng.standard_normal(rdm1_emb.shape)
    target = rdm1_emb + 0.5 * (noise + noise.transpose(0, 2, 1))
    mode = int(rng.integers(0, 3))
    if mode == 0:
        imp_idx, det_idx = list(range(nemb)), []
    elif mode == 1:
        imp_idx, det_idx = list(range(nlo)), []                       # impurity block only
    else:
        imp_idx, det_idx = list(range(nlo)), list(range(nlo, nemb))  # impurity block + bath diagonal
    rdg = bool(rng.random() < 0.3)
    nel = ne                                                # integers, as lattice.ncore + lattice.nval (slater.py:957-962)
    fit = slater.EmbFitDevice(ctx, target, L, basis, v, beta, nel, imp_idx, det_idx, Fk, Sk, remove_diag_grad=rdg)
    ref = F.EmbFit(target, mesh, basis, ov, beta, Fk, Sk, nel, imp_idx=imp_idx, det_idx=det_idx, remove_diag_grad=rdg)
    e_dV = float(np.abs(fit.d_dV.get().reshape(ref.dV.shape) - ref.dV).max())
    assert e_dV < 1e-12, (trial, mesh, nlo, nval, e_dV)
    worst["dV"] = max(worst["dV"], e_dV)
    for scale in (0.0, 0.05):
        p = scale * rng.standard_normal(v.length())
        # a T = 0 objective is only defined where the frontier of embH1 + V is not degenerate (tol_deg 1e-3 in the gradient)
        e, e_ref = fit.errfunc(p), ref.errfunc(p)
        assert abs(e - e_ref) < 1e-9 * max(1.0, abs(e_ref)), (trial, mesh, nlo, nval, beta, mode, scale, e, e_ref)
        g, g_ref = fit.gradfunc(p), (ref.gradfunc(p) if beta == np.inf else ref.gradfunc_ft(p))     # slater.py:1126-1141 / ftsystem.py:151-213
        e_g = float(np.abs(g - g_ref).max()) / max(1.0, float(np.abs(g_ref).max()))
        assert e_g < 1e-7, (trial, mesh, nlo, nval, beta, mode, rdg, scale, e_g)
        worst["err"], worst["grad"] = max(worst["err"], abs(e - e_ref)), max(worst["grad"], e_g)
    done += 1
# ---- the LATTICE fit (FitVcorFull, slater.py:1352-1682): the whole lattice re-diagonalised per parameter vector ---------------------
from libdmet_preview_amd import synth
from libdmet_preview_amd.routine.mfd import check_nelec
worst_full = {"err": 0.0, "grad": 0.0}
done_full = 0
for trial in range(max(1, trials // 2)):
    mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.5, 0.3, 0.1, 0.1]))
    nk = mesh[0] * mesh[1] * mesh[2]
    if nk < 2:
        mesh, nk = (2, 1, 1), 2
    nlo = 2 * int(rng.integers(1, 13))
    nval = int(rng.integers(1, nlo + 1))
    spin, filling = 2, 0.5
    beta = float(rng.uniform(5.0, 40.0)) if rng.random() < 0.7 else np.inf
    sysm = pipeline.SyntheticSystem(ctx, mesh, nlo, 0, nval, spin, seed=int(rng.integers(1, 1 << 30)), name="fullfit_stress")
    d_rhoR, mf = pipeline.mean_field_stage(ctx, sysm)
    d_basis, nemb, sig = pipeline.bath_stage(ctx, sysm, d_rhoR)
    basis = d_basis.get().reshape(spin, nk, nlo, nemb)
    Fk = sysm.d_Fock_k.get().reshape(spin, nk, nlo, nlo)
    L = Lattice(nlo, mesh)
    L.set_Ham_lo(fock_lo_R=sysm.Fock_R)
    L.val_idx, L.virt_idx, L.core_idx = list(range(nval)), list(range(nval, nlo)), []
    v = Hubbard.VcorLocal(False, False, nlo)
    ov = F.VcorLocal(False, False, nlo)
    mode = int(rng.integers(0, 3))
    if mode == 0 and True:
        imp_idx, det_idx, ibf = list(range(nemb)), [], True                    # imp + bath fit: objective only (slater.py:1510-1512)
        rho_k = R.R2k(d_rhoR.get().reshape(spin, nk, nlo, nlo), mesh)
        bk = R.R2k(basis, mesh)
        tgt = np.asarray([np.einsum("kpa,kpq,kqb->ab", bk[s].conj(), rho_k[s], bk[s]).real / nk for s in range(spin)])
    else:
        ibf = False
        imp_idx, det_idx = (list(range(nlo)), []) if mode == 1 else ([], list(range(nlo)))
        tgt = d_rhoR.get().reshape(spin, nk, nlo, nlo)[:, 0].copy()
    noise = 0.02 * rng.standard_normal(tgt.shape)
    target = tgt + 0.5 * (noise + noise.transpose(0, 2, 1))
    nelec = check_nelec(spin * nk * nlo * filling)[0]
    fit = slater.FullFitDevice(ctx, target, L, basis, v, beta, nelec, imp_idx, det_idx, ibf)
    ref = F.FullFit(target, mesh, basis, ov, beta, Fk, filling, imp_idx=None if ibf else imp_idx, det_idx=None if ibf else det_idx)
    for scale in (0.0, 0.05):
        p = scale * rng.standard_normal(v.length())
        if beta == np.inf:
            # T = 0: only defined where the lattice frontier is not degenerate under this potential
            ov.update(p)
            vm = ov.get(0, True)
            lv = np.sort(np.concatenate([np.linalg.eigvalsh(Fk[s, k] + vm[s]) for s in range(spin) for k in range(nk)]))
            if lv[nelec] - lv[nelec - 1] < 1e-6:
                continue
        e, e_ref = fit.errfunc(p), ref.errfunc(p)
        assert abs(e - e_ref) < 1e-9 * max(1.0, abs(e_ref)), ("full", trial, mesh, nlo, nval, beta, mode, scale, e, e_ref)
        worst_full["err"] = max(worst_full["err"], abs(e - e_ref))
        if beta != np.inf and not ibf:
            g, g_ref = fit.gradfunc(p), ref.gradfunc_ft(p)
            e_g = float(np.abs(g - g_ref).max()) / max(1.0, float(np.abs(g_ref).max()))
            assert e_g < 1e-7, ("full", trial, mesh, nlo, nval, beta, mode, scale, e_g)
            worst_full["grad"] = max(worst_full["grad"], e_g)
    done_full += 1
print("lattice-fit stress ok: %d lattice problems, worst |derr| %.1e, |dgrad| / max(1, |g|) %.1e" % (done_full, worst_full["err"], worst_full["grad"]))
print("fit stress ok: %d embedding problems (%d skipped: empty or full channel) in %.0f s, worst |ddV| %.1e, |derr| %.1e, |dgrad| / max(1, |g|) %.1e"
      % (done, skipped, time.time() - t0, worst["dV"], worst["err"], worst["grad"]))
