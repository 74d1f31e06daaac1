"""Randomised differential campaign of what round 6 added around the path for the BCS / GSO twins and the non-local potentials:
mfd.HFB, mfd.GHF, spinless.get_emb_Ham, bcs.embHam, the GSO / BCS embedding fits (objective + gradient), the lattice stage of the GSO
fit, the cell-resolved potential's dV/dparam and the k-resolved lattice fit -- the HIP path through the C ABI against the oracle
restatements (oracle/restate_bcs.py, restate_gso.py, restate_fit.py) on random lattices, potentials and parameters.
    STRESS_SEED=1 STRESS_TRIALS=30 python tools/twins_stress.py          (test infrastructure: imports the oracle)"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from oracle import restate_bcs as B
from oracle import restate_gso as G
from oracle import restate_fit as F
from libdmet_preview_amd import synth
from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.routine import mfd, spinless, slater, vcor as pvcor
from libdmet_preview_amd.dmet import Hubbard
from libdmet_preview_amd.system.lattice import Lattice

log.verbose = "RESULT"
seed, trials = int(os.environ.get("STRESS_SEED", "1")), int(os.environ.get("STRESS_TRIALS", "30"))
rng = np.random.default_rng(seed)
worst = {}


def note(key, dev, tol, ctx):
    worst[key] = max(worst.get(key, 0.0), float(dev))
    assert dev < tol, (key, dev, ctx)


class V3(object):
    def __init__(self, v):
        self.v = v

    def islocal(self):
        return True

    is_local = islocal

    def get(self, i=0, kspace=True):
        return self.v if (kspace or i == 0) else np.zeros_like(self.v)


t0 = time.time()
for trial in range(trials):
    mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.4, 0.3, 0.2, 0.1]))
    nk = mesh[0] * mesh[1] * mesh[2]
    if nk < 2:
        mesh, nk = (3, 1, 1), 3
    n = int(rng.integers(2, 7))
    ctxinfo = (trial, mesh, n)
    sd = lambda: int(rng.integers(1, 1 << 30))
    FR = synth.make_fock_R(mesh, n, spin=2, seed=sd())
    D_R = 0.3 * synth.make_fock_R(mesh, n, spin=1, seed=sd())[0]
    v = 0.2 * rng.standard_normal((3, n, n))
    v[0], v[1] = v[0] + v[0].T, v[1] + v[1].T
    mu = float(rng.uniform(-0.3, 0.3))
    beta = float(rng.choice([np.inf, 6.0, 15.0]))
    lo, hi = sorted(int(x) for x in rng.integers(0, n, size=2))
    val = list(range(lo, hi + 1))
    L = Lattice(n, mesh)
    L.val_idx, L.virt_idx, L.core_idx = val, [i for i in range(n) if i > hi], [i for i in range(n) if i < lo]
    Fk = R.R2k(FR, mesh)
    # ---- HFB ---------------------------------------------------------------------------------------------------------------
    L.set_Ham_lo(fock_lo_R=FR, hcore_lo_R=0.8 * FR)
    L.H0 = 0.1
    symm = bool(rng.integers(0, 2))
    GT, npart, E, res = mfd.HFB(L, V3(v), False, mu=mu, beta=beta, ires=True, symm=symm)
    oGT, on, oE, ores = B.HFB(mesh, Fk, FR, 0.8 * FR, v, mu, H0=0.1, beta=beta, symm=symm)
    note("hfb_rho", np.abs(GT - oGT).max(), 1e-9, ctxinfo)
    note("hfb_E", abs(E - oE) / max(1.0, abs(oE)), 1e-9, ctxinfo)
    note("hfb_ew", np.abs(res["e"] - ores["e"]).max(), 1e-10, ctxinfo)
    # ---- GHF on a triple (aa, -bb, ab) -----------------------------------------------------------------------------------------
    F3 = R.R2k(np.asarray([FR[0], -FR[1], D_R]), mesh)
    H3 = 0.7 * F3
    L.hcore_lo_k, L.fock_lo_k, L.fock_hf_lo_k = H3, F3, 0.9 * F3
    filling = float(rng.choice([0.5, 0.5, 0.4]))
    if (2 * n * nk * filling) % 1 > 1e-9:
        filling = 0.5
    GT, npart, E, res = mfd.GHF(L, V3(v), False, filling=filling, mu=mu, beta=beta, ires=True, symm=symm)
    oGT, on, oE, ores = G.GHF(mesh, H3, F3, v, mu, H0=0.1, filling=filling, beta=beta, symm=symm)
    gap = ores["gap"]
    if beta < np.inf or gap > 1e-6:                     # a degenerate frontier at T = 0 is a tie the two sides may break differently
        note("ghf_rho", np.abs(GT - oGT).max(), 1e-8, ctxinfo)
        note("ghf_E", abs(E - oE) / max(1.0, abs(oE)), 1e-8, ctxinfo)
    note("ghf_ew", np.abs(res["e"] - ores["e"]).max(), 1e-10, ctxinfo)
    # ---- GSO bath, embedding Hamiltonian and fit ---------------------------------------------------------------------------------
    GRho_k = ores["rho_k"]
    GRhoT = R.FFTtoT(GRho_k, mesh).real
    try:
        basis = spinless.get_emb_basis(L, GRhoT)
    except Exception:
        basis = None
    if basis is not None and basis.shape[-1] > 2 * (len(val) + len(L.virt_idx)):
        neo = basis.shape[-1]
        npair = neo * (neo + 1) // 2
        X = rng.standard_normal((5, npair)) / np.sqrt(5)
        H2 = (X.T @ X)[None]
        S3 = np.zeros((3, nk, n, n), dtype=complex)
        S3[0] = S3[1] = np.eye(n)
        L.ovlp_lo_k, L.rdm1_lo_k, L.JK_imp = S3, GRho_k, None
        ib = bool(rng.integers(0, 2))
        L.use_hcore_as_emb_ham = False
        av = bool(rng.integers(0, 2))
        Himp, _ = spinless.get_emb_Ham(L, basis, V3(v), mu, H2_given=H2, int_bath=ib, add_vcor=av)
        oH1, oov, oJK = G.gso_embHam1e(mesh, basis, H2, H3, 0.9 * F3 if ib else F3, S3, GRho_k, v, mu, int_bath=ib, add_vcor=av)
        note("gso_H1", np.abs(Himp.H1["cd"] - oH1).max(), 1e-9, ctxinfo)
        nimp = len(val) + len(L.virt_idx)
        vv = Hubbard.VcorLocal(False, True, n)
        ov = F.VcorLocal(False, True, n)
        noise = 0.05 * rng.standard_normal((neo, neo))
        target = spinless.foldRho_k(GRho_k, L.R2k_basis(basis)) + 0.5 * (noise + noise.T)
        mode = rng.choice(["all", "imp", "det"])
        kw = dict(imp_fit=True) if mode == "imp" else (dict(det=True) if mode == "det" else dict())
        vv.update(np.zeros(vv.length()))
        spinless.FitVcorEmb(target, L, basis, vv, mu, beta=beta, MaxIter=1, **kw)
        fit = spinless.FitVcorEmb.last_fit
        ofit = G.gso_emb_fit(target, mesh, basis, ov, mu, beta, F3, S3, nimp, **kw)
        p = 0.1 * rng.standard_normal(vv.length())
        ew_o = ofit._solve(p)[0][0]
        ne = neo // 2
        if beta < np.inf or ew_o[ne] - ew_o[ne - 1] > 1e-6:
            note("gso_fit_err", abs(fit.errfunc(p) - ofit.errfunc(p)), 1e-9, ctxinfo)
            og = ofit.gradfunc(p) if beta == np.inf else ofit.gradfunc_ft(p)
            note("gso_fit_grad", np.abs(fit.gradfunc(p) - og).max() / max(1.0, np.abs(og).max()), 1e-6, ctxinfo)
        # lattice stage (finite T only: analytic gradient)
        if beta < np.inf:
            tgt = GRhoT[0] + 0.5 * 0.04 * (lambda z: z + z.T)(rng.standard_normal((2 * n, 2 * n)))
            vv.update(0.05 * rng.standard_normal(vv.length()))
            spinless.FitVcorFull(tgt, L, basis, vv, mu, beta, None, MaxIter=1, imp_fit=True)
            ffit = spinless.FitVcorFull.last_fit
            offit = G.GsoFullFit(tgt, mesh, ov, mu, beta, F3, imp_idx=list(range(nimp)))
            note("gso_full_err", abs(ffit.errfunc(p) - offit.errfunc(p)), 1e-9, ctxinfo)
            og = offit.gradfunc_ft(p)
            note("gso_full_grad", np.abs(ffit.gradfunc(p) - og).max() / max(1.0, np.abs(og).max()), 1e-6, ctxinfo)
    # ---- cell-resolved potential: dV/dparam; k-resolved potential: lattice fit ------------------------------------------------------
    spin = int(rng.integers(1, 3))
    nb = int(rng.integers(2, 2 * n + 1))
    bas = rng.standard_normal((spin, nk, n, nb)) / np.sqrt(nk * n)
    idx = sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist())
    pv = pvcor.VcorNonLocal(spin == 1, False, L, idx_range=idx)
    ovn = F.VcorNonLocal(spin == 1, False, mesh, n, idx)
    got = slater.get_dV_dparam(pv, bas, None, L)
    note("nonlocal_dV", np.abs(got - F.get_dV_dparam(ovn, bas)).max(), 1e-11, ctxinfo)
    if beta < np.inf:
        L.set_Ham_lo(fock_lo_R=FR if spin == 2 else FR[0], hcore_lo_R=FR if spin == 2 else FR[0])
        kv = pvcor.VcorKpoints(spin == 1, False, L)
        okv = F.VcorKpoints(spin == 1, mesh, n)
        tgt = 0.5 * np.eye(n)[None].repeat(spin, 0) + 0.05 * (lambda z: z + z.transpose(0, 2, 1))(rng.standard_normal((spin, n, n)))
        full = slater.FullFitDevice(slater.get_ctx(), tgt, L, np.zeros((spin, nk, n, 1)), kv, beta,
                                    F.FullFit(tgt, mesh, np.zeros((spin, nk, n, 1)), okv, beta, Fk if spin == 2 else Fk[0], 0.5,
                                              imp_idx=list(range(n)), det_idx=[]).nelec, list(range(n)), [], False)
        ofull = F.FullFit(tgt, mesh, np.zeros((spin, nk, n, 1)), okv, beta, Fk if spin == 2 else Fk[0], 0.5, imp_idx=list(range(n)), det_idx=[])
        p = 0.2 * rng.standard_normal(kv.length())
        note("kpts_full_err", abs(full.errfunc(p) - ofull.errfunc(p)), 1e-9, ctxinfo)
        og = ofull.gradfunc_ft(p)
        note("kpts_full_grad", np.abs(full.gradfunc(p) - og).max() / max(1.0, np.abs(og).max()), 1e-6, ctxinfo)
out = {"seed": seed, "trials": trials, "seconds": round(time.time() - t0, 1), "worst": {k: float("%.3g" % x) for k, x in sorted(worst.items())}}
print(json.dumps(out))
