"""Randomised differential campaign of the mean-field + bath half of the path: routine.mfd.HF (batched k-point diagonalisation,
occupations, rho_k, k -> R fold) and routine.slater.get_emb_basis (Schmidt bath) through the C ABI against the oracle's restatement
(oracle/restate.py: HF, get_emb_basis) on random lattices -- meshes with odd and even axes, 2 .. 72 orbitals per cell (every
eigensolver family: Jacobi, LDS-resident, HBM-resident), restricted and unrestricted, T = 0 and T > 0, random fillings with a
non-degenerate frontier, random valence / virtual splits for the bath.
    STRESS_SEED=1 STRESS_TRIALS=60 python tools/meanfield_stress.py          (test infrastructure: imports the oracle)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from libdmet_preview_amd import synth
from libdmet_preview_amd.routine import mfd, slater
from libdmet_preview_amd.system.lattice import Lattice


class _Vcor(object):
    def __init__(self, v):
        self.value = v

    def islocal(self):
        return True

    def get(self, i=0, kspace=True):
        return self.value if (kspace or i == 0) else np.zeros_like(self.value)


rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "60"))
worst = {"ew": 0.0, "rho": 0.0, "bath": 0.0}
t0, done, skipped, skipped_bath = time.time(), 0, 0, 0
for trial in range(trials):
    mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4, 5, 6], size=3, p=[0.35, 0.25, 0.15, 0.1, 0.05, 0.1]))
    nk = mesh[0] * mesh[1] * mesh[2]
    if nk < 2:
        mesh = (2, 1, 1); nk = 2
    nlo = int(rng.choice([int(rng.integers(2, 17)), int(rng.integers(17, 73))]))
    if nk * nlo > 3000:
        nlo = max(2, 3000 // nk)
    restricted = bool(rng.random() < 0.5)
    spin = 1 if restricted else 2
    beta = np.inf if rng.random() < 0.75 else float(rng.uniform(5.0, 50.0))
    FR = synth.make_fock_R(mesh, nlo, spin=spin, seed=int(rng.integers(1, 1 << 30)))
    v = 0.1 * rng.standard_normal((spin if not restricted else 1, nlo, nlo))
    v = v + v.transpose(0, 2, 1)
    # a filling whose T = 0 frontier is not degenerate: electrons per spin channel = a gap of the sorted levels wider than 1e-4
    Fk = R.R2k(FR, mesh)
    vm = v if v.shape[0] == spin else np.repeat(v, spin, axis=0)
    ew_all = np.sort(np.concatenate([np.linalg.eigvalsh(Fk[s, k] + vm[s]) for s in range(spin) for k in range(nk)]))
    gaps = np.where(np.diff(ew_all) > 1e-4)[0]
    gaps = gaps[(gaps > len(ew_all) // 8) & (gaps < 7 * len(ew_all) // 8)]
    if len(gaps) == 0:
        skipped += 1
        continue
    nelec_levels = int(rng.choice(gaps)) + 1                 # number of occupied (spin, k, level) states
    filling = nelec_levels / float(spin * nk * nlo)
    L = Lattice(nlo, mesh)
    L.set_Ham_lo(fock_lo_R=FR)
    vc = _Vcor(v if not restricted else v[:1])
    rhoT, mu, E, res = mfd.HF(L, vc, filling, restricted, beta=beta, ires=True)
    rr, mur, Er, resr = R.HF(mesh, Fk, FR, FR, vc.get(0, True), filling, restricted, beta=beta, ires=True)
    e_ew = float(np.abs(res["e"] - resr["e"]).max())
    e_rho = float(np.abs(rhoT - rr).max())
    assert e_ew < 1e-10, (trial, mesh, nlo, restricted, beta, e_ew)
    if beta == np.inf:
        assert np.array_equal(res["mo_occ"], resr["mo_occ"]), (trial, mesh, nlo, restricted, "occupations differ")
    # T > 0: the reference's own brentq leaves mu uncertain to 1e-12 (1 + |mu|); the occupations inherit beta / 4 times that
    tol_rho = 1e-9 if beta == np.inf else 1e-9 + 0.25 * beta * 2e-12 * (1.0 + abs(mur)) * nlo
    assert e_rho < tol_rho, (trial, mesh, nlo, restricted, beta, filling, e_rho)
    worst["ew"], worst["rho"] = max(worst["ew"], e_ew), max(worst["rho"], e_rho)
    # ---- Schmidt bath from the reference's own density (so that a rho difference cannot leak into the bath comparison) ----
    nval = int(rng.integers(1, nlo + 1))
    perm = rng.permutation(nlo)
    val, virt = sorted(int(x) for x in perm[:nval]), sorted(int(x) for x in perm[nval:])
    L.val_idx, L.virt_idx, L.core_idx = val, virt, []
    rdm = rr if not restricted else rr[0]
    b = slater.get_emb_basis(L, rdm)
    ref, info = R.get_emb_basis(mesh, nlo, rdm, imp_idx=list(range(nlo)), val_idx=val, return_info=True)
    # the comparison is only defined where the bath is: singular values next to the cut-off (tol_bath = 1e-9) or to each other make the
    # kept subspace a matter of rounding in the reference itself
    sig = [np.sort(np.asarray(x))[::-1] for x in info["sigma"]]
    nb = min(info["nbath_s"])
    cond_ok = True
    for x in sig:
        if len(x) and (np.any((x > 1e-11) & (x < 1e-7)) or (1 <= nb < len(x) and x[nb - 1] - x[nb] < 1e-6 * max(x[0], 1e-300) and x[nb] > 1e-9)):
            cond_ok = False
    if not cond_ok:
        if os.environ.get("STRESS_VERBOSE"):
            print("bath not compared:", trial, mesh, nlo, nval, restricted, [x.tolist() for x in sig], info["nbath_s"])
        skipped_bath += 1
        done += 1
        continue
    assert b.shape == ref.shape, (trial, mesh, nlo, nval, b.shape, ref.shape)
    for s in range(b.shape[0]):
        B, Bref = b[s].reshape(-1, b.shape[-1]), ref[s].reshape(-1, ref.shape[-1])
        assert np.abs(B.T @ B - np.eye(B.shape[1])).max() < 1e-11, (trial, "orthonormality")
        d = np.sqrt(2.0) * np.linalg.norm(B - Bref @ (Bref.T @ B))
        worst["bath"] = max(worst["bath"], float(d))
        smin = min([float(x[:nb].min()) for x in sig if len(x) and nb >= 1] or [1.0])
        assert d < 1e-9 + 1e-15 / max(smin, 1e-300) ** 1, (trial, mesh, nlo, nval, restricted, d, [x.tolist() for x in sig], info["nbath_s"])
    done += 1
print("mean-field stress ok: %d lattices (%d skipped: no gap at the Fermi level; %d baths not compared: singular values at the cut-off) in %.0f s, "
      "worst |dew| %.1e, |drho| %.1e, bath projector distance %.1e" % (done, skipped, skipped_bath, time.time() - t0, worst["ew"], worst["rho"], worst["bath"]))
