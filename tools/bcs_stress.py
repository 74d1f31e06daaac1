"""Randomised differential campaign of the BCS / Nambu twin of the path: mfd.DiagBdG / DiagGHF
(batched 2n x 2n diagonalisations), bcs.embBasis (quasi-particle Schmidt bath), bcs_helper's Nambu folds and gradient tables, through
the C ABI against oracle/restate_bcs.py (reference: routine/mfd.py:429-641, routine/bcs.py:78-104, routine/bcs_helper.py) on random
lattices: meshes with odd and even axes, 1 .. 20 orbitals per cell, random normal and pairing potentials, random valence sets.
    STRESS_SEED=1 STRESS_TRIALS=60 python tools/bcs_stress.py          (test infrastructure: imports the oracle)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from oracle import restate_bcs as B
from libdmet_preview_amd import synth
from libdmet_preview_amd.routine import mfd, bcs, bcs_helper as bh
from libdmet_preview_amd.system.lattice import Lattice


class _Vcor(object):
    def __init__(self, v):
        self.value = v

    def get(self, i=0, kspace=True):
        return self.value if (kspace or i == 0) else np.zeros_like(self.value)

    def length(self):
        n = self.value.shape[-1]
        return n * (n + 1) + n * n


def _occ_proj(ew, ev):
    return np.einsum("kpm,km,kqm->kpq", ev, (ew < 0).astype(float), ev.conj())


rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "60"))
worst = {"ew": 0.0, "proj": 0.0, "bath": 0.0, "fold": 0.0, "dV": 0.0}
t0, done, gapless, cut = time.time(), 0, 0, 0
for trial in range(trials):
    mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4, 5], size=3, p=[0.4, 0.3, 0.15, 0.1, 0.05]))
    nk = mesh[0] * mesh[1] * mesh[2]
    if nk < 2:
        mesh, nk = (3, 1, 1), 3
    n = int(rng.integers(1, 21))
    if nk * n > 800:
        n = max(1, 800 // nk)
    FR = synth.make_fock_R(mesh, n, spin=2, seed=int(rng.integers(1, 1 << 30)))
    Fk = R.R2k(FR, mesh)
    v = 0.2 * rng.standard_normal((3, n, n))
    v[0], v[1] = v[0] + v[0].T, v[1] + v[1].T
    mu = float(rng.uniform(-0.3, 0.3))
    nval = int(rng.integers(1, n + 1))
    val = sorted(int(x) for x in rng.permutation(n)[:nval])
    L = Lattice(n, mesh)
    L.val_idx, L.virt_idx, L.core_idx = list(val), [], [i for i in range(n) if i not in val]
    vc = _Vcor(v)
    # ---- a3: BdG diagonalisation, quasi-particle projector ----
    ew, ev = mfd.DiagBdG(Fk, vc, mu)
    ewo, evo = B.DiagBdG(Fk, v, mu)
    e_ew = float(np.abs(ew - ewo).max())
    assert e_ew < 1e-10, (trial, mesh, n, e_ew)
    m = ev.shape[-1]
    assert np.abs(np.einsum("kpm,kpn->kmn", ev.conj(), ev) - np.eye(m)).max() < 1e-11, (trial, mesh, n, "eigenvectors not orthonormal")
    worst["ew"] = max(worst["ew"], e_ew)
    if np.abs(ewo).min() < 1e-6:
        gapless += 1                                         # a level at zero: the occupied projector is a matter of rounding
    else:
        e_p = float(np.abs(_occ_proj(ew, ev) - _occ_proj(ewo, evo)).max())
        assert e_p < 1e-9, (trial, mesh, n, e_p)
        worst["proj"] = max(worst["proj"], e_p)
    # ---- a8: quasi-particle bath from the oracle's generalised density ----
    GRho_k = _occ_proj(ewo, evo)
    GRho = R.FFTtoT(GRho_k, mesh).real
    o_basis, o_sigma, o_B, o_w = B.embBasis_proj(GRho, n, val)
    sig = np.sort(o_sigma)[::-1]
    ww = np.sort(o_w)[::-1]
    # the particle / hole split of the bath orders the singular vectors by particle weight (bcs.py:92-95): a tie at the split, or a
    # singular value at zero, leaves the split to rounding in the reference itself
    Bd = bcs.embBasis(L, GRho, only_return_bath=True)
    a, b = Bd.reshape(-1, Bd.shape[-1]), o_B.reshape(-1, o_B.shape[-1])
    assert np.abs(a.T @ a - np.eye(a.shape[1])).max() < 1e-11, (trial, mesh, n, nval, "bath not orthonormal")
    if sig.min() > 1e-7:
        d = float(np.sqrt(2.0) * np.linalg.norm(b - a @ (a.T @ b)))
        assert d < 1e-9 + 1e-15 / sig.min(), (trial, mesh, n, nval, d, sig.min())
        worst["bath"] = max(worst["bath"], d)
    else:
        cut += 1
    basis = bcs.embBasis(L, GRho)
    assert basis.shape == o_basis.shape, (trial, basis.shape, o_basis.shape)
    # ---- a8: Nambu folds and gradient tables on the ORACLE's basis (gauge fixed) ----
    D_R = 0.1 * rng.standard_normal((nk, n, n))
    H3 = np.asarray([FR[0], FR[1], D_R])
    for tag, fn, fo, Hm in (("ti3", bh.transform_trans_inv, B.transform_trans_inv, H3), ("ti1", bh.transform_trans_inv, B.transform_trans_inv, FR[0]),
                            ("loc3", bh.transform_local, B.transform_local, v), ("imp3", bh.transform_imp, B.transform_imp, v),
                            ("ie3", bh.transform_imp_env, B.transform_imp_env, H3)):
        (hA, hB), hD, e0 = fn(o_basis, L, Hm)
        (rA, rB), rD, r0 = fo(o_basis, mesh, Hm)
        e_f = max(float(np.abs(hA - rA).max()), float(np.abs(hB - rB).max()), float(np.abs(hD - rD).max()), abs(e0 - r0))
        assert e_f < 1e-10 * max(1.0, float(np.abs(rA).max())), (trial, mesh, n, nval, tag, e_f)
        worst["fold"] = max(worst["fold"], e_f)
    dV = bh.get_dV_dparam(o_basis, L, vc)
    e_dV = float(np.abs(dV - B.get_dV_dparam(o_basis, vc.length())).max())
    assert e_dV < 1e-12, (trial, mesh, n, nval, e_dV)
    worst["dV"] = max(worst["dV"], e_dV)
    done += 1
print("bcs stress ok: %d lattices in %.0f s (%d with a level at zero: projector not compared; %d with a singular value at zero: bath not compared), "
      "worst |dew| %.1e, projector %.1e, bath %.1e, folds %.1e, dV_dparam %.1e"
      % (done, time.time() - t0, gapless, cut, worst["ew"], worst["proj"], worst["bath"], worst["fold"], worst["dV"]))
