"""Randomised differential campaign of the GSO ("spinless", generalised spin orbital) twins of the path:
spinless.get_emb_basis (Schmidt bath of a generalised density with orthogonalised virtuals and particle-weight ordering) and
get_emb_eri_gso (DF transform with the partial particle-hole contraction), through the C ABI against oracle/restate_gso.py
(reference: routine/spinless.py:58-163, basis_transform/eri_transform.py:1104-1277) on random lattices and physical DF tensors.
    STRESS_SEED=1 STRESS_TRIALS=40 python tools/gso_stress.py          (test infrastructure: imports the oracle)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from oracle import restate_bcs as B
from oracle import restate_gso as G
from libdmet_preview_amd import synth
from libdmet_preview_amd.routine import spinless
from libdmet_preview_amd.basis_transform import eri_transform as et
from libdmet_preview_amd.system.lattice import Lattice, _UnitCell

rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "40"))
worst = {"bath": 0.0, "eri": 0.0}
t0, cut, odd = time.time(), 0, 0
for trial in range(trials):
    # ---- bath of a generalised density (quasi-particle vacuum of a random BdG problem) ----
    mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.45, 0.3, 0.15, 0.1]))
    nk = mesh[0] * mesh[1] * mesh[2]
    if nk < 2:
        mesh, nk = (3, 1, 1), 3
    n = int(rng.integers(2, 13))
    FR = synth.make_fock_R(mesh, n, spin=2, seed=int(rng.integers(1, 1 << 30)))
    v = 0.2 * rng.standard_normal((3, n, n))
    v[0], v[1] = v[0] + v[0].T, v[1] + v[1].T
    ewo, evo = B.DiagBdG(R.R2k(FR, mesh), v, float(rng.uniform(-0.3, 0.3)))
    GRho = R.FFTtoT(np.einsum("kpm,km,kqm->kpq", evo, (ewo < 0).astype(float), evo.conj()), mesh).real
    lo, hi = sorted(int(x) for x in rng.integers(0, n, size=2))
    val = list(range(lo, hi + 1))                       # valence orbitals contiguous: core below, virtual above (as in the reference's use)
    L = Lattice(n, mesh)
    L.val_idx, L.virt_idx, L.core_idx = val, [i for i in range(n) if i > hi], [i for i in range(n) if i < lo]
    imp = val + L.virt_idx
    nimp = 2 * len(imp)
    for vb in (True, False):
        try:
            ref, sigma, w = G.get_emb_basis_gso(GRho, n, val, imp, valence_bath=vb)
        except AssertionError:
            # an odd number of singular values above tol_bath: the reference refuses (spinless.py:113); so must the product
            try:
                spinless.get_emb_basis(L, GRho, valence_bath=vb)
            except Exception:
                odd += 1
                continue
            raise AssertionError((trial, mesh, n, val, vb, "reference refuses an odd nbath, the product does not"))
        nb = ref.shape[-1] - nimp
        s = np.sort(sigma)[::-1]
        if nb < len(s) and (s[nb - 1] < 1e-7 or s[nb] > 1e-11) if nb >= 1 else False:
            cut += 1                                    # a singular value next to tol_bath: the cut is a matter of rounding
            continue
        b = spinless.get_emb_basis(L, GRho, valence_bath=vb)
        assert b.shape == ref.shape, (trial, mesh, n, val, vb, b.shape, ref.shape)
        assert np.array_equal(b[..., :nimp], ref[..., :nimp]), (trial, "impurity block")
        if nb > 0:
            a2, r2 = b.reshape(-1, b.shape[-1])[:, nimp:], ref.reshape(-1, ref.shape[-1])[:, nimp:]
            assert np.abs(a2.T @ a2 - np.eye(nb)).max() < 1e-11, (trial, mesh, n, val, vb, "bath not orthonormal")
            d = float(np.sqrt(2.0) * np.linalg.norm(r2 - a2 @ (a2.T @ r2)))
            assert d < 1e-9 + 1e-15 / max(float(s[:nb].min()), 1e-300), (trial, mesh, n, val, vb, d)
            worst["bath"] = max(worst["bath"], d)
    # ---- GSO ERI on a random physical DF tensor ----
    mesh2 = tuple(int(x) for x in rng.choice([1, 2, 3], size=3, p=[0.5, 0.35, 0.15]))
    nk2 = mesh2[0] * mesh2[1] * mesh2[2]
    if nk2 < 2 or nk2 > 9:
        mesh2, nk2 = (2, 2, 1), 4
    nao, naux = int(rng.integers(2, 8)), int(rng.integers(2, 10))
    nlo = nao
    W0 = rng.standard_normal((naux, nk2, nao, nk2, nao)) * np.exp(-0.4 * np.arange(nk2))[None, :, None, None, None] \
        * np.exp(-0.4 * np.arange(nk2))[None, None, None, :, None]
    W0 = W0 + W0.transpose(0, 3, 4, 1, 2)
    ks = R.make_kpts_scaled(mesh2)
    blocks = R.df_blocks_from_W0(W0, mesh2, ks)
    cell = _UnitCell(nao)
    kpts = cell.get_abs_kpts(ks)
    mydf = et.GDFMemory(kpts, dict(blocks), naux)
    sp = int(rng.integers(1, 3))
    C = synth.make_C_ao_lo(mesh2, nao, nlo, spin=sp, seed=int(rng.integers(1, 1000)))
    C = C[0] if sp == 1 else C
    nemb = int(rng.integers(2, 11))
    basis = rng.standard_normal((nk2, 2 * nlo, nemb)) / np.sqrt(2 * nlo)
    get = lambda i, j: blocks[(i, j)]
    for kw in (dict(t_reversal_symm=True), dict(t_reversal_symm=False), dict(symmetry=1), dict(unit_eri=True)):
        e = et.get_emb_eri_gso(cell, mydf, C_ao_lo=C, basis=basis, **kw)
        ref = G.get_emb_eri_gso(mesh2, ks, get, naux, nao, C, basis, **kw)
        assert e.shape == ref.shape, (trial, kw, e.shape, ref.shape)
        err = float(np.abs(e - ref).max()) / max(1.0, float(np.abs(ref).max()))
        assert err < 1e-8, (trial, mesh2, nao, naux, nemb, sp, kw, err)
        worst["eri"] = max(worst["eri"], err)
print("gso stress ok: %d rounds in %.0f s (%d baths not compared: singular value at the cut-off; %d refused by both: odd nbath), "
      "worst bath projector distance %.1e, ERI %.1e" % (trials, time.time() - t0, cut, odd, worst["bath"], worst["eri"]))
