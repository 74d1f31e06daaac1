"""Randomised differential campaign of the basis algebra and the k <-> R folds: system.fourier
FFTtoK / FFTtoT / R2k / k2R on meshes with axes 1 .. 7 (full-mesh fused kernel and the generic path), basis_transform.make_basis
multiply_basis / transform_h1_to_lo / transform_rdm1_to_lo / transform_rdm1_to_ao with every spin-dimension combination, get_basis_k, and
eri_restore (4-fold -> 1-fold / 8-fold), through the C ABI against oracle/restate.py (reference: system/fourier.py:129-177,
basis_transform/make_basis.py, basis_transform/eri_transform.py:523-544).
    STRESS_SEED=1 STRESS_TRIALS=80 python tools/algebra_stress.py          (test infrastructure: imports the oracle)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from libdmet_preview_amd.system import fourier
from libdmet_preview_amd.basis_transform import make_basis as mb
from libdmet_preview_amd.basis_transform import eri_transform as et

rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "80"))
worst = {"fold": 0.0, "algebra": 0.0, "restore": 0.0}
t0 = time.time()


def cplx(*shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def chk(key, got, ref, tol, what):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    e = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
    assert e < tol, (what, e)
    worst[key] = max(worst[key], e)


for trial in range(trials):
    mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4, 5, 6, 7], size=3, p=[0.3, 0.2, 0.15, 0.12, 0.08, 0.1, 0.05]))
    nk = mesh[0] * mesh[1] * mesh[2]
    if nk > 150:
        mesh, nk = (mesh[0], mesh[1], 1), mesh[0] * mesh[1]
    n = int(rng.integers(1, 25))
    m = int(rng.integers(1, 25))
    spin = int(rng.integers(1, 3))
    # ---- a6: folds (real-space operators are real; k-space ones Hermitian in the cell index) ----
    A_R = rng.standard_normal((nk, n, m))
    chk("fold", fourier.FFTtoK(A_R, mesh), R.FFTtoK(A_R, mesh), 1e-12, ("FFTtoK", mesh, n, m))
    A_k = R.FFTtoK(A_R, mesh)
    chk("fold", fourier.FFTtoT(A_k, mesh), R.FFTtoT(A_k, mesh).real, 1e-12, ("FFTtoT", mesh, n, m))
    S_R = rng.standard_normal((spin, nk, n, n))
    chk("fold", fourier.R2k(S_R, mesh), R.R2k(S_R, mesh), 1e-12, ("R2k", mesh, n, spin))
    S_k = R.R2k(S_R, mesh)
    chk("fold", fourier.k2R(S_k, mesh), np.asarray(R.k2R(S_k, mesh)).real, 1e-12, ("k2R", mesh, n, spin))
    chk("fold", fourier.R2k(S_R[0], mesh), R.R2k(S_R[0], mesh), 1e-12, ("R2k 3d", mesh, n))
    # ---- a9 / a10: basis algebra with every spin-dimension combination ----
    nlo, nemb = int(rng.integers(1, 20)), int(rng.integers(1, 30))
    C = cplx(spin, nk, n, nlo)
    bk = cplx(spin, nk, nlo, nemb)
    chk("algebra", mb.multiply_basis(C, bk), R.multiply_basis(C, bk), 1e-12, ("multiply_basis", spin))
    chk("algebra", mb.multiply_basis(C[0], bk), R.multiply_basis(C[0], bk), 1e-12, ("multiply_basis mixed", spin))
    chk("algebra", mb.multiply_basis(C[0], bk[0]), R.multiply_basis(C[0], bk[0]), 1e-12, ("multiply_basis rhf", spin))
    basis_R = rng.standard_normal((spin, nk, nlo, nemb))
    phase = R.get_phase_R2k(mesh, R.make_kpts_scaled(mesh))
    chk("algebra", et.get_basis_k(basis_R, phase), R.get_basis_k(basis_R, phase), 1e-12, ("get_basis_k", mesh, spin))
    h = cplx(spin, nk, n, n)
    h = h + h.conj().transpose(0, 1, 3, 2)
    chk("algebra", mb.transform_h1_to_lo(h, C), R.transform_h1_to_lo(h, C), 1e-11, ("transform_h1_to_lo", spin))
    chk("algebra", mb.transform_h1_to_lo(h[0], C[0]), R.transform_h1_to_lo(h[0], C[0]), 1e-11, ("transform_h1_to_lo rhf",))
    S = cplx(nk, n, n)
    S = S + S.conj().transpose(0, 2, 1)
    chk("algebra", mb.transform_rdm1_to_lo(h, C, S), R.transform_rdm1_to_lo(h, C, S), 1e-11, ("transform_rdm1_to_lo", spin))
    d_lo = cplx(spin, nk, nlo, nlo)
    d_lo = d_lo + d_lo.conj().transpose(0, 1, 3, 2)
    chk("algebra", mb.transform_rdm1_to_ao(d_lo, C), R.transform_rdm1_to_ao(d_lo, C), 1e-11, ("transform_rdm1_to_ao", spin))
    # ---- a14: 4-fold -> 1-fold / 8-fold restore ----
    ne = int(rng.integers(1, 13))
    npair = ne * (ne + 1) // 2
    e4 = rng.standard_normal((npair, npair))
    e4 = e4 + e4.T
    for sym in (1, 4, 8):
        chk("restore", np.asarray(et.eri_restore(e4[None], sym, ne)), np.asarray(R.eri_restore(e4[None], sym, ne)), 1e-14, ("eri_restore", sym, ne))
    e3 = rng.standard_normal((3, npair, npair))
    chk("restore", np.asarray(et.eri_restore(e3, 1, ne)), np.asarray(R.eri_restore(e3, 1, ne)), 1e-14, ("eri_restore uhf", ne))
print("algebra stress ok: %d rounds in %.0f s, worst relative error: folds %.1e, basis algebra %.1e, eri_restore %.1e"
      % (trials, time.time() - t0, worst["fold"], worst["algebra"], worst["restore"]))
