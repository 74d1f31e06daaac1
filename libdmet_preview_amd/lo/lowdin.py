"""
Loewdin orthogonalisation of orbital sets (reference: libdmet/lo/lowdin.py:83-134), batched over spin and k on the
device: metric M = C^H S C (complex MFMA GEMMs), eigh (K1), X = sum_m v_m v_m^H / sqrt(e_m) over e_m > tol as ONE
occupation-weighted density build (dmk_occ_density with occ = e^-1/2), result (C f) X.
The IAO / PySCF orth_ao drivers of the same reference file are outside the path.
"""
import numpy as np

from libdmet_preview_amd._lib import get_ctx
from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.routine import mfd
from libdmet_preview_amd.basis_transform.make_basis import bgemm_dev


def _lowdin_dev(ctx, d_M, n, batch, tol):
    d_w, d_Vt = mfd.eigh_dev(ctx, d_M, n, batch)
    e = d_w.get()
    bad = e <= tol
    if bad.any():
        log.warn("_vec_lowdin has almost zero eigenvalues:\n%s", e[bad])
    occ = np.zeros_like(e)
    occ[~bad] = 1.0 / np.sqrt(e[~bad])
    return mfd.density_dev(ctx, d_Vt, ctx.to_device(occ), n, batch)


def _lowdin(s, tol=1e-14):
    """S^-1/2 on the span of the eigenvectors with eigenvalue > tol (lowdin.py:83-91)."""
    s = np.asarray(s)
    n = s.shape[-1]
    ctx = get_ctx()
    X = _lowdin_dev(ctx, ctx.to_device(s.reshape(1, n, n), np.complex128), n, 1, tol).get().reshape(n, n)
    return X if np.iscomplexobj(s) else np.ascontiguousarray(X.real)


def _batched(C, S, F, tol=1e-14):
    """C (b, p, m), S (b, p, p) or None (identity metric), F (b, m) or None  ->  (C F) (C^H S C)^-1/2."""
    ctx = get_ctx()
    b, p, m = C.shape
    d_C = ctx.to_device(C, np.complex128)
    if S is None:
        d_SC = d_C
    else:
        d_SC = bgemm_dev(ctx, "N", "N", p, m, p, b, ctx.to_device(S, np.complex128), p * p, d_C, p * m)
    d_M = bgemm_dev(ctx, "C", "N", m, m, p, b, d_C, p * m, d_SC, p * m)
    d_X = _lowdin_dev(ctx, d_M, m, b, tol)
    d_Cf = d_C if F is None else ctx.to_device(C * F[:, None, :], np.complex128)
    return bgemm_dev(ctx, "N", "N", p, m, m, b, d_Cf, p * m, d_X, m * m).get().reshape(b, p, m)


def _finish(out, *inputs):
    if any(np.iscomplexobj(np.asarray(x)) for x in inputs if x is not None and not np.isscalar(x)):
        return out
    return np.ascontiguousarray(out.real)


def _vec_lowdin(c, s=1, f=None):
    """Loewdin orthogonalisation for the metric c^H s c: returns (c f) x (lowdin.py:93-101)."""
    c = np.asarray(c)
    S = None if np.isscalar(s) and s == 1 else np.asarray(s)[None]
    if np.isscalar(s) and s != 1:
        S = (s * np.eye(c.shape[0]))[None]
    F = None if f is None else np.asarray(f)[None]
    return _finish(_batched(c[None], S, F)[0], c, s, f)


def vec_lowdin(C, S, f=None):
    """Loewdin orthogonalisation of orbitals with (spin and) k-points (lowdin.py:103-134); all (s, k) in one batch."""
    C, S = np.asarray(C), np.asarray(S)
    if f is not None:
        if C.ndim == 2:
            return _vec_lowdin(C, S, f)
        raise NotImplementedError("vec_lowdin with a scaling array f is only available for a single orbital set")
    if S.ndim == 3:
        nk = C.shape[-3]
        if C.ndim == 3:
            return _finish(_batched(C, S, None), C, S)
        spin = C.shape[0]
        Sb = np.broadcast_to(S, (spin,) + S.shape).reshape(spin * nk, *S.shape[1:])
        out = _batched(C.reshape(spin * nk, *C.shape[2:]), Sb, None)
        return _finish(out.reshape(C.shape), C, S)
    if C.ndim == 2:
        return _vec_lowdin(C, S)
    spin = C.shape[0]
    Sb = np.broadcast_to(S, (spin,) + S.shape)
    return _finish(_batched(C, Sb, None), C, S)


vec_lowdin_k = vec_lowdin


def _cano(s, tol=1e-12):
    """Canonical orthogonalisation factor of a metric (lo/lowdin.py:138-143): eigenvectors with eigenvalue > tol, each divided
    by sqrt(eigenvalue), as columns.  The eigenpairs come from the batched device eigensolver."""
    s = np.asarray(s)
    n = s.shape[-1]
    ctx = get_ctx()
    d_w, d_Vt = mfd.eigh_dev(ctx, ctx.to_device(s.reshape(1, n, n), np.complex128), n, 1)
    e, Vt = d_w.get().reshape(n), d_Vt.get().reshape(n, n)
    idx = e > tol
    log.debug(2, "canonical orthogonalization eigenvals:\n%s", e)
    X = Vt[idx].T / np.sqrt(e[idx])                                  # rows of Vt are the eigenvectors
    return X if np.iscomplexobj(s) else np.ascontiguousarray(X.real)


def _orth_cano(c, s, tol=1e-12, f=None):
    """(c f) . _cano(c^H s c) (lo/lowdin.py:145-156); the two tall products run as device GEMMs."""
    c = np.asarray(c)
    ctx = get_ctx()
    p, m = c.shape
    d_C = ctx.to_device(c.reshape(1, p, m), np.complex128)
    d_SC = d_C if s is None else bgemm_dev(ctx, "N", "N", p, m, p, 1, ctx.to_device(np.asarray(s).reshape(1, p, p), np.complex128),
                                           p * p, d_C, p * m)
    M = bgemm_dev(ctx, "C", "N", m, m, p, 1, d_C, p * m, d_SC, p * m).get().reshape(m, m)
    real = not (np.iscomplexobj(c) or (s is not None and np.iscomplexobj(np.asarray(s))))
    X = _cano(M.real if real else M, tol=tol)
    k = X.shape[-1]
    if k == 0:
        return np.zeros((p, 0), dtype=c.dtype)
    d_Cf = d_C if f is None else ctx.to_device((c * f).reshape(1, p, m), np.complex128)
    out = bgemm_dev(ctx, "N", "N", p, k, m, 1, d_Cf, p * m, ctx.to_device(np.asarray(X).reshape(1, m, k), np.complex128), m * k)
    out = out.get().reshape(p, k)
    return np.ascontiguousarray(out.real) if real and (f is None or not np.iscomplexobj(f)) else out
