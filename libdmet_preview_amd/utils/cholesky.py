"""
Modified (pivoted, incomplete) Cholesky decomposition with the reference's names (libdmet/utils/cholesky.py:21-131) over
dmk_modified_cholesky (csrc/cholesky.hip: the reference's loop operation by operation on the device, same pivot sequence).

    modified_cholesky(mat, max_error)        (n, n) symmetric positive semi-definite -> (nvec, n) vectors, mat ~ V^T V
    modified_cholesky_uhf([aa, bb, ab])      one pivot sequence over the stacked [[aa, ab], [ab^T, bb]] -> (nvec, 2 n)
    get_cderi_rhf(eri_s4, norb, tol)         4-fold ERI (npair, npair) -> (nchol, norb, norb)
    get_cderi_uhf([aa, bb, ab], norb, tol)   -> (2, nchol, norb, norb)
"""
import ctypes as C

import numpy as np

from libdmet_preview_amd._lib import lib, get_ctx
from libdmet_preview_amd.utils import logger as log


def _decompose(blocks, max_error):
    ctx = get_ctx()
    blocks = [np.ascontiguousarray(b, dtype=np.float64) for b in blocks]
    n = blocks[0].shape[0]
    for b in blocks:
        if b.shape != (n, n):
            raise ValueError("modified_cholesky: square blocks of one size expected, got %s" % (b.shape,))
        if not np.isfinite(b).all():
            raise ValueError("modified_cholesky: the matrix contains NaN / Inf")
    uhf = len(blocks) == 3
    d = [ctx.to_device(b) for b in blocks]
    max_vecs = 2 * n + 2
    d_vecs = ctx.empty(((2 if uhf else 1), max_vecs, n), np.float64)
    nvec, exhausted = C.c_int(0), C.c_int(0)
    ctx.check(lib.dmk_modified_cholesky(ctx.h, n, 1 if uhf else 0, d[0].ptr, d[1].ptr if uhf else None, d[2].ptr if uhf else None,
                                        float(max_error), max_vecs, d_vecs.ptr, C.byref(nvec), C.byref(exhausted)))
    if exhausted.value:
        log.warn("modified cholesky does not converge ...")
    return d_vecs, int(nvec.value), max_vecs, n


def modified_cholesky(mat, max_error=1e-6):
    """Vectors V (nvec, n) with mat ~ V^T V, pivots by largest residual diagonal (utils/cholesky.py:21-52)."""
    mat = np.asarray(mat)
    assert mat.ndim == 2
    d_vecs, nvec, _, n = _decompose([mat], max_error)
    return d_vecs.get()[0, :nvec]


def modified_cholesky_uhf(mat, max_error=1e-6):
    """The same for the spin-blocked matrix given as (aa, bb, ab): (nvec, 2 n), alpha columns first (utils/cholesky.py:54-105)."""
    assert len(mat) == 3 and np.ndim(mat[0]) == 2
    d_vecs, nvec, _, n = _decompose([mat[0], mat[1], mat[2]], max_error)
    v = d_vecs.get()
    return np.concatenate([v[0, :nvec], v[1, :nvec]], axis=1)


def _unpack(ctx, d_vecs, row0, nvec, norb):
    npair = norb * (norb + 1) // 2
    d_full = ctx.empty((nvec, norb, norb), np.float64)
    ctx.check(lib.dmk_sym_unpack(ctx.h, norb, nvec, d_vecs.offset(row0 * npair, (nvec, npair)).ptr, None, d_full.ptr))
    return d_full.get()


def get_cderi_rhf(eri, norb, tol=1e-8):
    """Cholesky vectors of a 4-fold ERI as symmetric (nchol, norb, norb) matrices (utils/cholesky.py:107-115)."""
    eri = np.asarray(eri)
    assert eri.ndim == 2
    d_vecs, nvec, max_vecs, n = _decompose([eri], tol)
    return _unpack(get_ctx(), d_vecs, 0, nvec, norb)


def get_cderi_uhf(eri, norb, tol=1e-8):
    """(2, nchol, norb, norb) from the (aa, bb, ab) blocks of a 4-fold ERI (utils/cholesky.py:117-128)."""
    assert len(eri) == 3 and np.ndim(eri[0]) == 2
    d_vecs, nvec, max_vecs, n = _decompose([eri[0], eri[1], eri[2]], tol)
    ctx = get_ctx()
    return np.asarray([_unpack(ctx, d_vecs, 0, nvec, norb), _unpack(ctx, d_vecs, max_vecs, nvec, norb)])
