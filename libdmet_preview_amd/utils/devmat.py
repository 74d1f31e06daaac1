"""
Small device-expression layer shared by the one-body folds (routine/slater_helper.py, routine/bcs_helper.py):
stacks of complex matrices (batch, r, c) in HBM, products through the batched complex MFMA GEMM
(dmk_zgemm_batched), R->k folds through dmk_fold_R2k.  Host arrays go in, small host results come out;
everything in between stays on the device.
"""
import numpy as np

from libdmet_preview_amd.basis_transform.make_basis import bgemm_dev
from libdmet_preview_amd.system import fourier
from libdmet_preview_amd.utils import logger as log


class Stack(object):
    __slots__ = ("d", "batch", "r", "c")

    def __init__(self, d, batch, r, c):
        self.d, self.batch, self.r, self.c = d, int(batch), int(r), int(c)

    def flat(self):
        """(batch, r, c) viewed as one (batch*r, c) matrix."""
        return Stack(self.d, 1, self.batch * self.r, self.c)

    def rows(self):
        """(batch, r, c) viewed as one (batch, r*c) matrix."""
        return Stack(self.d, 1, self.batch, self.r * self.c)

    def get(self):
        return self.d.get().reshape(self.batch, self.r, self.c)


def up(ctx, a):
    a = np.asarray(a)
    if a.ndim == 2:
        a = a[None]
    return Stack(ctx.to_device(a, np.complex128), a.shape[0], a.shape[1], a.shape[2])


def mm(ctx, opA, A, opB, B, alpha=1.0):
    """op(A[b]) op(B[b]); a stack of one matrix is broadcast over the other's batch."""
    batch = max(A.batch, B.batch)
    assert A.batch in (1, batch) and B.batch in (1, batch)
    M, K = (A.r, A.c) if opA == "N" else (A.c, A.r)
    K2, N = (B.r, B.c) if opB == "N" else (B.c, B.r)
    assert K == K2, (K, K2)
    sA = A.r * A.c if A.batch == batch else 0
    sB = B.r * B.c if B.batch == batch else 0
    return Stack(bgemm_dev(ctx, opA, opB, M, N, K, batch, A.d, sA, B.d, sB, alpha=alpha), batch, M, N)


def sum_batch(ctx, A):
    """sum_b A[b] as a (1 x batch) times (batch x r*c) product."""
    ones = Stack(ctx.to_device(np.ones((1, 1, A.batch)), np.complex128), 1, 1, A.batch)
    s = mm(ctx, "N", ones, "N", A.rows())
    return Stack(s.d, 1, A.r, A.c)


def fold(ctx, lattice, a):
    """R -> k of a real (ncells, r, c) stack (unnormalised, exp(-ik.R)), result stays on the device."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    nk, r, c = a.shape
    d = fourier.fold_R2k_dev(ctx.to_device(a), lattice.kmesh, 1, r * c)
    return Stack(d, nk, r, c)


def quad_trans_inv(ctx, lattice, CL, G, CR=None):
    """sum_{ij} CL[i]^T G[i - j] CR[j]  =  (1/nk) Re sum_k CL_k^H G_k CR_k."""
    nk = lattice.ncells
    Lk = fold(ctx, lattice, CL)
    Rk = Lk if CR is None else fold(ctx, lattice, CR)
    T = mm(ctx, "N", fold(ctx, lattice, G), "N", Rk)
    res = mm(ctx, "C", Lk.flat(), "N", T.flat(), alpha=1.0 / nk).get()[0]
    if np.abs(res.imag).max(initial=0.0) > 1e-7:
        log.warn("transform_trans_inv: has imag part %s", np.abs(res.imag).max())
    return np.ascontiguousarray(res.real)


def quad_local(ctx, CL, G, CR=None):
    """sum_i CL[i]^T G CR[i]."""
    L = up(ctx, CL)
    R = L if CR is None else up(ctx, CR)
    T = mm(ctx, "N", up(ctx, G), "N", R)
    return np.ascontiguousarray(mm(ctx, "T", L.flat(), "N", T.flat()).get()[0].real)


def quad_imp_env(ctx, CL, G, CR=None):
    """0.5 (sum_i CL[0]^T G[i] CR[i] + sum_i CL[i]^T G[i] CR[0])   (bcs_helper.py:363-370; i - 0 = i)."""
    L = up(ctx, CL)
    R = L if CR is None else up(ctx, CR)
    Gd = up(ctx, G)
    L0 = up(ctx, np.asarray(CL)[0])
    R0 = L0 if CR is None else up(ctx, np.asarray(CR)[0])
    s1 = sum_batch(ctx, mm(ctx, "N", Gd, "N", R))              # sum_i G[i] CR[i]
    s2 = sum_batch(ctx, mm(ctx, "T", L, "N", Gd))              # sum_i CL[i]^T G[i]
    r1 = mm(ctx, "T", L0, "N", s1).get()[0].real
    r2 = mm(ctx, "N", s2, "N", R0).get()[0].real
    return 0.5 * (r1 + r2)


def quad_k(ctx, Bk, Hk, Rk=None):
    """(1/nk) sum_k Bk[k]^H Hk[k] Rk[k] for host (nk, nlo, nb) / (nk, nlo, nlo) stacks; returns complex (nb, nbR)."""
    L = up(ctx, Bk)
    R = L if Rk is None else up(ctx, Rk)
    T = mm(ctx, "N", up(ctx, Hk), "N", R)
    return mm(ctx, "C", L.flat(), "N", T.flat(), alpha=1.0 / L.batch).get()[0]
