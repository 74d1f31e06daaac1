"""
Nambu / BCS algebra with the reference's entry points (libdmet/routine/bcs_helper.py):

  extractRdm / extractH1 / combineRdm / swapSpin / basisToCanonical / basisToSpin   :14-70   (block bookkeeping)
  separate_basis, contract_trans_inv, transform_trans_inv                             :176-207
  contract_local, transform_local, transform_local_grad*, transform_imp,
  contract_imp_env, transform_imp_env, get_dV_dparam                                  :248-430

The reference evaluates the translation-invariant folds as ncells^2 Python loops of
basis[i]^T H[i-j] basis[j].  Here every fold is a quadratic form of the *canonical* Nambu basis
C = basisToCanonical(basis) with the Nambu matrix G = [[H_A, D], [D^T, -H_B]]:

    M = C^T G C      ->   H_A' = M[:nb,:nb],  D' = M[:nb,nb:],  H_B' = -M[nb:,nb:]

and the translation-invariant one runs in k space, (1/nk) Re sum_k C_k^H G_k C_k (dmk_fold_R2k + the
batched complex MFMA GEMM), which is O(ncells) instead of O(ncells^2).  The gradient tensors are one
real GEMM C_s^T C_t over the cell index (dmk_dgemm_tn_acc_rect).
"""
import numpy as np

from libdmet_preview_amd._lib import lib, get_ctx
from libdmet_preview_amd.basis_transform.make_basis import bgemm_dev
from libdmet_preview_amd.system import fourier
from libdmet_preview_amd.utils import logger as log


# ---------------------------------------------------------------------------------------------
# block bookkeeping (pure re-indexing of small host arrays)
# ---------------------------------------------------------------------------------------------

def extractRdm(GRho):
    """GRho = [[rho_A, k_ba^dg], [k_ba, 1 - rho_B]]  ->  rho_A, rho_B, kappa_BA."""
    norbs = GRho.shape[0] // 2
    log.eassert(norbs * 2 == GRho.shape[0], "generalized density matrix dimension error")
    return GRho[:norbs, :norbs].copy(), np.eye(norbs) - GRho[norbs:, norbs:], GRho[norbs:, :norbs].copy()


def extractH1(GFock):
    norbs = GFock.shape[0] // 2
    log.eassert(norbs * 2 == GFock.shape[0], "generalized density matrix dimension error")
    return GFock[:norbs, :norbs].copy(), -GFock[norbs:, norbs:], GFock[norbs:, :norbs].copy()


def combineRdm(rhoA, rhoB, kappaAB):
    norbs = rhoA.shape[0]
    return np.block([[rhoA, -kappaAB], [-kappaAB.T, np.eye(norbs) - rhoB]])


def swapSpin(GRho):
    rhoA, rhoB, kappaBA = extractRdm(GRho)
    norbs = rhoA.shape[0]
    return np.block([[rhoB, -kappaBA], [-kappaBA.T, np.eye(norbs) - rhoA]])


def basisToCanonical(basis):
    assert basis.shape[0] == 2
    shape = list(basis.shape[1:])
    nbasis, nsites = shape[-1], shape[-2] // 2
    shape[-1] *= 2
    newbasis = np.empty(tuple(shape))
    newbasis[..., :nbasis] = basis[0]
    newbasis[..., :nsites, nbasis:] = basis[1, ..., nsites:, :]
    newbasis[..., nsites:, nbasis:] = basis[1, ..., :nsites, :]
    return newbasis


def basisToSpin(basis):
    shape = [2] + list(basis.shape)
    shape[-1] = shape[-1] // 2
    nbasis, nsites = shape[-1], shape[-2] // 2
    newbasis = np.empty(tuple(shape))
    newbasis[0] = basis[..., :nbasis]
    newbasis[1, ..., :nsites, :] = basis[..., nsites:, nbasis:]
    newbasis[1, ..., nsites:, :] = basis[..., :nsites, nbasis:]
    return newbasis


def separate_basis(basis):
    nscsites = basis.shape[2] // 2
    # VA, VB, UA, UB
    return basis[0, :, :nscsites], basis[1, :, :nscsites], basis[1, :, nscsites:], basis[0, :, nscsites:]


# ---------------------------------------------------------------------------------------------
# device building blocks: stacks of complex matrices (batch, r, c)
# ---------------------------------------------------------------------------------------------

class _Stack(object):
    __slots__ = ("d", "batch", "r", "c")

    def __init__(self, d, batch, r, c):
        self.d, self.batch, self.r, self.c = d, int(batch), int(r), int(c)

    def flat(self):
        """(batch, r, c) viewed as one (batch*r, c) matrix."""
        return _Stack(self.d, 1, self.batch * self.r, self.c)

    def rows(self):
        """(batch, r, c) viewed as one (batch, r*c) matrix."""
        return _Stack(self.d, 1, self.batch, self.r * self.c)

    def get(self):
        return self.d.get().reshape(self.batch, self.r, self.c)


def _up(ctx, a):
    a = np.asarray(a)
    if a.ndim == 2:
        a = a[None]
    return _Stack(ctx.to_device(a, np.complex128), a.shape[0], a.shape[1], a.shape[2])


def _mm(ctx, opA, A, opB, B, alpha=1.0):
    """op(A[b]) op(B[b]); a stack of one matrix is broadcast over the other's batch."""
    batch = max(A.batch, B.batch)
    assert A.batch in (1, batch) and B.batch in (1, batch)
    M, K = (A.r, A.c) if opA == "N" else (A.c, A.r)
    K2, N = (B.r, B.c) if opB == "N" else (B.c, B.r)
    assert K == K2, (K, K2)
    sA = A.r * A.c if A.batch == batch else 0
    sB = B.r * B.c if B.batch == batch else 0
    return _Stack(bgemm_dev(ctx, opA, opB, M, N, K, batch, A.d, sA, B.d, sB, alpha=alpha), batch, M, N)


def _sum_batch(ctx, A):
    """sum_b A[b] as a (1 x batch) times (batch x r*c) product."""
    ones = _Stack(ctx.to_device(np.ones((1, 1, A.batch)), np.complex128), 1, 1, A.batch)
    s = _mm(ctx, "N", ones, "N", A.rows())
    return _Stack(s.d, 1, A.r, A.c)


def _fold(ctx, lattice, a):
    """R -> k of a real (ncells, r, c) stack (unnormalised, exp(-ik.R)), result stays on the device."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    nk, r, c = a.shape
    d = fourier.fold_R2k_dev(ctx.to_device(a), lattice.kmesh, 1, r * c)
    return _Stack(d, nk, r, c)


def _quad_trans_inv(ctx, lattice, CL, G, CR=None):
    """sum_{ij} CL[i]^T G[i - j] CR[j]  =  (1/nk) Re sum_k CL_k^H G_k CR_k."""
    nk = lattice.ncells
    Lk = _fold(ctx, lattice, CL)
    Rk = Lk if CR is None else _fold(ctx, lattice, CR)
    T = _mm(ctx, "N", _fold(ctx, lattice, G), "N", Rk)
    res = _mm(ctx, "C", Lk.flat(), "N", T.flat(), alpha=1.0 / nk).get()[0]
    if np.abs(res.imag).max(initial=0.0) > 1e-7:
        log.warn("transform_trans_inv: has imag part %s", np.abs(res.imag).max())
    return np.ascontiguousarray(res.real)


def _quad_local(ctx, CL, G, CR=None):
    """sum_i CL[i]^T G CR[i]."""
    L = _up(ctx, CL)
    R = L if CR is None else _up(ctx, CR)
    T = _mm(ctx, "N", _up(ctx, G), "N", R)
    return np.ascontiguousarray(_mm(ctx, "T", L.flat(), "N", T.flat()).get()[0].real)


def _quad_imp_env(ctx, CL, G, CR=None):
    """0.5 (sum_i CL[0]^T G[i] CR[i] + sum_i CL[i]^T G[i] CR[0])   (bcs_helper.py:363-370; i - 0 = i)."""
    L = _up(ctx, CL)
    R = L if CR is None else _up(ctx, CR)
    Gd = _up(ctx, G)
    L0 = _up(ctx, np.asarray(CL)[0])
    R0 = L0 if CR is None else _up(ctx, np.asarray(CR)[0])
    s1 = _sum_batch(ctx, _mm(ctx, "N", Gd, "N", R))              # sum_i G[i] CR[i]
    s2 = _sum_batch(ctx, _mm(ctx, "T", L, "N", Gd))              # sum_i CL[i]^T G[i]
    r1 = _mm(ctx, "T", L0, "N", s1).get()[0].real
    r2 = _mm(ctx, "N", s2, "N", R0).get()[0].real
    return 0.5 * (r1 + r2)


# ---------------------------------------------------------------------------------------------
# contractions with the reference's signatures
# ---------------------------------------------------------------------------------------------

def contract_trans_inv(basisL, basisR, lattice, H):
    return _quad_trans_inv(get_ctx(), lattice, basisL, H, basisR)


def contract_local(basisL, basisR, lattice, H):
    return _quad_local(get_ctx(), basisL, H, basisR)


def contract_imp_env(basisL, basisR, lattice, H):
    return _quad_imp_env(get_ctx(), basisL, H, basisR)


def _split_H(H, single_ndim):
    H = np.asarray(H)
    if H.ndim == single_ndim:
        return H, H, np.zeros_like(H)
    elif H.shape[0] == 2:
        return H[0], H[1], np.zeros_like(H[0])
    elif H.shape[0] == 3:
        return H[0], H[1], H[2]
    raise ValueError("unknown shape of H: %s" % (H.shape,))


def _nambu(HA, HB, D, DT):
    """G = [[H_A, D], [D^T, -H_B]] with a leading cell axis when the blocks have one."""
    n = HA.shape[-1]
    G = np.zeros(HA.shape[:-2] + (2 * n, 2 * n))
    G[..., :n, :n] = HA
    G[..., n:, n:] = -HB
    G[..., :n, n:] = D
    G[..., n:, :n] = DT
    return G


def _hole_pair(basis, HB):
    """Y = [VB; UB] and blockdiag(H_B, H_B): tr(Y^T . Y) restores the +H_B terms of E0."""
    VA, VB, UA, UB = separate_basis(basis)
    Y = np.concatenate([VB, UB], axis=-2)
    n = HB.shape[-1]
    G = np.zeros(HB.shape[:-2] + (2 * n, 2 * n))
    G[..., :n, :n] = HB
    G[..., n:, n:] = HB
    return Y, G


def _nambu_transform(basis, HA, HB, D, DT, quad):
    basis = np.asarray(basis, dtype=np.float64)
    nb = basis.shape[-1]
    C = basisToCanonical(basis)
    M = quad(C, _nambu(HA, HB, D, DT))
    Y, GB = _hole_pair(basis, HB)
    E0 = np.trace(M[nb:, nb:]) + np.trace(quad(Y, GB))
    return np.asarray((M[:nb, :nb], -M[nb:, nb:])), np.ascontiguousarray(M[:nb, nb:]), E0


def transform_trans_inv(basis, lattice, H):
    HA, HB, D = _split_H(H, 3)
    ctx = get_ctx()
    return _nambu_transform(basis, HA, HB, D, lattice.transpose(D), lambda C, G: _quad_trans_inv(ctx, lattice, C, G))


def transform_local(basis, lattice, H):
    HA, HB, D = _split_H(H, 2)
    ctx = get_ctx()
    return _nambu_transform(basis, HA, HB, D, D.T, lambda C, G: _quad_local(ctx, C, G))


def transform_imp(basis, lattice, H):
    HA, HB, D = _split_H(H, 2)
    ctx = get_ctx()
    return _nambu_transform(basis, HA, HB, D, D.T, lambda C, G: _quad_local(ctx, C[:1], G))


def transform_imp_env(basis, lattice, H):
    HA, HB, D = _split_H(H, 3)
    ctx = get_ctx()
    return _nambu_transform(basis, HA, HB, D, lattice.transpose(D), lambda C, G: _quad_imp_env(ctx, C, G))


# ---------------------------------------------------------------------------------------------
# gradient of the embedded potential w.r.t. local vcor entries
# ---------------------------------------------------------------------------------------------

def _cell_gram(basis):
    """Gram tensors over the cell index: g[s][t][(r,p),(r',q)] = sum_c basis[s,c,r,p] basis[t,c,r',q]."""
    basis = np.ascontiguousarray(basis, dtype=np.float64)
    _, ncells, n2, nb = basis.shape
    ctx = get_ctx()
    d = ctx.to_device(basis)
    m = n2 * nb
    out = {}
    for s, t in ((0, 0), (0, 1), (1, 1)):
        d_g = ctx.zeros((m, m), np.float64)
        ctx.check(lib.dmk_dgemm_tn_acc_rect(ctx.h, m, m, ncells, 1.0, d.offset(s * ncells * m, (ncells, m)).ptr, m,
                                            d.offset(t * ncells * m, (ncells, m)).ptr, m, d_g.ptr, m))
        out[(s, t)] = d_g.get().reshape(n2, nb, n2, nb)
    return out


def _gram_pair(basisL, basisR):
    """sum_c L[c,i,p] R[c,j,q] -> (i, p, j, q) as one real GEMM over the cell index."""
    L = np.ascontiguousarray(basisL, dtype=np.float64)
    R = np.ascontiguousarray(basisR, dtype=np.float64)
    ncells = L.shape[0]
    mL, mR = L.shape[1] * L.shape[2], R.shape[1] * R.shape[2]
    ctx = get_ctx()
    d_L, d_R = ctx.to_device(L), ctx.to_device(R)
    d_g = ctx.zeros((mL, mR), np.float64)
    ctx.check(lib.dmk_dgemm_tn_acc_rect(ctx.h, mL, mR, ncells, 1.0, d_L.ptr, mL, d_R.ptr, mR, d_g.ptr, mR))
    return d_g.get().reshape(L.shape[1], L.shape[2], R.shape[1], R.shape[2])


def contract_local_grad(basisL, basisR, lattice):
    """sum_c L[c] (x) R[c]: (i, p, j, q) -> (i, j, p, q)."""
    return np.ascontiguousarray(_gram_pair(basisL, basisR).transpose(0, 2, 1, 3))


def contract_local_grad_DT(basisL, basisR, lattice):
    """sum_c L[c] (x) R[c]: (j, p, i, q) -> (i, j, p, q)   (the D^T terms of dV / dD)."""
    return np.ascontiguousarray(_gram_pair(basisL, basisR).transpose(2, 0, 1, 3))


def transform_local_grad(basis, lattice):
    """bcs_helper.py:285-316 with all twelve cell sums taken from three Gram GEMMs."""
    basis = np.asarray(basis, dtype=np.float64)
    n = basis.shape[2] // 2
    gram = _cell_gram(basis)
    lo, hi = slice(0, n), slice(n, 2 * n)
    # (spin, row half) of VA, VB, UA, UB inside basis
    part = {"VA": (0, lo), "VB": (1, lo), "UA": (1, hi), "UB": (0, hi)}

    def g(a, b, DT=False):
        (sa, ra), (sb, rb) = part[a], part[b]
        if (sa, sb) in gram:
            x = gram[(sa, sb)][ra, :, rb, :]                    # (i, p, j, q)
        else:
            x = gram[(sb, sa)][rb, :, ra, :].transpose(2, 3, 0, 1)
        return x.transpose(2, 0, 1, 3) if DT else x.transpose(0, 2, 1, 3)

    resA = (np.asarray([g("VA", "VA"), -g("UA", "UA")]), g("VA", "UA"), None)
    resB = (np.asarray([-g("UB", "UB"), g("VB", "VB")]), -g("UB", "VB"), None)
    resD = (np.asarray([g("VA", "UB") + g("UB", "VA", True), -g("VB", "UA", True) - g("UA", "VB")]),
            g("VA", "VB") + g("UB", "UA", True), None)
    return resA, resB, resD


def get_dV_dparam(basis, lattice, vcor):
    def sym_triu(a):
        a = a + a.transpose((1, 0, 2, 3))
        a[np.arange(a.shape[0]), np.arange(a.shape[1])] *= 0.5
        return a[np.triu_indices(a.shape[0])]

    nbasis = basis.shape[-1]
    dV_dp = np.empty((vcor.length(), nbasis * 2, nbasis * 2))
    resA, resB, resD = transform_local_grad(basis, lattice)
    flat = lambda x: x.reshape((-1,) + x.shape[-2:])
    dA = np.concatenate([sym_triu(resA[0][0]), sym_triu(resB[0][0]), flat(resD[0][0])], axis=0)
    dB = np.concatenate([sym_triu(resA[0][1]), sym_triu(resB[0][1]), flat(resD[0][1])], axis=0)
    dD = np.concatenate([sym_triu(resA[1]), sym_triu(resB[1]), flat(resD[1])], axis=0)
    for ip in range(vcor.length()):
        dV_dp[ip, :nbasis, :nbasis] = dA[ip]
        dV_dp[ip, nbasis:, nbasis:] = -dB[ip]
        dV_dp[ip, :nbasis, nbasis:] = dD[ip]
        dV_dp[ip, nbasis:, :nbasis] = dD[ip].T
    return dV_dp
