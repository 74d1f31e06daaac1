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
from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.utils.devmat import quad_trans_inv as _quad_trans_inv, quad_local as _quad_local, \
    quad_imp_env as _quad_imp_env


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


def _pair_symmetrised(t):
    """(n, n, m, m) derivative tensor of a SYMMETRIC local block -> (n (n + 1) / 2, m, m): entry (i <= j) is the response to the
    single parameter shared by elements (i, j) and (j, i), i.e. t[i, j] + t[j, i], and t[i, i] on the diagonal."""
    iu, ju = np.triu_indices(t.shape[0])
    both = t[iu, ju] + t[ju, iu]
    both[iu == ju] *= 0.5
    return both


def get_dV_dparam(basis, lattice, vcor):
    """dV_emb / dparam of the Nambu embedding potential (bcs_helper.py:390-428): parameters ordered as the upper triangles of the two
    normal blocks followed by all elements of the pairing block; rows of the three stacked derivative tables follow that order."""
    nbasis = basis.shape[-1]
    nparam = vcor.length()
    (nA, pA, _), (nB, pB, _), (nD, pD, _) = transform_local_grad(basis, lattice)      # (normal [A, B], pairing, E0) per block kind
    stack = lambda a, b, d: np.concatenate([_pair_symmetrised(a), _pair_symmetrised(b), d.reshape((-1,) + d.shape[-2:])], axis=0)
    dA, dB, dD = stack(nA[0], nB[0], nD[0]), stack(nA[1], nB[1], nD[1]), stack(pA, pB, pD)
    dV_dp = np.empty((nparam, nbasis * 2, nbasis * 2))
    # Nambu layout: [[dA, dD], [dD^T, -dB]] for every parameter at once
    dV_dp[:, :nbasis, :nbasis], dV_dp[:, nbasis:, nbasis:] = dA[:nparam], -dB[:nparam]
    dV_dp[:, :nbasis, nbasis:], dV_dp[:, nbasis:, :nbasis] = dD[:nparam], dD[:nparam].transpose(0, 2, 1)
    return dV_dp


# ---------------------------------------------------------------------------------------------
# root of a monotonic function (bcs_helper.py:72-129): the chemical-potential search of the BCS mean field
# ---------------------------------------------------------------------------------------------

def mono_fit(fn, y0, x0, thr, increase=True, dx=1.0, verbose=True):
    """x with |fn(x) - y0| < thr for a monotonic fn: unit steps from x0 until the target is bracketed, then false position with the
    split clamped to [0.2, 0.8] of the bracket.  The sequence of evaluation points is the reference's (the fitted value is carried
    from one DMET iteration to the next)."""
    if not increase:
        return mono_fit(lambda x: -fn(x), -y0, x0, thr, True)
    calls = [0]

    def evaluate(x):
        y = fn(x)
        if verbose:
            log.debug(1, "Iter %2d, x = %20.12f, f(x) = %20.12f", calls[0], x, y)
        calls[0] += 1
        return y

    if verbose:
        log.debug(0, "target f(x) = %20.12f", y0)
    hit = lambda y: abs(y - y0) < thr
    lo_x, lo_y = x0, evaluate(x0)
    if hit(lo_y):
        return lo_x
    step = -dx if lo_y > y0 else dx
    while True:                                                  # walk until the target lies between two consecutive points
        hi_x = lo_x + step
        hi_y = evaluate(hi_x)
        if hit(hi_y):
            return hi_x
        if (lo_y - y0) * (hi_y - y0) < 0:
            break
        lo_x, lo_y = hi_x, hi_y
    if lo_x > hi_x:
        lo_x, lo_y, hi_x, hi_y = hi_x, hi_y, lo_x, lo_y
    while hi_x - lo_x > 0.1 * thr:
        frac = min(max((y0 - lo_y) / (hi_y - lo_y), 0.2), 0.8)
        mid_x = lo_x * (1. - frac) + hi_x * frac
        mid_y = evaluate(mid_x)
        if hit(mid_y):
            return mid_x
        if (mid_y - y0) * (lo_y - y0) < 0:
            hi_x, hi_y = mid_x, mid_y
        else:
            lo_x, lo_y = mid_x, mid_y
    return 0.5 * (lo_x + hi_x)


def mono_fit_2(fn, y0, x0, thr, increase=True, dx=1.0, verbose=True, maxiter=1000):
    """mono_fit with Brent's method on the bracket (bcs_helper.py:131-174); a decreasing function goes to mono_fit like in the reference."""
    from scipy.optimize import brentq
    if not increase:
        return mono_fit(lambda x: -fn(x), -y0, x0, thr, True)
    if verbose:
        log.debug(0, "target f(x) = %20.12f", y0)
    lo_x, lo_y = x0, fn(x0)
    if abs(lo_y - y0) < thr:
        return lo_x
    step = -dx if lo_y > y0 else dx
    for _ in range(maxiter * 50):
        hi_x = lo_x + step
        hi_y = fn(hi_x)
        if abs(hi_y - y0) < thr:
            return hi_x
        if (lo_y - y0) * (hi_y - y0) < 0:
            break
        lo_x, lo_y = hi_x, hi_y
    else:
        raise RuntimeError("Cannot find the section.")
    left, right = min(lo_x, hi_x), max(lo_x, hi_x)
    root, info = brentq(lambda x: fn(x) - y0, left, right, xtol=thr, rtol=thr, maxiter=maxiter, full_output=True, disp=False)
    if not info.converged:
        log.warn("mono_fit_2: brentq fails. x: %s, y: %s", root, fn(root) - y0)
    return root
