"""
oracle/restate_bcs.py -- CPU restatement (numpy) of the Nambu / BCS twin of the hot path and of the
remaining small rows of SURVEY.md section 8a:

  a3   DiagGHF / DiagGHF_symm (routine/mfd.py:591-641), DiagBdG / DiagBdGsymm (:429-478)
  a7   HubPhSymm.basisMatching (dmet/HubPhSymm.py:37-48)
  a8   routine/bcs_helper.py (:14-70, :176-207, :248-316, :346-430) and bcs.embBasis (routine/bcs.py:25-135)
  a14  slater_helper.unit2emb (routine/slater_helper.py:494-528) and the caller's spin-block reorder
       (routine/slater.py:461-462)

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  Pinned against the reference itself through tests/golden/G7_bcs.npz
(oracle/gen_golden.py gen_G7, reference functions executed under oracle/shim.py).
"""
import itertools as it

import numpy as np
import scipy.linalg as la

from oracle.restate import CellArith


# ---------------------------------------------------------------------------------------------
# a3: generalised / Nambu diagonalisations
# ---------------------------------------------------------------------------------------------

def _neg_table(kmesh):
    ca = CellArith(kmesh)
    return [ca.cell_pos2idx(-ca.cell_idx2pos(i)) for i in range(ca.ncells)]


def _symm_fill(ew, ev, kmesh, solve):
    """mfd.py:461-476 / :629-640: the later member of a (k, -k) pair is the conjugate of the earlier."""
    neg = _neg_table(kmesh)
    done = set()
    for i in range(ew.shape[0]):
        if neg[i] in done:
            ew[i], ev[i] = ew[neg[i]], ev[neg[i]].conj()
        else:
            ew[i], ev[i] = solve(i)
            done.add(i)
    return ew, ev


def DiagGHF(GFock, vcor_mat, mu, kmesh=None):
    """mfd.py:591-610 (kmesh given: DiagGHF_symm :612-641).  Only the lower triangle is referenced."""
    G = np.array(GFock, dtype=np.complex128, copy=True)
    nk, nso, _ = G.shape
    nao = nso // 2
    G[:, :nao, :nao] += vcor_mat[0]
    G[:, nao:, nao:] += vcor_mat[1]
    G[:, nao:, :nao] += vcor_mat[2].conj().T
    if mu is not None:
        G[:, range(nao), range(nao)] -= mu
        G[:, range(nao, nso), range(nao, nso)] += mu
    ew = np.empty((nk, nso))
    ev = np.empty((nk, nso, nso), dtype=np.complex128)
    solve = lambda k: la.eigh(G[k], lower=True)
    if kmesh is None:
        for k in range(nk):
            ew[k], ev[k] = solve(k)
        return ew, ev
    return _symm_fill(ew, ev, kmesh, solve)


def DiagBdG(Fock, vcor_mat, mu, kmesh=None):
    """mfd.py:429-449 (kmesh given: DiagBdGsymm :451-478)."""
    Fock = np.asarray(Fock)
    if Fock.ndim == 3:
        Fock = np.asarray((Fock, Fock))
    nk, n = Fock.shape[-3], Fock.shape[-1]
    ew = np.empty((nk, 2 * n))
    ev = np.empty((nk, 2 * n, 2 * n), dtype=np.complex128)

    def solve(i):
        t = np.empty((2 * n, 2 * n), dtype=np.complex128)
        t[:n, :n] = Fock[0, i] + vcor_mat[0] - mu * np.eye(n)
        t[n:, n:] = -Fock[1, i] - vcor_mat[1] + mu * np.eye(n)
        t[:n, n:] = vcor_mat[2]
        t[n:, :n] = vcor_mat[2].conj().T
        return la.eigh(t)

    if kmesh is None:
        for i in range(nk):
            ew[i], ev[i] = solve(i)
        return ew, ev
    return _symm_fill(ew, ev, kmesh, solve)


# ---------------------------------------------------------------------------------------------
# a7: alpha / beta bath matching
# ---------------------------------------------------------------------------------------------

def basisMatching(basis):
    """dmet/HubPhSymm.py:37-48: S = A^T B = u g vt;  A' = A u, B' = B vt^T."""
    A, B = basis[0], basis[1]
    S = np.tensordot(A, B, axes=((0, 1), (0, 1)))
    u, gamma, vt = la.svd(S)
    return np.asarray([np.tensordot(A, u, axes=(2, 0)), np.tensordot(B, vt, axes=(2, 1))]), gamma


# ---------------------------------------------------------------------------------------------
# a14: unit ERI -> embedding ERI container
# ---------------------------------------------------------------------------------------------

def init_H2(norb, eri_symmetry, spin_dim=None):
    """slater_helper.py:444-471."""
    sd = () if spin_dim is None else (spin_dim,)
    npair = norb * (norb + 1) // 2
    if eri_symmetry == 1:
        return np.zeros(sd + (norb,) * 4)
    if eri_symmetry == 4:
        return np.zeros(sd + (npair, npair))
    if eri_symmetry == 8:
        return np.zeros(sd + (npair * (npair + 1) // 2,))
    raise ValueError("unknown ERI symmetry: %s" % eri_symmetry)


def unit2emb(H2_unit, neo):
    """slater_helper.py:494-518 (in-core branch)."""
    sp = H2_unit.shape[0]
    sym = {5: 1, 3: 4, 2: 8}.get(H2_unit.ndim)
    if sym is None:
        raise ValueError
    out = init_H2(neo, sym, spin_dim=sp)
    out[tuple(map(slice, H2_unit.shape))] = H2_unit
    return out


def reorder_spin_blocks(H2):
    """routine/slater.py:461-462: (aa, ab, bb) -> (aa, bb, ab)."""
    return H2[[0, 2, 1]] if H2.shape[0] == 3 else H2


# ---------------------------------------------------------------------------------------------
# a8: Nambu-matrix bookkeeping (bcs_helper.py:14-70)
# ---------------------------------------------------------------------------------------------

def extractRdm(GRho):
    n = GRho.shape[0] // 2
    assert 2 * n == GRho.shape[0]
    return GRho[:n, :n].copy(), np.eye(n) - GRho[n:, n:], GRho[n:, :n].copy()


def extractH1(GFock):
    n = GFock.shape[0] // 2
    assert 2 * n == GFock.shape[0]
    return GFock[:n, :n].copy(), -GFock[n:, n:], GFock[n:, :n].copy()


def combineRdm(rhoA, rhoB, kappaAB):
    n = rhoA.shape[0]
    return np.block([[rhoA, -kappaAB], [-kappaAB.T, np.eye(n) - rhoB]])


def swapSpin(GRho):
    rhoA, rhoB, kappaBA = extractRdm(GRho)
    n = rhoA.shape[0]
    return np.block([[rhoB, -kappaBA], [-kappaBA.T, np.eye(n) - rhoA]])


def basisToCanonical(basis):
    assert basis.shape[0] == 2
    shape = list(basis.shape[1:])
    nb, ns = shape[-1], shape[-2] // 2
    shape[-1] *= 2
    out = np.empty(tuple(shape))
    out[..., :nb] = basis[0]
    out[..., :ns, nb:] = basis[1, ..., ns:, :]
    out[..., ns:, nb:] = basis[1, ..., :ns, :]
    return out


def basisToSpin(basis):
    shape = [2] + list(basis.shape)
    shape[-1] //= 2
    nb, ns = shape[-1], shape[-2] // 2
    out = np.empty(tuple(shape))
    out[0] = basis[..., :nb]
    out[1, ..., :ns, :] = basis[..., ns:, nb:]
    out[1, ..., ns:, :] = basis[..., :ns, nb:]
    return out


def separate_basis(basis):
    """bcs_helper.py:176-180 -> VA, VB, UA, UB."""
    n = basis.shape[2] // 2
    return basis[0, :, :n], basis[1, :, :n], basis[1, :, n:], basis[0, :, n:]


# ---------------------------------------------------------------------------------------------
# a8: one-body folds into the embedding (bcs_helper.py:182-207, 248-268, 346-388)
# ---------------------------------------------------------------------------------------------

def lattice_transpose(kmesh, A):
    """system/lattice.py transpose: A^T[R] = A[-R]^T."""
    neg = _neg_table(kmesh)
    return np.asarray([A[neg[i]].T for i in range(A.shape[0])])


def contract_trans_inv(basisL, basisR, kmesh, H):
    ca = CellArith(kmesh)
    res = np.zeros((basisL.shape[2], basisR.shape[2]))
    for i, j in it.product(range(ca.ncells), repeat=2):
        res += basisL[i].T @ H[ca.subtract(i, j)] @ basisR[j]
    return res


def contract_local(basisL, basisR, kmesh, H):
    return sum(basisL[i].T @ H @ basisR[i] for i in range(basisL.shape[0]))


def contract_imp(basisL, basisR, kmesh, H):
    return basisL[0].T @ H @ basisR[0]


def contract_imp_env(basisL, basisR, kmesh, H):
    ca = CellArith(kmesh)
    r1 = sum(basisL[0].T @ H[i] @ basisR[i] for i in range(ca.ncells))
    r2 = sum(basisL[i].T @ H[ca.subtract(i, 0)] @ basisR[0] for i in range(ca.ncells))
    return 0.5 * (r1 + r2)


def _split_H(H, nd_single):
    if H.ndim == nd_single:
        return H, H, np.zeros_like(H)
    if H.shape[0] == 2:
        return H[0], H[1], np.zeros_like(H[0])
    if H.shape[0] == 3:
        return H[0], H[1], H[2]
    raise ValueError("unknown shape of H: %s" % (H.shape,))


def _nambu_fold(basis, kmesh, H, nd_single, contract, tr):
    VA, VB, UA, UB = separate_basis(basis)
    HA, HB, D = _split_H(H, nd_single)
    DT = tr(D)
    c = lambda L, h, R: contract(L, R, kmesh, h)
    rA = c(VA, HA, VA) - c(UB, HB, UB) + c(VA, D, UB) + c(UB, DT, VA)
    rB = c(VB, HB, VB) - c(UA, HA, UA) - c(VB, DT, UA) - c(UA, D, VB)
    rD = c(VA, HA, UA) - c(UB, HB, VB) + c(VA, D, VB) + c(UB, DT, UA)
    E0 = np.trace(c(UA, HA, UA) + c(UB, HB, UB) + c(UA, D, VB) + c(VB, DT, UA))
    return np.asarray((rA, rB)), rD, E0


def transform_trans_inv(basis, kmesh, H):
    return _nambu_fold(basis, kmesh, H, 3, contract_trans_inv, lambda D: lattice_transpose(kmesh, D))


def transform_local(basis, kmesh, H):
    return _nambu_fold(basis, kmesh, H, 2, contract_local, lambda D: D.T)


def transform_imp(basis, kmesh, H):
    return _nambu_fold(basis, kmesh, H, 2, contract_imp, lambda D: D.T)


def transform_imp_env(basis, kmesh, H):
    return _nambu_fold(basis, kmesh, H, 3, contract_imp_env, lambda D: lattice_transpose(kmesh, D))


# ---------------------------------------------------------------------------------------------
# a8: derivative of the embedded potential w.r.t. the local vcor entries (bcs_helper.py:270-430)
# ---------------------------------------------------------------------------------------------

def contract_local_grad(basisL, basisR):
    """sum_c L[c,i,p] R[c,j,q] -> (i, j, p, q)."""
    return np.einsum("cip,cjq->ijpq", basisL, basisR)


def contract_local_grad_DT(basisL, basisR):
    """sum_c L[c,j,p] R[c,i,q] -> (i, j, p, q)."""
    return np.einsum("cjp,ciq->ijpq", basisL, basisR)


def transform_local_grad(basis):
    VA, VB, UA, UB = separate_basis(basis)
    g, gT = contract_local_grad, contract_local_grad_DT
    A = (np.asarray([g(VA, VA), -g(UA, UA)]), g(VA, UA), None)
    B = (np.asarray([-g(UB, UB), g(VB, VB)]), -g(UB, VB), None)
    D = (np.asarray([g(VA, UB) + gT(UB, VA), -gT(VB, UA) - g(UA, VB)]), g(VA, VB) + gT(UB, UA), None)
    return A, B, D


def get_dV_dparam(basis, nparam):
    def sym_triu(a):
        a = a + a.transpose((1, 0, 2, 3))
        a[np.arange(a.shape[0]), np.arange(a.shape[1])] *= 0.5
        return a[np.triu_indices(a.shape[0])]

    nb = basis.shape[-1]
    rA, rB, rD = transform_local_grad(basis)
    flat = lambda x: x.reshape((-1,) + x.shape[-2:])
    cols = []
    for which in range(3):      # -> H_A, H_B, D blocks of the embedded Nambu matrix
        pick = (lambda r: r[0][0]) if which == 0 else (lambda r: r[0][1]) if which == 1 else (lambda r: r[1])
        cols.append(np.concatenate([sym_triu(pick(rA)), sym_triu(pick(rB)), flat(pick(rD))], axis=0))
    dA, dB, dD = cols
    out = np.empty((nparam, 2 * nb, 2 * nb))
    for ip in range(nparam):
        out[ip, :nb, :nb] = dA[ip]
        out[ip, nb:, nb:] = -dB[ip]
        out[ip, :nb, nb:] = dD[ip]
        out[ip, nb:, :nb] = dD[ip].T
    return out


# ---------------------------------------------------------------------------------------------
# a8: BCS embedding basis (routine/bcs.py:25-135)
# ---------------------------------------------------------------------------------------------

def embBasis_proj(GRho, nscsites, val_idx, localize_bath=None):
    """bcs.py:78-104 (the branch without "sites").  Returns basis, sigma, B.  `localize_bath='scdm'`: the bath orbitals rotated by
    routine/localizer.localize_bath between the SVD and the particle-weight sorting (bcs.py:84-88)."""
    ncells = GRho.shape[0]
    n, nval = nscsites, len(val_idx)
    basis = np.zeros((2, ncells, 2 * n, n + nval))
    cols = list(val_idx) + [i + n for i in val_idx]
    A = GRho[1:].reshape(n * (ncells - 1) * 2, 2 * n)[:, cols]
    u, sigma, vt = la.svd(A, full_matrices=False)
    B = u.reshape((ncells - 1, 2 * n, 2 * nval))
    if localize_bath is not None:
        assert localize_bath == "scdm"
        from oracle.restate import localize_bath_scdm
        B = localize_bath_scdm(B)
    basis[0, 0, :n, :n] = np.eye(n)
    basis[1, 0, :n, :n] = np.eye(n)
    w = np.diag(np.tensordot(B[:, :n], B[:, :n], axes=((0, 1), (0, 1))))
    order = np.argsort(w, kind="mergesort")[::-1]
    oA, oB = order[:nval], order[nval:]
    basis[0, 1:, :, n:] = B[:, :, oA]
    basis[1, 1:, :n, n:] = B[:, n:, oB]
    basis[1, 1:, n:, n:] = B[:, :n, oB]
    return basis, sigma, B, w


def MatSqrt(M):
    """routine/slater.py:38-50."""
    ew, ev = la.eigh(M)
    if ew[0] < 0:
        ew = ew + 1e-11
    assert (ew >= 0).all()
    return ev @ np.diag(np.sqrt(ew))


def orthonormalizeBasis(b):
    """routine/slater.py:59-78."""
    ov = np.tensordot(b, b, axes=((0, 1), (0, 1)))
    if np.allclose(ov - np.diag(np.diag(ov)), 0.0):
        return np.tensordot(b, np.diag(1.0 / np.sqrt(np.diag(ov))), axes=(2, 0))
    ew, ev = la.eigh(ov)
    ew, ev = ew[::-1], ev[:, ::-1]
    return np.tensordot(np.tensordot(b, ev, axes=(2, 0)), np.diag(ew ** -0.5), axes=(2, 0))


def embBasis_phsymm(GRho, nscsites):
    """bcs.py:109-135: quasiparticle bath; particle part -> alpha, hole part -> beta."""
    ncells, n = GRho.shape[0], nscsites
    basis = np.empty((2, ncells, 2 * n, 2 * n))
    A1 = MatSqrt(GRho[0])
    basis[0] = orthonormalizeBasis(np.dot(GRho, la.inv(A1.T)))
    Gh = -GRho
    Gh[0] += np.eye(2 * n)
    A2 = MatSqrt(Gh[0])
    BA2 = orthonormalizeBasis(np.dot(Gh, la.inv(A2.T)))
    basis[1, :, :n], basis[1, :, n:] = BA2[:, n:], BA2[:, :n]
    return basis


# ---------------------------------------------------------------------------------------------
# BCS embedding Hamiltonian of model lattices (routine/bcs.py:137-318); golden G28
# ---------------------------------------------------------------------------------------------

def bcs_embHam(kmesh, basis, hcore_R, LatH2, vcor_mat, mu, ImpJK=None, fitting=False):
    """The branch the reference implements: local basis, bare bath, hcore as the embedding Hamiltonian, 'local' lattice ERI.
    Returns (H1 {"cd", "cc"}, H0, ccdd), (H1energy, H0energy)."""
    n, nb = vcor_mat.shape[-1], basis.shape[-1]
    for s in range(2):
        assert np.abs(basis[s, 0, :n, :n] - np.eye(n)).max() < 1e-10                # bcs.py:200-202
    ccdd = np.zeros((3,) + (nb,) * 4)
    ccdd[:, :n, :n, :n, :n] = LatH2
    cd, cc, H0 = transform_trans_inv(basis, kmesh, hcore_R)                          # transform_trans_inv_sparse, thr 1e-7
    v = np.array(vcor_mat, copy=True)
    v[0] -= mu * np.eye(n)
    v[1] -= mu * np.eye(n)
    terms = [(+1, transform_local(basis, kmesh, v))]
    if not fitting:
        terms.append((-1, transform_imp(basis, kmesh, vcor_mat)))
    if ImpJK is not None:
        terms.append((-1, transform_imp(basis, kmesh, ImpJK)))
    cd, cc = np.array(cd), np.array(cc)
    for sign, (tcd, tcc, t0) in terms:
        cd, cc, H0 = cd + sign * np.asarray(tcd), cc + sign * tcc, H0 + sign * t0
    ecd, ecc, e0 = transform_imp_env(basis, kmesh, hcore_R)
    return ({"cd": cd, "cc": cc[None]}, H0, ccdd), ({"cd": np.asarray(ecd), "cc": np.asarray(ecc)[None]}, e0)


# ---------------------------------------------------------------------------------------------
# BCS vcor fit in the embedding space (routine/bcs.py:356-530); golden G30
# ---------------------------------------------------------------------------------------------

def nambu(A, B, D):
    """[[A, D], [D^T, -B]] (bcs.py:389-392, 398-401)."""
    nb = A.shape[-1]
    M = np.empty((2 * nb, 2 * nb))
    M[:nb, :nb], M[nb:, nb:], M[:nb, nb:], M[nb:, :nb] = A, -B, D, D.T
    return M


def bcs_emb_fit(GRho, kmesh, basis, vcor, mu, beta, fock_R, mu0=0.0, fix_mu=True):
    """The objective of bcs.FitVcorEmb as an oracle.restate_fit.EmbFit on ONE Nambu block of dimension 2 nbasis: every entry
    fitted, nbasis levels filled at T = 0 (the reference's `ew < 0` on a particle-hole symmetric spectrum), Fermi function around
    the fixed mu0 at finite T, |dGRho| / sqrt(2); the constant part holds the fold of the Fock stripe and of -mu."""
    from oracle.restate_fit import EmbFit
    n, nb = basis.shape[2] // 2, basis.shape[-1]
    (HA, HB), HD, _ = transform_trans_inv(basis, kmesh, fock_R)
    shift = np.zeros((3, n, n))
    shift[0] = shift[1] = -mu * np.eye(n)
    (A0, B0), D0, _ = transform_local(basis, kmesh, shift)
    g = vcor.gradient()
    table = []
    for ip in range(vcor.length()):
        (dA, dB), dD, _ = transform_local(basis, kmesh, g[ip])
        table.append(nambu(dA, dB, dD))
    dim = 2 * nb
    tl = np.tril_indices(dim)
    fit = EmbFit.__new__(EmbFit)
    fit.C_act, fit.spin, fit.nb, fit.norm = None, 1, dim, np.sqrt(2.0)
    fit.beta, fit.nelec, fit.mu0, fit.fix_mu, fit.tol_deg = beta, nb, mu0, (fix_mu if beta < np.inf else False), 1e-3
    fit.vcor, fit.remove_diag_grad = vcor, False
    idx = list(range(dim))
    fit.fit_idx = idx
    fit.imp_mesh, fit.det_mesh = np.ix_(idx, idx), ([], [])
    fit.imp_fill, fit.det_fill = (slice(dim), slice(dim)), (range(dim, dim), range(dim, dim))
    fit.embH1 = (nambu(np.asarray(HA), np.asarray(HB), np.asarray(HD)) + nambu(A0, B0, D0))[None]
    fit.ovlp = np.eye(dim)[None]
    fit.dV = np.asarray([m[tl] for m in table])[:, None, :]
    fit.tril = tl
    fit.target = np.array(GRho, copy=True)[None]
    return fit


# ---------------------------------------------------------------------------------------------
# Hartree-Fock-Bogoliubov lattice mean field (routine/mfd.py:480-590) and the lattice stage of the BCS fit (bcs.py:319-343, 532-562);
# golden G31
# ---------------------------------------------------------------------------------------------

def HFB(kmesh, Fock_k, Fock_R, H1_R, vcor_mat, mu, H0=0.0, beta=np.inf, fix_mu=False, symm=False):
    """Returns GRhoT, n, E, {"e", "rho_k", "gap", "homo", "lumo"}."""
    from oracle.restate import FFTtoT, fermi_smearing_occ, find_mu
    ew, ev = DiagBdG(Fock_k, vcor_mat, mu, kmesh=kmesh if symm else None)
    ew_sorted = np.sort(ew, axis=None, kind='mergesort')
    mu_ref = 0.0
    if beta == np.inf:
        occ = (ew < mu_ref).astype(float)
    else:
        if not fix_mu:
            # find_mu_by_density(0.5, ...) = find_mu(0.5 norb, ..., tol = 1e-12 norb)  (ftsystem.py:107-113)
            mu_ref = find_mu(0.5 * ew_sorted.size, ew_sorted, beta, mu0=mu_ref, tol=1e-12 * ew_sorted.size)
        occ = fermi_smearing_occ(mu_ref, ew, beta)
    GRho = np.einsum('kpm,km,kqm->kpq', ev, occ, ev.conj())
    GRhoT = FFTtoT(GRho, kmesh)
    n = Fock_R.shape[-1]
    FT = Fock_R if Fock_R.ndim == 4 else np.asarray((Fock_R, Fock_R))
    HT = H1_R if H1_R.ndim == 4 else np.asarray((H1_R, H1_R))
    rA, rB, kBA = GRhoT[:, :n, :n].copy(), np.eye(n) - GRhoT[:, n:, n:], GRhoT[:, n:, :n].copy()
    rB[1:] -= np.eye(n)
    npart = np.trace(rA[0]) + np.trace(rB[0])
    E = 0.5 * np.sum((FT[0] + HT[0]) * rA + (FT[1] + HT[1]) * rB) + H0
    E += 0.5 * np.sum(vcor_mat[0] * rA[0] + vcor_mat[1] * rB[0] + 2 * vcor_mat[2] * kBA[0])
    homo = ew_sorted[max(np.searchsorted(ew_sorted, mu_ref, side='right') - 1, 0)]
    lumo = ew_sorted[min(np.searchsorted(ew_sorted, mu_ref, side='left'), len(ew_sorted) - 1)]
    return GRhoT, npart, E, {"e": ew, "rho_k": GRho, "gap": lumo - homo, "homo": homo, "lumo": lumo}


def foldRho_bcs(GRho, kmesh, basis):
    """bcs.py:319-343: sum_{ij} C_i^T GRho[i - j] C_j with the canonical basis C (ncells, 2 n, 2 nbasis)."""
    from oracle.restate import cartesian_prod
    cells = [tuple(c) for c in cartesian_prod([np.arange(x) for x in kmesh])]
    where = dict((tuple(c), i) for i, c in enumerate(cells))
    n, nb = basis.shape[2] // 2, basis.shape[-1]
    C = np.empty((len(cells), 2 * n, 2 * nb))
    C[:, :, :nb] = basis[0]
    C[:, :n, nb:], C[:, n:, nb:] = basis[1, :, n:], basis[1, :, :n]
    res = np.zeros((2 * nb, 2 * nb))
    size = np.asarray(kmesh)
    for i, ci in enumerate(cells):
        for j, cj in enumerate(cells):
            res += C[i].T @ GRho[where[tuple((np.asarray(ci) - np.asarray(cj)) % size)]] @ C[j]
    return res


def bcs_full_errfunc(GRho_target, kmesh, basis, vcor, mu, beta, Fock_k, Fock_R):
    """errfunc of bcs.FitVcorFull without a reference point (bcs.py:545-554)."""
    def errfunc(param):
        vcor.update(param)
        GRhoT = HFB(kmesh, Fock_k, Fock_R, Fock_R, vcor.get(), mu, beta=beta)[0]
        return la.norm(foldRho_bcs(GRhoT, kmesh, basis) - GRho_target) / np.sqrt(2.0)
    return errfunc
