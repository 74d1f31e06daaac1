"""
oracle/restate_gso.py -- CPU restatement (numpy) of the generalised-spin-orbital (GSO, "spinless") twins of the path,
SURVEY.md section 8(f) rank 4:

  spinless._get_emb_basis_svd     routine/spinless.py:58-163     Schmidt bath of the generalised density matrix
  spinless._get_emb_basis_eig     routine/spinless.py:166-275    ... from the eigenvectors of the env-env block
  spinless._get_emb_basis_ph      routine/spinless.py:351-423    ... particle / hole projections, canonical orthogonalisation
                                                                 (both pinned by tests/golden/G18_branches.npz, gen_G18)
  spinless.get_emb_basis_opt      routine/spinless.py:274-349    bath_opt: integer electron number of the embedding space
                                                                 (pinned by tests/golden/G19_bath_opt.npz, gen_G19)
  eri_transform.get_emb_eri_gso   basis_transform/eri_transform.py:1104-1250
  _Lij_s4_to_eri_gso              basis_transform/eri_transform.py:1252-1310  (aaaa + bbbb - aabb - bbaa)

  spinless.get_emb_Ham            routine/spinless.py:431-725 over spinless_helper.py:261-440 (pinned by tests/golden/G27_gso_embham.npz)

TEST INFRASTRUCTURE ONLY.  Pinned against tests/golden/G12_gso.npz (oracle/gen_golden.py gen_G12: the reference's
own functions under oracle/shim.py; the ERI driver over the restated PySCF primitives of oracle/shim.py).
"""
import numpy as np
import scipy.linalg as la

from oracle.restate import (KPT_DIFF_TOL, add_spin_dim, eri_restore, get_basis_k, get_phase_R2k, get_weights_t_reversal,
                            kpt_member, max_abs, multiply_basis, pack_tril, transform_ao_to_emb, vec_lowdin)


def get_emb_basis_gso(rdm1, nlo, val_idx, imp_idx, valence_bath=True, tol_bath=1e-9, nbath=None, localize_bath=None):
    """routine/spinless.py:58-163 (orth = True); `localize_bath='scdm'` rotates the orthonormalised bath before the particle-hole
    sorting (:139-146)."""
    rdm1 = np.asarray(rdm1)
    ncells, nso, _ = rdm1.shape
    assert nso == 2 * nlo
    val2 = list(val_idx) + [i + nlo for i in val_idx]
    imp2 = list(imp_idx) + [i + nlo for i in imp_idx]
    bath_cols = val2 if valence_bath else imp2
    env, virt, alpha = [], [], []
    for R in range(ncells):
        for s in range(2):
            for i in range(nlo):
                idx = R * nso + s * nlo + i
                if idx not in bath_cols:
                    env.append(idx)
                    virt.append(idx in imp2)
                    alpha.append(s == 0)
    nimp = len(imp2)
    A = rdm1.reshape(ncells * nso, nso)[env][:, bath_cols]
    u, sigma, vt = la.svd(A, full_matrices=False)
    if nbath is None:
        nbath = int((sigma >= tol_bath).sum())
    assert nbath % 2 == 0
    B = u[:, :nbath].copy()
    B[np.asarray(virt)] = 0.0
    B = vec_lowdin(B)
    if localize_bath is not None:
        assert localize_bath == "scdm"
        from oracle.restate import localize_bath_scdm
        B = localize_bath_scdm(B)
    w = np.einsum("ai,ai->i", B[np.asarray(alpha)], B[np.asarray(alpha)])
    order = np.argsort(w, kind="mergesort")[::-1]
    basis = np.zeros((ncells * nso, nimp + nbath))
    basis[imp2, :nimp] = np.eye(nimp)
    basis[env, nimp:] = B[:, order]
    return basis.reshape(ncells, nso, nimp + nbath), sigma, w


def _gso_index_sets(ncells, nlo, val_idx, imp_idx, valence_bath):
    """The index bookkeeping shared by the three bath flavours (routine/spinless.py:86-106, 196-216, 376-398)."""
    nso = 2 * nlo
    val2 = list(val_idx) + [i + nlo for i in val_idx]
    imp2 = list(imp_idx) + [i + nlo for i in imp_idx]
    bath_cols = val2 if valence_bath else imp2
    env, virt, alpha, virt_idx = [], [], [], []
    for R in range(ncells):
        for s in range(2):
            for i in range(nlo):
                idx = R * nso + s * nlo + i
                if idx not in bath_cols:
                    env.append(idx)
                    virt.append(idx in imp2)
                    if idx in imp2:
                        virt_idx.append(idx)
                    alpha.append(s == 0)
    return imp2, bath_cols, env, np.asarray(virt, dtype=bool), np.asarray(alpha, dtype=bool), virt_idx


def get_emb_basis_gso_eig(kmesh, rdm1, nlo, val_idx, imp_idx, valence_bath=True, tol_bath=1e-9, localize_bath=None):
    """routine/spinless.py:166-275 (kind = 'eig', orth = True): eigenvectors of the env-env block of the generalised density
    matrix whose eigenvalues are neither 0 nor 1 (lattice.expand restated by CellArith.expand)."""
    rdm1 = np.asarray(rdm1)
    ncells, nso, _ = rdm1.shape
    imp2, bath_cols, env, virt, alpha, _ = _gso_index_sets(ncells, nlo, val_idx, imp_idx, valence_bath)
    nimp = len(imp2)
    from oracle.restate import CellArith
    big = CellArith(kmesh).expand(rdm1[None])[0]                                                       # lattice.expand
    ew, ev = la.eigh(big[env][:, env])
    keep = [i for i, e in enumerate(ew) if abs(e) > tol_bath and abs(1 - e) > tol_bath]
    B = ev[:, keep].copy()
    nbath = B.shape[-1]
    assert nbath % 2 == 0
    B[virt] = 0.0
    B = vec_lowdin(B)
    if localize_bath is not None:
        assert localize_bath == "scdm"
        from oracle.restate import localize_bath_scdm
        B = localize_bath_scdm(B)
    w = np.einsum("ai,ai->i", B[alpha], B[alpha])
    order = np.argsort(w, kind="mergesort")[::-1]
    basis = np.zeros((ncells * nso, nimp + nbath))
    basis[imp2, :nimp] = np.eye(nimp)
    basis[env, nimp:] = B[:, order]
    return basis.reshape(ncells, nso, nimp + nbath), ew[keep]


def get_emb_basis_gso_ph(rdm1, nlo, val_idx, imp_idx, valence_bath=True, tol_bath=1e-9):
    """routine/spinless.py:351-423 (kind = 'ph'): particle and hole projections of the bath columns plus the local virtuals,
    canonically orthogonalised (lo/lowdin.py:138-148: eigenvectors of the overlap above tol / sqrt(eigenvalue))."""
    rdm1 = np.asarray(rdm1)
    ncells, nso, _ = rdm1.shape
    imp2, bath_cols, env, virt, alpha, virt_idx = _gso_index_sets(ncells, nlo, val_idx, imp_idx, valence_bath)
    bath_p = rdm1.reshape(ncells * nso, nso)[:, bath_cols]
    rdm1_h = -rdm1.copy()
    rdm1_h[0, range(nso), range(nso)] += 1.0
    bath_h = rdm1_h.reshape(ncells * nso, nso)[:, bath_cols]
    nval = len(bath_cols) * 2
    nvirt = len(virt_idx)
    nbasis = nval + nvirt
    basis = np.zeros((ncells * nso, nbasis))
    basis[virt_idx, range(nbasis - nvirt, nbasis)] = 1.0
    basis[:, :nval // 2] = bath_p
    basis[:, nval // 2:nval] = bath_h
    e, v = la.eigh(basis.conj().T @ basis)
    idx = e > tol_bath
    out = basis @ (v[:, idx] / np.sqrt(e[idx]))
    return out.reshape(ncells, nso, -1), e


def get_emb_basis_opt(kmesh, rdm1_R, basis, keep_imp_identity=False, nimp=None, tol=1e-6):
    """routine/spinless.py:274-349: shift mu (scipy brentq on [-1, 0] or [0, 1], xtol = rtol = tol) such that the span of the top
    nemb eigenvectors of  B B^T - mu D  (D = lattice.expand(rdm1_R)) holds an integer number of electrons.  The electron number
    is the trace of the folded density matrix (foldRho_k = the full-space  tr(E^T D E), :288).  Returns (basis, mu, nelec)."""
    from scipy import optimize as opt
    from oracle.restate import CellArith
    rdm1_R = np.asarray(rdm1_R).real
    basis = np.asarray(basis)
    nemb = basis.shape[-1]
    B = basis.reshape(-1, nemb)
    D = CellArith(kmesh).expand(rdm1_R[None])[0]
    count = lambda E: float(np.einsum("ai,ab,bi->", E, D, E))
    nelec = count(B)
    target = np.round(nelec)
    if abs(nelec - target) < tol:
        return basis, None, nelec
    lval, rval = (-1.0, 0.0) if nelec < target else (1.0, 0.0)
    P = B @ B.conj().T

    def top(mu):
        ew, ev = la.eigh(P - mu * D)
        return ev[:, -nemb:]

    res = opt.brentq(lambda mu: count(top(mu)) - target, lval, rval, xtol=tol, rtol=tol, maxiter=1000, full_output=True, disp=False)
    mu = res[0]
    ev = top(mu)
    if keep_imp_identity:                                                   # :326-341
        BR = B[:, :nimp]
        for i in range(ev.shape[-1]):
            v = ev[:, i]
            v = v - BR @ (v @ BR)
            nv = la.norm(v)
            if nv > tol and BR.shape[-1] < nemb:
                BR = np.hstack((BR, (v / nv)[:, None]))
        out = BR.reshape(basis.shape)
    else:
        out = ev.reshape(basis.shape)
    return out, mu, count(out.reshape(-1, nemb))


def Lij_s4_to_eri_gso(Lij_s4, eri, weight=1, t_reversal_symm=False):
    """eri_transform.py:1252-1283."""
    if t_reversal_symm:
        parts = [Lij_s4.real] if weight == 1 else [Lij_s4.real, Lij_s4.imag]
        for P in parts:
            a, b = P
            eri[0] += weight * (a.T @ a + b.T @ b - a.T @ b - b.T @ a)
    else:
        a, b = Lij_s4
        ab = -(a.conj().T @ b)
        eri[0] += a.conj().T @ a + b.conj().T @ b + ab + ab.conj().T


def get_emb_eri_gso(kmesh, kpts_scaled, get_block, naux, nao, C_ao_lo, basis, symmetry=4, unit_eri=False,
                    t_reversal_symm=True, kconserv_tol=KPT_DIFF_TOL):
    """eri_transform.py:1104-1250, in-core branch."""
    kscaled = np.asarray(kpts_scaled, dtype=float)
    nk = len(kscaled)
    C_ao_lo = add_spin_dim(np.asarray(C_ao_lo), 2)
    if unit_eri:
        C_ao_emb = C_ao_lo / (nk ** 0.75)
    else:
        nlo = basis.shape[1] // 2
        bk = get_basis_k(basis[None], get_phase_R2k(kmesh, kscaled))[0]
        C_ao_emb = multiply_basis(C_ao_lo, np.asarray((bk[:, :nlo], bk[:, nlo:]))) / (nk ** 0.75)
    spin, _, _, nemb = C_ao_emb.shape
    npair = nemb * (nemb + 1) // 2
    weights = get_weights_t_reversal(kscaled) if t_reversal_symm else np.ones(nk, dtype=int)
    eri = np.zeros((1, npair, npair), dtype=float if t_reversal_symm else np.complex128)
    for kL in range(nk):
        if weights[kL] <= 0:
            continue
        Lij_s4 = np.zeros((spin, naux, npair), dtype=np.complex128)
        visited = np.zeros(nk, dtype=bool)
        for i in range(nk):
            if visited[i]:
                continue
            visited[i] = True
            for j in range(nk):
                kc = -kscaled[i] + kscaled[j] + kscaled[kL]
                if max_abs(np.round(kc) - kc) > kconserv_tol:
                    continue
                if t_reversal_symm:
                    jm = kpt_member(-kscaled[j], kscaled)[0]
                Lpq = np.asarray(get_block(i, j), dtype=np.complex128).reshape(naux, nao * nao)
                Lij = transform_ao_to_emb(Lpq, C_ao_emb, i, j)
                if t_reversal_symm and (not visited[jm]):
                    Lij = Lij + Lij.transpose(0, 1, 3, 2)
                Lij_s4 += pack_tril(Lij)
                if t_reversal_symm:
                    visited[jm] = True
        Lij_s4_to_eri_gso(Lij_s4, eri, weight=weights[kL], t_reversal_symm=t_reversal_symm)
    return eri_restore(eri.real, symmetry, nemb)


# ---------------------------------------------------------------------------------------------
# GSO embedding Hamiltonian (routine/spinless.py:431-725 over routine/spinless_helper.py:261-440); golden G27
# ---------------------------------------------------------------------------------------------

def spin_orbital_matrix(H):
    """(2 | 3, ..., n, n) blocks (aa, bb[, ab]) -> (..., 2n, 2n) with the ba block the conjugate transpose of ab: the quadratic
    forms of spinless_helper.py:349-440 are B^H M B with the stacked basis B = [B_a; B_b]."""
    H = np.asarray(H)
    assert H.shape[0] in (2, 3)
    n = H.shape[-1]
    M = np.zeros(H.shape[1:-2] + (2 * n, 2 * n), dtype=H.dtype)
    M[..., :n, :n], M[..., n:, n:] = H[0], H[1]
    if H.shape[0] == 3:
        M[..., :n, n:] = H[2]
        M[..., n:, :n] = np.swapaxes(H[2].conj(), -1, -2)
    return M


def transform_trans_inv_k_gso(basis_k, H_k):
    """spinless_helper.py:349-381: (1/nk) Re sum_k B_k^H M_k B_k."""
    M = spin_orbital_matrix(H_k)
    return np.einsum('kpa,kpq,kqb->ab', basis_k.conj(), M, basis_k).real / basis_k.shape[0]


def transform_local_gso(basis, H):
    """spinless_helper.py:383-409: every cell sees the same block matrix."""
    return np.einsum('Rpa,pq,Rqb->ab', basis, spin_orbital_matrix(H), basis)


def transform_imp_gso(basis, H):
    """spinless_helper.py:411-436: cell 0 only."""
    return transform_local_gso(basis[:1], H)


def pair_rows(idx, neo):
    """Positions, in the lower-triangle pair list of neo orbitals, of the pairs drawn from `idx` (spinless_helper.py:261-286)."""
    r, c = np.tril_indices(neo)
    return np.nonzero(np.isin(r, idx) & np.isin(c, idx))[0]


def unit2emb_gso(H2_unit, neo):
    """spinless_helper.py:288-313: (aa, bb, ab) unit ERI into the alpha-alpha, beta-beta, alpha-beta and beta-alpha pair blocks of
    the 4-fold embedding ERI, alpha = the first nao embedding orbitals, beta the next nao."""
    nao = int(np.sqrt(H2_unit.shape[-1] * 2))
    pa, pb = pair_rows(np.arange(nao), neo), pair_rows(np.arange(nao, 2 * nao), neo)
    npair = neo * (neo + 1) // 2
    out = np.zeros((npair, npair))
    out[np.ix_(pa, pa)], out[np.ix_(pb, pb)] = H2_unit[0], H2_unit[1]
    out[np.ix_(pa, pb)], out[np.ix_(pb, pa)] = H2_unit[2], H2_unit[2].T
    return out


def transform_eri_local_gso(basis, H2_unit):
    """spinless_helper.py:319-347: per cell the aa, bb and ab (+ transposed) four-index transforms, 4-fold packed."""
    ncells, nso, neo = basis.shape
    nao = nso // 2
    tl = np.tril_indices(nao)
    full = []
    for blk in H2_unit:
        f = np.zeros((nao, nao, nao, nao))
        sq = np.zeros((nao, nao, blk.shape[1]))
        sq[tl[0], tl[1]] = blk
        sq[tl[1], tl[0]] = blk
        f[:, :, tl[0], tl[1]] = sq
        f[:, :, tl[1], tl[0]] = sq
        full.append(f)
    t = np.tril_indices(neo)
    out = np.zeros((len(t[0]), len(t[0])))
    for R in range(ncells):
        a, b = basis[R, :nao], basis[R, nao:]
        for f, (l, r) in ((full[0], (a, a)), (full[1], (b, b))):
            out += np.einsum('ijkl,ip,jq,kr,ls->pqrs', f, l, l, r, r, optimize=True)[t[0], t[1]][:, t[0], t[1]]
        ab = np.einsum('ijkl,ip,jq,kr,ls->pqrs', full[2], a, a, b, b, optimize=True)[t[0], t[1]][:, t[0], t[1]]
        out += ab + ab.T
    return out


def gso_embHam1e(kmesh, basis, H2_emb, hcore_k, fock_k, ovlp_k, rdm1_k, vcor_mat, mu, int_bath=True, add_vcor=False, JK_imp=None,
                 use_hcore_as_emb_ham=False, fitting=False, hcore_add=None, hcore_custom=None):
    """spinless.py:560-725, Hartree-Fock branches.  `fock_k` is what the reference folds in the branch taken (fock_hf_lo_k of an
    ab-initio lattice with an interacting bath, the lattice Fock otherwise).  Returns H1 (1, neo, neo), ovlp_emb, JK_core."""
    from oracle.restate import R2k
    from oracle.restate_ham import get_veff
    basis_k = R2k(basis, kmesh)
    n = basis.shape[1] // 2
    hcore_emb = transform_trans_inv_k_gso(basis_k, hcore_k if hcore_custom is None else hcore_custom)
    ovlp_emb = transform_trans_inv_k_gso(basis_k, ovlp_k)
    local_jk = lambda: get_veff(np.einsum('kpa,kpq,kqb->ab', basis_k.conj(), rdm1_k, basis_k).real / basis_k.shape[0],
                                H2_emb, hyb=1.0, ghf=True)
    extra = 0.0 if hcore_add is None else transform_imp_gso(basis, hcore_add)
    if int_bath:
        H1 = transform_trans_inv_k_gso(basis_k, fock_k) + extra - local_jk()
        JK_core = H1 - hcore_emb
    else:
        add_vcor = True
        if use_hcore_as_emb_ham:
            H1, JK_core = hcore_emb + extra, None
        else:
            H1 = transform_trans_inv_k_gso(basis_k, fock_k) - local_jk() + extra
            JK_core = H1 - hcore_emb
    H1 = H1 + transform_local_gso(basis, np.asarray([-mu * np.eye(n), mu * np.eye(n)]))
    if add_vcor:
        H1 = H1 + transform_local_gso(basis, vcor_mat)
        if not fitting:
            H1 = H1 - transform_imp_gso(basis, vcor_mat)
        if JK_imp is not None:
            H1 = H1 - transform_imp_gso(basis, JK_imp)
    return H1[None], ovlp_emb, JK_core


# ---------------------------------------------------------------------------------------------
# GSO vcor fit in the embedding space (routine/spinless.py:1090-1430); golden G29
# ---------------------------------------------------------------------------------------------

def get_dV_dparam_gso(vcor, basis, compact=True):
    """spinless.py:1090-1127: transform_local of every parameter's gradient blocks (aa, bb, ab)."""
    g = vcor.gradient()
    nb = basis.shape[-1]
    tl = np.tril_indices(nb)
    full = np.asarray([transform_local_gso(basis, g[ip]) for ip in range(vcor.length())])
    return full[:, tl[0], tl[1]] if compact else full


def gso_emb_fit(rho, kmesh, basis, vcor, mu, beta, fock_k, ovlp_k, nimp, nelec=None, imp_fit=False, det=False, mu0=None, fix_mu=False,
                tol_deg=1e-3):
    """The objective of spinless.FitVcorEmb as an oracle.restate_fit.EmbFit on one generalised block: GSO operators, half filling,
    fitted spatial indices doubled to alpha + beta, |drho| / sqrt(2)."""
    from oracle.restate import R2k
    from oracle.restate_fit import EmbFit
    nb, n = basis.shape[-1], basis.shape[1] // 2
    fit = EmbFit.__new__(EmbFit)
    fit.C_act, fit.spin, fit.nb, fit.norm = None, 1, nb, np.sqrt(2.0)
    fit.beta, fit.nelec, fit.mu0, fit.fix_mu, fit.tol_deg = beta, (nb // 2 if nelec is None else nelec), mu0, fix_mu, tol_deg
    fit.vcor, fit.remove_diag_grad = vcor, False
    doubled = list(range(nimp)) + [i + nimp for i in range(nimp)]
    imp_idx, det_idx = (doubled, []) if imp_fit else (([], doubled) if det else (list(range(nb)), []))
    fit.fit_idx = imp_idx + det_idx
    ni, nidx = len(imp_idx), len(fit.fit_idx)
    fit.imp_mesh, fit.det_mesh = np.ix_(imp_idx, imp_idx), (det_idx, det_idx)
    fit.imp_fill, fit.det_fill = (slice(ni), slice(ni)), (range(ni, nidx), range(ni, nidx))
    basis_k = R2k(basis, kmesh)
    H = transform_trans_inv_k_gso(basis_k, fock_k) + transform_local_gso(basis, np.asarray([-mu * np.eye(n), mu * np.eye(n)]))
    fit.embH1, fit.ovlp = H[None], transform_trans_inv_k_gso(basis_k, ovlp_k)[None]
    fit.dV = get_dV_dparam_gso(vcor, basis)[:, None, :]
    fit.tril = np.tril_indices(nb)
    fit.target = np.zeros((1, nidx, nidx))
    fit.target[0][fit.imp_fill] = rho[fit.imp_mesh]
    fit.target[0][fit.det_fill] = rho[fit.det_mesh]
    return fit


# ---------------------------------------------------------------------------------------------
# generalised Hartree-Fock lattice mean field (routine/mfd.py:735-858); golden G33
# ---------------------------------------------------------------------------------------------

def GHF(kmesh, H1_k, Fock_k, vcor_mat, mu, H0=0.0, filling=0.5, mu0=None, beta=np.inf, symm=True, fix_mu=False, nfrac=None, ph_trans=False,
        tol_deg=1e-6):
    """Returns GRhoT, n, E, result dict.  H1_k / Fock_k: (3, nk, n, n) triples, or with `ph_trans` any ((spin,) nk, n, n)."""
    from oracle.restate import FFTtoT, assignocc, check_nelec
    from oracle.restate_bcs import DiagGHF
    def ph(H):
        H = np.asarray(H)
        HA, HB = (H, H) if H.ndim == 3 else (H[0], H[min(1, H.shape[0] - 1)])
        HD = H[2] if (H.ndim == 4 and H.shape[0] == 3) else np.zeros_like(HA)
        return np.asarray([HA, -HB, HD]), np.einsum('kii->', HB).real / HA.shape[0]
    GH0 = H0
    if ph_trans:
        H1_k, a = ph(H1_k)
        Fock_k, b = ph(Fock_k)
        GH0 = 0.5 * (a + b) + H0
    GFock, GH1 = spin_orbital_matrix(Fock_k), spin_orbital_matrix(H1_k)
    nk = GFock.shape[0]
    ew, ev = DiagGHF(GFock, vcor_mat, mu, kmesh=kmesh if symm else None)
    GFock = GFock + spin_orbital_matrix(np.asarray(vcor_mat))[None]
    nelec = check_nelec(ew.size * filling, None)[0]
    ew_sorted = np.sort(ew, axis=None, kind='mergesort')
    ncore, nvirt = (0, 0) if nfrac is None else (nelec - nfrac, ew.size - (nelec + nfrac))
    if mu0 is None:
        mu0 = 0.5 * (ew_sorted[nelec - 1] + ew_sorted[nelec])
    occ, mu_quasi, nerr = assignocc(ew, nelec, beta, mu0, fix_mu=fix_mu, thr_deg=tol_deg, ncore=ncore, nvirt=nvirt)
    GRho = np.einsum('kpm,km,kqm->kpq', ev, occ, ev.conj())
    GRhoT = FFTtoT(GRho, kmesh)
    n = GRhoT.shape[-1] // 2
    rA, rB = GRhoT[:, :n, :n], np.eye(n) - GRhoT[:, n:, n:]
    npart = (np.trace(rA[0]) + np.trace(rB[0])).real
    E = (0.5 / nk) * np.einsum('kij,kji->', GFock + GH1, GRho).real + GH0
    homo = ew_sorted[max(np.searchsorted(ew_sorted, mu_quasi, side='right') - 1, 0)]
    lumo = ew_sorted[min(np.searchsorted(ew_sorted, mu_quasi, side='left'), len(ew_sorted) - 1)]
    return GRhoT, npart, E, {"e": ew, "rho_k": GRho, "mo_occ": occ, "gap": lumo - homo, "homo": homo, "lumo": lumo, "mu_quasi": mu_quasi,
                             "nerr": nerr}


# ---------------------------------------------------------------------------------------------
# lattice stage of the GSO fit (routine/spinless.py:1431-1769); golden G35
# ---------------------------------------------------------------------------------------------

class GsoFullFit(object):
    """errfunc / gradfunc_ft of spinless.FitVcorFull for an impurity-block or diagonal fit (the imp + bath form has no gradient in
    the reference): every k point's generalised Fock + spin-orbital vcor, half filling with the quasiparticle level searched from 0,
    cell-0 density on the doubled indices, |drho| / sqrt(2); `bogo_only` leaves the normal blocks out."""

    def __init__(self, rho, kmesh, vcor, mu, beta, fock_k, imp_idx=(), det_idx=(), fix_mu=False, bogo_only=False):
        from oracle.restate import check_nelec
        self.vcor, self.beta, self.fix_mu = vcor, beta, fix_mu
        n = fock_k.shape[-1]
        GF = spin_orbital_matrix(fock_k).astype(complex)
        GF[:, range(n), range(n)] -= mu
        GF[:, range(n, 2 * n), range(n, 2 * n)] += mu
        self.GFock, self.nk, self.nso = GF, GF.shape[0], 2 * n
        dbl = lambda idx: list(idx) + [i + n for i in idx]
        imp, det = dbl(imp_idx), dbl(det_idx)
        self.fit_idx = imp + det
        ni, nidx = len(imp), len(imp) + len(det)
        self.imp_mesh, self.det_mesh = np.ix_(imp, imp), (det, det)
        self.imp_fill, self.det_fill = (slice(ni), slice(ni)), (range(ni, nidx), range(ni, nidx))
        self.mask = np.ones((nidx, nidx))
        if bogo_only:
            hi, hd = ni // 2, len(det) // 2
            for lo, up in ((0, hi), (hi, ni), (ni, ni + hd), (ni + hd, nidx)):
                self.mask[lo:up, lo:up] = 0.0
        self.target = np.zeros((nidx, nidx))
        self.target[self.imp_fill] = rho[self.imp_mesh]
        self.target[self.det_fill] = rho[self.det_mesh]
        self.target *= self.mask
        self.nelec = check_nelec(self.nk * self.nso * 0.5, None)[0]
        g = vcor.gradient()
        tl = np.tril_indices(self.nso)
        self.dV = spin_orbital_matrix(np.asarray([g[:, 0], g[:, 1], g[:, 2]]))[:, tl[0], tl[1]]

    def _solve(self, param):
        from oracle.restate import assignocc
        self.vcor.update(param)
        H = self.GFock + spin_orbital_matrix(np.asarray(self.vcor.get()))[None]
        ew = np.empty((self.nk, self.nso))
        ev = np.empty((self.nk, self.nso, self.nso), dtype=complex)
        for k in range(self.nk):
            ew[k], ev[k] = la.eigh(H[k])
        occ, mu_q, _ = assignocc(ew, self.nelec, self.beta, 0.0, fix_mu=self.fix_mu)
        rhoT = (np.einsum('kpm,km,kqm->pq', ev, occ, ev.conj()) / self.nk).real
        rho1 = np.zeros_like(self.target)
        rho1[self.imp_fill] = rhoT[self.imp_mesh]
        rho1[self.det_fill] = rhoT[self.det_mesh]
        return ew, ev, mu_q, rho1 * self.mask - self.target

    def errfunc(self, param):
        return la.norm(self._solve(param)[3]) / np.sqrt(2.0)

    def gradfunc_ft(self, param):
        from oracle.restate_fit import get_dw_dv
        ew, ev, mu_q, drho = self._solve(param)
        val = la.norm(drho)
        res = 0.0
        for k in range(self.nk):
            dw = get_dw_dv(ew[k][None], ev[k][None], drho[None], mu_q, self.beta, fix_mu=self.fix_mu, fit_idx=self.fit_idx, compact=True)
            res = res + self.dV.dot(dw.ravel())
        return res.real / (2.0 * val * np.sqrt(2.0) * self.nk)


class GsoFullFitMu(GsoFullFit):
    """errfunc of spinless.FitVcorFull_mu (spinless.py:2030-2073): the particle chemical potential is searched, for every trial
    potential, so that tr rho_aa - tr rho_bb + nao of the cell-0 density equals 2 nao filling (bracketing + Brent, 1e-6, from
    `mu_start` with step `dx`), then the objective of GsoFullFit at that mu."""

    def __init__(self, rho, kmesh, vcor, filling, beta, fock_k, mu_start, dx=0.1, **kw):
        GsoFullFit.__init__(self, rho, kmesh, vcor, 0.0, beta, fock_k, **kw)
        self.bare, self.filling, self.mu_start, self.dx = self.GFock.copy(), filling, mu_start, dx

    def with_mu(self, m):
        n = self.nso // 2
        self.GFock = self.bare.copy()
        self.GFock[:, range(n), range(n)] -= m
        self.GFock[:, range(n, 2 * n), range(n, 2 * n)] += m

    def solve_mu(self, param):
        from scipy.optimize import brentq
        from oracle.restate import assignocc
        n = self.nso // 2
        target = self.nso * self.filling

        def nelec_phys(m):
            self.with_mu(m)
            self.vcor.update(param)
            H = self.GFock + spin_orbital_matrix(np.asarray(self.vcor.get()))[None]
            pairs = [la.eigh(H[k]) for k in range(self.nk)]
            ews, evs = np.asarray([p[0] for p in pairs]), np.asarray([p[1] for p in pairs])
            occ = assignocc(ews, self.nelec, self.beta, 0.0, fix_mu=self.fix_mu)[0]
            w = (np.abs(evs[:, :n]) ** 2).sum(axis=1) - (np.abs(evs[:, n:]) ** 2).sum(axis=1)
            return np.dot(w.ravel(), occ.ravel()) / self.nk + n
        x, y = self.mu_start, nelec_phys(self.mu_start)
        if abs(y - target) < 1e-6:
            return x
        step = -self.dx if y > target else self.dx
        while True:
            x1 = x + step
            y1 = nelec_phys(x1)
            if abs(y1 - target) < 1e-6:
                return x1
            if (y - target) * (y1 - target) < 0:
                break
            x, y = x1, y1
        return brentq(lambda m: nelec_phys(m) - target, min(x, x1), max(x, x1), xtol=1e-6, rtol=1e-6, maxiter=20, disp=False)

    def errfunc(self, param):
        self.with_mu(self.solve_mu(param))
        return GsoFullFit.errfunc(self, param)
