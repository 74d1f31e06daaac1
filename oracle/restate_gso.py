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
