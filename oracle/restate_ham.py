"""
oracle/restate_ham.py -- CPU restatement (numpy) of SURVEY.md section 8(f) rank 1: the embedding
one-body Hamiltonian and the ERI x density contraction that completes `get_emb_Ham`:

  slater_helper.transform_trans_inv[_k], transform_local, transform_imp, transform_imp_env,
  transform_4idx, transform_eri_local                              routine/slater_helper.py:22-156
  slater.transform_h1 / foldRho_k / foldRho                        routine/slater.py:682-704
  scf._get_jk / _get_veff, slater.get_veff                         solver/scf.py:255-352, routine/slater.py:477-523
  slater.__embHam1e (HF branches), get_emb_Ham                     routine/slater.py:525-680, 320-370
  integral.get_eri_format                                          system/integral.py:883-927

PySCF primitive restated here: `pyscf.scf.hf.dot_eri_dm(eri, dm, hermi, with_j, with_k)` in the
convention the reference documents at solver/scf.py:268-270 (J: ijkl,kl->ij; K: ijkl,il->jk), which
is also what its UIHF code relies on (the explicit transpose of the alpha-beta block at :325-327).

TEST INFRASTRUCTURE ONLY (never imported by the product package).  Pinned against the reference
through tests/golden/G8_embham.npz (oracle/gen_golden.py gen_G8: reference get_emb_Ham, _get_jk,
get_veff, transform_* executed under oracle/shim.py with this module's dot_eri_dm bound in).
"""
import itertools as it

import numpy as np

from oracle.restate import CellArith, restore, R2k
from oracle.restate_bcs import unit2emb   # noqa: F401  (same container helper)


# ---------------------------------------------------------------------------------------------
# one-body folds (slater_helper.py:22-115)
# ---------------------------------------------------------------------------------------------

def transform_trans_inv(basis, kmesh, H, symmetric=True):
    ca = CellArith(kmesh)
    nc, nb = ca.ncells, basis.shape[-1]
    res = np.zeros((nb, nb))
    if symmetric:
        for i in range(nc):
            res += basis[i].T @ H[0] @ basis[i]
        for i, j in it.combinations(range(nc), 2):
            t = basis[i].T @ H[ca.subtract(i, j)] @ basis[j]
            res += t + t.T
    else:
        for i, j in it.product(range(nc), repeat=2):
            res += basis[i].T @ H[ca.subtract(i, j)] @ basis[j]
    return res


def transform_trans_inv_k(basis_k, H_k):
    nk, nlo, nb = basis_k.shape
    res = np.zeros((nb, nb), dtype=np.complex128)
    for k in range(nk):
        res += basis_k[k].conj().T @ H_k[k] @ basis_k[k]
    return res.real / float(nk)


def transform_local(basis, H):
    return sum(basis[i].T @ H @ basis[i] for i in range(basis.shape[0]))


def transform_imp(basis, H):
    return basis[0].T @ H @ basis[0]


def transform_imp_env(basis, H):
    res = sum(basis[i].T @ H[i] @ basis[0] for i in range(basis.shape[0]))
    return 0.5 * (res + res.T)


def transform_4idx(v, ip, jq, kr, ls):
    return np.einsum('ijkl,ip,jq,kr,ls->pqrs', v, ip, jq, kr, ls, optimize=True)


def transform_eri_local(basis, H2):
    """slater_helper.py:133-156: sum over cells of the 4-index transform of a cell-local ERI."""
    if basis.ndim == 3:
        basis = basis[None]
    spin, nc, n, nb = basis.shape
    res = np.zeros((spin * (spin + 1) // 2, nb, nb, nb, nb))
    if H2.ndim == 4:
        H2 = H2[None] if spin == 1 else [H2, H2, H2]
    for i in range(nc):
        res[0] += transform_4idx(H2[0], *([basis[0, i]] * 4))
        if spin == 2:
            res[1] += transform_4idx(H2[1], *([basis[1, i]] * 4))
            res[2] += transform_4idx(H2[2], basis[0, i], basis[0, i], basis[1, i], basis[1, i])
    return res


def _add_spin_dim(H, spin, non_spin_dim=3):
    H = np.asarray(H)
    if H.ndim == non_spin_dim:
        H = H[None]
    if H.shape[0] < spin:
        H = np.asarray((H[0],) * spin)
    return H


def transform_h1(H1_k, basis_k):
    """slater.py:682-689 (alias foldRho_k :704)."""
    spin, nb = basis_k.shape[0], basis_k.shape[-1]
    H1_k = _add_spin_dim(H1_k, spin)
    return np.asarray([transform_trans_inv_k(basis_k[s], H1_k[s]) for s in range(spin)])


foldRho_k = transform_h1


def foldRho(rho, kmesh, basis):
    return np.asarray([transform_trans_inv(basis[s], kmesh, rho[s]) for s in range(rho.shape[0])])


# ---------------------------------------------------------------------------------------------
# ERI x density (solver/scf.py:255-352)
# ---------------------------------------------------------------------------------------------

def get_eri_format(eri, nao):
    """system/integral.py:883-927."""
    eri = np.asarray(eri)
    npair = nao * (nao + 1) // 2
    s1, s4, s8 = nao ** 4, npair * npair, npair * (npair + 1) // 2
    if eri.ndim == 5:
        return 's1', eri.size // s1
    if eri.ndim == 4 and eri.size == s1:
        return 's1', 0
    if eri.ndim == 3:
        return 's4', eri.size // s4
    if eri.ndim == 2 and eri.size == s4:
        return 's4', 0
    if eri.ndim == 2 and eri.size == s8:
        return 's8', 1
    if eri.ndim == 1 and eri.size == s8:
        return 's8', 0
    raise ValueError("Unknown ERI shape %s, nao %s" % (eri.shape, nao))


def _unpack_s8(eri, nao):
    npair = nao * (nao + 1) // 2
    e4 = np.zeros((npair, npair))
    i2 = np.tril_indices(npair)
    e4[i2] = eri.ravel()
    e4[(i2[1], i2[0])] = eri.ravel()
    return e4


def _to_s4(eri, nao):
    """ao2mo.restore(4, .) for s1 or s4 input."""
    eri = np.asarray(eri)
    npair = nao * (nao + 1) // 2
    if eri.size == npair * npair:
        return eri.reshape(npair, npair)
    ia, ib = np.tril_indices(nao)
    return eri.reshape((nao,) * 4)[ia, ib][:, ia, ib]


def dot_eri_dm(eri, dm, hermi=0, with_j=True, with_k=True):
    """PySCF hf.dot_eri_dm, documented convention J: ijkl,kl->ij  K: ijkl,il->jk; eri in s1 / s4 / s8 storage."""
    dm = np.asarray(dm, dtype=np.double)
    nao = dm.shape[-1]
    dms = dm.reshape(-1, nao, nao)
    eri = np.asarray(eri)
    npair = nao * (nao + 1) // 2
    if eri.size == nao ** 4:
        e1 = eri.reshape((nao,) * 4)
    elif eri.size == npair * npair:
        e1 = restore(1, eri.reshape(npair, npair), nao)
    else:
        e1 = restore(1, _unpack_s8(eri, nao), nao)
    vj = np.einsum('ijkl,xkl->xij', e1, dms).reshape(dm.shape) if with_j else None
    vk = np.einsum('ijkl,xil->xjk', e1, dms).reshape(dm.shape) if with_k else None
    return vj, vk


def get_jk(dm, eri, with_j=True, with_k=True):
    """solver/scf.py:255-335 (_get_jk)."""
    dm = np.asarray(dm, dtype=np.double)
    if dm.ndim == 2:
        dm = dm[None]
    spin, nao = dm.shape[0], dm.shape[-1]
    eri = np.asarray(eri, dtype=np.double)
    fmt, sd = get_eri_format(eri, nao)
    if sd == 0:
        eri, sd = eri[None], 1
    if spin == 1 or sd == 1:
        return dot_eri_dm(eri[0], dm, 1, with_j, with_k)
    if sd != 3:
        raise ValueError
    assert spin == 2
    vj00, vk00 = dot_eri_dm(_to_s4(eri[0], nao), dm[0], 1, with_j, with_k)
    vj11, vk11 = dot_eri_dm(_to_s4(eri[1], nao), dm[1], 1, with_j, with_k)
    eab = _to_s4(eri[2], nao)
    vj01 = dot_eri_dm(eab, dm[1], 1, with_j, False)[0]
    vj10 = dot_eri_dm(eab.T, dm[0], 1, with_j, False)[0]
    return np.asarray(((vj00, vj11), (vj01, vj10))), np.asarray((vk00, vk11))


def get_veff(rdm1, eri, hyb=1.0, hyb_j=1.0, ghf=False):
    """slater.py:477-523; ghf: ONE (nso, nso) density against a spinless ERI, J - hyb K with J scaled by hyb_j
    (solver/scf.py:732-740 _get_veff_ghf for the HF case)."""
    rdm1 = np.asarray(rdm1)
    if ghf:
        assert rdm1.ndim == 2
        vj, vk = get_jk(rdm1, eri, with_j=True, with_k=hyb != 0.0)
        if hyb == 1.0:
            return vj[0] - vk[0]
        vj0 = vj[0] if hyb_j == 1.0 else vj[0] * hyb_j
        return vj0 if hyb == 0.0 else vj0 - (vk[0] * hyb)
    if rdm1.ndim == 2:
        rdm1 = rdm1[None]
    spin = rdm1.shape[0]
    if hyb == 1.0:
        vj, vk = get_jk(rdm1, eri)
        return vj - vk * 0.5 if spin == 1 else vj[0] + vj[1] - vk
    if hyb == 0.0:
        vj = get_jk(rdm1, eri, with_j=True, with_k=False)[0]
        return vj if spin == 1 else vj[0] + vj[1]
    vj, vk = get_jk(rdm1, eri)
    return vj - vk * (hyb * 0.5) if spin == 1 else vj[0] + vj[1] - vk * hyb


# ---------------------------------------------------------------------------------------------
# embedding Hamiltonian (slater.py:320-370, 525-680), Hartree-Fock branches
# ---------------------------------------------------------------------------------------------

def embHam1e(kmesh, basis, H2_emb, hcore_k, fock_k, ovlp_k, rdm1_k, vcor_mat=None, int_bath=True, add_vcor=False,
             JK_imp=None, use_hcore_as_emb_ham=False, fitting=False):
    """Returns H1, ovlp_emb, JK_core.  fock_k must already be hcore_lo_k + vhf_lo_k for ab-initio lattices."""
    spin = basis.shape[0]
    basis_k = np.asarray([R2k(basis[s], kmesh) for s in range(spin)])
    hcore_emb = transform_h1(hcore_k, basis_k)
    ovlp_emb = transform_h1(ovlp_k, basis_k)
    if ovlp_emb.ndim == 3 and ovlp_emb.shape[0] == 1:
        ovlp_emb = ovlp_emb[0]
    if int_bath:
        rdm1_emb = foldRho_k(rdm1_k, basis_k)
        H1 = transform_h1(fock_k, basis_k)
        H1 = H1 - get_veff(rdm1_emb, H2_emb)
        JK_core = H1 - hcore_emb
    else:
        add_vcor = True
        if use_hcore_as_emb_ham:
            H1, JK_core = hcore_emb, None
        else:
            H1 = transform_h1(fock_k, basis_k)
            if JK_imp is not None:
                JK_emb = np.asarray([transform_imp(basis[s], JK_imp if JK_imp.ndim == 2 else JK_imp[s])
                                     for s in range(spin)])
            else:
                JK_emb = get_veff(foldRho_k(rdm1_k, basis_k), H2_emb)
            H1 = H1 - JK_emb
            JK_core = H1 - hcore_emb
    if add_vcor:
        H1 = np.array(H1, copy=True)
        for s in range(spin):
            H1[s] += transform_local(basis[s], vcor_mat[s])
            if not fitting:
                H1[s] -= transform_imp(basis[s], vcor_mat[s])
    return H1, ovlp_emb, JK_core
