"""
Schmidt-decomposition bath with the reference's entry point
(libdmet/routine/slater.py:98-318: get_emb_basis / embBasis, SVD and eig flavours).

The SVD flavour never materialises `lattice.expand(rdm1)` (slater.py:167-171): the env x imp block is
gathered on the device from the stripe by index arithmetic, factorised (Householder QR + Jacobi SVD,
dmk_bath_svd), thresholded on the host (`(sigma >= tol_bath).sum()`, slater.py:181-185) and
orthogonalised / scattered by dmk_bath_assemble (slater.py:200-213, lo/lowdin.py:83-101).

get_emb_Ham (slater.py:320-704) completes the exit of the path: H2 from the DF transform (or the cell-local
4-index transform for models), H1 = basis^H (hcore + vhf) basis - JK_emb with JK_emb from dmk_jk_s4.
"""
import numpy as np

from libdmet_preview_amd._lib import lib, mesh3, get_ctx
from libdmet_preview_amd.basis_transform.eri_transform import get_emb_eri, get_unit_eri
from libdmet_preview_amd.routine.slater_helper import *       # noqa: F401,F403  (reference: slater.py:35)
from libdmet_preview_amd.routine.slater_helper import (transform_trans_inv, transform_trans_inv_k, transform_local,
                                                       transform_imp, transform_eri_local, unit2emb)
from libdmet_preview_amd.solver.scf import _get_jk, _get_veff
from libdmet_preview_amd.system import integral
from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.utils.misc import add_spin_dim


def _index_sets(lattice, imp_idx, val_idx, valence_bath):
    ncells, nlo = lattice.ncells, lattice.nscsites
    imp_idx_bath = list(val_idx) if valence_bath else list(imp_idx)
    bath_set, imp_set = set(imp_idx_bath), set(imp_idx)
    env_idx = [i for i in range(ncells * nlo) if i not in bath_set]
    virt_mask = [i in imp_set for i in env_idx]
    return imp_idx_bath, np.asarray(env_idx, dtype=np.int32), np.asarray(virt_mask, dtype=np.int32)


def get_emb_basis(lattice, rho=None, local=True, kind='svd', **kwargs):
    """Embedding basis C_lo_eo for a Slater determinant, shape (spin, ncells, nlo, nemb)."""
    if rho is None:
        rho = lattice.rdm1_lo_R
    if not local:
        raise NotImplementedError("non-local (particle-hole symmetric) bath is outside the HIP path")
    rho = np.asarray(rho).real
    if kind == 'svd':
        return _get_emb_basis_svd(lattice, rho, **kwargs)
    elif kind == 'eig':
        return _get_emb_basis_eig(lattice, rho, **kwargs)
    raise ValueError("get_emb_basis: Unknown kind %s" % kind)


embBasis = get_emb_basis


def bath_svd_dev(ctx, kmesh, nlo, d_rdm1_s, d_env, nenv, d_col, nb):
    d_sigma = ctx.empty((nb,), np.float64)
    d_U = ctx.empty((nenv, nb), np.float64)
    ctx.check(lib.dmk_bath_svd(ctx.h, mesh3(kmesh), int(nlo), d_rdm1_s.ptr, d_env.ptr, int(nenv), d_col.ptr,
                               int(nb), d_sigma.ptr, d_U.ptr))
    return d_sigma, d_U


def bath_assemble_dev(ctx, d_U, nenv, nb, nbath, d_virt, orth, d_env, d_imp, nimp, nsites, ncol, d_basis):
    ctx.check(lib.dmk_bath_assemble(ctx.h, d_U.ptr, int(nenv), int(nb), int(nbath), d_virt.ptr, 1 if orth else 0,
                                    d_env.ptr, d_imp.ptr, int(nimp), int(nsites), int(ncol), d_basis.ptr))
    return d_basis


def _get_emb_basis_svd(lattice, rdm1, **kwargs):
    imp_idx = kwargs.get("imp_idx", lattice.imp_idx)
    val_idx = kwargs.get("val_idx", lattice.val_idx)
    valence_bath = kwargs.get("valence_bath", True)
    orth = kwargs.get("orth", True)
    tol_bath = kwargs.get("tol_bath", 1e-9)
    nbath = kwargs.get("nbath", None)
    if kwargs.get("localize_bath", None) is not None:
        raise NotImplementedError("localize_bath is outside the HIP path")

    ncells, nlo = lattice.ncells, lattice.nscsites
    imp_idx = list(imp_idx)
    imp_idx_bath, env_idx, virt_mask = _index_sets(lattice, imp_idx, val_idx, valence_bath)
    nimp = len(imp_idx)
    rdm1 = np.asarray(rdm1)
    if rdm1.ndim == 3:
        rdm1 = rdm1[np.newaxis]
    assert rdm1.shape[-3:] == (ncells, nlo, nlo)
    spin = rdm1.shape[0]
    # same nbath_final seeds as the two branches of slater.py:167-175
    nbath_final = len(imp_idx_bath) if np.max(imp_idx_bath) >= nlo - 1 else nlo
    nb, nenv = len(imp_idx_bath), len(env_idx)
    nsites, ncol = ncells * nlo, nimp * 2

    ctx = get_ctx()
    d_env = ctx.to_device(env_idx, np.int32)
    d_col = ctx.to_device(np.asarray(imp_idx_bath), np.int32)
    d_virt = ctx.to_device(virt_mask, np.int32)
    d_imp = ctx.to_device(np.asarray(imp_idx), np.int32)
    d_rdm1 = ctx.to_device(rdm1, np.float64)
    basis = np.zeros((spin, nsites, ncol))
    for s in range(spin):
        d_sigma, d_U = bath_svd_dev(ctx, lattice.kmesh, nlo, d_rdm1.offset(s * ncells * nlo * nlo, (ncells, nlo, nlo)),
                                    d_env, nenv, d_col, nb)
        sigma = d_sigma.get()
        nbath_s = int((sigma >= tol_bath).sum()) if nbath is None else int(nbath)
        nzero = int(np.sum(np.abs(sigma[:nbath_s]) < tol_bath))
        log.debug(0, "Zero singular values number: %s", nzero)
        if nzero > 0:
            log.warn("Zero singular value exists, \nthis may cause numerical instability.")
        d_basis = ctx.empty((nsites, ncol), np.float64)
        bath_assemble_dev(ctx, d_U, nenv, nb, nbath_s, d_virt, orth, d_env, d_imp, nimp, nsites, ncol, d_basis)
        basis[s] = d_basis.get()
        nbath_final = min(nbath_final, nbath_s)
    log.debug(0, "nimp : %d", nimp)
    log.debug(0, "nbath: %d", nbath_final)
    return np.ascontiguousarray(basis[:, :, :nimp + nbath_final]).reshape(spin, ncells, nlo, nimp + nbath_final)


def _get_emb_basis_eig(lattice, rdm1, **kwargs):
    """Eigen-decomposition of the env-env block (slater.py:224-318); model-size systems
    (needs the expanded (ncells*nlo)^2 matrix like the reference)."""
    imp_idx = kwargs.get("imp_idx", lattice.imp_idx)
    val_idx = kwargs.get("val_idx", lattice.val_idx)
    valence_bath = kwargs.get("valence_bath", True)
    orth = kwargs.get("orth", True)
    tol_bath = kwargs.get("tol_bath", 1e-9)
    ncells, nlo = lattice.ncells, lattice.nscsites
    imp_idx = list(imp_idx)
    imp_idx_bath, env_idx, virt_mask = _index_sets(lattice, imp_idx, val_idx, valence_bath)
    nimp, nenv = len(imp_idx), len(env_idx)
    rdm1 = np.asarray(rdm1)
    if rdm1.ndim == 3:
        rdm1 = rdm1[np.newaxis]
    spin = rdm1.shape[0]
    if nenv > 1024:
        raise NotImplementedError("eig bath: env dimension %d exceeds the batched eigensolver limit" % nenv)
    env_env = lattice.expand(rdm1)[:, env_idx][:, :, env_idx]
    ctx = get_ctx()
    d_A = ctx.to_device(env_env, np.float64)
    d_w = ctx.empty((spin, nenv), np.float64)
    d_Vt = ctx.empty((spin, nenv, nenv), np.float64)
    ctx.check(lib.dmk_eigh_batched_real(ctx.h, nenv, spin, d_A.ptr, d_w.ptr, d_Vt.ptr))
    ew, Vt = d_w.get(), d_Vt.get()
    keep = [[i for i, e in enumerate(ew[s]) if abs(e) > tol_bath and abs(1 - e) > tol_bath] for s in range(spin)]
    nb = len(keep[0])
    if any(len(k) != nb for k in keep):
        raise ValueError("eig bath: spin sectors give different numbers of bath orbitals")
    nsites = ncells * nlo
    basis = np.zeros((spin, nsites, nimp + nb))
    d_env = ctx.to_device(env_idx, np.int32)
    d_virt = ctx.to_device(virt_mask, np.int32)
    d_imp = ctx.to_device(np.asarray(imp_idx), np.int32)
    for s in range(spin):
        if nb == 0:
            basis[s, imp_idx, :nimp] = np.eye(nimp)
            continue
        U = np.ascontiguousarray(Vt[s][keep[s]].T)          # (nenv, nb) columns = kept eigenvectors
        d_U = ctx.to_device(U, np.float64)
        d_basis = ctx.empty((nsites, nimp + nb), np.float64)
        bath_assemble_dev(ctx, d_U, nenv, nb, nb, d_virt, orth, d_env, d_imp, nimp, nsites, nimp + nb, d_basis)
        basis[s] = d_basis.get()
    return basis.reshape(spin, ncells, nlo, nimp + nb)


# ---------------------------------------------------------------------------------------------
# embedding Hamiltonian (routine/slater.py:320-704)
# ---------------------------------------------------------------------------------------------

def transform_h1(H1_k, basis_k):
    """(spin, nbasis, nbasis) = per spin (1/nk) Re sum_k basis_k^H H1_k basis_k   (slater.py:682-689)."""
    basis_k = np.asarray(basis_k)
    spin, nbasis = basis_k.shape[0], basis_k.shape[-1]
    H1_k = add_spin_dim(H1_k, spin, non_spin_dim=3)
    H1 = np.empty((spin, nbasis, nbasis))
    for s in range(spin):
        H1[s] = transform_trans_inv_k(basis_k[s], H1_k[s])
    return H1


foldRho_k = transform_h1


def foldRho(rho, lattice, basis):
    """Density matrix into the embedding space from the real-space stripe (slater.py:691-701)."""
    spin, nbasis = rho.shape[0], basis.shape[-1]
    rdm1_emb = np.empty((spin, nbasis, nbasis))
    for s in range(spin):
        rdm1_emb[s] = transform_trans_inv(basis[s], lattice, rho[s])
    return rdm1_emb


def get_veff(rdm1, eri, hyb=1.0, ghf=False, hyb_j=1.0):
    """Effective potential of the embedding Hamiltonian (slater.py:477-523); rdm1 is spin traced if restricted."""
    if ghf:
        raise NotImplementedError("the GHF effective potential is outside the HIP path")
    rdm1 = np.asarray(rdm1)
    if rdm1.ndim == 2:
        rdm1 = rdm1[None]
    spin = rdm1.shape[0]
    if hyb == 1.0:      # HF
        veff = _get_veff(rdm1, eri)
    elif hyb == 0.0:    # pure DFT, J only
        vj = _get_jk(rdm1, eri, with_j=True, with_k=False)[0]
        veff = vj if spin == 1 else vj[0] + vj[1]
    else:               # hybrid DFT
        vj, vk = _get_jk(rdm1, eri, with_j=True, with_k=True)
        veff = vj - vk * (hyb * 0.5) if spin == 1 else vj[0] + vj[1] - vk * hyb
    return veff


def _embHam2e(lattice, basis, vcor, local, int_bath=True, last_aabb=True, **kwargs):
    """H2_emb (slater.py:372-475): DF transform for ab-initio lattices, cell-local 4-index transform for models."""
    nbasis, spin = basis.shape[-1], basis.shape[0]
    if lattice.is_model:
        LatH2 = lattice.getH2(compact=False, kspace=False)
        log.eassert(local, "non local bath is outside the HIP path")
        if lattice.H2_format == 'local':
            if int_bath:
                H2 = transform_eri_local(basis, lattice, LatH2)
            else:
                H2 = unit2emb(np.asarray((LatH2,) * (spin * (spin + 1) // 2)), nbasis)
        elif lattice.H2_format == "spin local":
            if int_bath:
                raise NotImplementedError
            H2 = unit2emb(np.asarray(LatH2), nbasis)
        else:
            raise NotImplementedError("H2_format %s is outside the HIP path" % lattice.H2_format)
    else:
        opts = dict(kscaled_center=kwargs.get("kscaled_center", None), symmetry=lattice.eri_symmetry,
                    max_memory=kwargs.get("max_memory", None), swap_idx=kwargs.get("swap_idx", None),
                    t_reversal_symm=kwargs.get("t_reversal_symm", True), incore=kwargs.get("incore", True),
                    fout=kwargs.get("fout", "H2.h5"), use_mpi=kwargs.get("use_mpi", False))
        if int_bath:
            H2 = get_emb_eri(lattice.cell, lattice.df, C_ao_lo=lattice.C_ao_lo, basis=basis, **opts)
            if last_aabb and isinstance(H2, np.ndarray) and H2.shape[0] == 3:
                H2 = H2[[0, 2, 1]]
        else:
            H2 = get_unit_eri(lattice.cell, lattice.df, C_ao_lo=lattice.C_ao_lo, **opts)
            if last_aabb and isinstance(H2, np.ndarray) and H2.shape[0] == 3:
                H2 = H2[[0, 2, 1]]
            H2 = unit2emb(H2, nbasis)
    if isinstance(H2, np.ndarray):
        log.info("H2 memory allocated size = %d MB", H2.size * 8. / 1024 / 1024)
    return H2


def _embHam1e(lattice, basis, vcor, H2_emb, int_bath=True, add_vcor=False, **kwargs):
    """H1_emb and ovlp_emb (slater.py:525-680), Hartree-Fock branches; sets lattice.JK_core."""
    for k in ("dft", "qsgw", "vxc_dc"):
        if kwargs.get(k, False):
            raise NotImplementedError("%s embedding Hamiltonian is outside the HIP path" % k)
    spin = basis.shape[0]
    basis_k = lattice.R2k_basis(basis)
    hcore_k = lattice.getH1(kspace=True)
    fock_k = lattice.getFock(kspace=True)
    ovlp_k = lattice.get_ovlp(kspace=True)
    JK_imp = lattice.get_JK_imp()
    if not isinstance(H2_emb, np.ndarray):
        H2_emb = np.asarray(H2_emb["ccdd"])

    hcore_emb = transform_h1(hcore_k, basis_k)
    ovlp_emb = transform_h1(ovlp_k, basis_k)
    if ovlp_emb.ndim == 3 and ovlp_emb.shape[0] == 1:
        ovlp_emb = ovlp_emb[0]

    if int_bath:
        rdm1_emb = foldRho_k(lattice.rdm1_lo_k, basis_k)
        if not lattice.is_model:
            fock_k = lattice.hcore_lo_k + lattice.vhf_lo_k
        H1 = transform_h1(fock_k, basis_k)
        # subtract JK_emb = rho_kl [2 (ij||kl) - (il||jk)], all indices in the embedding basis
        H1 -= get_veff(rdm1_emb, H2_emb)
        lattice.JK_core = H1 - hcore_emb
    else:
        add_vcor = True
        if lattice.use_hcore_as_emb_ham:
            H1 = hcore_emb
            lattice.JK_core = None
        else:
            H1 = transform_h1(fock_k, basis_k)
            if JK_imp is not None:
                JK_imp = np.asarray(JK_imp)
                if JK_imp.ndim == 2:
                    JK_emb = np.asarray([transform_imp(basis[s], lattice, JK_imp) for s in range(spin)])
                else:
                    JK_emb = np.asarray([transform_imp(basis[s], lattice, JK_imp[s]) for s in range(spin)])
            else:
                JK_emb = get_veff(foldRho_k(lattice.rdm1_lo_k, basis_k), H2_emb)
            H1 -= JK_emb
            lattice.JK_core = H1 - hcore_emb

    if add_vcor:
        log.eassert(vcor.islocal(), "nonlocal correlation potential cannot be treated in this routine")
        for s in range(spin):
            H1[s] += transform_local(basis[s], lattice, vcor.get()[s])
            if not "fitting" in kwargs or not kwargs["fitting"]:
                H1[s] -= transform_imp(basis[s], lattice, vcor.get()[s])
    return H1, ovlp_emb


def get_emb_Ham(lattice, basis, vcor, local=True, **kwargs):
    """
    Embedding Hamiltonian (slater.py:320-370): two-body part first (JK_emb needs it), then the one-body part.

    Kwargs: incore, H2_given (ndarray), int_bath, add_vcor, fitting, and the ERI-transform options.
    Returns (ImpHam, None) with ImpHam an integral.Integral.
    """
    basis = np.asarray(basis)
    spin, nbasis = basis.shape[0], basis.shape[-1]
    log.info("Two-body part")
    H2_given = kwargs.get("H2_given", None)
    if H2_given is None:
        if kwargs.get("H2_fname", None) is not None:
            raise NotImplementedError("H2_fname (HDF5) is not available; pass H2_given")
        H2 = _embHam2e(lattice, basis, vcor, local, **kwargs)
    else:
        log.debug(1, "Using specified H2 array.")
        H2 = H2_given
    log.info("One-body part")
    H1, ovlp_emb = _embHam1e(lattice, basis, vcor, H2, **kwargs)
    H0 = lattice.getH0()
    if isinstance(H2, np.ndarray):
        H2 = {"ccdd": H2}
    ImpHam = integral.Integral(nbasis, spin == 1, False, H0, {"cd": H1}, H2, ovlp=ovlp_emb)
    return ImpHam, None


embHam = get_emb_Ham
