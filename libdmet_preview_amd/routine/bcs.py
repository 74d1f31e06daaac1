"""
BCS (Nambu) embedding basis with the reference's entry point (libdmet/routine/bcs.py:25-135):

  embBasis(lattice, GRho, local=True)     projective bath  (:33-107, the branch without "sites")
  embBasis(lattice, GRho, local=False)    quasiparticle bath, Phys. Rev. B 93, 035126 (:109-135)

The projective bath is the Slater Schmidt step on the generalised density matrix: GRho is treated as
the stripe of a lattice with 2*nscsites orbitals per cell, so the env x imp block is gathered and
factorised by dmk_bath_svd exactly like routine/slater.py; the particle weights (:92) and the
alpha / beta column assignment (:93-103) run in dmk_bcs_weight / dmk_bcs_assemble.
"""
import numpy as np

from libdmet_preview_amd._lib import lib, get_ctx
from libdmet_preview_amd.routine.bcs_helper import *          # noqa: F401,F403  (reference re-exports them)
from libdmet_preview_amd.routine.slater import bath_svd_dev, complete_null_columns
from libdmet_preview_amd.utils import logger as log


def embBasis(lattice, GRho, local=True, **kwargs):
    if local:
        return _embBasis_proj(lattice, GRho, **kwargs)
    else:
        return _embBasis_phsymm(lattice, GRho, **kwargs)


get_emb_basis = embBasis


def emb_basis_proj_dev(ctx, kmesh, ncells, nscsites, val_idx, d_GRho, localize_bath=None):
    """Device form: d_GRho (ncells, 2n, 2n) f64 -> d_basis (2, ncells, 2n, n+nval), sigma, w, order, d_U.  `localize_bath`: the
    bath orbitals are rotated by routine/localizer.localize_bath between the SVD and the particle weights (bcs.py:84-88)."""
    n, nval = int(nscsites), len(val_idx)
    nenv, nb = (ncells - 1) * 2 * n, 2 * nval
    cols = np.asarray(list(val_idx) + [i + n for i in val_idx], dtype=np.int32)
    env = np.arange(2 * n, 2 * n * ncells, dtype=np.int32)
    d_sigma, d_U = bath_svd_dev(ctx, kmesh, 2 * n, d_GRho, ctx.to_device(env), nenv, ctx.to_device(cols), nb)
    complete_null_columns(ctx, d_sigma.get().reshape(-1)[:nb], d_U, nenv, nb)     # bcs.py:46, 84: every column is kept
    if localize_bath is not None:
        from libdmet_preview_amd.routine import localizer
        d_U.set(np.ascontiguousarray(localizer.localize_bath(d_U.get().reshape(ncells - 1, 2 * n, nb), method=localize_bath)
                                     .reshape(nenv, nb)))
    d_w = ctx.empty((nb,), np.float64)
    ctx.check(lib.dmk_bcs_weight(ctx.h, ncells - 1, 2 * n, n, nb, d_U.ptr, d_w.ptr))
    w = d_w.get()
    order = np.argsort(w, kind='mergesort')[::-1]
    d_basis = ctx.empty((2, ncells, 2 * n, n + nval), np.float64)
    d_order = ctx.to_device(np.ascontiguousarray(order, dtype=np.int32))      # keep alive across the launch
    ctx.check(lib.dmk_bcs_assemble(ctx.h, int(ncells), n, nval, d_U.ptr, d_order.ptr, d_basis.ptr))
    return d_basis, d_sigma.get(), w, order, d_U


def _embBasis_proj(lattice, GRho, **kwargs):
    ncells, nscsites, nval = lattice.ncells, lattice.nscsites, lattice.nval
    if "sites" in kwargs:
        # the reference's "sites" branch reads an undefined name (bcs.py:44-45) and cannot run
        raise NotImplementedError('keyword "sites" is not supported')
    loc_method = kwargs.get("localize_bath", None)
    if loc_method is not None:
        log.eassert(lattice.is_model, "Only model is currently supported for localization of bath.")
    GRho = np.ascontiguousarray(np.asarray(GRho).real, dtype=np.float64)
    log.eassert(GRho.shape == (ncells, 2 * nscsites, 2 * nscsites), "GRho must be (ncells, 2*nscsites, 2*nscsites)")
    ctx = get_ctx()
    d_basis, sigma, w, order, d_U = emb_basis_proj_dev(ctx, lattice.kmesh, ncells, nscsites, lattice.val_idx,
                                                       ctx.to_device(GRho), localize_bath=loc_method)
    log.debug(0, "Zero singular values number: %s", np.sum(np.abs(sigma) < 1e-8))
    log.debug(1, "Singular values:\n%s", sigma)
    w1 = w[order]
    wA, wB = w1[:nval], 1.0 - w1[nval:]
    log.debug(0, "particle character:\nspin A max %.2f min %.2f mean %.2f"
              "\nspin B max %.2f min %.2f mean %.2f", np.max(wA), np.min(wA), np.average(wA),
              np.max(wB), np.min(wB), np.average(wB))
    log.info("Bath coupling strength\n%s\n%s", sigma[order[:nval]], sigma[order[nval:]])
    if kwargs.get("only_return_bath", False):
        return d_U.get().reshape((ncells - 1, nscsites * 2, nval * 2))
    return d_basis.get()


def _eigh_real(ctx, A):
    """Symmetric eigenproblem of one small real matrix on the device: ew ascending, ev columns."""
    m = A.shape[-1]
    d_w = ctx.empty((1, m), np.float64)
    d_Vt = ctx.empty((1, m, m), np.float64)
    d_A = ctx.to_device(A, np.float64)
    ctx.check(lib.dmk_eigh_batched_real(ctx.h, m, 1, d_A.ptr, d_w.ptr, d_Vt.ptr))
    return d_w.get()[0], np.ascontiguousarray(d_Vt.get()[0].T)


def _inv_sqrt_factor(ctx, M):
    """inv(A^T) for A = MatSqrt(M) = ev sqrt(ew) (routine/slater.py:38-50): ev diag(ew^-1/2)."""
    log.eassert(np.abs(M - M.T).max() < 1e-10, "matrix must be symmetric")
    ew, ev = _eigh_real(ctx, M)
    if ew[0] < 0:
        ew = ew + 1e-11
    log.eassert((ew >= 0).all(), "matrix must be positive definite")
    log.check(ew[0] > 1e-10, "small eigenvalue for rho_imp,"
              "cut-off is recommended\nthe first 5 eigenvalues are %s", ew[:5])
    return ev / np.sqrt(ew)[None, :]


def _rotate_rows(ctx, d_b, nrow, m, X):
    """b (nrow, m) @ X (m, m) on the device."""
    d_out = ctx.empty((nrow, m), np.float64)
    d_X = ctx.to_device(X, np.float64)
    ctx.check(lib.dmk_dgemm_nn_small(ctx.h, int(nrow), m, m, d_b.ptr, d_X.ptr, 0, d_out.ptr))
    ctx.sync()                                   # d_X goes out of scope here
    return d_out


def _orthonormalize_dev(ctx, d_b, nrow, m):
    """routine/slater.py:59-78 on a device (nrow, m) block."""
    d_ov = ctx.zeros((m, m), np.float64)
    ctx.check(lib.dmk_dgemm_tn_acc_rect(ctx.h, m, m, int(nrow), 1.0, d_b.ptr, m, d_b.ptr, m, d_ov.ptr, m))
    ov = d_ov.get()
    log.debug(1, "basis overlap is\n%s", ov)
    if np.allclose(ov - np.diag(np.diag(ov)), 0.):
        return _rotate_rows(ctx, d_b, nrow, m, np.diag(1. / np.sqrt(np.diag(ov))))
    ew, ev = _eigh_real(ctx, ov)
    ew, ev = ew[::-1], ev[:, ::-1]
    return _rotate_rows(ctx, d_b, nrow, m, ev * (ew ** (-0.5))[None, :])


def _embBasis_phsymm(lattice, GRho, **kwargs):
    """BCS bath from quasiparticle embedding.  Phys. Rev. B, 93, 035126 (2016)."""
    if "sites" in kwargs:
        log.error('keyword "sites" not supported.')
    log.eassert(lattice.nval == lattice.nscsites, "Non-local bath does not support truncation.")
    ncells, nscsites = lattice.ncells, lattice.nscsites
    m = 2 * nscsites
    GRho = np.ascontiguousarray(np.asarray(GRho).real, dtype=np.float64)
    ctx = get_ctx()
    basis = np.empty((2, ncells, m, m))
    # particle part -> alpha spin
    d_G = ctx.to_device(GRho.reshape(ncells * m, m))
    d_BA1 = _rotate_rows(ctx, d_G, ncells * m, m, _inv_sqrt_factor(ctx, GRho[0]))
    basis[0] = _orthonormalize_dev(ctx, d_BA1, ncells * m, m).get().reshape(ncells, m, m)
    # hole part -> beta spin
    GRho_h = -GRho
    GRho_h[0] += np.eye(m)
    d_Gh = ctx.to_device(GRho_h.reshape(ncells * m, m))
    d_BA2 = _rotate_rows(ctx, d_Gh, ncells * m, m, _inv_sqrt_factor(ctx, GRho_h[0]))
    BA2 = _orthonormalize_dev(ctx, d_BA2, ncells * m, m).get().reshape(ncells, m, m)
    basis[1, :, :nscsites], basis[1, :, nscsites:] = BA2[:, nscsites:], BA2[:, :nscsites]
    return basis


# ---- BCS embedding Hamiltonian of model lattices (routine/bcs.py:137-318) --------------------------------------------------

def _embHam2e(lattice, basis, vcor, local, int_bath=False, last_aabb=True, **kwargs):
    """Two-body part: ({"ccdd", "cccd", "cccc"}, one-body terms from it or None, H0 from it).  The reference implements the local
    basis of a model with a bare bath: the lattice ERI sits on the impurity block of all three ccdd spin blocks, the anomalous
    terms are zero (bcs.py:198-229); the ERI is zero-padded on the device (dmk_pad_block_f64)."""
    from libdmet_preview_amd.routine.slater_helper import unit2emb
    n, nb = lattice.nscsites, basis.shape[-1]
    if not lattice.is_model:
        raise NotImplementedError                                   # bcs.py:253-254: no ab-initio BCS Hamiltonian in the reference
    if not local:
        raise NotImplementedError("quasiparticle (non-local) BCS Hamiltonian: libdmet.integral.integral_nonlocal_emb is outside the HIP path")
    if "sites" in kwargs:
        raise NotImplementedError('keyword "sites" is not supported')
    LatH2 = np.asarray(lattice.getH2(compact=False, kspace=False))
    for s in range(2):
        log.eassert(np.abs(basis[s, 0, :n, :n] - np.eye(n)).max() < 1e-10, "the embedding basis is not local")
    if lattice.H2_format != "local":
        # 'nearest' / 'full' write into an undefined name in the reference (bcs.py:214-225); anything else is its ValueError
        if lattice.H2_format in ("nearest", "full"):
            raise NotImplementedError("BCS model ERI in format %s" % lattice.H2_format)
        raise ValueError
    if int_bath:
        raise NotImplementedError
    ccdd = unit2emb(np.asarray((LatH2,) * 3), nb)
    log.info("H2 memory allocated size = %d MB", ccdd.size * 2 * 8. / 1024 / 1024)
    return {"ccdd": ccdd, "cccd": np.zeros((2,) + (nb,) * 4), "cccc": np.zeros((1,) + (nb,) * 4)}, None, 0.


def _embHam1e(lattice, basis, vcor, mu, H2_emb, int_bath=False, add_vcor=False, **kwargs):
    """One-body part (bcs.py:256-316): ((H1 {"cd", "cc"}, H0), (H1 for the energy, its H0)); all Nambu folds on the device
    (bcs_helper: one quadratic form of the canonical basis each)."""
    log.eassert(vcor.islocal(), "nonlocal correlation potential cannot be treated in this routine")
    n = lattice.nscsites
    hcore_R = np.asarray(lattice.getH1(kspace=False))
    ImpJK = lattice.getImpJK()
    if int_bath or not lattice.use_hcore_as_emb_ham:
        raise NotImplementedError                                   # bcs.py:278-287: only the bare bath on hcore exists
    lattice.JK_core = None
    cd, cc, H0 = transform_trans_inv(basis, lattice, hcore_R)       # noqa: F405
    cd, cc = np.array(cd), np.array(cc)
    shifted = np.array(vcor.get(), copy=True)
    shifted[0] -= mu * np.eye(n)
    shifted[1] -= mu * np.eye(n)
    terms = [(+1.0, transform_local(basis, lattice, shifted))]      # noqa: F405  vcor - mu everywhere ...
    if not kwargs.get("fitting", False):
        terms.append((-1.0, transform_imp(basis, lattice, np.asarray(vcor.get()))))   # noqa: F405  ... vcor off the impurity
    if ImpJK is not None:
        terms.append((-1.0, transform_imp(basis, lattice, np.asarray(ImpJK))))        # noqa: F405
    for sign, (tcd, tcc, t0) in terms:
        cd, cc, H0 = cd + sign * np.asarray(tcd), cc + sign * np.asarray(tcc), H0 + sign * t0
    ecd, ecc, e0 = transform_imp_env(basis, lattice, hcore_R)       # noqa: F405
    return ({"cd": cd, "cc": cc[np.newaxis]}, H0), ({"cd": np.asarray(ecd), "cc": np.asarray(ecc)[np.newaxis]}, e0)


def embHam(lattice, basis, vcor, mu, local=True, **kwargs):
    """BCS embedding Hamiltonian (bcs.py:137-155): (integral.Integral(norb, False, True, ...), (H1 for the energy, its H0))."""
    from libdmet_preview_amd.system import integral
    basis = np.asarray(basis)
    log.info("Two-body part")
    Int2e, Int1e_from2e, H0_from2e = _embHam2e(lattice, basis, vcor, local, **kwargs)
    log.info("One-body part")
    (Int1e, H0_from1e), (Int1e_energy, H0_energy_from1e) = _embHam1e(lattice, basis, vcor, mu, Int2e, **kwargs)
    if Int1e_from2e is not None:
        for target in (Int1e, Int1e_energy):
            target["cd"] += Int1e_from2e["cd"]
            target["cc"] += Int1e_from2e["cc"]
    return integral.Integral(basis.shape[-1], False, True, H0_from1e + H0_from2e, Int1e, Int2e), (Int1e_energy, H0_energy_from1e + H0_from2e)


get_emb_Ham = embHam


# ---- BCS correlation-potential fit in the embedding space (routine/bcs.py:356-530) -----------------------------------------

def _nambu(A, B, D):
    """[[A, D], [D^T, -B]] (bcs.py:389-392)."""
    nb = A.shape[-1]
    M = np.empty((2 * nb, 2 * nb))
    M[:nb, :nb], M[nb:, nb:], M[:nb, nb:], M[nb:, :nb] = A, -np.asarray(B), D, np.asarray(D).T
    return M


def FitVcorEmb(GRho, lattice, basis, vcor, mu, beta=np.inf, MaxIter=300, CG_check=False, BFGS=False, diff_criterion=None,
               imp_fit=False, fit_idx=None, **kwargs):
    """
    Fit the correlation potential (normal + pairing blocks) in the Nambu embedding space (bcs.py:356-530): minimise
    |GRho_emb[vcor] - GRho|_F / sqrt(2) over the parameters, GRho_emb the lower half of the spectrum of
    embH + V(param), a symmetric matrix of dimension 2 nbasis.  It is the Slater fit on one block -- all entries fitted, nbasis
    levels filled at T = 0, the Fermi function around the fixed `mu0` (default 0, `fix_mu` default True) at finite T -- so the
    objective and both analytic gradients run on the device (slater.EmbFitDevice with the Nambu operators handed over).
    """
    from libdmet_preview_amd.routine import slater
    log.eassert(imp_fit is False and fit_idx is None, "Only imp+bath fit is supported.")
    basis = np.asarray(basis, dtype=np.float64)
    param_begin = vcor.param.copy()
    n, nb = lattice.nscsites, basis.shape[-1]
    fock_R = np.asarray(lattice.getH1(kspace=False) if lattice.use_hcore_as_emb_ham else lattice.getFock(kspace=False))
    (HA, HB), HD, _ = transform_trans_inv(basis, lattice, fock_R)                                # noqa: F405
    shift = np.zeros((3, n, n))
    shift[0] = shift[1] = -mu * np.eye(n)                        # the zero potential with -mu on both normal blocks (bcs.py:403-409)
    (A0, B0), D0, _ = transform_local(basis, lattice, shift)                                     # noqa: F405
    embH = _nambu(np.asarray(HA), np.asarray(HB), np.asarray(HD)) + _nambu(np.asarray(A0), np.asarray(B0), np.asarray(D0))
    if not vcor.restricted and vcor.bogoliubov and not getattr(vcor, "bogo_res", False) and getattr(vcor, "_v_idx_diag", None) is None:
        table = get_dV_dparam(basis, lattice, vcor)             # noqa: F405  every parameter at once (bcs_helper.py:390-430)
    else:
        g = np.asarray(vcor.gradient())
        table = np.empty((vcor.length(), 2 * nb, 2 * nb))
        for ip in range(vcor.length()):
            (dA, dB), dD, _ = transform_local(basis, lattice, g[ip])                             # noqa: F405
            table[ip] = _nambu(np.asarray(dA), np.asarray(dB), np.asarray(dD))
    dim = 2 * nb
    tl = np.tril_indices(dim)
    ctx = get_ctx()
    d_dV = ctx.to_device(np.ascontiguousarray(table[:, tl[0], tl[1]]).reshape(vcor.length(), 1, len(tl[0])))
    finite = beta < np.inf
    fit = slater.EmbFitDevice(ctx, np.asarray(GRho)[np.newaxis], lattice, None, vcor, beta, nb, list(range(dim)), [],
                              None, None, mu0=kwargs.get("mu0", 0.0) if finite else None,
                              fix_mu=kwargs.get("fix_mu", True) if finite else False, eigh=kwargs.get("eigh", "jacobi"),
                              operators=(embH, np.eye(dim)), dV_table=d_dV, norm=np.sqrt(2.0))
    kwargs = dict(kwargs)
    kwargs.pop("mu0", None), kwargs.pop("fix_mu", None)
    return slater.drive_emb_fit(fit, vcor, param_begin, beta, MaxIter, CG_check, BFGS, diff_criterion, kwargs, FitVcorEmb,
                                grad_check_steps=(1e-4, 1e-5))


def FitVcorTwoStep(GRho, lattice, basis, vcor, mu, beta=np.inf, MaxIter1=300, MaxIter2=0, kinetic=False, CG_check=False, BFGS=False,
                   serial=True, method='CG', ytol=1e-7, gtol=5e-3, **kwargs):
    """Main wrapper of the BCS fit (bcs.py:621-664): the embedding-space stage on a copy of `vcor`; returns (vcor_new, err_end).  The
    kinetic-energy variant (`kinetic`) runs FitVcorFullK instead of both stages."""
    from copy import deepcopy
    vcor_new = deepcopy(vcor)
    log.result("Using two-step vcor fitting")
    if kinetic:
        log.check(MaxIter1 > 0, "Embedding fitting with kinetic energy minimization does not work!\nSkipping Embedding fitting")
        vcor_new, err_begin, err_end = FitVcorFullK(GRho, lattice, basis, vcor_new, mu, MaxIter=max(MaxIter2, 1))
        log.result("residue (begin) = %20.12f", err_begin)
        log.result("residue (end)   = %20.12f", err_end)
        return vcor_new, err_end
    log.eassert(MaxIter1 > 0 or MaxIter2 > 0, "FitVcorTwoStep: no stage to run (MaxIter1 = MaxIter2 = 0)")
    err_begin = None
    if MaxIter1 > 0:
        log.info("Impurity model stage max %d steps", MaxIter1)
        log.info("Finite temperature used in fitting? beta = %15.6f ", beta)
        vcor_new, err_begin, err_end = FitVcorEmb(GRho, lattice, basis, vcor_new, mu, beta=beta, MaxIter=MaxIter1, CG_check=CG_check,
                                                  serial=serial, BFGS=BFGS, method=method, ytol=ytol, gtol=gtol, **kwargs)
        log.info("Embedding Stage:\nbegin %20.12f    end %20.12f" % (err_begin, err_end))
    if MaxIter2 > 0:
        log.info("Full lattice stage  max %d steps", MaxIter2)
        vcor_new, err_begin2, err_end = FitVcorFull(GRho, lattice, basis, vcor_new, mu, MaxIter=MaxIter2, beta=beta, method=method,
                                                    ytol=ytol, gtol=gtol)
        log.info("Full Lattice Stage:\nbegin %20.12f    end %20.12f" % (err_begin2, err_end))
        err_begin = err_begin2 if err_begin is None else err_begin
    log.result("residue (begin) = %20.12f", err_begin)
    log.result("residue (end)   = %20.12f", err_end)
    return vcor_new, err_end


# ---- lattice stage of the BCS fit (routine/bcs.py:319-346, 532-562) --------------------------------------------------------

def foldRho(GRho, Lat, basis, thr=1e-7):
    """Generalised density stripe (ncells, 2 n, 2 n) into the Nambu embedding space (2 nbasis, 2 nbasis): sum_ij C_i^T GRho[i - j] C_j
    with the canonical basis C (bcs.py:319-343).  The stripe of a density matrix is Hermitian, so the double sum runs in k space
    on the device (slater_helper.transform_trans_inv); the reference's `thr` only skips blocks that are zero."""
    from libdmet_preview_amd.routine import slater_helper
    return slater_helper.transform_trans_inv(basisToCanonical(np.asarray(basis, dtype=np.float64)), Lat, np.asarray(GRho).real)   # noqa: F405


def foldRho_k(GRho_k, basis_k):
    from libdmet_preview_amd.routine import slater_helper
    return slater_helper.transform_trans_inv_k(np.asarray(basis_k), np.asarray(GRho_k))


def FitVcorFull(GRho, lattice, basis, vcor, mu, beta=np.inf, MaxIter=20, method='CG', ytol=1e-7, gtol=1e-2, **kwargs):
    """Lattice stage of the BCS fit (bcs.py:532-562): every evaluation is a full HFB mean field of the lattice (mfd.HFB: BdG
    matrices of all k on the device) folded into the Nambu embedding space; the gradient is numerical, as in the reference.  The
    reference re-centres the fold on a reference density through the minimiser's callback -- the fold is linear, so the value is
    the same and the callback is not needed here."""
    from libdmet_preview_amd.routine import mfd
    from libdmet_preview_amd.routine.fit import minimize
    quiet = log.verbose

    def errfunc(param, ref=None):
        vcor.update(param)
        log.verbose = "RESULT"
        try:
            GRhoT = mfd.HFB(lattice, vcor, False, mu=mu, beta=beta)[0]
        finally:
            log.verbose = quiet
        return np.linalg.norm(foldRho(GRhoT, lattice, basis, thr=1e-8) - GRho) / np.sqrt(2.0)

    err_begin = errfunc(vcor.param)
    param, err_end, pattern, gnorm_res = minimize(errfunc, vcor.param, MaxIter, method=method, ytol=ytol, gtol=gtol, **kwargs)
    vcor.update(param)
    FitVcorFull.last_errfunc = errfunc
    return vcor, err_begin, err_end


def FitVcorFullK(GRho, lattice, basis, vcor, mu, MaxIter, **kwargs):
    """Kinetic-energy form of the lattice fit (bcs.py:564-619): maximise the mean-field kinetic energy plus the constraint that the
    cell-0 blocks of the HFB density equal the impurity blocks of `GRho`, with the analytic gradient -dRho . dV/dparam and SciPy's
    default quasi-Newton driver, like the reference.  Every evaluation is an mfd.HFB of the lattice on the device.  Returns
    (vcor, constraint at the start, constraint at the end)."""
    from scipy.optimize import minimize as scipy_minimize
    from libdmet_preview_amd.routine import mfd
    n = lattice.nscsites
    FockT = np.asarray(lattice.getFock(kspace=False))
    rhoA_t, rhoB_t, kappa_t = extractRdm(np.asarray(GRho))                                       # noqa: F405
    target = rhoA_t[:n, :n], rhoB_t[:n, :n], kappa_t[:n, :n]
    quiet = log.verbose

    def mean_field(param):
        vcor.update(param)
        log.verbose = "RESULT"
        try:
            return mfd.HFB(lattice, vcor, False, mu=mu, beta=np.inf)[0]
        finally:
            log.verbose = quiet

    def costfunc(param, v=False):
        GRhoT = mean_field(param)
        blocks = [extractRdm(x) for x in GRhoT]                                                   # noqa: F405
        rhoAT, rhoBT = np.asarray([b[0] for b in blocks]), np.asarray([b[1] for b in blocks])
        kinetic = np.sum((rhoAT + rhoBT) * FockT)
        vm = np.asarray(vcor.get())
        dk = blocks[0][2] - target[2]
        constraint = (np.sum((vm[0] - mu * np.eye(n)) * (rhoAT[0] - target[0])) + np.sum((vm[1] - mu * np.eye(n)) * (rhoBT[0] - target[1]))
                      + np.sum(vm[2] * dk.T) + np.sum(vm[2].T * dk))
        return (kinetic, constraint) if v else -(kinetic + constraint)

    def grad(param):
        rhoA0, rhoB0, kappa0 = extractRdm(mean_field(param)[0])                                    # noqa: F405
        dRho = np.asarray([rhoA0 - target[0], rhoB0 - target[1], 2 * (kappa0.T - target[2].T)])
        return -np.tensordot(np.asarray(vcor.gradient()), dRho, axes=((1, 2, 3), (0, 1, 2)))

    ke_begin, c_begin = costfunc(vcor.param, v=True)
    log.info("begin: \\nkinetic energy = %20.12f    constraint = %20.12f", ke_begin, c_begin)
    param = scipy_minimize(costfunc, vcor.param, jac=grad).x
    ke_end, c_end = costfunc(param, v=True)
    log.info("end: \\nkinetic energy = %20.12f    constraint = %20.12f", ke_end, c_end)
    vcor.update(param)
    FitVcorFullK.last_cost = (costfunc, grad)
    return vcor, c_begin, c_end
