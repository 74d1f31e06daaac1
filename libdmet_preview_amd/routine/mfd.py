"""
Mean-field routines with the reference's entry points (libdmet/routine/mfd.py), the batched
k-point diagonalisation, density build and k->R fold running on the MI355X through libdmetk:

  DiagRHF / DiagUHF / DiagRHF_symm / DiagUHF_symm   mfd.py:33-108  -> dmk_eigh_batched (all k, s in one launch)
  DiagGHF[_symm] / DiagBdG[symm]                    mfd.py:591-641, 429-478 -> same kernel on (2 nlo) matrices
  HF                                                mfd.py:235-427 -> + dmk_occ_density + dmk_fold_k2R
  assignocc / check_nelec                           mfd.py:860-957 (host: sort + scalar root find)

Occupations stay on the host like the reference's brentq (SURVEY.md section 2.2 last row).
"""
import numpy as np

from libdmet_preview_amd._lib import lib, get_ctx
from libdmet_preview_amd.routine import ftsystem
from libdmet_preview_amd.settings import IMAG_DISCARD_TOL
from libdmet_preview_amd.system import fourier
from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.utils.misc import max_abs, add_spin_dim

try:
    from collections.abc import Iterable
except ImportError:  # pragma: no cover
    from collections import Iterable


# ---------------------------------------------------------------------------------------------
# device-resident core
# ---------------------------------------------------------------------------------------------

def eigh_dev(ctx, d_F, n, batch, d_add=None, add_group=0):
    """ew (batch, n) ascending, Vt (batch, n, n) with ROW m = eigenvector m; both device arrays."""
    d_w = ctx.empty((batch, n), np.float64)
    d_Vt = ctx.empty((batch, n, n), np.complex128)
    ctx.check(lib.dmk_eigh_batched(ctx.h, int(n), int(batch), d_F.ptr, d_add.ptr if d_add is not None else None,
                                   int(add_group), d_w.ptr, d_Vt.ptr))
    return d_w, d_Vt


def density_dev(ctx, d_Vt, d_occ, n, batch):
    d_rho = ctx.empty((batch, n, n), np.complex128)
    ctx.check(lib.dmk_occ_density(ctx.h, int(n), int(batch), d_Vt.ptr, d_occ.ptr, d_rho.ptr))
    return d_rho


def _vt_to_ev(ctx, d_Vt, n, batch):
    d_ev = ctx.empty((batch, n, n), np.complex128)
    ctx.check(lib.dmk_transpose_c128(ctx.h, int(n), int(n), int(batch), d_Vt.ptr, d_ev.ptr))
    return d_ev


def _vcor_mat(vcor, nspin_needed):
    """vcor.get(i, True): one (2|3, nlo, nlo) matrix for every k (routine/vcor.py:36-47)."""
    if vcor is None:
        return None
    v = np.asarray(vcor.get(0, True))
    if v.ndim == 2:
        v = v[None]
    return np.ascontiguousarray(v[:nspin_needed].real, dtype=np.float64)


def _diag(Fock, vcor, spin, symm_neg=None):
    """Shared body of Diag*: returns ew (spin, nk, n), ev (spin, nk, n, n) as numpy (reference layout)."""
    ctx = get_ctx()
    Fock = np.asarray(Fock)
    nk, n = Fock.shape[-3], Fock.shape[-1]
    d_F = ctx.to_device(Fock.reshape(spin * nk, n, n), np.complex128)
    v = _vcor_mat(vcor, spin)
    d_add = ctx.to_device(v) if v is not None else None
    d_w, d_Vt = eigh_dev(ctx, d_F, n, spin * nk, d_add, nk)
    ew = d_w.get().reshape(spin, nk, n)
    ev = _vt_to_ev(ctx, d_Vt, n, spin * nk).get().reshape(spin, nk, n, n)
    if symm_neg is not None:
        # k / -k symmetry of the reference: the later member of a pair takes the conjugate (mfd.py:56-66)
        computed = set()
        for i in range(nk):
            ni = int(symm_neg[i])
            if ni in computed:
                ew[:, i] = ew[:, ni]
                ev[:, i] = ev[:, ni].conj()
            else:
                computed.add(i)
    return ew, ev


def DiagRHF(Fock, vcor, **kwargs):
    Fock = np.asarray(Fock)
    if Fock.ndim == 3:
        Fock = Fock[np.newaxis]
    ew, ev = _diag(Fock[:1], vcor, 1)
    return ew[0], ev[0]


def DiagRHF_symm(Fock, vcor, lattice, **kwargs):
    Fock = np.asarray(Fock)
    if Fock.ndim == 3:
        Fock = Fock[np.newaxis]
    neg = [lattice.cell_pos2idx(-lattice.cell_idx2pos(i)) for i in range(Fock.shape[-3])]
    ew, ev = _diag(Fock[:1], vcor, 1, symm_neg=neg)
    return ew[0], ev[0]


def DiagUHF(Fock, vcor, **kwargs):
    Fock = np.asarray(Fock)
    if Fock.ndim == 3:
        Fock = np.asarray((Fock, Fock))
    return _diag(Fock[:2], vcor, 2)


def DiagUHF_symm(Fock, vcor, lattice, **kwargs):
    Fock = np.asarray(Fock)
    if Fock.ndim == 3:
        Fock = np.asarray((Fock, Fock))
    neg = [lattice.cell_pos2idx(-lattice.cell_idx2pos(i)) for i in range(Fock.shape[-3])]
    return _diag(Fock[:2], vcor, 2, symm_neg=neg)


def _diag_nambu(A, add, symm_lattice=None):
    """Batched eigh of (nk, m, m) complex matrices + one real (m, m) shift shared by all k; lower triangle only."""
    ctx = get_ctx()
    nk, m = A.shape[0], A.shape[-1]
    d_A = ctx.to_device(A, np.complex128)
    d_add = ctx.to_device(add[None], np.float64)
    d_w, d_Vt = eigh_dev(ctx, d_A, m, nk, d_add, nk)
    ew = d_w.get().reshape(nk, m)
    ev = _vt_to_ev(ctx, d_Vt, m, nk).get().reshape(nk, m, m)
    if symm_lattice is not None:
        computed = set()
        for i in range(nk):
            ni = symm_lattice.cell_pos2idx(-symm_lattice.cell_idx2pos(i))
            if ni in computed:
                ew[i], ev[i] = ew[ni], ev[ni].conj()
            else:
                computed.add(i)
    return ew, ev


def _ghf_shift(vcor, nao, mu):
    """The k-independent part of mfd.py:597-608: vcor blocks (lower triangle convention) and -/+ mu."""
    v = np.asarray(vcor.get(0, True))
    if np.iscomplexobj(v) and max_abs(v.imag) > 0.0:
        raise NotImplementedError("complex correlation potential is outside the HIP path")
    v = v.real
    add = np.zeros((2 * nao, 2 * nao))
    add[:nao, :nao] = v[0]
    add[nao:, nao:] = v[1]
    add[nao:, :nao] = v[2].T
    add[:nao, nao:] = v[2]
    if mu is not None:
        add[range(nao), range(nao)] -= mu
        add[range(nao, 2 * nao), range(nao, 2 * nao)] += mu
    return add


def DiagGHF(GFock, vcor, mu, **kwargs):
    """mfd.py:591-610: generalised (spin-orbital) Fock, one eigh per k on the device."""
    GFock = np.asarray(GFock)
    nao = GFock.shape[-1] // 2
    return _diag_nambu(GFock, _ghf_shift(vcor, nao, mu))


def DiagGHF_symm(GFock, vcor, mu, lattice, **kwargs):
    """mfd.py:612-641."""
    GFock = np.asarray(GFock)
    nao = GFock.shape[-1] // 2
    return _diag_nambu(GFock, _ghf_shift(vcor, nao, mu), symm_lattice=lattice)


def _bdg_matrix(Fock):
    """Block-diagonal (F_a, -F_b) part of the BdG matrix (mfd.py:439-447); vcor and mu enter as the shift."""
    Fock = np.asarray(Fock)
    if Fock.ndim == 3:
        Fock = np.asarray((Fock, Fock))
    nk, n = Fock.shape[-3], Fock.shape[-1]
    A = np.zeros((nk, 2 * n, 2 * n), dtype=np.complex128)
    A[:, :n, :n] = Fock[0]
    A[:, n:, n:] = -Fock[1]
    return A, n


def _bdg_shift(vcor, n, mu):
    v = np.asarray(vcor.get(0, True))
    if np.iscomplexobj(v) and max_abs(v.imag) > 0.0:
        raise NotImplementedError("complex correlation potential is outside the HIP path")
    v = v.real
    add = np.zeros((2 * n, 2 * n))
    add[:n, :n] = v[0] - mu * np.eye(n)
    add[n:, n:] = -v[1] + mu * np.eye(n)
    add[:n, n:] = v[2]
    add[n:, :n] = v[2].T
    return add


def DiagBdG(Fock, vcor, mu, **kwargs):
    """mfd.py:429-449."""
    A, n = _bdg_matrix(Fock)
    return _diag_nambu(A, _bdg_shift(vcor, n, mu))


def DiagBdGsymm(Fock, vcor, mu, lattice, **kwargs):
    """mfd.py:451-478."""
    A, n = _bdg_matrix(Fock)
    return _diag_nambu(A, _bdg_shift(vcor, n, mu), symm_lattice=lattice)


# ---------------------------------------------------------------------------------------------
# occupations (host)
# ---------------------------------------------------------------------------------------------

def check_nelec(nelec, ncells=None, tol=1e-5):
    nelec_round = int(np.round(nelec))
    if abs(nelec - nelec_round) > tol:
        log.warn("HF: nelec is rounded to integer nelec = %d (original %.2f)", nelec_round, nelec)
    nelec = nelec_round
    if ncells is None:
        nelec_per_cell = None
    else:
        nelec_per_cell = nelec / float(ncells)
        if abs(nelec_per_cell - np.round(nelec_per_cell)) > tol:
            log.warn("HF: nelec per cell (%.5f) is not an integer.", nelec_per_cell)
        else:
            nelec_per_cell = int(np.round(nelec_per_cell))
    return nelec, nelec_per_cell


def assignocc(ew, nelec, beta, mu0=0.0, fix_mu=False, thr_deg=1e-6, Sz=None, fit_tol=1e-12,
              f_occ=ftsystem.fermi_smearing_occ, ncore=0, nvirt=0):
    """Occupation numbers of a mean field (mfd.py:887-957). nelec is per spin for RHF, total for UHF."""
    ew = np.asarray(ew)
    if (Sz is None) and (not isinstance(nelec, Iterable)):
        if beta < np.inf:
            if ncore == 0 and nvirt == 0:
                ew_sorted = np.sort(ew, axis=None, kind="mergesort")
                if fix_mu:
                    mu = mu0
                else:
                    mu = ftsystem.find_mu(nelec, ew_sorted, beta, mu0=mu0, tol=fit_tol, f_occ=f_occ)
                ewocc = f_occ(mu, ew, beta)
                nerr = abs(np.sum(ewocc) - nelec)
            else:
                idx = np.argsort(ew, axis=None, kind="mergesort")
                ew_sorted = ew.ravel()[idx]
                idx_re = np.argsort(idx, kind="mergesort")
                if fix_mu:
                    mu = mu0
                else:
                    mu = ftsystem.find_mu(nelec, ew_sorted, beta, mu0=mu0, tol=fit_tol, f_occ=f_occ,
                                          ncore=ncore, nvirt=nvirt)
                ewocc = f_occ(mu, ew_sorted, beta, ncore=ncore, nvirt=nvirt)[idx_re]
                ewocc = ewocc.reshape(ew.shape)
                nerr = abs(np.sum(ewocc) - nelec)
        else:
            ew_sorted = np.sort(ew, axis=None, kind="mergesort")
            nelec = check_nelec(nelec, None)[0]
            if np.sum(ew < mu0 - thr_deg) <= nelec and np.sum(ew <= mu0 + thr_deg) >= nelec:
                mu = mu0
            else:
                mu = 0.5 * (ew_sorted[nelec - 1] + ew_sorted[nelec])
            ewocc = 1.0 * (ew < mu - thr_deg)
            nremain_elec = nelec - np.sum(ewocc)
            if nremain_elec > 0:
                remain_orb = np.logical_and(ew <= mu + thr_deg, ew >= mu - thr_deg)
                nremain_orb = np.sum(remain_orb)
                log.warn("degenerate HOMO-LUMO, assign fractional occupation\n"
                         "%d electrons assigned to %d orbitals", nremain_elec, nremain_orb)
                ewocc += (float(nremain_elec) / nremain_orb) * remain_orb
            nerr = 0.0
    else:
        spin = ew.shape[0]
        assert spin == 2
        if not isinstance(nelec, Iterable):
            nelec = [(nelec + Sz) * 0.5, (nelec - Sz) * 0.5]
        if not isinstance(mu0, Iterable):
            mu0 = [mu0 for s in range(spin)]
        ewocc = np.empty_like(ew)
        mu = np.zeros((spin,))
        nerr = np.zeros((spin,))
        for s in range(2):
            ewocc[s], mu[s], nerr[s] = assignocc(ew[s], nelec[s], beta, mu0[s], fix_mu=fix_mu, thr_deg=thr_deg,
                                                 fit_tol=fit_tol, f_occ=f_occ, ncore=ncore, nvirt=nvirt)
    return ewocc, mu, nerr


# ---------------------------------------------------------------------------------------------
# HF
# ---------------------------------------------------------------------------------------------

def _mu0_guess(ew_sorted, nelec):
    if nelec <= 0:
        return ew_sorted[0]
    elif nelec >= len(ew_sorted):
        return ew_sorted[-1]
    return 0.5 * (ew_sorted[nelec - 1] + ew_sorted[nelec])


def HF(lattice, vcor, filling, restricted, mu0=None, beta=np.inf, ires=False, scf=False, use_hcore=None,
       **kwargs):
    """
    RHF and UHF routine for lattice problems (mfd.py:235-427).

    Returns rho (spin, ncells, nao, nao), mu, E (per cell, with vcor contribution) and, with
    ires=True, a dict with keys gap, e, coef, nerr, rho_k, E0, E, mo_occ, homo, lumo.
    kwargs: symm, fix_mu, tol_deg, nfrac.
    """
    log.eassert(beta >= 0, "beta cannot be negative")
    if scf:
        raise NotImplementedError("scf=True needs a PySCF KSCF object (out of scope of the HIP path)")
    if use_hcore is None:
        use_hcore = lattice.use_hcore_as_emb_ham
    if use_hcore:
        Fock = lattice.getH1(kspace=True)
        FockT = H1T = lattice.getH1(kspace=False)
    else:
        Fock = lattice.getFock(kspace=True)
        FockT = lattice.getFock(kspace=False)
        H1T = lattice.getH1(kspace=False)

    ctx = get_ctx()
    Fock = np.asarray(Fock)
    symm = kwargs.get("symm", False)
    if restricted:
        log.info("Restricted Hartree-Fock")
        if Fock.ndim == 3:
            Fock = Fock[np.newaxis]
        Fock = Fock[:1]
        spin = 1
    else:
        log.info("Unrestricted Hartree-Fock")
        if Fock.ndim == 3:
            Fock = np.asarray((Fock, Fock))
        Fock = Fock[:2]
        spin = 2
    nkpts, n = Fock.shape[-3], Fock.shape[-1]
    kmesh = lattice.kmesh

    # ---- diagonalisation: every (s, k) in one launch; eigenvectors stay on the device ------
    d_F = ctx.to_device(Fock.reshape(spin * nkpts, n, n), np.complex128)
    v = _vcor_mat(vcor, spin)
    d_add = ctx.to_device(v) if v is not None else None
    d_w, d_Vt = eigh_dev(ctx, d_F, n, spin * nkpts, d_add, nkpts)
    ew = d_w.get().reshape(spin, nkpts, n)
    if symm:
        neg = [lattice.cell_pos2idx(-lattice.cell_idx2pos(i)) for i in range(nkpts)]
        computed = set()
        for i in range(nkpts):
            if neg[i] in computed:
                ew[:, i] = ew[:, neg[i]]
            else:
                computed.add(i)

    # ---- occupancy (host) --------------------------------------------------------------------
    if isinstance(filling, Iterable):
        nelec = [ew.size * filling[0] * 0.5, ew.size * filling[1] * 0.5]
        nelec[0], nelec[1] = check_nelec(nelec[0], None)[0], check_nelec(nelec[1], None)[0]
        ew_sorted = [np.sort(ew[s], axis=None, kind="mergesort") for s in range(2)]
        if mu0 is None:
            mu0 = [_mu0_guess(ew_sorted[0], nelec[0]), _mu0_guess(ew_sorted[1], nelec[1])]
    else:
        nelec = ew.size * filling
        nelec = check_nelec(nelec, None)[0]
        ew_sorted = np.sort(ew, axis=None, kind="mergesort")
        if mu0 is None:
            mu0 = _mu0_guess(ew_sorted, nelec)

    fix_mu = kwargs.get("fix_mu", False)
    tol_deg = kwargs.get("tol_deg", 1e-6)
    nfrac = kwargs.get("nfrac", None)
    if nfrac is None:
        ncore = nvirt = 0
    else:
        if restricted:
            ncore = nelec - nfrac
            nvirt = ew.size - (nelec + nfrac)
        else:
            ncore = (nelec // 2 - nfrac)
            nvirt = ew.size // 2 - (nelec // 2 + nfrac)
    ewocc, mu, nerr = assignocc(ew, nelec, beta, mu0, fix_mu=fix_mu, thr_deg=tol_deg, ncore=ncore, nvirt=nvirt)

    # ---- density matrix: rho_k = (ev occ) ev^H, rhoT = k2R(rho_k) --------------------------------
    d_occ = ctx.to_device(ewocc.reshape(spin * nkpts, n), np.float64)
    d_rho = density_dev(ctx, d_Vt, d_occ, n, spin * nkpts)
    d_imax = ctx.zeros((1,), np.float64)
    d_rhoT = fourier.fold_k2R_dev(d_rho.reshape(spin, nkpts, n * n), kmesh, spin, n * n, imag_max=d_imax)
    rhoT = d_rhoT.get().reshape(spin, nkpts, n, n)
    imag = float(d_imax.get()[0])
    if imag > IMAG_DISCARD_TOL:
        log.warn("k2R: non-zero imaginary part: %15.8g", imag)

    # ---- energy -------------------------------------------------------------------------------
    FockT = add_spin_dim(FockT, spin)
    H1T = add_spin_dim(H1T, spin)
    vcorT = None if vcor is None else np.asarray(vcor.get(0, kspace=False))
    if vcor is not None and not vcor.islocal():
        raise NotImplementedError("non-local vcor")
    if spin == 1:
        E0 = np.sum((FockT + H1T) * rhoT) + lattice.getH0()
        E = E0 + (np.sum(vcorT[0] * rhoT[0, 0]) if vcorT is not None else 0.0)
    else:
        E0 = 0.5 * np.sum((FockT + H1T) * rhoT) + lattice.getH0()
        E = E0 + (0.5 * np.sum(vcorT[0] * rhoT[0, 0] + vcorT[1] * rhoT[1, 0]) if vcorT is not None else 0.0)
    if max_abs(np.imag(E)) > IMAG_DISCARD_TOL:
        log.warn("E.imag = %e", np.imag(E))
    E = float(np.real(E))

    if ires:
        rho = d_rho.get().reshape(spin, nkpts, n, n)
        ev = _vt_to_ev(ctx, d_Vt, n, spin * nkpts).get().reshape(spin, nkpts, n, n)
        if symm:
            computed = set()
            for i in range(nkpts):
                if neg[i] in computed:
                    ev[:, i] = ev[:, neg[i]].conj()
                else:
                    computed.add(i)
        if isinstance(mu, Iterable):
            homo, lumo = [], []
            for s in range(2):
                hi = max(np.searchsorted(ew_sorted[s], mu[s], side="right") - 1, 0)
                li = min(np.searchsorted(ew_sorted[s], mu[s], side="left"), len(ew_sorted[s]) - 1)
                homo.append(ew_sorted[s][hi])
                lumo.append(ew_sorted[s][li])
            gap = np.array((lumo[0] - homo[0], lumo[1] - homo[1]))
            homo, lumo = tuple(homo), tuple(lumo)
        else:
            hi = max(np.searchsorted(ew_sorted, mu, side="right") - 1, 0)
            li = min(np.searchsorted(ew_sorted, mu, side="left"), len(ew_sorted) - 1)
            homo, lumo = ew_sorted[hi], ew_sorted[li]
            gap = lumo - homo
        res = {"gap": gap, "e": ew, "coef": ev, "nerr": nerr, "rho_k": rho, "E0": E0, "E": E,
               "mo_occ": ewocc, "homo": homo, "lumo": lumo}
        return rhoT, mu, E, res
    return rhoT, mu, E
