"""
Finite-temperature occupation helpers behind the reference's names (libdmet/routine/ftsystem.py:24-105).

The production path never calls the host functions below: occupations and the chemical potential of a mean field
are computed on the device by libdmetk's `dmk_assign_occ` (csrc/occ.hip, reached through `mfd.assignocc`).  What
stays here is (i) the elementwise Fermi function on host arrays, which the vcor-fit gradient evaluates on a few
hundred embedding levels, and (ii) a host root finder for the cases the device kernel does not cover -- a
user-supplied smearing function `f_occ`, or frozen core / virtual levels (`ncore`, `nvirt`).
"""
import numpy as np
from scipy.optimize import brentq

from libdmet_preview_amd.utils import logger as log

FIT_TOL = 1e-12
ZERO_TOL = 1e-10
_EXP_CUTOFF = 100.0        # levels with beta (e - mu) >= 100 are empty (ftsystem.py:43)


def fermi_smearing_occ(mu, mo_energy, beta, ncore=0, nvirt=0):
    """1 / (exp(beta (e - mu)) + 1), exactly 0 beyond the cut-off; `mu` is a scalar or one value per leading (spin)
    entry of `mo_energy`.  The first `ncore` / last `nvirt` entries of a 1-d spectrum are pinned to 1 / 0."""
    e = np.asarray(mo_energy, dtype=float)
    level = np.asarray(mu, dtype=float).reshape((-1,) + (1,) * (e.ndim - 1))
    x = beta * (e - level)
    with np.errstate(over="ignore", invalid="ignore"):
        occ = np.where(x < _EXP_CUTOFF, 1.0 / (np.exp(np.minimum(x, _EXP_CUTOFF)) + 1.0), 0.0)
    if ncore or nvirt:
        if e.ndim != 1:
            raise AssertionError("frozen levels need a flat, sorted spectrum")
        occ[:ncore] = 1.0
        if nvirt:
            occ[e.size - nvirt:] = 0.0
    return occ


def find_mu(nelec, mo_energy, beta, mu0=None, f_occ=fermi_smearing_occ, tol=FIT_TOL, ncore=0, nvirt=0):
    """Chemical potential at which `f_occ` puts `nelec` electrons on the ascending levels `mo_energy` (host fall-back
    of dmk_assign_occ).  The electron count is monotone in mu, so any sign-changing bracket holds the unique root: the
    bracket starts one thermal width outside the two frontier levels and doubles outwards until the count straddles
    nelec; Brent's method then polishes it to `tol`."""
    e = np.asarray(mo_energy, dtype=float).ravel()
    excess = lambda level: float(np.sum(f_occ(level, e, beta, ncore=ncore, nvirt=nvirt))) - nelec
    top = e.size - 1
    below = min(max(int(round(nelec)) - 1, 0), top)
    above = min(max(int(round(nelec)), 0), top)
    width = 1.0 / beta
    lo, hi = e[below] - width, e[above] + width
    reach = max(width, 1.0)
    for _ in range(64):
        f_lo, f_hi = excess(lo), excess(hi)
        if f_lo <= 0.0 <= f_hi:
            break
        if f_lo > 0.0:
            lo -= reach
        if f_hi < 0.0:
            hi += reach
        reach *= 2.0
    else:
        raise ValueError("find_mu: no chemical potential gives %s electrons on %d levels" % (nelec, e.size))
    if f_lo == 0.0:
        return lo
    if f_hi == 0.0:
        return hi
    root, info = brentq(excess, lo, hi, xtol=tol, rtol=tol, maxiter=10000, full_output=True, disp=False)
    if not info.converged:
        log.warn("fitting mu (fermi level) brentq fails.")
    return root


def gaussian_smearing_occ(mu, mo_energy, beta, ncore=0, nvirt=0):
    """erfc((e - mu) beta) / 2 (ftsystem.py:56-70); same broadcasting of `mu` as fermi_smearing_occ."""
    from scipy.special import erfc
    e = np.asarray(mo_energy, dtype=float)
    level = np.asarray(mu, dtype=float).reshape((-1,) + (1,) * (e.ndim - 1))
    return 0.5 * erfc((e - level) * beta)


def find_mu_by_density(density, mo_energy, beta, mu0=None, f_occ=fermi_smearing_occ, tol=FIT_TOL, ncore=0, nvirt=0):
    """find_mu for an electron density per level (ftsystem.py:107-113)."""
    norb = np.asarray(mo_energy).size
    return find_mu(density * norb, mo_energy, beta, mu0=mu0, f_occ=f_occ, tol=tol * norb, ncore=ncore, nvirt=nvirt)


# ---- finite-temperature mean field of ONE matrix and its responses (ftsystem.py:115-297), on the device ------------------------------
# kernel / make_rdm1 are the single-matrix form of the lattice mean field (dmk_eigh_batched, dmk_assign_occ, dmk_occ_density);
# get_rho_grad / get_dw_dv are the response formulas FitVcorEmb's gradients use, as stand-alone functions with the reference's
# signatures (the fit itself keeps them fused with its own resident buffers, routine/slater.py EmbFitDevice).

def kernel(h, nelec, beta, mu0=None, fix_mu=False):
    """(mo_energy, mo_coeff, mo_occ, mu) of a Hermitian matrix at inverse temperature beta (ftsystem.py:115-122)."""
    from libdmet_preview_amd._lib import get_ctx
    from libdmet_preview_amd.routine import mfd
    h = np.asarray(h)
    n = h.shape[-1]
    ctx = get_ctx()
    d_w, d_Vt = mfd.eigh_dev(ctx, ctx.to_device(np.ascontiguousarray(h).reshape(1, n, n), np.complex128), n, 1)
    mo_energy = d_w.get()[0]
    vt = d_Vt.get()[0]                                     # rows = eigenvectors
    mo_coeff = np.ascontiguousarray(vt.T if np.iscomplexobj(h) else vt.T.real)
    if fix_mu:
        mu = mu0
        mo_occ = fermi_smearing_occ(mu, mo_energy, beta)
    else:
        _, mu, _ = mfd.assignocc_dev(ctx, d_w, nelec, beta, mu0=mu0)
        mo_occ = fermi_smearing_occ(mu, mo_energy, beta)
    return mo_energy, mo_coeff, mo_occ, mu


def make_rdm1(mo_coeff, mo_occ):
    """(C occ) C^H (ftsystem.py:124-125) through dmk_occ_density."""
    from libdmet_preview_amd._lib import get_ctx
    from libdmet_preview_amd.routine import mfd
    C = np.asarray(mo_coeff)
    n = C.shape[-1]
    ctx = get_ctx()
    d_Vt = ctx.to_device(np.ascontiguousarray(C.T).reshape(1, n, C.shape[0]), np.complex128)
    out = mfd.density_dev(ctx, d_Vt, ctx.to_device(np.ascontiguousarray(mo_occ, dtype=np.float64).reshape(1, n)), n, 1).get()[0]
    return out if np.iscomplexobj(C) else np.ascontiguousarray(out.real)


def get_h_random(norb, seed=None):
    """Random symmetric test matrix from numpy's global generator (ftsystem.py:127-132)."""
    if seed is not None:
        np.random.seed(seed)
    h = np.random.random((norb, norb))
    return h + h.T.conj()


def get_h_random_deg(norb, deg_orbs=[], deg_energy=[], seed=None):
    """The same with prescribed degenerate levels: the spectrum of get_h_random with the levels `deg_orbs[i]` set to `deg_energy[i]`
    (ftsystem.py:134-145); eigenpairs from the device eigensolver, recomposed by dmk_occ_density with the levels as weights."""
    from libdmet_preview_amd._lib import get_ctx
    from libdmet_preview_amd.routine import mfd
    h = get_h_random(norb, seed)
    ctx = get_ctx()
    d_w, d_Vt = mfd.eigh_dev(ctx, ctx.to_device(h.reshape(1, norb, norb), np.complex128), norb, 1)
    e = d_w.get()[0]
    for orbs, val in zip(deg_orbs, deg_energy):
        e[orbs] = val
    return np.ascontiguousarray(mfd.density_dev(ctx, d_Vt, ctx.to_device(e.reshape(1, norb)), norb, 1).get()[0].real)


def _occupation_kernel(mo_energy, mu, beta, sign):
    """K[p][q] = sign (f_p - f_q) / (e_p - e_q) with the degenerate limit sign (-beta) f_p (1 - f_q) (ftsystem.py:170-182, 244-257),
    one spin channel, from dmk_fit_kmat."""
    import ctypes as C
    from libdmet_preview_amd._lib import lib, get_ctx
    ctx = get_ctx()
    e = np.ascontiguousarray(mo_energy, dtype=np.float64).reshape(1, -1)
    n = e.shape[1]
    f = np.ascontiguousarray(fermi_smearing_occ(mu, e[0], beta), dtype=np.float64).reshape(1, n)
    d_K, d_e, d_f = ctx.empty((1, n, n), np.float64), ctx.to_device(e), ctx.to_device(f)       # (named: alive until the read-back)
    ctx.check(lib.dmk_fit_kmat(ctx.h, n, 1, d_e.ptr, d_f.ptr, float(beta), 0, d_K.ptr))
    return sign * d_K.get()[0], f[0]


def get_rho_grad(mo_energy, mo_coeff, mu, beta, fix_mu=True, compact=False):
    """d rho_{ij} / d v_{kl}, kl over the lower triangle of a symmetric potential (ftsystem.py:147-221): (npair, norb, norb), or
    (npair, npair) with compact=True.  The norb^4 contraction -(C* x C) K (C x C*) runs as two device products
    (dmk_zgemm_batched); the index shuffles, the symmetrisation and the chemical-potential term (rank one) are host bookkeeping."""
    from libdmet_preview_amd._lib import get_ctx
    from libdmet_preview_amd.utils import devmat
    C = np.asarray(mo_coeff)
    n = C.shape[-1]
    K, f = _occupation_kernel(mo_energy, mu, beta, -1.0)          # (f_q - f_p) / (e_p - e_q), degenerate limit +beta f (1 - f)
    ctx = get_ctx()
    scr = (C.conj()[:, None, :] * C[None, :, :]).reshape(n * n, n)                       # [(l m), p] = C*_lp C_mp
    A = devmat.mm(ctx, "N", devmat.up(ctx, scr), "N", devmat.up(ctx, K))                  # (l m), q
    G = devmat.mm(ctx, "N", A, "T", devmat.up(ctx, scr), alpha=-1.0).get()[0]             # (l m), (n s)
    g = G.reshape(n, n, n, n).transpose(0, 3, 1, 2)                                       # [l, s, m, n]
    g = g + g.transpose(1, 0, 2, 3)
    g[np.arange(n), np.arange(n)] *= 0.5
    tl = np.tril_indices(n)
    g = g[tl]
    if not fix_mu:
        ff = f * (1.0 - f)
        fsum = ff.sum()
        if abs(fsum) > ZERO_TOL:
            drho_dmu = make_rdm1(C, ff) * beta
            mg = np.einsum('ki,li,i->kl', C.conj(), C, ff) / fsum
            mg = mg + mg.T
            mg[np.arange(n), np.arange(n)] *= 0.5
            g = g + mg[tl][:, None, None] * drho_dmu[None]
    if not np.iscomplexobj(C):
        g = g.real
    if compact:
        g = g.transpose(1, 2, 0)[tl].transpose(1, 0)
    return np.ascontiguousarray(g)


def get_dw_dv(mo_energy, mo_coeff, drho, mu, beta, fix_mu=True, compact=False, fit_idx=None):
    """d w / d v for w = |rho[fit, fit] - target|^2 through the finite-T response (ftsystem.py:223-297): (spin, norb, norb) or the
    doubled-off-diagonal tril packing (spin, npair).  Per spin: C (C[fit]^T 2 drho C[fit]* o K) C^T on the device + the
    chemical-potential term."""
    from libdmet_preview_amd._lib import get_ctx
    from libdmet_preview_amd.utils import devmat
    mo_energy, mo_coeff, drho = np.asarray(mo_energy), np.asarray(mo_coeff), np.asarray(drho)
    if mo_coeff.ndim == 2:
        mo_energy, mo_coeff, drho = mo_energy[None], mo_coeff[None], drho[None]
    spin, _, n = mo_coeff.shape
    fit_idx = list(range(n)) if fit_idx is None else list(fit_idx)
    mus = np.asarray(mu, dtype=float).reshape(-1)
    ctx = get_ctx()
    out = np.zeros((spin, n, n), dtype=mo_coeff.dtype)
    for s in range(spin):
        C = mo_coeff[s]
        K, f = _occupation_kernel(mo_energy[s], mus[s if mus.size > 1 else 0], beta, 1.0)
        Cf = devmat.up(ctx, C[fit_idx])
        t = devmat.mm(ctx, "T", Cf, "N", devmat.mm(ctx, "N", devmat.up(ctx, 2.0 * drho[s]), "N", devmat.up(ctx, C[fit_idx].conj()))).get()[0] * K
        full = devmat.mm(ctx, "N", devmat.up(ctx, C.conj()), "N", devmat.mm(ctx, "N", devmat.up(ctx, t), "T", devmat.up(ctx, C))).get()[0]
        if not fix_mu:
            ff = f * (1.0 - f)
            fsum = ff.sum()
            if abs(fsum) > ZERO_TOL:
                drho_dmu = make_rdm1(C, ff)
                dw_dmu = np.einsum('ij,ij->', drho[s], drho_dmu[np.ix_(fit_idx, fit_idx)]) * 2.0 * beta
                full = full + drho_dmu * (dw_dmu / fsum)
        out[s] = full if np.iscomplexobj(out) else full.real
    if compact:
        tl = np.tril_indices(n)
        packed = np.asarray([m[tl] for m in out]) * 2.0
        packed[:, np.cumsum([0] + list(range(2, n + 1)))] *= 0.5
        return packed
    return out
