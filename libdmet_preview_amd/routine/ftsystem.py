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
