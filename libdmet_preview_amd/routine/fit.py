"""
Optimiser drivers of the correlation-potential fit with the reference's entry point
(libdmet/routine/fit.py:17-45 `minimize`; fit.py:47-187, fit_helper.py:174-484).

Host control flow only: the objective / gradient callables they drive are the device evaluations of
routine/slater.py (FitVcorEmb).  Same algorithms and stopping rules as the reference so that the iterates
coincide: steepest descent, Polak-Ribiere(+) conjugate gradient and BFGS, each with the bounded scalar line
search `minimize_scalar(bounds=(0, scale))`, scale = max(|mean of the last two steps|, min_step), and the
Nelder-Mead fallback when the bounded search lands above f(0).  Norms are max-abs (fit_helper.py:33).
Round 6: the trust-region Newton-CG driver (fit.py:217-288 over fit_helper.py:486-668): Steihaug's truncated CG for the
subproblem, Hessian-vector products from central differences of the gradient.  CIAH needs PySCF's solver and raises.
"""
import numpy as np
from scipy.optimize import minimize_scalar, fmin

from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.utils.misc import max_abs

norm = max_abs


def _numeric_grad(fn, callback, eps, diag_idx=None):
    """Central differences, one parameter at a time (fit.py:166-184); diag_idx groups have their mean removed."""
    def grad(x):
        if callback is not None:
            ref = callback(x)
            fn1 = lambda x1: fn(x1, ref=ref)
        else:
            fn1 = fn
        g = np.empty(len(x))
        for ix in range(len(x)):
            dx = np.zeros_like(x)
            dx[ix] = eps
            g[ix] = (0.5 / eps) * (fn1(x + dx) - fn1(x - dx))
        if diag_idx is not None:
            for idx in diag_idx:
                g[idx] -= np.average(g[idx])
        return g
    return grad


def _line_search(phi, f0, steps, min_step, xatol, fallback=True):
    """Bounded Brent search on [0, scale]; returns (step, value)."""
    scale = max(abs(np.average(steps[-2:])), min_step)
    res = minimize_scalar(phi, bounds=(0.0, scale), method="bounded", options={"maxiter": 100, "xatol": xatol})
    if res.fun > f0:
        # the bounded search can return a local minimum above f(0)
        if fallback:
            xopt, fopt, _, _, _ = fmin(phi, 0.0, disp=False, xtol=xatol * 0.1, full_output=True)
            if fopt <= f0:
                return xopt[0], fopt
            res_fun = fopt
        else:
            res_fun = res.fun
        log.warn("line search fails, resulting value  %20.12f is\nlarger than the previous step value %20.12f",
                 res_fun, f0)
        return 0.0, f0
    return res.x, res.fun


def minimize_SD(fn, x0, MaxIter=300, fgrad=None, callback=None, ytol=1e-7, gtol=1e-3, dx_tol=1e-7, **kwargs):
    """Steepest descent with the damped direction h * 10 / (1 + h.h), h = 10 g / y (fit.py:47-150)."""
    eps = kwargs.get("eps", 1e-5)
    init_step, min_step, xatol = kwargs.get("init_step", 1.0), kwargs.get("min_step", 0.1), kwargs.get("xatol", 1e-5)
    if fgrad is None:
        fgrad = _numeric_grad(fn, callback, eps)
    x = x0
    y = fn(x)
    steps = [init_step]
    pattern = 0
    g = None
    for it in range(MaxIter):
        if y < ytol * 0.1 and it != 0:
            pattern = 1
            break
        g = fgrad(x)
        if norm(g) < min(1e-5, gtol):
            pattern = 2
            break
        h = 10 * g / y
        dx = h * 10 / (1 + np.sum(h * h))
        if callback is None:
            ray_fn = kwargs.get("ray_fn", None)
            phi = ray_fn(x, -dx) if ray_fn is not None else (lambda step: fn(x - step * dx))
        else:
            ref_ = callback(x)
            phi = lambda step: fn(x - step * dx, ref_)
        step, y_new = _line_search(phi, y, steps, min_step, xatol, fallback=False)
        steps.append(step)
        dx = dx * step
        if y_new > y * 1.5:
            pattern = 3
            break
        if abs(y - y_new) < ytol and norm(g) < gtol:
            pattern = 3
            x = x - dx
            y = y_new
            break
        if norm(dx) < dx_tol:
            pattern = 3
            x = x - dx
            y = y_new
            break
        x = x - dx
        y = y_new
        log.debug(0, "%4d %20.12f %20.12f %20.12f %15.3e", it, y, norm(g), norm(dx), step)
    return x, y, pattern, norm(g)


def _downhill(fn, x0, method, MaxIter, fgrad, callback, ytol, gtol, dx_tol, **kwargs):
    """CG (Polak-Ribiere with restart at beta < 0) or BFGS + line search (fit_helper.py:174-484)."""
    eps = kwargs.get("eps", 1e-5)
    init_step, min_step, xatol = kwargs.get("init_step", 1.0), kwargs.get("min_step", 0.1), kwargs.get("xatol", 1e-5)
    if fgrad is None:
        fgrad = _numeric_grad(fn, callback, eps, kwargs.get("diag_idx", None))
    ray_fn = kwargs.get("ray_fn", None)
    f = lambda x: fn(np.copy(x))
    fp = lambda x: np.asarray(fgrad(np.copy(x)))
    xk = np.asarray(x0).flatten()
    gfk = fp(xk)
    old_fval = f(xk)
    steps = [init_step]
    k = 0
    nfev, ngev = 1, 1
    bfgs = (method == 'BFGS')
    if bfgs:
        N = len(xk)
        I = np.eye(N, dtype=int)
        Hk = I
    else:
        pk = -gfk
    gnorm = norm(gfk)
    while k < MaxIter and (not bfgs or gnorm > gtol):
        if bfgs:
            pk = -np.dot(Hk, gfk)
        else:
            deltak = np.dot(gfk, gfk)
        # `ray_fn(x, p)` (optional, FitVcorEmb passes it): phi(t) = fn(x + t p) from an objective that is cheaper along
        # a fixed ray (the potential is linear in the parameters: two passes over dV_dparam serve the whole search)
        phi = ray_fn(xk, pk) if ray_fn is not None else (lambda step: f(xk + step * pk))
        alpha_k, new_fval = _line_search(phi, old_fval, steps, min_step, xatol)
        steps.append(alpha_k)
        dy = abs(new_fval - old_fval)
        norm_dx = norm(pk) * alpha_k
        if not bfgs and abs(norm_dx) < dx_tol:
            log.debug(0, "CG: dx (%20.12g) < %15.8g reached.", norm_dx, dx_tol)
            break
        old_fval = new_fval
        xkp1 = xk + alpha_k * pk
        gfkp1 = fp(xkp1)
        ngev += 1
        yk = gfkp1 - gfk
        if bfgs:
            sk = xkp1 - xk
        else:
            beta_k = max(0, np.dot(yk, gfkp1) / deltak)
            pk = -gfkp1 + beta_k * pk
        xk, gfk = xkp1, gfkp1
        gnorm = norm(gfk)
        if not bfgs:
            log.debug(0, "%4d %20.12f %20.12f %20.12f %15.3e", k, old_fval, gnorm, norm_dx, alpha_k)
        if callback is not None:
            callback(xk)
        k += 1
        if gnorm < gtol:
            log.debug(0, "%s: gnorm (%20.12g) < %15.8g reached.", method, gnorm, gtol)
            break
        if dy < ytol:
            log.debug(0, "%s: dy (%20.12g) < %15.8g reached.", method, dy, ytol)
            break
        if bfgs:
            log.debug(0, "%4d %20.12f %20.12f %20.12f %15.3e", k, old_fval, gnorm, norm_dx, alpha_k)
            if abs(norm_dx) < dx_tol:
                break
            if not np.isfinite(old_fval):
                break
            ys = np.dot(yk, sk)
            rhok = 1.0 / ys if ys != 0.0 else 1000.0
            if np.isinf(rhok):
                rhok = 1000.0
            A1 = I - sk[:, np.newaxis] * yk[np.newaxis, :] * rhok
            A2 = I - yk[:, np.newaxis] * sk[np.newaxis, :] * rhok
            Hk = np.dot(A1, np.dot(Hk, A2)) + (rhok * sk[:, np.newaxis] * sk[np.newaxis, :])
    if k >= MaxIter:
        log.info("Warning: Maximum number of iterations has been exceeded.")
    log.info("         Current function value: %f" % old_fval)
    log.info("         Iterations: %d" % k)
    return xk, old_fval, 3, norm(gfk)


def minimize_CG(fn, x0, MaxIter=300, fgrad=None, callback=None, ytol=1e-7, gtol=1e-3, dx_tol=1e-7, **kwargs):
    return _downhill(fn, x0, 'CG', MaxIter, fgrad, callback, ytol, gtol, dx_tol, **kwargs)


def minimize_BFGS(fn, x0, MaxIter=300, fgrad=None, callback=None, ytol=1e-7, gtol=1e-3, dx_tol=1e-7, **kwargs):
    return _downhill(fn, x0, 'BFGS', MaxIter, fgrad, callback, ytol, gtol, dx_tol, **kwargs)


def _boundary_steps(z, d, radius):
    """The two t with |z + t d| = radius, ascending."""
    a, b, c = np.dot(d, d), 2.0 * np.dot(z, d), np.dot(z, z) - radius ** 2
    root = np.sqrt(b * b - 4.0 * a * c)
    # numerically stable pair of roots (no cancellation in -b +- root)
    aux = b + np.copysign(root, b)
    return tuple(sorted([-aux / (2.0 * a), -2.0 * c / aux]))


def _steihaug(g, hessp, radius, model):
    """Truncated conjugate gradient for min_p g.p + p.B p / 2 inside |p| <= radius (Steihaug 1983; the subproblem solver the
    reference takes from SciPy, fit_helper.py:18, 488): stops at the boundary on negative curvature or when an iterate leaves
    the region, else when the residual drops below min(0.5, sqrt|g|) |g|.  Returns (p, hits_boundary)."""
    gnorm = np.linalg.norm(g)
    tol = min(0.5, np.sqrt(gnorm)) * gnorm
    z = np.zeros_like(g)
    if gnorm < tol:
        return z, False
    r, d = g, -g
    while True:
        Bd = hessp(d)
        dBd = np.dot(d, Bd)
        if dBd <= 0:
            ta, tb = _boundary_steps(z, d, radius)
            pa, pb = z + ta * d, z + tb * d
            return (pa if model(pa) < model(pb) else pb), True
        r2 = np.dot(r, r)
        alpha = r2 / dBd
        z_next = z + alpha * d
        if np.linalg.norm(z_next) >= radius:
            return z + _boundary_steps(z, d, radius)[1] * d, True
        r_next = r + alpha * Bd
        if np.linalg.norm(r_next) < tol:
            return z_next, False
        d = -r_next + (np.dot(r_next, r_next) / r2) * d
        z, r = z_next, r_next


def minimize_NCG(fn, x0, MaxIter=300, fgrad=None, callback=None, ytol=1e-7, gtol=1e-3, dx_tol=1e-7, **kwargs):
    """Trust-region Newton-CG (fit.py:217-288).  Radii scale with sqrt(nx): initial 1e-5, maximal 3e-3, acceptance eta 1e-3;
    Hessian-vector products (g(x + eps p) - g(x - eps p)) / (2 eps), eps 1e-5; without `fgrad` the gradient is numerical.
    The loop is the textbook one (Nocedal & Wright, algorithm 4.1) with the reference's three extra exits (value below
    ytol / 10, step below dx_tol, iteration count).  Returns (x, y, 3, max|g|)."""
    x = np.asarray(x0, dtype=float).flatten()
    nx = x.shape[0]
    radius = kwargs.get("initial_trust_radius", 1e-5) * np.sqrt(nx)
    radius_max = kwargs.get("max_trust_radius", 3e-3) * np.sqrt(nx)
    eta, eps = kwargs.get("eta", 0.001), kwargs.get("eps", 1e-5)
    if not (0 <= eta < 0.25):
        raise Exception('invalid acceptance stringency')
    if radius <= 0 or radius_max <= 0 or radius >= radius_max:
        raise ValueError('trust radii: 0 < initial (%g) < max (%g) required' % (radius, radius_max))
    if fgrad is None:
        fgrad = _numeric_grad(fn, callback, eps)
    log.debug(0, "NCG: initial_trust_radius: %.2e", radius)
    log.debug(0, "NCG: max_trust_radius: %.2e", radius_max)
    log.debug(0, "  Iter           Value               Grad                 Step              Radius\n"
                 "-----------------------------------------------------------------------------------------")

    def local_model(xc):
        fc, gc = fn(xc), np.asarray(fgrad(xc), dtype=float)
        hp = lambda p: (np.asarray(fgrad(xc + p * eps), dtype=float) - np.asarray(fgrad(xc - p * eps), dtype=float)) * (0.5 / eps)
        return fc, gc, hp

    f, g, hp = local_model(x)
    k, flag = 0, 0
    while np.linalg.norm(g) >= gtol:
        quad = lambda p: f + np.dot(g, p) + 0.5 * np.dot(p, hp(p))
        try:
            p, on_boundary = _steihaug(g, hp, radius, quad)
        except np.linalg.LinAlgError:
            flag = 3
            break
        predicted = f - quad(p)
        x_new = x + p
        f_new, g_new, hp_new = local_model(x_new)
        if predicted <= 0:
            flag = 2
            break
        rho = (f - f_new) / predicted
        if rho < 0.25:
            radius *= 0.25
        elif rho > 0.75 and on_boundary:
            radius = min(1.75 * radius, radius_max)
        x_old = x
        if rho > eta:
            x, f, g, hp = x_new, f_new, g_new, hp_new
        if callback is not None:
            callback(np.copy(x))
        step = norm(x - x_old)
        log.debug(0, "%4d %20.12f %20.12f %20.12f %15.3e", k, f, np.linalg.norm(g), step, radius)
        k += 1
        if np.linalg.norm(g) < gtol:
            log.debug(0, "NCG: g = 0 condition reached.")
            break
        if f < ytol * 0.1:
            log.debug(0, "NCG: y = 0 condition reached.")
            break
        if abs(step) < dx_tol:
            log.debug(0, "NCG: dx = 0 condition reached.")
            break
        if k >= MaxIter:
            flag = 1
            break
    if flag:
        log.warn("Warning: %s", ("", "Maximum number of iterations has been exceeded.",
                                 "A bad approximation caused failure to predict improvement.",
                                 "A linalg error occurred, such as a non-psd Hessian.")[flag])
    log.info("         Current function value: %f" % f)
    log.info("         Iterations: %d" % k)
    return x, f, 3, norm(g)


def minimize(fn, x0, MaxIter=300, fgrad=None, callback=None, method='CG', ytol=1e-7, gtol=1e-3, dx_tol=1e-7, **kwargs):
    """Main wrapper for the minimisers: returns (x, y, converge_pattern, gnorm)."""
    log.info("%s used in minimizer", method)
    method = method.lower().strip()
    if method == 'cg':
        driver = minimize_CG
    elif method == 'bfgs':
        driver = minimize_BFGS
    elif method == 'sd':
        driver = minimize_SD
    elif method == 'trust-ncg':
        driver = minimize_NCG
    elif method == 'ciah':
        raise NotImplementedError("minimiser ciah needs PySCF's CIAH solver (pyscf.soscf.ciah); use CG, BFGS, SD or trust-ncg")
    else:
        raise ValueError("Unknown method %s" % method)
    return driver(fn, x0, MaxIter=MaxIter, fgrad=fgrad, callback=callback, ytol=ytol, gtol=gtol, dx_tol=dx_tol, **kwargs)
