"""
oracle/fit_driver_campaign.py -- TEST INFRASTRUCTURE, THIS CONTAINER ONLY (imports the reference from /root/reference through
oracle/shim.py; nothing here travels to the GPU box or is imported by the product).

Randomised differential campaign of the HOST control flow of the vcor fit: the optimiser drivers of
libdmet_preview_amd/routine/fit.py (CG / BFGS with the bounded line search, steepest descent, numerical gradients) against the
reference's own drivers (libdmet/routine/fit.py:17-215, fit_helper.py:174-484) run side by side on random analytic objectives --
random positive-definite quadratics under a square root, Rosenbrock chains, quartics -- with random dimensions, starting points,
tolerances and line-search parameters.  The golden G9 pins five such runs; this covers the parameter space.
    python oracle/fit_driver_campaign.py [seed] [trials]
"""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import shim


def objectives(rng):
    n = int(rng.integers(2, 13))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        M = rng.standard_normal((n, n))
        A = M @ M.T + n * np.eye(n) * float(rng.uniform(0.05, 1.0))
        b = rng.standard_normal(n)
        c = float(0.5 * b @ np.linalg.solve(A, b) + rng.uniform(0.5, 5.0))
        fn = lambda x: float(np.sqrt(0.5 * x @ A @ x - b @ x + c))
        fg = lambda x: (A @ x - b) / (2.0 * fn(x))
        name = "sqrt-quadratic"
    elif kind == 1:
        n = max(n, 2)
        fn = lambda x: float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2) + 1e-3)
        fg = lambda x: np.concatenate([[0.0], 200.0 * (x[1:] - x[:-1] ** 2)]) + \
            np.concatenate([-400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1]), [0.0]])
        name = "rosenbrock"
    else:
        w = rng.uniform(0.5, 3.0, n)
        s = rng.standard_normal(n)
        fn = lambda x: float(np.sum(w * (x - s) ** 4) + 0.1 * np.sum((x - s) ** 2) + 0.05)
        fg = lambda x: 4.0 * w * (x - s) ** 3 + 0.2 * (x - s)
        name = "quartic"
    return n, fn, fg, name


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    shim.install()
    shim.quiet()
    from libdmet.routine import fit as rfit                      # the reference
    from libdmet_preview_amd.routine import fit as pfit          # the product's host drivers
    rng = np.random.default_rng(seed)
    worst, t0, counts = 0.0, time.time(), {}
    for trial in range(trials):
        n, fn, fg, name = objectives(rng)
        method = ["CG", "BFGS", "SD"][int(rng.integers(0, 3))]
        analytic = bool(rng.random() < 0.7)
        x0 = rng.standard_normal(n) * float(rng.uniform(0.2, 1.5))
        kw = dict(method=method, ytol=float(10.0 ** rng.uniform(-10, -6)), gtol=float(10.0 ** rng.uniform(-6, -3)),
                  dx_tol=float(10.0 ** rng.uniform(-9, -6)))
        if rng.random() < 0.5:
            kw.update(init_step=float(rng.uniform(0.3, 2.0)), min_step=float(rng.uniform(0.02, 0.3)), xatol=float(10.0 ** rng.uniform(-7, -4)))
        mi = int(rng.integers(3, 40))
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            xr, yr, pr, gr = rfit.minimize(fn, x0.copy(), mi, fg if analytic else None, **kw)
            xp, yp, pp, gp = pfit.minimize(fn, x0.copy(), mi, fg if analytic else None, **kw)
        e = max(float(np.abs(np.asarray(xp) - np.asarray(xr)).max()), abs(float(yp) - float(yr)))
        assert e < 1e-8 * max(1.0, float(np.abs(xr).max())) and int(pp) == int(pr), (trial, name, n, method, analytic, kw, mi, e, pp, pr)
        assert abs(float(gp) - float(gr)) < 1e-6 * max(1.0, abs(float(gr))), (trial, name, method, gp, gr)
        worst = max(worst, e)
        counts[(name, method)] = counts.get((name, method), 0) + 1
    print("fit driver campaign ok: %d runs against the reference's drivers in %.0f s, worst |dx|, |dy| = %.1e; runs per (objective, method): %s"
          % (trials, time.time() - t0, worst, {"%s/%s" % k: v for k, v in sorted(counts.items())}))


if __name__ == "__main__":
    main()
