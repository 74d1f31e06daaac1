"""
Host control flow of the vcor fit (no GPU needed): the optimiser drivers of libdmet_preview_amd/routine/fit.py against
iterates captured from the reference's routine/fit.py on analytic objectives (golden G9, `opt/*`), and the
VcorLocal parametrisation of libdmet_preview_amd/dmet/Hubbard.py against the reference's.
"""
import numpy as np
import pytest

from tests.test_oracle_fit import VC


def _objectives(g):
    A, b = g["opt/A"], g["opt/b"]
    quad = lambda x: float(np.sqrt(0.5 * x @ A @ x - b @ x + 20.0))
    qgrad = lambda x: (A @ x - b) / (2.0 * quad(x))
    rosen = lambda x: float(np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2) + 1e-3)
    rgrad = lambda x: np.concatenate([[0.0], 200.0 * (x[1:] - x[:-1] ** 2)]) + \
        np.concatenate([-400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1]), [0.0]])
    return {"quad_cg": (quad, qgrad, dict(method="CG"), 60), "quad_cg_num": (quad, None, dict(method="CG"), 60),
            "quad_sd": (quad, qgrad, dict(method="SD"), 60), "rosen_cg": (rosen, rgrad, dict(method="CG"), 25),
            "quad_bfgs": (quad, qgrad, dict(method="BFGS"), 60)}


@pytest.mark.parametrize("tag", ["quad_cg", "quad_cg_num", "quad_sd", "rosen_cg", "quad_bfgs"])
def test_minimize_matches_reference_iterates(golden, tag):
    from libdmet_preview_amd.routine import fit
    g = golden("G9_vcorfit.npz")
    fn, fg, kw, mi = _objectives(g)[tag]
    x, y, pat, gn = fit.minimize(fn, g["opt/%s_x0" % tag].copy(), mi, fg, **kw)
    xr, (yr, patr, gnr) = g["opt/%s_x" % tag], g["opt/%s_res" % tag]
    assert np.abs(x - xr).max() < 1e-9, (x, xr)
    assert abs(y - yr) < 1e-11 and int(pat) == int(patr) and abs(gn - gnr) < 1e-8


@pytest.mark.parametrize("tag", ["quad_cg", "quad_sd", "rosen_cg", "quad_bfgs"])
def test_minimize_ray_objective_reproduces_reference_iterates(golden, tag):
    """`ray_fn(x, p)` (the cheaper objective along a search ray that FitVcorEmb passes: routine/fit.py `_downhill`,
    `minimize_SD`) must not change the iterates: with phi(t) = fn(x + t p) the drivers still reproduce the reference's, and
    every line search goes through the ray objective."""
    from libdmet_preview_amd.routine import fit
    g = golden("G9_vcorfit.npz")
    fn, fg, kw, mi = _objectives(g)[tag]
    made = []

    def ray_fn(x, p):
        x, p = np.array(x, dtype=float), np.array(p, dtype=float)
        made.append(0)

        def phi(t):
            made[-1] += 1
            return fn(x + float(np.asarray(t).ravel()[0]) * p)
        return phi
    x, y, pat, gn = fit.minimize(fn, g["opt/%s_x0" % tag].copy(), mi, fg, ray_fn=ray_fn, **kw)
    xr, (yr, patr, gnr) = g["opt/%s_x" % tag], g["opt/%s_res" % tag]
    assert np.abs(x - xr).max() < 1e-9 and abs(y - yr) < 1e-11 and int(pat) == int(patr)
    assert len(made) > 0 and min(made) > 0


NCG_RUNS = {"quad_ncg": ("quad", True, dict(), 60), "quad_ncg_num": ("quad", False, dict(), 60),
            "quad_ncg_wide": ("quad", True, dict(initial_trust_radius=0.05, max_trust_radius=0.5), 60),
            "rosen_ncg": ("rosen", True, dict(initial_trust_radius=0.02, max_trust_radius=0.3), 40)}


@pytest.mark.parametrize("tag", sorted(NCG_RUNS))
def test_trust_ncg_matches_reference_iterates(golden, tag):
    """The trust-region Newton-CG driver (routine/fit.py:217-288 over fit_helper.py:486-668 and SciPy's Steihaug subproblem)
    against results captured from the reference (golden G21 `opt/*`): analytic and numerical gradient, default radii (the run
    ends on the iteration count with the radius still growing) and wide ones (converged), a non-convex objective."""
    from libdmet_preview_amd.routine import fit
    g = golden("G21_fit_options.npz")
    objs = _objectives(g)
    which, analytic, kw, mi = NCG_RUNS[tag]
    fn, fg = objs["quad_cg" if which == "quad" else "rosen_cg"][:2]
    x, y, pat, gn = fit.minimize(fn, g["opt/%s_x0" % tag].copy(), mi, fg if analytic else None, method="trust-ncg", **kw)
    xr, (yr, patr, gnr) = g["opt/%s_x" % tag], g["opt/%s_res" % tag]
    assert np.abs(x - xr).max() < 1e-9, (x, xr)
    assert abs(y - yr) < 1e-11 and int(pat) == int(patr) == 3 and abs(gn - gnr) < 1e-8


def test_trust_ncg_argument_checks():
    from libdmet_preview_amd.routine import fit
    f, g = (lambda x: float(x @ x) + 1.0), (lambda x: 2.0 * x)
    with pytest.raises(ValueError):
        fit.minimize(f, np.ones(3), fgrad=g, method="trust-ncg", initial_trust_radius=1.0, max_trust_radius=0.5)
    with pytest.raises(Exception):
        fit.minimize(f, np.ones(3), fgrad=g, method="trust-ncg", eta=0.3)
    # a start at the minimum: no step is taken
    x, y, pat, gn = fit.minimize(f, np.zeros(3), fgrad=g, method="trust-ncg")
    assert np.array_equal(x, np.zeros(3)) and y == 1.0 and gn == 0.0
    # negative curvature: the subproblem walks to the boundary and the value still decreases
    sad = lambda x: float(x[0] ** 2 - x[1] ** 2 + 0.1 * x[1] ** 4) + 5.0
    gsad = lambda x: np.array([2.0 * x[0], -2.0 * x[1] + 0.4 * x[1] ** 3])
    x, y, pat, gn = fit.minimize(sad, np.array([0.3, 0.01]), 80, gsad, method="trust-ncg", initial_trust_radius=0.05, max_trust_radius=0.5)
    assert y < sad(np.array([0.3, 0.01])) - 1.0 and abs(abs(x[1]) - np.sqrt(5.0)) < 1e-3


def test_minimize_rejects_unknown_method():
    from libdmet_preview_amd.routine import fit
    with pytest.raises(ValueError):
        fit.minimize(lambda x: 0.0, np.zeros(2), method="nope")
    with pytest.raises(NotImplementedError):
        fit.minimize(lambda x: 0.0, np.zeros(2), method="ciah")


@pytest.mark.parametrize("tag", sorted(VC))
@pytest.mark.parametrize("itag", ["all", "sub"])
def test_vcor_local(golden, tag, itag):
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G9_vcorfit.npz")
    key = "vcor/%s_%s" % (tag, itag)
    v = Hubbard.VcorLocal(nscsites=5, idx_range=None if itag == "all" else [1, 3, 4], **VC[tag])
    p = g[key + "/param"]
    assert v.length() == len(p) and v.islocal() and v.is_local()
    assert np.array_equal(v.get(), np.zeros_like(g[key + "/value"]))       # starts from zero parameters
    v.update(p)
    assert np.array_equal(v.get(), g[key + "/value"])
    assert np.array_equal(v.get(3, kspace=False), np.zeros_like(g[key + "/value"]))
    assert np.array_equal(v.gradient(), g[key + "/grad"])
    assert np.array_equal(np.asarray(v.diag_indices()), g[key + "/diag"])
    # assign() projects a matrix back onto the parameters
    w = Hubbard.VcorLocal(nscsites=5, idx_range=None if itag == "all" else [1, 3, 4], **VC[tag])
    w.assign(g[key + "/value"])
    assert np.abs(w.param - p).max() < 1e-14
    assert "fitted orbitals" in w.show() and "nao 5" in w.show()


@pytest.mark.parametrize("kw", [dict(restricted=True, bogoliubov=False), dict(restricted=False, bogoliubov=False),
                                dict(restricted=True, bogoliubov=True), dict(restricted=True, bogoliubov=True, ghf=True),
                                dict(restricted=False, bogoliubov=True), dict(restricted=False, bogoliubov=True, bogo_res=True)])
@pytest.mark.parametrize("idx", [None, [1, 3, 4]])
def test_vcor_grad_entries_match_dense_gradient(kw, idx):
    """VcorLocal.grad_entries() lists exactly the non-zeros of gradient(), in np.nonzero order."""
    from libdmet_preview_amd.dmet import Hubbard
    v = Hubbard.VcorLocal(nscsites=5, idx_range=idx, **kw)
    g = v.gradient()
    nz = np.nonzero(g)
    P, B, I, J, V = v.grad_entries()
    assert np.array_equal(P, nz[0]) and np.array_equal(B, nz[1]) and np.array_equal(I, nz[2]) and np.array_equal(J, nz[3])
    assert np.array_equal(V, g[nz])


def test_mono_fit_iterates_equal_the_reference(golden):
    """bcs_helper.mono_fit (bcs_helper.py:72-129): the chemical-potential search of the BCS mean field -- every evaluation point
    and the result are the reference's (golden G32)."""
    from libdmet_preview_amd.routine.bcs_helper import mono_fit
    g = golden("G32_bcs_driver.npz")
    for tag, fn, y0, x0, thr, inc in (("cubic", lambda x: x ** 3 + x, 2.5, 0.0, 1e-9, True), ("tanh", lambda x: np.tanh(0.3 * x), -0.7, 1.0, 1e-7, True),
                                      ("dec", lambda x: -np.arctan(x), 0.4, 3.0, 1e-8, False)):
        trace = []
        x = mono_fit(lambda z: (trace.append(z), fn(z))[1], y0, x0, thr, increase=inc)
        assert x == float(g["mono/" + tag]) and np.array_equal(np.asarray(trace), g["mono/" + tag + "_trace"])
    assert mono_fit(lambda x: 2.0 * x, 1.0, 0.5, 1e-12) == 0.5                      # the start already hits the target


def test_mono_fit_2_iterates_equal_the_reference(golden):
    """bcs_helper.mono_fit_2 (bcs_helper.py:131-174: bracketing + Brent), the chemical-potential search of the GSO mean field."""
    from libdmet_preview_amd.routine.spinless_helper import mono_fit_2
    g = golden("G34_gso_driver.npz")
    for tag, fn, y0, x0, thr, inc in (("cubic", lambda x: x ** 3 + x, 2.5, 0.0, 1e-9, True), ("tanh", lambda x: np.tanh(0.3 * x), -0.7, 1.0, 1e-7, True),
                                      ("dec", lambda x: -np.arctan(x), 0.4, 3.0, 1e-8, False)):
        trace = []
        x = mono_fit_2(lambda z: (trace.append(z), fn(z))[1], y0, x0, thr, increase=inc)
        assert x == float(g["mono2/" + tag]) and np.array_equal(np.asarray(trace), g["mono2/" + tag + "_trace"])
    with pytest.raises(RuntimeError):
        mono_fit_2(lambda x: 0.0, 1.0, 0.0, 1e-9, maxiter=1)


AF_CASES = [("a", (2, 2), dict()), ("b", (2, 2), dict(polar=0.3)), ("c", (2, 2), dict(bogoliubov=True, rand=0.02)), ("d", (2, 2), dict(bogoliubov=True, rand=0.05, d_wave=True)),
            ("e", (4,), dict(bogoliubov=True, rand=0.01, bogo_res=True)), ("f", (2, 2), dict(trace_zero=True)), ("g", (2, 1, 2), dict(polar=-0.2, bogoliubov=True, rand=0.03, d_wave=True)),
            ("h", (2,), dict(subA=[0], subB=[2], subP=[1]))]


def test_starting_potentials_equal_the_reference(golden):
    """dmet/Hubbard.py:482-549 AFInitGuess / PMInitGuess (and the BCS / GSO wrappers): parameters and matrices of golden G36,
    the pairing noise from the reference's fixed seed included."""
    from libdmet_preview_amd.dmet import Hubbard, HubbardBCS, HubbardGSO
    g = golden("G36_init_guess.npz")
    for tag, size, kw in AF_CASES:
        v = Hubbard.AFInitGuess(size, 4.0, 0.4, **kw)
        assert np.array_equal(np.asarray(v.param), g["af/%s/param" % tag]) and np.array_equal(np.asarray(v.get()), g["af/%s/value" % tag]), tag
    for tag, size, kw in (("a", (2, 2), dict()), ("b", (3,), dict(rand=0.1))):
        v = Hubbard.PMInitGuess(size, 4.0, 0.4, **kw)
        assert np.array_equal(np.asarray(v.param), g["pm/%s/param" % tag]) and np.array_equal(np.asarray(v.get()), g["pm/%s/value" % tag]), tag
    assert Hubbard.PMInitGuess((2,), 4.0, 0.5, bogoliubov=True).get().shape == (3, 2, 2)
    for mod in (HubbardBCS, HubbardGSO):
        v = mod.AFInitGuess((2, 2), 4.0, 0.4, rand=0.02)
        assert np.array_equal(np.asarray(v.param), g["af/c/param"])
    with pytest.raises(Exception):
        Hubbard.BipartiteSquare((3,))


@pytest.mark.parametrize("tag,res,bogo", [("r", True, False), ("u", False, False), ("rb", True, True), ("ub", False, True)])
def test_vcor_restricted_equals_the_reference(golden, tag, res, bogo):
    """dmet/Hubbard.py:788-938 VcorRestricted (full potential on active orbitals, diagonal on core orbitals): value, the non-zeros of
    gradient() and the sparse entries the device table builder reads, bit-exact (golden G36)."""
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G36_init_guess.npz")
    v = Hubbard.VcorRestricted(res, bogo, [0, 2, 3], [1, 4])
    p = g["vr/%s/param" % tag]
    assert v.length() == len(p) and v.is_local()
    v.update(p)
    gr = v.gradient()
    assert np.array_equal(v.get(), g["vr/%s/value" % tag]) and gr.shape == tuple(g["vr/%s/grad_shape" % tag])
    assert np.array_equal(np.asarray(np.nonzero(gr)), g["vr/%s/grad_nz" % tag])
    assert np.array_equal(np.asarray(v.grad_entries()[:4]), g["vr/%s/grad_nz" % tag]) and np.all(v.grad_entries()[4] == 1.0)
    w = Hubbard.VcorRestricted(res, bogo, [0, 2, 3], [1, 4])
    w.assign(v.get())
    assert np.abs(np.asarray(w.param) - p).max() < 1e-14
    if not res and bogo:
        with pytest.raises(NotImplementedError):
            Hubbard.VcorRestricted(False, True, [0, 1], [2], bogo_res=True)


def test_symmetry_adapted_potentials_equal_the_reference(golden):
    """dmet/Hubbard.py:940-1494 VcorSymm / VcorSymmSpin / VcorSymmBogo in every mode the reference implements: value and gradient
    against golden G36 (1e-14: products of the irrep orbitals), gradient = Jacobian of the value, diag_indices."""
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G36_init_guess.npz")
    Q, Qb = g["vs/Q"], g["vs/Qb"]
    Ca, Cb, idx = [Q[:, :2], Q[:, 2:]], [Qb[:, :2], Qb[:, 2:]], [0, 2, 3, 5, 6]
    makers = [("symm", lambda: Hubbard.VcorSymm(False, False, 7, Ca, idx_range=idx)),
              ("spin", lambda: Hubbard.VcorSymmSpin(False, False, 7, Ca, Cb, idx_range=idx)),
              ("spin_bres", lambda: Hubbard.VcorSymmSpin(False, True, 7, Ca, Cb, idx_range=idx, bogo_res=True)),
              ("spin_b", lambda: Hubbard.VcorSymmSpin(False, True, 7, Ca, Cb, idx_range=idx)),
              ("bogo_res", lambda: Hubbard.VcorSymmBogo(False, True, 7, Ca, Cb, idx_range=idx, bogo_res=True)),
              ("bogo", lambda: Hubbard.VcorSymmBogo(False, True, 7, Ca, Cb, idx_range=idx))]
    for tag, make in makers:
        v = make()
        p = g["vs/%s/param" % tag]
        assert v.length() == len(p) and v.is_local() and np.abs(v.get()).max() == 0.0
        v.update(p)
        assert np.abs(v.get() - g["vs/%s/value" % tag]).max() < 1e-14, tag
        gr = v.gradient()
        if "vs/%s/grad" % tag in g:
            assert np.abs(gr - g["vs/%s/grad" % tag]).max() < 1e-14, tag
        assert np.abs(np.tensordot(p, gr, axes=(0, 0)) - v.get()).max() < 1e-14, tag
    assert np.array_equal(np.asarray(Hubbard.VcorSymm(False, False, 7, Ca, idx_range=idx).diag_indices()), g["vs/symm/diag"])
    assert Hubbard.VcorSymmSpin(False, False, 7, Ca, Cb, idx_range=idx).diag_indices() is None
    for bad in (lambda: Hubbard.VcorSymm(True, False, 7, Ca, idx_range=idx), lambda: Hubbard.VcorSymm(False, True, 7, Ca, idx_range=idx),
                lambda: Hubbard.VcorSymmSpin(True, False, 7, Ca, Cb, idx_range=idx), lambda: Hubbard.VcorSymmBogo(False, False, 7, Ca, Cb, idx_range=idx)):
        with pytest.raises(NotImplementedError):
            bad()


def test_vcor_diagonal_helpers_equal_the_reference(golden):
    """slater.addDiag / vcor_diag_average / make_vcor_trace_unchanged (slater.py:757-818) and the GSO forms spinless.addDiag /
    keep_vcor_trace_fixed (spinless.py:739-752): parameters after every step equal the reference's (golden G36)."""
    from libdmet_preview_amd.dmet import Hubbard, HubbardGSO
    from libdmet_preview_amd.routine import slater, spinless
    g = golden("G36_init_guess.npz")
    for tag, res, bogo, rng_idx in (("u", False, False, [1, 2]), ("rb", True, True, None)):
        v = Hubbard.VcorLocal(res, bogo, 4, idx_range=[0, 1, 2] if rng_idx else None)
        old = Hubbard.VcorLocal(res, bogo, 4, idx_range=[0, 1, 2] if rng_idx else None)
        v.update(np.array(g["vd/%s/p_new" % tag]))
        old.update(np.array(g["vd/%s/p_old" % tag]))
        assert np.abs(slater.vcor_diag_average(v, idx_range=rng_idx) - g["vd/%s/ave" % tag]).max() < 1e-15
        Hubbard.addDiag(v, [0.3, -0.2, 0.0][: v.get().shape[0]] if not res else 0.25, idx_range=rng_idx)
        assert np.abs(np.asarray(v.param) - g["vd/%s/after_add" % tag]).max() < 1e-15
        Hubbard.make_vcor_trace_unchanged(v, old, idx_range=rng_idx)
        assert np.abs(np.asarray(v.param) - g["vd/%s/after_trace" % tag]).max() < 1e-15
    v, old = Hubbard.VcorLocal(False, True, 3), Hubbard.VcorLocal(False, True, 3)
    v.update(np.array(g["vd/gso/p_new"]))
    old.update(np.array(g["vd/gso/p_old"]))
    spinless.addDiag(v, 0.4)
    assert np.abs(np.asarray(v.param) - g["vd/gso/after_add"]).max() < 1e-15
    HubbardGSO.keep_vcor_trace_fixed(v, old)
    assert np.abs(np.asarray(v.param) - g["vd/gso/after_trace"]).max() < 1e-15
    assert Hubbard.VcorZeros is Hubbard.VcorLocal and Hubbard.VcorLocal_new is Hubbard.VcorLocal


def test_particle_hole_symmetric_potentials_equal_the_reference(golden):
    """dmet/HubPhSymm.py:114-295 VcorLocalPhSymm (with pairing, with a distance cut), VcorDCAPhSymm and InitGuess: value, gradient and
    the projected starting parameters equal the reference's (golden G36); the sparse entries reproduce the dense gradient."""
    from libdmet_preview_amd.dmet import HubPhSymm as HP
    from libdmet_preview_amd.dmet.Hubbard import BipartiteSquare
    g = golden("G36_init_guess.npz")
    makers = (("loc22", lambda: HP.VcorLocalPhSymm(4.0, False, (2, 2), *BipartiteSquare((2, 2)))),
              ("loc22b", lambda: HP.VcorLocalPhSymm(4.0, True, (2, 2), *BipartiteSquare((2, 2)))),
              ("loc4r", lambda: HP.VcorLocalPhSymm(3.0, True, (4,), *BipartiteSquare((4,)), r=1.0)),
              ("dca22", lambda: HP.VcorDCAPhSymm(4.0, (2, 2), *BipartiteSquare((2, 2)))),
              ("dca4", lambda: HP.VcorDCAPhSymm(2.0, (4,), *BipartiteSquare((4,)))))
    for tag, make in makers:
        v = make()
        p = g["ph/%s/param" % tag]
        assert v.length() == len(p) and v.is_local(), tag
        v.update(p)
        assert np.array_equal(v.get(), g["ph/%s/value" % tag]), tag
        gr = v.gradient()
        assert np.array_equal(gr, g["ph/%s/grad" % tag]), tag
        P, B, I, J, S = v.grad_entries()
        dense = np.zeros_like(gr)
        dense[P, B, I, J] = S
        assert np.array_equal(dense, gr)
    for tag, kw in (("a", dict()), ("b", dict(polar=0.7)), ("c", dict(r=1.0))):
        v = HP.InitGuess((2, 2), 4.0, **kw)
        assert np.array_equal(np.asarray(v.param), g["ph/init_%s/param" % tag]) and np.array_equal(v.get(), g["ph/init_%s/value" % tag]), tag
