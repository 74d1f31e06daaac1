"""
Pins oracle/restate_fit.py (vcor least-squares fit in the embedding space) against
tests/golden/G9_vcorfit.npz: the reference's VcorLocal, get_dV_dparam and the errfunc / gradfunc closures of
FitVcorEmb captured at fixed parameter vectors under oracle/shim.py.  CPU only.
"""
import numpy as np
import pytest

from oracle import restate as R
from oracle import restate_fit as F

CASES = ["uhf_231", "rhf_411", "uhf_222"]
VC = {"r": dict(restricted=True, bogoliubov=False), "u": dict(restricted=False, bogoliubov=False),
      "rb": dict(restricted=True, bogoliubov=True), "rbg": dict(restricted=True, bogoliubov=True, ghf=True),
      "ub": dict(restricted=False, bogoliubov=True), "ubr": dict(restricted=False, bogoliubov=True, bogo_res=True)}


def fit_runs(nb):
    return [("t0", np.inf, dict()), ("ft", 15.0, dict()), ("ft_fixmu", 15.0, dict(fix_mu=True, mu0=0.1)),
            ("t0_imp", np.inf, dict(imp_fit=True)), ("t0_det", np.inf, dict(det=True)),
            ("t0_idx", np.inf, dict(imp_idx=[0, 1], det_idx=[nb - 1])), ("t0_rdg", np.inf, dict(remove_diag_grad=True))]


def fit_inputs(g, name):
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    FR, basis, target = g[name + "/Fock_R"], g[name + "/basis"], g[name + "/target"]
    val = [int(x) for x in g[name + "/val"]]
    spin, nlo = basis.shape[0], FR.shape[-1]
    Fk = R.R2k(FR, mesh)
    Sk = np.asarray([np.eye(nlo, dtype=complex)] * basis.shape[1])
    ncore = min(val)
    nelec = ncore + len(val) if spin == 1 else [ncore + len(val)] * 2
    return mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec


def idx_sets(kw, nimp, nb):
    """imp_idx / det_idx resolution of slater.py:985-1005."""
    if kw.get("imp_fit"):
        return list(range(nimp)), []
    if kw.get("det"):
        return [], list(range(nimp))
    return kw.get("imp_idx"), kw.get("det_idx")


@pytest.mark.parametrize("tag", sorted(VC))
@pytest.mark.parametrize("itag", ["all", "sub"])
def test_G9_vcor_local(golden, tag, itag):
    g = golden("G9_vcorfit.npz")
    key = "vcor/%s_%s" % (tag, itag)
    v = F.VcorLocal(nscsites=5, idx_range=None if itag == "all" else [1, 3, 4], **VC[tag])
    p = g[key + "/param"]
    assert v.length() == len(p)
    v.update(p)
    assert np.array_equal(v.get(), g[key + "/value"])
    assert np.array_equal(v.gradient(), g[key + "/grad"])
    assert np.array_equal(np.asarray(v.diag_indices()), g[key + "/diag"])


@pytest.mark.parametrize("name", CASES)
def test_G9_objective_and_gradient(golden, name):
    g = golden("G9_vcorfit.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    nb = basis.shape[-1]
    v = F.VcorLocal(spin == 1, False, nlo, idx_range=val)
    assert np.abs(F.get_dV_dparam(v, basis) - g[name + "/dV_dparam"]).max() < 1e-13
    assert np.abs(F.get_dV_dparam(v, basis, compact=False) - g[name + "/dV_dparam_full"]).max() < 1e-13
    for tag, beta, kw in fit_runs(nb):
        imp_idx, det_idx = idx_sets(kw, nlo - min(val), nb)
        fit = F.EmbFit(target, mesh, basis, v, beta, Fk if spin == 2 else Fk[0], Sk, nelec, imp_idx=imp_idx, det_idx=det_idx,
                       mu0=kw.get("mu0"), fix_mu=kw.get("fix_mu", False), remove_diag_grad=kw.get("remove_diag_grad", False))
        key = "%s/%s" % (name, tag)
        grad = fit.gradfunc if beta == np.inf else fit.gradfunc_ft
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-12, key
            assert np.abs(grad(p) - gr).max() < 1e-9 * max(1.0, np.abs(gr).max()), key
        # the reference's converged parameters are a (local) minimum of the restated objective too
        pfit, (e0, e1) = g[key + "/param"], g[key + "/err"]
        assert abs(fit.errfunc(np.zeros_like(pfit)) - e0) < 1e-12
        assert abs(fit.errfunc(pfit) - e1) < 1e-11
        assert e1 <= e0 + 1e-12


FULL_RUNS = [("bath_t0", "target_emb", np.inf, dict()), ("imp_t0", "target_loc", np.inf, dict(imp_fit=True)),
             ("det_ft", "target_loc", 12.0, dict(det=True)), ("idx_ft", "target_loc", 12.0, dict(imp_idx=[0, 1], det_idx=[-1]))]


@pytest.mark.parametrize("name", ["uhf_231", "rhf_411"])
def test_G10_full_lattice_objective(golden, name):
    g = golden("G10_vcorfit_full.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    FR, basis = g[name + "/Fock_R"], g[name + "/basis"]
    val = [int(x) for x in g[name + "/val"]]
    spin, nlo = basis.shape[0], FR.shape[-1]
    Fk = R.R2k(FR, mesh)
    for tag, tkey, beta, kw in FULL_RUNS:
        v = F.VcorLocal(spin == 1, False, nlo, idx_range=val)
        kw = dict(kw)
        if kw.get("det_idx") == [-1]:
            kw["det_idx"] = [nlo - 1]
        imp_idx, det_idx = idx_sets(kw, nlo - min(val), None)
        fit = F.FullFit(g[name + "/" + tkey], mesh, basis, v, beta, Fk if spin == 2 else Fk[0], 0.5, imp_idx=imp_idx, det_idx=det_idx)
        key = "%s/%s" % (name, tag)
        for p, e in zip(g[key + "/probe"], g[key + "/probe_err"]):
            assert abs(fit.errfunc(p) - e) < 1e-11, key
        assert abs(fit.errfunc(np.zeros(v.length())) - g[key + "/err"][0]) < 1e-11
        assert abs(fit.errfunc(g[key + "/param"]) - g[key + "/err"][1]) < 1e-10


GRAD_RUNS = [("imp_ft", 10.0, dict(imp_fit=True)), ("det_ft", 12.0, dict(det=True)),
             ("idx_ft_fixmu", 8.0, dict(imp_idx=[0, 1], det_idx=[-1], fix_mu=True)),
             ("imp_ft_bfgs", 15.0, dict(imp_fit=True, method="BFGS"))]
GRAD_CASES = ["uhf_231", "rhf_411", "rhf_222"]


@pytest.mark.parametrize("name", GRAD_CASES)
def test_G14_full_lattice_gradient(golden, name):
    """FitVcorFull.gradfunc_ft (slater.py:1480-1640) captured from the reference at fixed parameter vectors."""
    g = golden("G14_vcorfit_full_grad.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    FR, basis = g[name + "/Fock_R"], g[name + "/basis"]
    val = [int(x) for x in g[name + "/val"]]
    spin, nlo = basis.shape[0], FR.shape[-1]
    Fk = R.R2k(FR, mesh)
    for tag, beta, kw in GRAD_RUNS:
        v = F.VcorLocal(spin == 1, False, nlo, idx_range=val)
        kw = dict(kw)
        if kw.get("det_idx") == [-1]:
            kw["det_idx"] = [nlo - 1]
        imp_idx, det_idx = idx_sets(kw, nlo - min(val), None)
        fit = F.FullFit(g[name + "/target_loc"], mesh, basis, v, beta, Fk if spin == 2 else Fk[0], 0.5, imp_idx=imp_idx,
                        det_idx=det_idx, fix_mu=kw.get("fix_mu", False))
        key = "%s/%s" % (name, tag)
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-11, key
            assert np.abs(fit.gradfunc_ft(p) - gr).max() < 1e-9 * max(1.0, np.abs(gr).max()), key
        if kw.get("fix_mu"):
            # with the chemical potential fixed the reference's formula is the exact derivative (central differences); with
            # a floating mu it normalises the mu response per k point (slater.py:1359-1362 flags it), so only parity holds
            p = g[key + "/probe"][0]
            e = np.zeros_like(p)
            e[1] = 1e-5
            fd = (fit.errfunc(p + e) - fit.errfunc(p - e)) / 2e-5
            assert abs(fd - fit.gradfunc_ft(p)[1]) < 1e-6, key
        assert abs(fit.errfunc(g[key + "/param"]) - g[key + "/err"][1]) < 1e-10


# ---- round 6: FitVcorEmb options (golden G21) ------------------------------------------------------------------------

OPT_CASES = ["uhf_231", "rhf_411"]


def option_runs(g, name, nimp):
    q, P = g[name + "/C_act"], g[name + "/P_full"]
    return [("idem_t0", np.inf, dict(idem_fit=True)), ("idem_ft", 15.0, dict(idem_fit=True)),
            ("cact_t0", np.inf, dict(C_act=q)), ("cact_ft", 15.0, dict(C_act=q)),
            ("cact_imp_t0", np.inf, dict(C_act=q[:, :nimp], imp_fit=True)),
            ("pact_t0", np.inf, dict(P_full=P)), ("pact_ft", 15.0, dict(P_full=P)),
            ("pact_cact_ft_fixmu", 15.0, dict(P_full=P, C_act=q, fix_mu=True, mu0=0.1))]


@pytest.mark.parametrize("name", OPT_CASES)
def test_G21_rdm1_idem_and_projector(golden, name):
    g = golden("G21_fit_options.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    for btag, beta in (("t0", np.inf), ("ft", 15.0)):
        assert np.abs(F.get_rdm1_idem(target, nelec, beta) - g["%s/idem_%s" % (name, btag)]).max() < 1e-12
    nel_k = g[name + "/nelec_k"]
    nel_k = int(nel_k) if nel_k.ndim == 0 else [int(x) for x in nel_k]
    assert np.abs(F.get_rdm1_idem(g[name + "/rdm1_k"], nel_k, np.inf) - g[name + "/idem_k_t0"]).max() < 1e-12
    assert np.abs(F.get_rdm1_idem(g[name + "/rdm1_k"], nel_k, 9.0) - g[name + "/idem_k_ft"]).max() < 1e-11
    Sk_s = np.asarray([Sk] * spin)
    Pf = F.get_active_projector_full(list(g[name + "/P_act"]), Sk_s)
    assert np.abs(Pf - g[name + "/P_full"]).max() < 1e-13
    v = F.VcorLocal(spin == 1, False, nlo, idx_range=val)
    assert np.abs(F.get_dV_dparam(v, basis, P_full=Pf, kmesh=mesh) - g[name + "/dV_dparam_pact"]).max() < 1e-13


@pytest.mark.parametrize("name", OPT_CASES)
def test_G21_option_objectives_and_gradients(golden, name):
    g = golden("G21_fit_options.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    nb = basis.shape[-1]
    v = F.VcorLocal(spin == 1, False, nlo, idx_range=val)
    for tag, beta, kw in option_runs(g, name, nlo - min(val)):
        kw = dict(kw)
        imp_idx, det_idx = idx_sets(kw, nlo - min(val), nb)
        kw.pop("imp_fit", None)
        fit = F.EmbFit(target, mesh, basis, v, beta, Fk if spin == 2 else Fk[0], Sk, nelec, imp_idx=imp_idx, det_idx=det_idx, **kw)
        key = "%s/%s" % (name, tag)
        grad = fit.gradfunc if beta == np.inf else fit.gradfunc_ft
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-12, key
            assert np.abs(grad(p) - gr).max() < 1e-9 * max(1.0, np.abs(gr).max()), key
        pfit, (e0, e1) = g[key + "/param"], g[key + "/err"]
        assert abs(fit.errfunc(np.zeros_like(pfit)) - e0) < 1e-12 and abs(fit.errfunc(pfit) - e1) < 1e-11


@pytest.mark.parametrize("name", OPT_CASES)
def test_G21_drho_dparam(golden, name):
    g = golden("G21_fit_options.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    v = F.VcorLocal(spin == 1, False, nlo, idx_range=val)
    for tag, kw in (("drho_dparam", dict()), ("drho_dparam_fixmu", dict(fix_mu=True, mu0=0.1))):
        fit = F.EmbFit(target, mesh, basis, v, 15.0, Fk if spin == 2 else Fk[0], Sk, nelec, **kw)
        got = fit.drho_dparam(g["%s/%s_param" % (name, tag)])
        ref = g["%s/%s" % (name, tag)]
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
        # and it IS the derivative of the embedding density: central differences of the restated density
        p0 = g["%s/%s_param" % (name, tag)]
        tl = np.tril_indices(basis.shape[-1])
        for ip in (0, len(p0) - 1):
            dp = np.zeros_like(p0)
            dp[ip] = 1e-5
            rp = fit._solve(p0 + dp)[4] + fit.target
            rm = fit._solve(p0 - dp)[4] + fit.target
            fd = np.asarray([((rp[s] - rm[s]) / 2e-5)[tl] for s in range(spin)])
            assert np.abs(fd - ref[:, ip]).max() < 1e-6


# ---- round 6: the cell-resolved potential VcorNonLocal and the fit it drives (golden G23) ---------------------------------

NONLOCAL_MODES = [("r", True, False, False), ("u", False, False, False), ("rb", True, True, False), ("ub_res", False, True, True),
                  ("ub", False, True, False)]
NONLOCAL_LATTICES = [("m411", (4, 1, 1), 3, None), ("m231", (2, 3, 1), 4, [1, 2]), ("m333", (3, 3, 3), 2, None), ("m221", (2, 2, 1), 3, [0, 2])]
NONLOCAL_FITS = ["uhf_231", "rhf_411", "rhf_222"]


@pytest.mark.parametrize("lat", NONLOCAL_LATTICES, ids=[x[0] for x in NONLOCAL_LATTICES])
@pytest.mark.parametrize("mode", NONLOCAL_MODES, ids=[x[0] for x in NONLOCAL_MODES])
def test_G23_nonlocal_tables(golden, lat, mode):
    """Integer bookkeeping: bit-exact against the reference's closures (value, the non-zeros of gradient(), assign)."""
    g = golden("G23_vcor_nonlocal.npz")
    (lname, mesh, nlo, idx), (mname, res, bogo, bres) = lat, mode
    key = "tab/%s/%s" % (lname, mname)
    v = F.VcorNonLocal(res, bogo, mesh, nlo, idx, bres)
    p = g[key + "/param"]
    assert v.length() == len(p)
    v.update(p)
    assert np.array_equal(v.value, g[key + "/value"])
    assert np.abs(v.value_k - g[key + "/value_k"]).max() < 1e-13
    assert np.abs(v.get(1, True) - g[key + "/get_k1"]).max() < 1e-13 and np.array_equal(v.get(1, False), g[key + "/get_R1"])
    gr = v.gradient()
    assert gr.shape == tuple(g[key + "/grad_shape"])
    assert np.array_equal(np.asarray(np.nonzero(gr)), g[key + "/grad_nz"]) and np.array_equal(gr[np.nonzero(gr)], g[key + "/grad_val"])
    v.assign(g[key + "/assign_in"])
    assert np.array_equal(v.param, g[key + "/assign_param"])


@pytest.mark.parametrize("name", NONLOCAL_FITS)
def test_G23_nonlocal_dV_and_objectives(golden, name):
    g = golden("G23_vcor_nonlocal.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    nb = basis.shape[-1]
    v = F.VcorNonLocal(spin == 1, False, mesh, nlo, val)
    assert np.abs(F.get_dV_dparam(v, basis) - g[name + "/dV_compact"]).max() < 1e-13
    assert np.abs(F.get_dV_dparam(v, basis, compact=False) - g[name + "/dV_full"]).max() < 1e-13
    for tag, beta, kw in [("t0", np.inf, {}), ("ft", 15.0, {}), ("imp_t0", np.inf, dict(imp_fit=True))]:
        imp_idx, det_idx = idx_sets(kw, nlo - min(val), nb)
        fit = F.EmbFit(target, mesh, basis, v, beta, Fk if spin == 2 else Fk[0], Sk, nelec, imp_idx=imp_idx, det_idx=det_idx)
        key = "%s/%s" % (name, tag)
        grad = fit.gradfunc if beta == np.inf else fit.gradfunc_ft
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-12, key
            assert np.abs(grad(p) - gr).max() < 1e-9 * max(1.0, np.abs(gr).max()), key
        pfit, (e0, e1) = g[key + "/param"], g[key + "/err"]
        assert abs(fit.errfunc(np.zeros_like(pfit)) - e0) < 1e-12 and abs(fit.errfunc(pfit) - e1) < 1e-11


# ---- round 6: the k-point-resolved potential VcorKpoints and the FitVcorFull branch that fits it (golden G24) -------------

KPTS_MESHES = [("m411", (4, 1, 1), 3), ("m231", (2, 3, 1), 2), ("m333", (3, 3, 3), 2), ("m221", (2, 2, 1), 4), ("m511", (5, 1, 1), 3)]
KPTS_RUNS = [("ft_imp", 15.0, dict(imp_fit=True)), ("ft_det", 15.0, dict(det=True)), ("ft_imp_fixmu", 15.0, dict(imp_fit=True, fix_mu=True)),
             ("t0_num", np.inf, dict(imp_fit=True, num_grad=True))]


@pytest.mark.parametrize("lat", KPTS_MESHES, ids=[x[0] for x in KPTS_MESHES])
@pytest.mark.parametrize("res", [True, False], ids=["r", "u"])
def test_G24_kpoints_tables(golden, lat, res):
    g = golden("G24_vcor_kpoints.npz")
    lname, mesh, nlo = lat
    key = "tab/%s/%s" % (lname, "r" if res else "u")
    v = F.VcorKpoints(res, mesh, nlo)
    p = g[key + "/param"]
    assert v.length() == len(p)
    v.update(p)
    assert np.array_equal(v.value, g[key + "/value"]) and np.array_equal(v.get(2), g[key + "/get2"])
    assert [[int(x) for x in r if x >= 0] for r in g[key + "/kpts_map"]] == v.kpts_map
    assert list(g[key + "/nparam_kpts"]) == v.nparam_kpts


@pytest.mark.parametrize("name", NONLOCAL_FITS)
def test_G24_full_fit_objective_and_gradient(golden, name):
    g = golden("G24_vcor_kpoints.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    basis, FR, target = g[name + "/basis"], g[name + "/Fock_R"], g[name + "/target"]
    spin, nlo = basis.shape[0], FR.shape[-1]
    Fk = R.R2k(FR, mesh)
    for tag, beta, kw in KPTS_RUNS:
        v = F.VcorKpoints(spin == 1, mesh, nlo)
        fit = F.FullFit(target, mesh, basis, v, beta, Fk if spin == 2 else Fk[0], 0.5, imp_idx=list(range(nlo)) if kw.get("imp_fit") else [],
                        det_idx=list(range(nlo)) if kw.get("det") else [], fix_mu=kw.get("fix_mu", False))
        key = "%s/%s" % (name, tag)
        for i, p in enumerate(g[key + "/probe"]):
            assert abs(fit.errfunc(p) - g[key + "/probe_err"][i]) < 1e-12, key
            if key + "/probe_grad" in g:
                gr = g[key + "/probe_grad"][i]
                assert np.abs(fit.gradfunc_ft(p) - gr).max() < 1e-10 * max(1.0, np.abs(gr).max()), key
        pfit, (e0, e1) = g[key + "/param"], g[key + "/err"]
        assert abs(fit.errfunc(np.zeros_like(pfit)) - e0) < 1e-12 and abs(fit.errfunc(pfit) - e1) < 1e-10
