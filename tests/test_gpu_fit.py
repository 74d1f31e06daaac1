"""
GPU parity tests (-m gpu) for SURVEY.md section 8(f) rank 2: the correlation-potential least-squares fit in the
embedding space.  HIP path through the C ABI against oracle/restate_fit.py and the golden fixture G9 (the reference's
FitVcorEmb objective / gradient closures at fixed parameters, and its converged fits).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import restate as R
from oracle import restate_fit as F
from tests.test_oracle_fit import CASES, fit_runs, fit_inputs, idx_sets


@pytest.fixture(scope="module")
def ctx():
    from libdmet_preview_amd import _lib
    return _lib.get_ctx()


def _lattice(mesh, nlo, val, Fk, spin):
    from libdmet_preview_amd.system.lattice import Lattice
    L = Lattice(int(nlo), mesh)
    L.val_idx = list(val)
    L.virt_idx = [i for i in range(nlo) if i > max(val)]
    L.core_idx = [i for i in range(nlo) if i < min(val)]
    L.fock_lo_k = L.hcore_lo_k = Fk if spin == 2 else Fk[0]
    return L


@pytest.mark.parametrize("M,N", [(1, 1), (7, 300), (33, 2049), (100, 5000), (321, 777)])
def test_dgemv2(ctx, M, N):
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(M * N)
    A, xr, xc = rng.standard_normal((M, N + 3)), rng.standard_normal(N), rng.standard_normal(M)
    dA, dxr, dxc = ctx.to_device(A), ctx.to_device(xr), ctx.to_device(xc)
    dyr, dyc = ctx.empty((M,), np.float64), ctx.empty((N,), np.float64)
    ctx.check(lib.dmk_dgemv2(ctx.h, M, N, dA.ptr, N + 3, dxr.ptr, dxc.ptr, dyr.ptr, dyc.ptr))
    assert np.abs(dyr.get() - A[:, :N] @ xr).max() < 1e-12 * np.sqrt(N) * 10
    assert np.abs(dyc.get() - A[:, :N].T @ xc).max() < 1e-12 * np.sqrt(M) * 10
    y1 = dyr.get().copy()
    ctx.check(lib.dmk_dgemv2(ctx.h, M, N, dA.ptr, N + 3, dxr.ptr, None, dyr.ptr, None))
    assert np.array_equal(dyr.get(), y1)                                  # fixed reduction order: bit-reproducible


@pytest.mark.parametrize("opA,opB", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_dgemm_batched(ctx, opA, opB):
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(opA * 2 + opB)
    M, N, K, nbat = 70, 45, 130, 3
    A = rng.standard_normal((nbat, K, M) if opA else (nbat, M, K))
    B = rng.standard_normal((nbat, N, K) if opB else (nbat, K, N))
    C0 = rng.standard_normal((nbat, M, N))
    dA, dB, dC = ctx.to_device(A), ctx.to_device(B), ctx.to_device(C0)
    ctx.check(lib.dmk_dgemm_batched(ctx.h, opA, opB, M, N, K, nbat, 0.7, dA.ptr, A.shape[2], A[0].size, dB.ptr, B.shape[2],
                                    B[0].size, -0.5, dC.ptr, N, M * N))
    ref = 0.7 * np.einsum("bmk,bkn->bmn", A.transpose(0, 2, 1) if opA else A, B.transpose(0, 2, 1) if opB else B) - 0.5 * C0
    assert np.abs(dC.get() - ref).max() < 1e-12


def test_small_fit_kernels(ctx):
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(4)
    n, nbat = 13, 2
    full = rng.standard_normal((nbat, n, n))
    il = np.tril_indices(n)
    d_full, d_tril = ctx.to_device(full), ctx.empty((nbat, len(il[0])), np.float64)
    ctx.check(lib.dmk_sym_fold(ctx.h, n, nbat, d_full.ptr, d_tril.ptr))
    sym = full + full.transpose(0, 2, 1)
    sym[:, np.arange(n), np.arange(n)] *= 0.5
    assert np.abs(d_tril.get() - sym[:, il[0], il[1]]).max() < 1e-15
    add = rng.standard_normal((nbat, len(il[0])))
    d_out = ctx.empty((nbat, n, n), np.float64)
    d_add = ctx.to_device(add)
    ctx.check(lib.dmk_sym_unpack(ctx.h, n, nbat, d_tril.ptr, d_add.ptr, d_out.ptr))
    ref = np.zeros((nbat, n, n))
    ref[:, il[0], il[1]] = d_tril.get() + add
    ref = ref + np.tril(ref, -1).transpose(0, 2, 1)
    assert np.array_equal(d_out.get(), ref)
    ri, ci = np.array([3, 0, 12, 5], dtype=np.int32), np.array([1, 1, 7], dtype=np.int32)
    d_g = ctx.empty((4, 3), np.float64)
    d_ri, d_ci = ctx.to_device(ri), ctx.to_device(ci)
    ctx.check(lib.dmk_gather2d_f64(ctx.h, 4, 3, d_ri.ptr, d_ci.ptr, d_full.ptr, n, d_g.ptr))
    assert np.array_equal(d_g.get(), full[0][np.ix_(ri, ci)])
    ctx.check(lib.dmk_gather2d_f64(ctx.h, n, 3, None, d_ci.ptr, d_full.ptr, n, d_out.ptr))
    assert np.array_equal(d_out.get().ravel()[:n * 3].reshape(n, 3), full[0][:, ci])
    a, b = rng.standard_normal((5, 9)), rng.standard_normal((5, 9))
    da, db, dc, dss = ctx.to_device(a), ctx.to_device(b), ctx.empty((5, 9), np.float64), ctx.empty((1,), np.float64)
    ctx.check(lib.dmk_ewise_mul(ctx.h, 0, 5, 9, da.ptr, db.ptr, dc.ptr))
    assert np.array_equal(dc.get(), a * b)
    d_b0 = ctx.to_device(b[:, 0].copy())
    ctx.check(lib.dmk_ewise_mul(ctx.h, 1, 5, 9, da.ptr, d_b0.ptr, dc.ptr))
    assert np.array_equal(dc.get(), a * b[:, :1])
    ctx.check(lib.dmk_sub_sumsq(ctx.h, 45, da.ptr, db.ptr, dc.ptr, dss.ptr))
    assert np.array_equal(dc.get(), a - b) and abs(dss.get()[0] - np.sum((a - b) ** 2)) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_dV_dparam(ctx, golden, name):
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G9_vcorfit.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    L = _lattice(mesh, nlo, val, Fk, spin)
    v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
    assert np.abs(slater.get_dV_dparam(v, basis, None, L) - g[name + "/dV_dparam"]).max() < 1e-13
    assert np.abs(slater.get_dV_dparam(v, basis, None, L, compact=False) - g[name + "/dV_dparam_full"]).max() < 1e-13


@pytest.mark.parametrize("name", CASES)
def test_fit_objective_gradient_and_result(ctx, golden, name):
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G9_vcorfit.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    nb = basis.shape[-1]
    L = _lattice(mesh, nlo, val, Fk, spin)
    for tag, beta, kw in fit_runs(nb):
        key = "%s/%s" % (name, tag)
        v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
        vfit, e0, e1 = slater.FitVcorEmb(target, L, basis, v, beta, MaxIter=40, **kw)
        fit = slater.FitVcorEmb.last_fit
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-11, key
            assert np.abs(fit.gradfunc(p) - gr).max() < 1e-8 * max(1.0, np.abs(gr).max()), key
        pref, (r0, r1) = g[key + "/param"], g[key + "/err"]
        assert abs(e0 - r0) < 1e-11, key
        assert abs(e1 - r1) < 1e-7, (key, e1, r1)
        assert np.abs(vfit.param - pref).max() < 2e-4, (key, np.abs(vfit.param - pref).max())
        assert vfit is v and e1 <= e0


def test_fit_nonorthogonal_overlap_and_numeric_gradient(ctx, golden):
    """Generalised eigenproblem path (embedding overlap != 1) against the oracle; num_grad drives the same objective."""
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    from libdmet_preview_amd import synth
    g = golden("G9_vcorfit.npz")
    name = "uhf_231"
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    nb = basis.shape[-1]
    SR = np.zeros(FR.shape[1:])
    SR[0] = np.eye(nlo)
    SR = SR + 0.02 * synth.make_fock_R(mesh, nlo, spin=1, seed=9)[0]
    Sk = R.R2k(SR, mesh)
    L = _lattice(mesh, nlo, val, Fk, spin)
    L.ovlp_lo_k = Sk
    v = Hubbard.VcorLocal(False, False, nlo, idx_range=val)
    slater.FitVcorEmb(target, L, basis, v, np.inf, MaxIter=3)
    fit = slater.FitVcorEmb.last_fit
    assert fit.d_X is not None
    vo = F.VcorLocal(False, False, nlo, idx_range=val)
    ofit = F.EmbFit(target, mesh, basis, vo, np.inf, Fk, Sk, nelec)
    oft = F.EmbFit(target, mesh, basis, vo, 12.0, Fk, Sk, nelec)
    rng = np.random.default_rng(8)
    for p in 0.1 * rng.standard_normal((2, v.length())):
        assert abs(fit.errfunc(p) - ofit.errfunc(p)) < 1e-11
        assert np.abs(fit.gradfunc(p) - ofit.gradfunc(p)).max() < 1e-8
    v2 = Hubbard.VcorLocal(False, False, nlo, idx_range=val)
    slater.FitVcorEmb(target, L, basis, v2, 12.0, MaxIter=2)
    fit2 = slater.FitVcorEmb.last_fit
    p = 0.1 * rng.standard_normal(v.length())
    assert abs(fit2.errfunc(p) - oft.errfunc(p)) < 1e-11
    assert np.abs(fit2.gradfunc(p) - oft.gradfunc_ft(p)).max() < 1e-8
    # numerical gradient option and the SciPy cross-check run through the same device objective
    v3 = Hubbard.VcorLocal(False, False, nlo, idx_range=val)
    _, e0, e1 = slater.FitVcorEmb(target, L, basis, v3, np.inf, MaxIter=2, num_grad=True)
    assert e1 <= e0
    v4 = Hubbard.VcorLocal(False, False, nlo, idx_range=val)
    _, e0, e1 = slater.FitVcorEmb(target, L, basis, v4, np.inf, MaxIter=3, CG_check=True)
    assert e1 <= e0


@pytest.mark.parametrize("name", ["uhf_231", "rhf_411"])
def test_fit_options_vs_reference(ctx, golden, name):
    """FitVcorEmb options of slater.py:969-1058 on the device against the reference's own closures and fits (golden G21):
    idem_fit (get_rdm1_idem on the target), C_act (residual projected on active orbitals; the fused T = 0 objective is bypassed),
    P_act (active-space projector inside dV_dparam, alone and together with C_act and a fixed mu), and the trust-region
    Newton-CG driver on the device objective."""
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    from tests.test_oracle_fit import option_runs
    g = golden("G21_fit_options.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    L = _lattice(mesh, nlo, val, Fk, spin)
    L.ovlp_lo_k = Sk if spin == 1 else np.asarray([Sk] * 2)
    P_act = list(g[name + "/P_act"])
    assert np.abs(slater.get_active_projector_full(P_act, L.ovlp_lo_k) - g[name + "/P_full"]).max() < 1e-13
    v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
    dV = slater.get_dV_dparam(v, basis, None, L, P_act=g[name + "/P_full"])
    assert np.abs(dV - g[name + "/dV_dparam_pact"]).max() < 1e-13
    runs = [(t, b, {("P_act" if k == "P_full" else k): (P_act if k == "P_full" else x) for k, x in kw.items()})
            for t, b, kw in option_runs(g, name, nlo - min(val))]
    runs += [("ncg_t0", np.inf, dict(method="trust-ncg")), ("ncg_ft", 15.0, dict(method="trust-ncg"))]
    for tag, beta, kw in runs:
        key = "%s/%s" % (name, tag)
        v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
        vfit, e0, e1 = slater.FitVcorEmb(target, L, basis, v, beta, MaxIter=40 if "ncg" not in tag else 12, **kw)
        fit = slater.FitVcorEmb.last_fit
        if "C_act" in kw:
            assert fit._fused is None
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-11, key
            assert np.abs(fit.gradfunc(p) - gr).max() < 1e-8 * max(1.0, np.abs(gr).max()), key
        pref, (r0, r1) = g[key + "/param"], g[key + "/err"]
        assert abs(e0 - r0) < 1e-11, key
        assert abs(e1 - r1) < 1e-7, (key, e1, r1)
        assert np.abs(vfit.param - pref).max() < 2e-4, (key, np.abs(vfit.param - pref).max())
        assert vfit is v and e1 <= e0


@pytest.mark.parametrize("name", ["uhf_231", "rhf_411"])
def test_rdm1_idem_and_drho_dparam(ctx, golden, name):
    """slater_helper.get_rdm1_idem (embedding and k-space densities, T = 0 and smeared) and the return_drho_dparam branch of
    FitVcorEmb (slater.py:1227-1261: the finite-T response of the embedding density to every parameter, batched nb^3 products on
    the device instead of the npair x npair response matrix) against the reference's outputs (golden G21)."""
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G21_fit_options.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    for btag, beta in (("t0", np.inf), ("ft", 15.0)):
        assert np.abs(slater.get_rdm1_idem(target, nelec, beta) - g["%s/idem_%s" % (name, btag)]).max() < 1e-12
    nel_k = g[name + "/nelec_k"]
    nel_k = int(nel_k) if nel_k.ndim == 0 else [int(x) for x in nel_k]
    got = slater.get_rdm1_idem(g[name + "/rdm1_k"], nel_k, np.inf)
    assert got.dtype == np.complex128 and np.abs(got - g[name + "/idem_k_t0"]).max() < 1e-12
    assert np.abs(slater.get_rdm1_idem(g[name + "/rdm1_k"], nel_k, 9.0) - g[name + "/idem_k_ft"]).max() < 1e-11
    L = _lattice(mesh, nlo, val, Fk, spin)
    for tag, kw in (("drho_dparam", dict()), ("drho_dparam_fixmu", dict(fix_mu=True, mu0=0.1))):
        v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
        v.update(g["%s/%s_param" % (name, tag)])
        got = slater.FitVcorEmb(target, L, basis, v, 15.0, return_drho_dparam=True, **kw)
        ref = g["%s/%s" % (name, tag)]
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-10 * max(1.0, np.abs(ref).max()), tag
        # chunking of the parameter batches does not change the numbers
        again = slater.FitVcorEmb.last_fit.drho_dparam(v.param, chunk=3)
        assert np.abs(again - got).max() < 1e-13
    with pytest.raises(AssertionError):
        slater.FitVcorEmb(target, L, basis, v, np.inf, return_drho_dparam=True)
    # use_drho_dparam only logs the norms and fits as usual; test_grad logs and fits
    v2 = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
    _, e0, e1 = slater.FitVcorEmb(target, L, basis, v2, 15.0, MaxIter=2, use_drho_dparam=True, test_grad=True)
    assert e1 <= e0


def test_pipeline_fit_round_trip(ctx):
    """Hidden-parameter round trip on a small synthetic system: the fit must drive the error towards zero."""
    from libdmet_preview_amd import pipeline
    sysm = pipeline.SyntheticSystem(ctx, (3, 2, 1), 8, 6, 3, 2, seed=31, name="t")
    out = pipeline.iteration(ctx, sysm)
    res = pipeline.vcor_fit_stage(ctx, sysm, out["basis"], out["nemb"], out["emb_ham"]["rdm1_emb"], MaxIter=60)
    assert res["err_end"] < 1e-2 * res["err_begin"], res
    assert res["objective_evals"] > 0 and res["gradient_evals"] > 0


@pytest.mark.parametrize("name", ["uhf_231", "rhf_411"])
def test_fit_full_lattice_and_two_step(ctx, golden, name):
    """FitVcorFull (device objective, numerical gradient) and FitVcorTwoStep against the reference's runs (golden G10)."""
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    from tests.test_oracle_fit import FULL_RUNS
    g = golden("G10_vcorfit_full.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    FR, basis = g[name + "/Fock_R"], g[name + "/basis"]
    val = [int(x) for x in g[name + "/val"]]
    spin, nlo = basis.shape[0], FR.shape[-1]
    Fk = R.R2k(FR, mesh)
    L = _lattice(mesh, nlo, val, Fk, spin)
    for tag, tkey, beta, kw in FULL_RUNS:
        kw = dict(kw)
        if kw.get("det_idx") == [-1]:
            kw["det_idx"] = [nlo - 1]
        key = "%s/%s" % (name, tag)
        v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
        vfit, e0, e1 = slater.FitVcorFull(g[name + "/" + tkey], L, basis, v, beta, 0.5, MaxIter=2, num_grad=True, **kw)
        fit = slater.FitVcorFull.last_fit
        pfit = np.array(vfit.param)                       # errfunc updates the vcor object in place, like the reference's
        for p, e in zip(g[key + "/probe"], g[key + "/probe_err"]):
            assert abs(fit.errfunc(p) - e) < 1e-10, key
        r0, r1 = g[key + "/err"]
        assert abs(e0 - r0) < 1e-10 and abs(e1 - r1) < 1e-6, (key, e1, r1)
        assert np.abs(pfit - g[key + "/param"]).max() < 1e-3, key
    v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
    v2, e_end = slater.FitVcorTwoStep(g[name + "/target_emb"], L, basis, v, np.inf, 0.5, MaxIter1=5, MaxIter2=1, num_grad=True)
    assert v2 is not v and abs(e_end - float(g[name + "/twostep_err"])) < 1e-6
    assert np.abs(v2.param - g[name + "/twostep_param"]).max() < 1e-3
    with pytest.raises(NotImplementedError):
        slater.FitVcorFull(g[name + "/target_emb"], L, basis, v, 12.0, 0.5, MaxIter=1)


@pytest.mark.parametrize("name", ["uhf_231", "rhf_411", "rhf_222"])
def test_fit_full_lattice_analytic_gradient(ctx, golden, name):
    """FitVcorFull with the analytic finite-T lattice gradient on the device against the reference's gradfunc_ft
    values and fit results (golden G14)."""
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    from tests.test_oracle_fit import GRAD_RUNS
    g = golden("G14_vcorfit_full_grad.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    FR, basis = g[name + "/Fock_R"], g[name + "/basis"]
    val = [int(x) for x in g[name + "/val"]]
    spin, nlo = basis.shape[0], FR.shape[-1]
    Fk = R.R2k(FR, mesh)
    L = _lattice(mesh, nlo, val, Fk, spin)
    for tag, beta, kw in GRAD_RUNS:
        kw = dict(kw)
        if kw.get("det_idx") == [-1]:
            kw["det_idx"] = [nlo - 1]
        key = "%s/%s" % (name, tag)
        v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
        vfit, e0, e1 = slater.FitVcorFull(g[name + "/target_loc"], L, basis, v, beta, 0.5, MaxIter=4, **kw)
        fit = slater.FitVcorFull.last_fit
        assert fit.ngev > 0
        pfit = np.array(vfit.param)
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-10, key
            assert np.abs(fit.gradfunc(p) - gr).max() < 1e-9 * max(1.0, np.abs(gr).max()), key
        r0, r1 = g[key + "/err"]
        assert abs(e0 - r0) < 1e-10 and abs(e1 - r1) < 1e-6, (key, e1, r1)
        assert np.abs(pfit - g[key + "/param"]).max() < 1e-4, key


@pytest.mark.parametrize("n,batch", [(1, 1), (2, 3), (7, 2), (31, 1), (32, 2), (33, 2), (100, 3), (256, 2), (300, 1)])
def test_eigh_jacobi(ctx, n, batch):
    """Multi-CU block Jacobi eigensolver, cold and warm start, against LAPACK."""
    import ctypes as C
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(100 * n + batch)
    A = rng.standard_normal((batch, n, n))
    A = A + A.transpose(0, 2, 1)
    if n >= 7:
        A[0, 3] = A[0, 2]                                   # a (nearly) repeated row / column pair: close eigenvalues
        A[0, :, 3] = A[0, :, 2]
        A[0] = 0.5 * (A[0] + A[0].T)
    dA, dw, dV = ctx.to_device(A), ctx.empty((batch, n), np.float64), ctx.empty((batch, n, n), np.float64)
    sw = C.c_int()
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA.ptr, None, dw.ptr, dV.ptr, C.byref(sw)))
    w, V = dw.get(), dV.get()
    scale = max(1.0, np.abs(A).max() * n ** 0.5)
    for b in range(batch):
        assert np.abs(w[b] - np.linalg.eigvalsh(A[b])).max() < 1e-13 * scale * 10
        assert np.abs(V[b] @ V[b].T - np.eye(n)).max() < 1e-13
        assert np.abs(V[b] @ A[b] @ V[b].T - np.diag(w[b])).max() < 1e-12 * scale
        assert (np.diff(w[b]) >= 0).all()
    # warm start on a nearby matrix: fewer sweeps, same accuracy; output may alias the warm-start input
    P = 1e-4 * rng.standard_normal((batch, n, n))
    A2 = A + P + P.transpose(0, 2, 1)
    dA2 = ctx.to_device(A2)
    sw2 = C.c_int()
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA2.ptr, dV.ptr, dw.ptr, dV.ptr, C.byref(sw2)))
    w2, V2 = dw.get(), dV.get()
    for b in range(batch):
        assert np.abs(w2[b] - np.linalg.eigvalsh(A2[b])).max() < 1e-13 * scale * 10
        assert np.abs(V2[b] @ A2[b] @ V2[b].T - np.diag(w2[b])).max() < 1e-12 * scale
    assert sw2.value <= sw.value
    if n >= 100:
        assert sw2.value < sw.value


@pytest.mark.parametrize("n,batch", [(7, 2), (33, 3), (100, 2), (256, 2), (300, 1)])
@pytest.mark.parametrize("kind", ["random", "degenerate", "cluster"])
def test_eigh_warm_refinement(ctx, n, batch, kind):
    """Warm start of dmk_eigh_jacobi_real: the refinement fast path (sweeps_out == 0) and its fall-back to the sweeps both
    return a basis that passes the same budgets as the cold solver, for perturbations from rounding size to O(1) rotations,
    exactly repeated levels that the perturbation splits, and a tight cluster."""
    import ctypes as C
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(7 * n + batch)
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    if kind == "random":
        A = rng.standard_normal((batch, n, n))
        A = A + A.transpose(0, 2, 1)
    else:
        lam = np.repeat(np.linspace(-3, 3, (n + 1) // 2), 2)[:n] if kind == "degenerate" else np.linspace(-3, 3, n)
        if kind == "cluster":
            lam[: max(n // 4, 2)] = -3 + 1e-7 * np.arange(max(n // 4, 2))
        A = np.stack([Q @ np.diag(lam) @ Q.T] * batch)
        A = 0.5 * (A + A.transpose(0, 2, 1))
    dA, dw, dV = ctx.to_device(A), ctx.empty((batch, n), np.float64), ctx.empty((batch, n, n), np.float64)
    sw = C.c_int()
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA.ptr, None, dw.ptr, dV.ptr, C.byref(sw)))
    V0 = dV.get()
    fast = 0
    for eps in (0.0, 1e-9, 1e-6, 1e-4, 1e-2):
        P = eps * rng.standard_normal((batch, n, n))
        A2 = A + P + P.transpose(0, 2, 1)
        dA2, dV0 = ctx.to_device(A2), ctx.to_device(V0)
        ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA2.ptr, dV0.ptr, dw.ptr, dV0.ptr, C.byref(sw)))   # output aliases V0
        w, V = dw.get(), dV0.get()
        fast += sw.value == 0
        scale = max(1.0, np.abs(A2).max() * n ** 0.5)
        for b in range(batch):
            assert np.abs(w[b] - np.linalg.eigvalsh(A2[b])).max() < 1e-12 * scale, (kind, eps)
            assert np.abs(V[b] @ V[b].T - np.eye(n)).max() < 1e-13, (kind, eps)
            assert np.abs(V[b] @ A2[b] @ V[b].T - np.diag(w[b])).max() < 1e-12 * scale, (kind, eps)
            assert (np.diff(w[b]) >= 0).all()
    if kind == "random":
        assert fast >= 3            # rounding-size and small perturbations of a generic matrix take the fast path


def test_eigh_warm_refinement_nonfinite(ctx):
    """A NaN in the matrix must not be accepted by the fast path (fmax would drop it): the call falls through to the
    sweeps, which report no convergence."""
    import ctypes as C
    from libdmet_preview_amd._lib import lib, DmkError
    n = 64
    rng = np.random.default_rng(3)
    A = rng.standard_normal((1, n, n))
    A = A + A.transpose(0, 2, 1)
    dA, dw, dV = ctx.to_device(A), ctx.empty((1, n), np.float64), ctx.empty((1, n, n), np.float64)
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, 1, dA.ptr, None, dw.ptr, dV.ptr, None))
    A[0, 5, 7] = A[0, 7, 5] = np.nan
    dA2 = ctx.to_device(A)
    with pytest.raises(DmkError):
        ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, 1, dA2.ptr, dV.ptr, dw.ptr, dV.ptr, None))


def test_eigh_jacobi_rejects(ctx):
    import ctypes as C
    from libdmet_preview_amd._lib import lib, DmkError
    d = ctx.zeros((4, 4), np.float64)
    with pytest.raises(DmkError):
        ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, 600, 1, d.ptr, None, d.ptr, d.ptr, None))
    with pytest.raises(DmkError):
        ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, 64, 200, d.ptr, None, d.ptr, d.ptr, None))


def test_fit_c5_shape_objective_and_gradient_vs_oracle(ctx):
    """The vcor fit at the C5 embedding shape (nlo 200, 56 valence orbitals -> nemb 256, UHF, VcorLocal on the valence
    orbitals: 3192 parameters, dV_dparam 1.68 GB) on a small k-mesh: objective and analytic gradient of the device fit
    against oracle/restate_fit.py at seeded parameter vectors.  bench.py measures this fit at the same shape; G9 pins only
    small cases.  reference: routine/slater.py:1040-1154 (errfunc, gradfunc), :851-907 (get_dV_dparam)."""
    from libdmet_preview_amd import pipeline
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    from libdmet_preview_amd.system.lattice import Lattice
    mesh, nlo, nval, spin = (2, 2, 1), 200, 56, 2
    sysm = pipeline.SyntheticSystem(ctx, mesh, nlo, 0, nval, spin, seed=5, name="fitC5")
    d_rhoR, mf = pipeline.mean_field_stage(ctx, sysm)
    d_basis, nemb, _ = pipeline.bath_stage(ctx, sysm, d_rhoR)
    assert nemb == 256
    nk = sysm.nk
    basis = d_basis.get().reshape(spin, nk, nlo, nemb)
    Fk = sysm.d_Fock_k.get().reshape(spin, nk, nlo, nlo)
    Sk = np.asarray([np.eye(nlo)] * nk)
    L = Lattice(nlo, mesh)
    L.val_idx, L.virt_idx, L.core_idx = list(range(nval)), list(range(nval, nlo)), []
    L.fock_lo_k = L.hcore_lo_k = Fk
    rng = np.random.default_rng(9)
    v = Hubbard.VcorLocal(False, False, nlo, idx_range=list(range(nval)))
    assert v.length() == 3192
    bk = R.R2k(basis, mesh)
    rdm1_emb = np.asarray([np.einsum("kpa,kpq,kqb->ab", bk[s].conj(), R.R2k(d_rhoR.get().reshape(spin, nk, nlo, nlo), mesh)[s], bk[s]).real / nk
                           for s in range(spin)])
    ne = [int(round(np.trace(rdm1_emb[s]))) for s in range(spin)]
    noise = 0.02 * rng.standard_normal(rdm1_emb.shape)
    target = rdm1_emb + 0.5 * (noise + noise.transpose(0, 2, 1))
    fit = slater.EmbFitDevice(ctx, target, L, basis, v, np.inf, ne, list(range(nemb)), [], Fk, Sk)
    ov = F.VcorLocal(False, False, nlo, idx_range=list(range(nval)))
    ref = F.EmbFit(target, mesh, basis, ov, np.inf, Fk, Sk, ne)
    assert np.abs(fit.d_dV.get().reshape(ref.dV.shape) - ref.dV).max() < 1e-12
    for scale in (0.0, 0.03):
        p = scale * rng.standard_normal(3192)
        e, e_ref = fit.errfunc(p), ref.errfunc(p)
        assert abs(e - e_ref) < 1e-10, (scale, e, e_ref)
        gq, g_ref = fit.gradfunc(p), ref.gradfunc(p)
        assert np.abs(gq - g_ref).max() < 1e-8 * max(1.0, np.abs(g_ref).max()), (scale, np.abs(gq - g_ref).max())
    # the FUSED native objective (dmk_fit_objective: one call per evaluation) along a line-search ray: against the oracle, against
    # the chain of separate calls on the same ray, and through its fall-back when the enqueued refinement cannot settle in time
    assert fit._fused is not None and fit.fused_calls >= 1
    x0, d = 0.03 * rng.standard_normal(3192), 2e-4 * rng.standard_normal(3192)     # line-search sized steps
    phi = fit.errfunc_ray(x0, d)
    ts = [0.0, 1.0, 0.37, 0.6, 0.52, 0.55]
    phi(0.0)                                                                         # settle the basis at the start of the ray
    fb0 = fit.fused_fallbacks
    vals = [phi(t) for t in ts]
    n_fused = fit.fused_calls
    for t, e in zip(ts, vals):
        assert abs(e - ref.errfunc(x0 + t * d)) < 1e-10, t
    gq, g_ref = fit.gradfunc(x0 + ts[-1] * d), ref.gradfunc(x0 + ts[-1] * d)         # gradient from the fused forward's state
    assert np.abs(gq - g_ref).max() < 1e-8 * max(1.0, np.abs(g_ref).max())
    chain = slater.EmbFitDevice(ctx, target, L, basis, v, np.inf, ne, list(range(nemb)), [], Fk, Sk)
    chain._fused = None                                                              # the separate-call path
    phi_c = chain.errfunc_ray(x0, d)
    assert max(abs(phi_c(t) - e) for t, e in zip(ts, vals)) < 1e-12
    assert chain.fused_calls == 0 and fit.fused_fallbacks - fb0 <= 2, (fit.fused_fallbacks, fb0, fit._fused.npass)   # most took the fast path
    # a big jump with ONE enqueued pass cannot verify: status 1 -> the synchronous solver takes over, same value
    fit._fused.npass = 1
    big = x0 + 4000.0 * d
    e_big = fit.errfunc(big)
    assert fit.fused_fallbacks >= 1 and abs(e_big - ref.errfunc(big)) < 1e-10
    assert fit.fused_calls > n_fused


@pytest.mark.parametrize("name", ["uhf_231", "rhf_411"])
def test_fit_with_restricted_core_potential(ctx, golden, name):
    """Hubbard.VcorRestricted (dmet/Hubbard.py:788-938) under the device fit: the dV table from its sparse entries equals the oracle's
    from the dense gradient, and a short fit lowers the residual."""
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G9_vcorfit.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    L = _lattice(mesh, nlo, val, Fk, spin)
    active, core = val[:-1], [val[-1]] + [i for i in range(nlo) if i not in val]
    v = Hubbard.VcorRestricted(spin == 1, False, active, core, nscsites=nlo)
    dV = slater.get_dV_dparam(v, basis, None, L)
    assert dV.shape == (v.length(), spin, basis.shape[-1] * (basis.shape[-1] + 1) // 2)
    assert np.abs(dV - F.get_dV_dparam(v, basis)).max() < 1e-13
    v.update(np.zeros(v.length()))
    vfit, e0, e1 = slater.FitVcorEmb(target, L, basis, v, np.inf, MaxIter=15)
    assert vfit is v and e1 < e0


def test_particle_hole_symmetric_driver(ctx):
    """dmet/HubPhSymm.py: HartreeFock at half filling with the particle-hole symmetric starting potential, and FitVcor (two-step) with
    VcorLocalPhSymm -- signed sparse gradient entries through the device dV builder against the oracle's dense route."""
    from libdmet_preview_amd.dmet import HubPhSymm as HP
    from libdmet_preview_amd.dmet.Hubbard import BipartiteSquare
    from libdmet_preview_amd.routine import mfd, slater
    from libdmet_preview_amd import synth
    from libdmet_preview_amd.system.lattice import Lattice
    mesh, cs, U = (6, 6, 1), (2, 2), 4.0
    H1 = synth.hubbard_h1_R(mesh, cs)
    n = H1.shape[-1]
    L = Lattice(n, mesh)
    L.val_idx, L.virt_idx, L.core_idx = list(range(n)), [], []
    L.set_Ham_lo(fock_lo_R=H1, hcore_lo_R=H1)
    v = HP.InitGuess(cs, U)
    rho, mu = HP.HartreeFock(L, v, U)
    rho2, mu2, E2 = mfd.HF(L, v, 0.5, False, mu0=U / 2)
    assert np.array_equal(rho, rho2) and mu == mu2 and abs(mu - U / 2) < 1e-8          # particle-hole symmetry pins mu at U / 2
    assert abs(np.trace(rho[0][0]) + np.trace(rho[1][0]) - n) < 1e-9
    basis = slater.get_emb_basis(L, rho)
    dV = slater.get_dV_dparam(v, basis, None, L)
    assert np.abs(dV - F.get_dV_dparam(v, basis)).max() < 1e-13
    rng = np.random.default_rng(2)
    target = slater.foldRho(rho, L, basis)
    noise = 0.03 * rng.standard_normal(target.shape)
    target = target + 0.5 * (noise + noise.transpose(0, 2, 1))
    w = HP.VcorLocalPhSymm(U, False, cs, *BipartiteSquare(cs))
    w.update(np.array(v.param))
    vnew, err = HP.FitVcor(target, L, basis, w, np.inf, MaxIter1=20, MaxIter2=0)
    assert vnew is not w and np.isfinite(err)
