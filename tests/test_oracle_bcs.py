"""
Pins oracle/restate_bcs.py (rows a3 GHF/BdG, a7 basisMatching, a8 BCS twin, a14 unit2emb of SURVEY.md
section 8a) against tests/golden/G7_bcs.npz, captured from the reference under oracle/shim.py.  CPU only.
"""
import numpy as np
import pytest

from oracle import restate as R
from oracle import restate_bcs as B

CASES = ["c611", "c441", "c222"]


def col_sign_dev(a, b):
    """max over columns of min(|a - b|, |a + b|): equality up to a sign per column."""
    a = a.reshape(-1, a.shape[-1])
    b = b.reshape(-1, b.shape[-1])
    return max(min(np.abs(a[:, j] - b[:, j]).max(), np.abs(a[:, j] + b[:, j]).max()) for j in range(a.shape[1]))


def _case(g, name):
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    FR = g[name + "/Fock_R"]
    Fk = np.asarray([R.FFTtoK(FR[s], mesh) for s in range(2)])
    return mesh, FR, Fk, g[name + "/vcor"], float(g[name + "/mu"]), [int(x) for x in g[name + "/val"]]


@pytest.mark.parametrize("name", CASES)
def test_G7_bdg_ghf(golden, name):
    g = golden("G7_bcs.npz")
    mesh, FR, Fk, v, mu, val = _case(g, name)
    for symm in (False, True):
        ew, ev = B.DiagBdG(Fk, v, mu, kmesh=mesh if symm else None)
        t = "bdg_symm" if symm else "bdg"
        assert np.abs(ew - g["%s/%s_ew" % (name, t)]).max() < 1e-11
        rho = np.einsum("kpm,km,kqm->kpq", ev, (ew < 0).astype(float), ev.conj())
        assert np.abs(rho - g["%s/%s_GRho_k" % (name, t)]).max() < 1e-10
    GFk = R.FFTtoK(g[name + "/GFock_R"], mesh)
    for symm, mu_ in ((False, mu), (True, mu), (False, None)):
        ew, ev = B.DiagGHF(GFk, v, mu_, kmesh=mesh if symm else None)
        t = "ghf_symm" if symm else ("ghf" if mu_ is not None else "ghf_nomu")
        assert np.abs(ew - g["%s/%s_ew" % (name, t)]).max() < 1e-11
        if mu_ is not None:
            rho = np.einsum("kpm,km,kqm->kpq", ev, (ew < 0).astype(float), ev.conj())
            assert np.abs(rho - g["%s/%s_rho_k" % (name, t)]).max() < 1e-10


@pytest.mark.parametrize("name", CASES)
def test_G7_nambu_bookkeeping(golden, name):
    g = golden("G7_bcs.npz")
    G0 = g[name + "/GRho"][0]
    rA, rB, kBA = B.extractRdm(G0)
    assert np.array_equal(np.asarray([rA, rB, kBA]), g[name + "/extractRdm"])
    assert np.array_equal(np.asarray(B.extractH1(G0)), g[name + "/extractH1"])
    assert np.array_equal(B.combineRdm(rA, rB, -kBA.T), g[name + "/combineRdm"])
    assert np.array_equal(B.swapSpin(G0), g[name + "/swapSpin"])
    basis = g[name + "/basis_proj"]
    can = B.basisToCanonical(basis)
    assert np.array_equal(can, g[name + "/canonical"])
    assert np.array_equal(B.basisToSpin(can), g[name + "/toSpin"])
    assert np.array_equal(B.basisToSpin(can), basis)


@pytest.mark.parametrize("name", CASES)
def test_G7_emb_basis(golden, name):
    g = golden("G7_bcs.npz")
    mesh, FR, Fk, v, mu, val = _case(g, name)
    n = FR.shape[-1]
    GRho = g[name + "/GRho"]
    basis, sigma, Bm, w = B.embBasis_proj(GRho, n, val)
    ref = g[name + "/basis_proj"]
    assert basis.shape == ref.shape
    assert np.array_equal(basis[:, 0], ref[:, 0])
    for s in range(2):
        assert col_sign_dev(basis[s][1:, :, n:], ref[s][1:, :, n:]) < 1e-9
    if name + "/basis_phsymm" in g:
        ph = B.embBasis_phsymm(GRho, n)
        refp = g[name + "/basis_phsymm"]
        for s in range(2):
            a, b = ph[s].reshape(-1, 2 * n), refp[s].reshape(-1, 2 * n)
            assert np.abs(a @ a.T - b @ b.T).max() < 1e-9


@pytest.mark.parametrize("name", CASES)
def test_G7_folds(golden, name):
    g = golden("G7_bcs.npz")
    mesh, FR, Fk, v, mu, val = _case(g, name)
    basis = g[name + "/basis_proj"]
    GF_R = g[name + "/GFock_R"]
    n = FR.shape[-1]
    D_R = GF_R[:, :n, n:]
    H3 = np.asarray([FR[0], FR[1], D_R])
    todo = [("ti3", B.transform_trans_inv, H3), ("ti2", B.transform_trans_inv, FR), ("ti1", B.transform_trans_inv, FR[0]),
            ("loc3", B.transform_local, v), ("loc2", B.transform_local, v[:2]), ("loc1", B.transform_local, v[0]),
            ("imp3", B.transform_imp, v), ("imp1", B.transform_imp, v[0]),
            ("ie3", B.transform_imp_env, H3), ("ie1", B.transform_imp_env, FR[0])]
    for tag, fn, H in todo:
        (hA, hB), hD, e0 = fn(basis, mesh, H)
        assert np.abs(np.asarray([hA, hB, hD]) - g["%s/%s_H" % (name, tag)]).max() < 1e-11, tag
        assert abs(e0 - float(g["%s/%s_E0" % (name, tag)])) < 1e-10, tag
    dV = B.get_dV_dparam(basis, n * (n + 1) + n * n)
    assert np.abs(dV - g[name + "/dV_dparam"]).max() < 1e-12
    gA, gB, gD = B.transform_local_grad(basis)
    assert np.abs(gD[0] - g[name + "/grad_D_A"]).max() < 1e-12
    assert np.abs(gD[1] - g[name + "/grad_D_D"]).max() < 1e-12


@pytest.mark.parametrize("tag", ["match", "match2"])
def test_G7_basis_matching(golden, tag):
    g = golden("G7_bcs.npz")
    out, gamma = B.basisMatching(g[tag + "/in"])
    ref = g[tag + "/out"]
    nb = ref.shape[-1]
    a, b = out.reshape(2, -1, nb), ref.reshape(2, -1, nb)
    for j in range(nb):
        sgn = np.sign(np.dot(a[0][:, j], b[0][:, j]))
        assert np.abs(a[0][:, j] - sgn * b[0][:, j]).max() < 1e-10
        assert np.abs(a[1][:, j] - sgn * b[1][:, j]).max() < 1e-10     # the pair flips together
    S = np.tensordot(out[0], out[1], axes=((0, 1), (0, 1)))
    assert np.abs(S - np.diag(gamma)).max() < 1e-12


def test_G7_unit2emb(golden):
    g = golden("G7_bcs.npz")
    neo = int(g["u2e/neo"])
    for k in ("4", "1", "8"):
        assert np.array_equal(B.unit2emb(g["u2e/in" + k], neo), g["u2e/out" + k])
    x = g["u2e/in4"]
    assert np.array_equal(B.reorder_spin_blocks(x), x[[0, 2, 1]])
    with pytest.raises(ValueError):
        B.unit2emb(np.zeros((1, 2, 2, 2)), neo)


@pytest.mark.parametrize("name", CASES)
def test_G18_complex_vcor_in_bdg_and_ghf(golden, name):
    """A COMPLEX local correlation potential (routine/mfd.py:439-447, 597-608): the restatement against the reference's own
    DiagBdG(symm) / DiagGHF(_symm) (golden G18, gen_G18)."""
    g7, g = golden("G7_bcs.npz"), golden("G18_branches.npz")
    mesh, FR, Fk, _, _, val = _case(g7, name)
    v, mu = g[name + "/vcor_complex"], float(g[name + "/mu"])
    for symm in (False, True):
        ew, ev = B.DiagBdG(Fk, v, mu, kmesh=mesh if symm else None)
        t = "bdg_symm" if symm else "bdg"
        assert np.abs(ew - g["%s/%s_ew" % (name, t)]).max() < 1e-11
        rho = np.einsum("kpm,km,kqm->kpq", ev, (ew < 0).astype(float), ev.conj())
        assert np.abs(rho - g["%s/%s_GRho_k" % (name, t)]).max() < 1e-10
    GFk = R.FFTtoK(g7[name + "/GFock_R"], mesh)
    for symm, mu_ in ((False, mu), (True, mu), (False, None)):
        ew, ev = B.DiagGHF(GFk, v, mu_, kmesh=mesh if symm else None)
        t = "ghf_symm" if symm else ("ghf" if mu_ is not None else "ghf_nomu")
        assert np.abs(ew - g["%s/%s_ew" % (name, t)]).max() < 1e-11
        if mu_ is not None:
            rho = np.einsum("kpm,km,kqm->kpq", ev, (ew < 0).astype(float), ev.conj())
            assert np.abs(rho - g["%s/%s_rho_k" % (name, t)]).max() < 1e-10


# ---- round 6: the BCS embedding Hamiltonian of model lattices (golden G28) -------------------------------------------------

BCS_HAM = ["c611", "c441"]
BCS_HAM_RUNS = [("nib", False, False), ("nib_fit", True, False), ("nib_jk", False, True)]


@pytest.mark.parametrize("name", BCS_HAM)
def test_G28_bcs_embedding_hamiltonian(golden, name):
    g, g7 = golden("G28_bcs_embham.npz"), golden("G7_bcs.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    basis, v, mu = g7[name + "/basis_proj"], g7[name + "/vcor"], float(g7[name + "/mu"])
    for tag, fitting, with_jk in BCS_HAM_RUNS:
        (H1, H0, ccdd), (He, e0) = B.bcs_embHam(mesh, basis, g[name + "/H3_R"], g[name + "/LatH2"], v, mu,
                                                ImpJK=g[name + "/JK_imp"] if with_jk else None, fitting=fitting)
        k = "%s/%s" % (name, tag)
        assert np.abs(H1["cd"] - g[k + "_cd"]).max() < 1e-12 and np.abs(H1["cc"] - g[k + "_cc"]).max() < 1e-12
        assert abs(H0 - float(g[k + "_H0"])) < 1e-12 and np.array_equal(ccdd, g[k + "_ccdd"])
        assert np.abs(He["cd"] - g[k + "_ecd"]).max() < 1e-12 and np.abs(He["cc"] - g[k + "_ecc"]).max() < 1e-12
        assert abs(e0 - float(g[k + "_eH0"])) < 1e-12


# ---- round 6: the BCS vcor fit in the embedding space (golden G30) ---------------------------------------------------------

BCS_FIT_RUNS = [("t0", np.inf, False), ("ft", 15.0, False), ("hcore_ft", 15.0, True)]


@pytest.mark.parametrize("name,n", [("c611", 2), ("c441", 4)])
def test_G30_bcs_fit_objective_and_gradient(golden, name, n):
    from oracle import restate_fit as F
    g, g7, g28 = golden("G30_bcs_fit.npz"), golden("G7_bcs.npz"), golden("G28_bcs_embham.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    basis, mu, H3 = g7[name + "/basis_proj"], float(g7[name + "/mu"]), g28[name + "/H3_R"]
    for vtag, res in (("u", False), ("r", True)):
        for tag, beta, hcore in BCS_FIT_RUNS:
            v = F.VcorLocal(res, True, n)
            fit = B.bcs_emb_fit(g[name + "/target"], mesh, basis, v, mu, beta, H3 if hcore else 1.1 * H3)
            key = "%s/%s_%s" % (name, vtag, tag)
            grad = fit.gradfunc if beta == np.inf else fit.gradfunc_ft
            for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
                assert abs(fit.errfunc(p) - e) < 1e-12, key
                assert np.abs(grad(p) - gr).max() < 1e-10 * max(1.0, np.abs(gr).max()), key
            pfit, (e0, e1) = g[key + "/param"], g[key + "/err"]
            assert abs(fit.errfunc(np.zeros_like(pfit)) - e0) < 1e-12 and abs(fit.errfunc(pfit) - e1) < 1e-11


# ---- round 6: Hartree-Fock-Bogoliubov mean field and the lattice stage of the BCS fit (golden G31) ---------------------------

HFB_RUNS = [("t0", np.inf, dict()), ("t0_symm", np.inf, dict(symm=True)), ("ft", 8.0, dict()), ("ft_fix", 8.0, dict(fix_mu=True)),
            ("ft_symm", 8.0, dict(symm=True)), ("t0_hcore", np.inf, dict(use_hcore=True))]


@pytest.mark.parametrize("name", ["c611", "c441", "c222"])
def test_G31_hfb(golden, name):
    g, g7 = golden("G31_hfb.npz"), golden("G7_bcs.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    FR, v, mu = g7[name + "/Fock_R"], g7[name + "/vcor"], float(g7[name + "/mu"])
    Fk = R.R2k(FR, mesh)
    for tag, beta, kw in HFB_RUNS:
        kw = dict(kw)
        f = 0.7 if kw.pop("use_hcore", False) else 1.0
        GT, n, E, res = B.HFB(mesh, f * Fk, f * FR, 0.7 * FR, v, mu, H0=0.3, beta=beta, **kw)
        key = "%s/%s" % (name, tag)
        assert np.abs(GT - g[key + "/GRhoT"]).max() < 1e-12 and abs(n - float(g[key + "/n"])) < 1e-12 and abs(E - float(g[key + "/E"])) < 1e-12
        assert np.abs(res["e"] - g[key + "/ew"]).max() < 1e-12
        assert np.abs(np.asarray([res["gap"], res["homo"], res["lumo"]]) - g[key + "/edges"]).max() < 1e-12


@pytest.mark.parametrize("name,n", [("c611", 2), ("c441", 4)])
def test_G31_bcs_full_fit_objective(golden, name, n):
    from oracle import restate_fit as F
    g, g7, g30 = golden("G31_hfb.npz"), golden("G7_bcs.npz"), golden("G30_bcs_fit.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    FR, mu, basis = g7[name + "/Fock_R"], float(g7[name + "/mu"]), g7[name + "/basis_proj"]
    assert np.abs(B.foldRho_bcs(g7[name + "/GRho"], mesh, basis) - g[name + "/foldRho"]).max() < 1e-12
    for tag, beta in (("t0", np.inf), ("ft", 8.0)):
        ef = B.bcs_full_errfunc(g30[name + "/target"], mesh, basis, F.VcorLocal(False, True, n), mu, beta, R.R2k(FR, mesh), FR)
        key = "%s/full_%s" % (name, tag)
        for p, e in zip(g[key + "/probe"], g[key + "/probe_err"]):
            assert abs(ef(p) - e) < 1e-12
        assert abs(ef(g[key + "/p0"]) - g[key + "/err"][0]) < 1e-12 and abs(ef(g[key + "/param"]) - g[key + "/err"][1]) < 1e-11
