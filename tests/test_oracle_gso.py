"""
Pins oracle/restate_gso.py (GSO twins) against tests/golden/G12_gso.npz captured from the
reference (spinless.get_emb_basis, get_emb_eri_gso) under oracle/shim.py.  CPU only.
"""
import numpy as np
import pytest

from oracle import restate as R
from oracle import restate_gso as G

BATH = [("c611", 2, [0, 1]), ("c441", 4, [0, 1, 2, 3]), ("c222", 5, [1, 2, 3])]
ERI = ["m311", "m221", "m222"]


def col_sign_dev(a, b):
    a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    return max(min(np.abs(a[:, j] - b[:, j]).max(), np.abs(a[:, j] + b[:, j]).max()) for j in range(a.shape[1]))


@pytest.mark.parametrize("name,n,val", BATH)
def test_G12_gso_bath(golden, name, n, val):
    g7, g = golden("G7_bcs.npz"), golden("G12_gso.npz")
    GRho = g7[name + "/GRho"]
    imp = list(val) + [i for i in range(n) if i > max(val)]
    for key, vb in (("basis", True), ("basis_full", False)):
        b, sigma, w = G.get_emb_basis_gso(GRho, n, val, imp, valence_bath=vb)
        ref = g["bath/%s/%s" % (name, key)]
        assert b.shape == ref.shape
        nimp = 2 * len(imp)
        assert np.array_equal(b[..., :nimp], ref[..., :nimp])
        assert col_sign_dev(b[..., nimp:], ref[..., nimp:]) < 1e-9


def eri_inputs(g, name):
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    W0, basis = g[name + "/W0"], g[name + "/basis"]
    ks = R.make_kpts_scaled(mesh)
    blocks = R.df_blocks_from_W0(W0, mesh, ks)
    return mesh, ks, blocks, W0.shape[0], W0.shape[2], basis


@pytest.mark.parametrize("name", ERI)
def test_G12_gso_eri(golden, name):
    g = golden("G12_gso.npz")
    mesh, ks, blocks, naux, nao, basis = eri_inputs(g, name)
    get = lambda i, j: blocks[(i, j)]
    for spin in (1, 2):
        st = "%s/s%d" % (name, spin)
        C = g[st + "/C_ao_lo"]
        for tr, key in ((True, "eri_tr"), (False, "eri_notr")):
            e = G.get_emb_eri_gso(mesh, ks, get, naux, nao, C, basis, t_reversal_symm=tr)
            assert np.abs(e - g[st + "/" + key]).max() < 1e-10 * max(1.0, np.abs(e).max())
        assert np.abs(G.get_emb_eri_gso(mesh, ks, get, naux, nao, C, basis, symmetry=1) - g[st + "/eri_s1"]).max() < 1e-10
        assert np.abs(G.get_emb_eri_gso(mesh, ks, get, naux, nao, C, basis, unit_eri=True) - g[st + "/eri_unit"]).max() < 1e-10


# ---- golden G18 (gen_G18): the 'eig' / 'ph' flavours of the GSO bath --------------------------------------------------------
G18_CASES = [("c611", (6, 1, 1), 2, [0, 1]), ("c441", (4, 4, 1), 4, [0, 1, 2, 3]), ("c222", (2, 2, 2), 5, [1, 2, 3])]


def _same_span(a, b, tol):
    """Two column sets span the same space (bases differ by a rotation inside degenerate groups)."""
    a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    assert a.shape == b.shape
    return np.abs(a @ a.T - b @ b.T).max() < tol


@pytest.mark.parametrize("name,mesh,n,val", G18_CASES)
def test_G18_gso_eig_and_ph_bath(golden, name, mesh, n, val):
    """routine/spinless.py:166-275 ('eig') and :351-423 ('ph') against what the reference's own functions returned."""
    from oracle import restate_gso as G
    g, g7 = golden("G18_branches.npz"), golden("G7_bcs.npz")
    GRho = g7[name + "/GRho"]
    imp = list(val) + [i for i in range(n) if i > max(val)]                      # Lattice.imp_idx = val + virt (system/lattice.py:119-120)
    for vb, tag in ((True, "val"), (False, "full")):
        be, _ = G.get_emb_basis_gso_eig(mesh, GRho, n, val, imp, valence_bath=vb)
        ref = g["%s/gso_eig_%s" % (name, tag)]
        assert be.shape == ref.shape
        nimp = 2 * len(imp)
        assert np.array_equal(be[..., :nimp], ref[..., :nimp])                     # impurity columns: the identity block
        assert _same_span(be[..., nimp:], ref[..., nimp:], 1e-10)
        bp, _ = G.get_emb_basis_gso_ph(GRho, n, val, imp, valence_bath=vb)
        refp = g["%s/gso_ph_%s" % (name, tag)]
        assert bp.shape == refp.shape and _same_span(bp, refp, 1e-10)
        B = bp.reshape(-1, bp.shape[-1])
        assert np.abs(B.T @ B - np.eye(B.shape[-1])).max() < 1e-10


# ---- golden G19 (gen_G19): bath_opt, the embedding space rotated to an integer electron number ------------------------------------
@pytest.mark.parametrize("name,mesh,n,val", G18_CASES)
def test_G19_bath_opt(golden, name, mesh, n, val):
    """routine/spinless.py:274-349 against what the reference returned on Fermi-smeared (metallic) generalised density matrices:
    the rotated space (projector; the columns are eigenvectors of a nearly degenerate cluster), its electron number, and the
    keep_imp_identity variant column by column (its columns are fixed by the Gram-Schmidt order up to the eigenvector signs)."""
    from oracle import restate_gso as G
    from oracle.restate import CellArith
    g = golden("G19_bath_opt.npz")
    imp = list(val) + [i for i in range(n) if i > max(val)]
    for tag in ("a", "b"):
        GRho = g["%s/%s/GRho" % (name, tag)]
        D = CellArith(mesh).expand(GRho[None])[0]
        for vb in ("val", "full"):
            key = "%s/%s/%s" % (name, tag, vb)
            b0, ref, refk = g[key + "/basis_svd"], g[key + "/basis_opt"], g[key + "/basis_opt_keep"]
            out, mu, nelec = G.get_emb_basis_opt(mesh, GRho, b0)
            assert mu is not None and abs(nelec - round(nelec)) < 1e-5
            R = ref.reshape(-1, ref.shape[-1])
            assert abs(np.trace(R.T @ D @ R) - nelec) < 1e-9
            assert _same_span(out, ref, 1e-8)
            outk, muk, _ = G.get_emb_basis_opt(mesh, GRho, b0, keep_imp_identity=True, nimp=len(imp))
            assert abs(muk - mu) < 1e-12 and _same_span(outk, refk, 1e-8)
            assert np.array_equal(outk[..., :len(imp)], refk[..., :len(imp)])
    # an integer electron number already (the gapped T = 0 matrices of G7 without core orbitals): returned unchanged (:292-293)
    g7 = golden("G7_bcs.npz")
    GRho0 = g7[name + "/GRho"]
    b = G.get_emb_basis_gso(GRho0, n, val, imp)
    b = b[0] if isinstance(b, tuple) else b
    same, mu0, nelec0 = G.get_emb_basis_opt(mesh, GRho0, b)
    if min(val) == 0:
        assert mu0 is None and same is b and abs(nelec0 - round(nelec0)) < 1e-6


# ---- round 6: the GSO embedding Hamiltonian (golden G27) -------------------------------------------------------------------

GSO_HAM = ["c611", "c441", "c222"]


def gso_ham_inputs(g, name):
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    basis = g[name + "/basis"]
    n, nk = basis.shape[1] // 2, basis.shape[0]
    S3 = np.zeros((3, nk, n, n), dtype=complex)
    S3[0] = S3[1] = np.eye(n)
    return mesh, basis, g[name + "/H2"], g[name + "/H3_k"], g[name + "/F3_k"], S3, g[name + "/GRho_k"], g[name + "/vcor"], 0.37


def gso_ham_runs(g, name):
    JK3, add2, cust = g[name + "/JK_imp"], g[name + "/hcore_add"], g[name + "/hcore_custom"]
    return [("ib", dict()), ("ib_vcor", dict(add_vcor=True)), ("ib_vcor_fit", dict(add_vcor=True, fitting=True)),
            ("ib_add", dict(hcore_add=add2)), ("ib_custom", dict(hcore_custom=cust)),
            ("nib", dict(int_bath=False)), ("nib_jk", dict(int_bath=False, JK_imp=JK3)), ("nib_add", dict(int_bath=False, hcore_add=add2)),
            ("nib_hcore", dict(int_bath=False, use_hcore_as_emb_ham=True, hcore_add=add2))]


@pytest.mark.parametrize("name", GSO_HAM)
def test_G27_gso_embedding_hamiltonian(golden, name):
    g = golden("G27_gso_embham.npz")
    mesh, basis, H2, H3, F3, S3, rk, v, mu = gso_ham_inputs(g, name)
    for tag, kw in gso_ham_runs(g, name):
        fock = 0.9 * F3 if kw.get("int_bath", True) else F3           # fock_hf_lo_k with an interacting bath (spinless.py:647-650)
        H1, ov, JKc = G.gso_embHam1e(mesh, basis, H2, H3, fock, S3, rk, v, mu, **kw)
        assert np.abs(H1 - g["%s/%s_H1" % (name, tag)]).max() < 1e-12, tag
        assert np.abs(ov - g["%s/%s_ovlp" % (name, tag)]).max() < 1e-12, tag
        key = "%s/%s_JK_core" % (name, tag)
        assert (np.abs(JKc - g[key]).max() < 1e-12) if key in g else JKc is None
    bk = R.R2k(basis, mesh)
    assert np.abs(G.transform_trans_inv_k_gso(bk, F3) - g[name + "/ti_k3"]).max() < 1e-12
    assert np.abs(G.transform_trans_inv_k_gso(bk, F3[:2]) - g[name + "/ti_k2"]).max() < 1e-12
    for k3, k2, fn in (("loc3", "loc2", G.transform_local_gso), ("imp3", "imp2", G.transform_imp_gso)):
        assert np.abs(fn(basis, v) - g[name + "/" + k3]).max() < 1e-12 and np.abs(fn(basis, v[:2]) - g[name + "/" + k2]).max() < 1e-12
    assert np.array_equal(G.unit2emb_gso(g[name + "/unit"], basis.shape[-1]), g[name + "/unit2emb"])
    if name + "/eri_local" in g:
        assert np.abs(G.transform_eri_local_gso(basis, g[name + "/unit"]) - g[name + "/eri_local"]).max() < 1e-12


# ---- round 6: the GSO vcor fit in the embedding space (golden G29) ---------------------------------------------------------

GSO_FIT = [("c611", 2, [0, 1]), ("c441", 4, [0, 1, 2, 3]), ("c222", 5, [1, 2, 3])]
GSO_FIT_RUNS = [("t0", np.inf, dict()), ("ft", 15.0, dict()), ("imp_t0", np.inf, dict(imp_fit=True)), ("det_ft", 15.0, dict(det=True)),
                ("fixmu_ft", 15.0, dict(fix_mu=True, mu0=0.05)), ("hcore_t0", np.inf, dict(hcore=True))]


@pytest.mark.parametrize("name,n,val", GSO_FIT)
def test_G29_gso_fit_objective_and_gradient(golden, name, n, val):
    from oracle import restate_fit as F
    g, g27 = golden("G29_gso_fit.npz"), golden("G27_gso_embham.npz")
    mesh, basis, H2, H3, F3, S3, rk, vmat, mu = gso_ham_inputs(g27, name)
    v = F.VcorLocal(False, True, n)
    assert np.abs(G.get_dV_dparam_gso(v, basis) - g[name + "/dV_compact"]).max() < 1e-13
    assert np.abs(G.get_dV_dparam_gso(v, basis, compact=False) - g[name + "/dV_full"]).max() < 1e-13
    nimp = len(val) + len([i for i in range(n) if i > max(val)])
    for tag, beta, kw in GSO_FIT_RUNS:
        kw = dict(kw)
        fock = H3 if kw.pop("hcore", False) else F3
        fit = G.gso_emb_fit(g[name + "/target"], mesh, basis, v, mu, beta, fock, S3, nimp, **kw)
        key = "%s/%s" % (name, tag)
        grad = fit.gradfunc if beta == np.inf else fit.gradfunc_ft
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-12, key
            assert np.abs(grad(p) - gr).max() < 1e-10 * max(1.0, np.abs(gr).max()), key
        pfit, (e0, e1) = g[key + "/param"], g[key + "/err"]
        assert abs(fit.errfunc(np.zeros_like(pfit)) - e0) < 1e-12 and abs(fit.errfunc(pfit) - e1) < 1e-11


# ---- round 6: generalised Hartree-Fock lattice mean field (golden G33) -----------------------------------------------------

GHF_RUNS = [("t0", np.inf, dict()), ("t0_nosymm", np.inf, dict(symm=False)), ("ft", 9.0, dict()), ("ft_fix", 9.0, dict(fix_mu=True, mu0=0.05)),
            ("t0_hcore", np.inf, dict(use_hcore=True)), ("ft_f04", 9.0, dict(filling=0.4)), ("ft_nfrac", 9.0, dict(nfrac=2))]


@pytest.mark.parametrize("name", GSO_HAM)
def test_G33_ghf(golden, name):
    g, g27, g7 = golden("G33_ghf.npz"), golden("G27_gso_embham.npz"), golden("G7_bcs.npz")
    mesh = tuple(int(x) for x in g27[name + "/mesh"])
    H3, F3, v = g27[name + "/H3_k"], g27[name + "/F3_k"], g27[name + "/vcor"]
    for tag, beta, kw in GHF_RUNS:
        kw = dict(kw)
        hcore = kw.pop("use_hcore", False)
        GT, n, E, res = G.GHF(mesh, H3, H3 if hcore else F3, v, 0.37, H0=0.3, beta=beta, **kw)
        key = "%s/%s" % (name, tag)
        assert np.abs(GT - g[key + "/GRhoT"]).max() < 1e-12 and abs(n - float(g[key + "/n"])) < 1e-12 and abs(E - float(g[key + "/E"])) < 1e-12
        assert np.abs(res["e"] - g[key + "/ew"]).max() < 1e-12 and np.abs(res["mo_occ"] - g[key + "/occ"]).max() < 1e-12
        edges = np.asarray([res["gap"], res["homo"], res["lumo"], res["mu_quasi"], res["nerr"]])
        assert np.abs(edges - g[key + "/edges"]).max() < 1e-12
    FR = g7[name + "/Fock_R"]
    GT, n, E, res = G.GHF(mesh, R.R2k(0.7 * FR, mesh), R.R2k(FR, mesh), v, 0.37, H0=0.3, ph_trans=True)
    assert np.abs(GT - g[name + "/ph/GRhoT"]).max() < 1e-12 and abs(E - float(g[name + "/ph/E"])) < 1e-12


# ---- round 6: the lattice stage of the GSO fit (golden G35) ----------------------------------------------------------------

GSO_FULL_RUNS = [("ft_imp", 12.0, dict(imp_fit=True), 10), ("ft_det", 12.0, dict(det=True), 10), ("ft_bogo", 12.0, dict(imp_fit=True, bogo_only=True), 10),
                 ("ft_fixmu", 12.0, dict(imp_fit=True, fix_mu=True), 10), ("t0_num", np.inf, dict(imp_fit=True, num_grad=True), 2)]


@pytest.mark.parametrize("name,n,val", GSO_FIT)
def test_G35_gso_lattice_fit_objective_and_gradient(golden, name, n, val):
    from oracle import restate_fit as F
    g, g27 = golden("G35_gso_full_fit.npz"), golden("G27_gso_embham.npz")
    mesh = tuple(int(x) for x in g27[name + "/mesh"])
    idx = list(range(len(val) + len([i for i in range(n) if i > max(val)])))
    for tag, beta, kw, _ in GSO_FULL_RUNS:
        fit = G.GsoFullFit(g[name + "/target"], mesh, F.VcorLocal(False, True, n), 0.37, beta, g27[name + "/F3_k"],
                           imp_idx=idx if kw.get("imp_fit") else [], det_idx=idx if kw.get("det") else [],
                           fix_mu=kw.get("fix_mu", False), bogo_only=kw.get("bogo_only", False))
        key = "%s/%s" % (name, tag)
        for i, p in enumerate(g[key + "/probe"]):
            assert abs(fit.errfunc(p) - g[key + "/probe_err"][i]) < 1e-12, key
            if key + "/probe_grad" in g:
                gr = g[key + "/probe_grad"][i]
                assert np.abs(fit.gradfunc_ft(p) - gr).max() < 1e-10 * max(1.0, np.abs(gr).max()), key
        assert abs(fit.errfunc(g[key + "/p0"]) - g[key + "/err"][0]) < 1e-12 and abs(fit.errfunc(g[key + "/param"]) - g[key + "/err"][1]) < 1e-10


@pytest.mark.parametrize("name,n,val", GSO_FIT)
def test_G37_gso_lattice_fit_with_mu_first_objective(golden, name, n, val):
    """spinless.FitVcorFull_mu: the objective at the starting parameters -- the one evaluation whose inner chemical-potential search
    starts from a known point -- against the reference (golden G37; the search stops at 1e-6 in the electron number)."""
    from oracle import restate_fit as F
    g, g27, g35 = golden("G37_gso_full_fit_mu.npz"), golden("G27_gso_embham.npz"), golden("G35_gso_full_fit.npz")
    mesh = tuple(int(x) for x in g27[name + "/mesh"])
    idx = list(range(len(val) + len([i for i in range(n) if i > max(val)])))
    for tag, filling, kw in (("ft_imp", 0.5, dict(imp_idx=idx)), ("ft_det", 0.45, dict(det_idx=idx)), ("ft_bogo", 0.55, dict(imp_idx=idx, bogo_only=True))):
        fit = G.GsoFullFitMu(g35[name + "/target"], mesh, F.VcorLocal(False, True, n), filling, 12.0, g27[name + "/F3_k"], 0.37, **kw)
        key = "%s/%s" % (name, tag)
        assert abs(fit.errfunc(g[key + "/p0"]) - g[key + "/err"][0]) < 1e-5, key
