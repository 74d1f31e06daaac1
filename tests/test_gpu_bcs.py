"""
GPU parity tests (-m gpu) for the rows of SURVEY.md section 8a completed after the ERI path: a3 GHF / BdG
diagonalisations, a7 basisMatching, a8 BCS twin (bcs_helper folds, embBasis), a14 unit2emb.  The HIP path,
through the C ABI, against oracle/restate_bcs.py and the golden fixture G7 captured from the reference.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import restate as R
from oracle import restate_bcs as B
from tests.test_oracle_bcs import CASES, _case, col_sign_dev


@pytest.fixture(scope="module")
def ctx():
    from libdmet_preview_amd import _lib
    return _lib.get_ctx()


def _lattice(mesh, nlo, val):
    from libdmet_preview_amd.system.lattice import Lattice
    L = Lattice(int(nlo), mesh)
    L.val_idx, L.virt_idx, L.core_idx = list(val), [], [i for i in range(nlo) if i not in val]
    return L


class _Vcor(object):
    def __init__(self, v):
        self.value = v

    def get(self, i=0, kspace=True):
        return self.value if (kspace or i == 0) else np.zeros_like(self.value)

    def length(self):
        n = self.value.shape[-1]
        return n * (n + 1) + n * n


def _occ_proj(ew, ev):
    return np.einsum("kpm,km,kqm->kpq", ev, (ew < 0).astype(float), ev.conj())


@pytest.mark.parametrize("name", CASES)
def test_bdg_ghf(ctx, golden, name):
    from libdmet_preview_amd.routine import mfd
    g = golden("G7_bcs.npz")
    mesh, FR, Fk, v, mu, val = _case(g, name)
    L = _lattice(mesh, FR.shape[-1], val)
    vc = _Vcor(v)
    for symm in (False, True):
        ew, ev = (mfd.DiagBdGsymm(Fk, vc, mu, L) if symm else mfd.DiagBdG(Fk, vc, mu))
        t = "bdg_symm" if symm else "bdg"
        assert np.abs(ew - g["%s/%s_ew" % (name, t)]).max() < 1e-10
        assert np.abs(_occ_proj(ew, ev) - g["%s/%s_GRho_k" % (name, t)]).max() < 1e-10
        m = ev.shape[-1]
        assert np.abs(np.einsum("kpm,kpn->kmn", ev.conj(), ev) - np.eye(m)).max() < 1e-12
    # restricted input (one Fock for both spins) takes the same path
    ew1, _ = mfd.DiagBdG(Fk[0], vc, mu)
    ewo, _ = B.DiagBdG(Fk[0], v, mu)
    assert np.abs(ew1 - ewo).max() < 1e-10
    GFk = R.FFTtoK(g[name + "/GFock_R"], mesh)
    for symm, mu_ in ((False, mu), (True, mu), (False, None)):
        ew, ev = (mfd.DiagGHF_symm(GFk, vc, mu_, L) if symm else mfd.DiagGHF(GFk, vc, mu_))
        t = "ghf_symm" if symm else ("ghf" if mu_ is not None else "ghf_nomu")
        assert np.abs(ew - g["%s/%s_ew" % (name, t)]).max() < 1e-10
        if mu_ is not None:
            assert np.abs(_occ_proj(ew, ev) - g["%s/%s_rho_k" % (name, t)]).max() < 1e-10


@pytest.mark.parametrize("name", CASES)
def test_bdg_ghf_complex_local_vcor(ctx, golden, name):
    """A COMPLEX local correlation potential in DiagBdG(symm) / DiagGHF(_symm) (routine/mfd.py:439-447, 597-608): the
    reference's own results (golden G18)."""
    from libdmet_preview_amd.routine import mfd
    g7, g = golden("G7_bcs.npz"), golden("G18_branches.npz")
    mesh, FR, Fk, _, _, val = _case(g7, name)
    L = _lattice(mesh, FR.shape[-1], val)
    v, mu = g[name + "/vcor_complex"], float(g[name + "/mu"])
    vc = _Vcor(v)
    for symm in (False, True):
        ew, ev = (mfd.DiagBdGsymm(Fk, vc, mu, L) if symm else mfd.DiagBdG(Fk, vc, mu))
        t = "bdg_symm" if symm else "bdg"
        assert np.abs(ew - g["%s/%s_ew" % (name, t)]).max() < 1e-10
        assert np.abs(_occ_proj(ew, ev) - g["%s/%s_GRho_k" % (name, t)]).max() < 1e-10
    GFk = R.FFTtoK(g7[name + "/GFock_R"], mesh)
    for symm, mu_ in ((False, mu), (True, mu), (False, None)):
        ew, ev = (mfd.DiagGHF_symm(GFk, vc, mu_, L) if symm else mfd.DiagGHF(GFk, vc, mu_))
        t = "ghf_symm" if symm else ("ghf" if mu_ is not None else "ghf_nomu")
        assert np.abs(ew - g["%s/%s_ew" % (name, t)]).max() < 1e-10
        if mu_ is not None:
            assert np.abs(_occ_proj(ew, ev) - g["%s/%s_rho_k" % (name, t)]).max() < 1e-10


@pytest.mark.parametrize("name", CASES)
def test_mfd_mpi_ghf_symm(ctx, golden, name):
    """routine/mfd_mpi.py twin: irreducible k points from the integer tables, reference call signature
    (cell, GFock, vcor_mat, mu, kpairs, kidx); single rank here, the sharded sum is covered by the gloo test."""
    from libdmet_preview_amd.routine import mfd_mpi
    g = golden("G7_bcs.npz")
    mesh, FR, Fk, v, mu, val = _case(g, name)
    GFk = R.FFTtoK(g[name + "/GFock_R"], mesh)
    ks = R.make_kpts_scaled(mesh)
    cell = type("Cell", (), {"get_scaled_kpts": staticmethod(lambda k: np.asarray(k))})()
    kpairs, kidx = mfd_mpi.get_kpairs_kidx(cell, ks)
    ew, ev = mfd_mpi.DiagGHF_symm(cell, GFk, np.asarray(v), mu, kpairs, kidx)
    assert np.abs(ew - g[name + "/ghf_symm_ew"]).max() < 1e-10
    assert np.abs(_occ_proj(ew, ev) - g[name + "/ghf_symm_rho_k"]).max() < 1e-10


@pytest.mark.parametrize("name", CASES)
def test_bcs_emb_basis(ctx, golden, name):
    from libdmet_preview_amd.routine import bcs
    g = golden("G7_bcs.npz")
    mesh, FR, Fk, v, mu, val = _case(g, name)
    n = FR.shape[-1]
    L = _lattice(mesh, n, val)
    GRho = g[name + "/GRho"]
    basis = bcs.embBasis(L, GRho)
    ref = g[name + "/basis_proj"]
    assert basis.shape == ref.shape
    assert np.array_equal(basis[:, 0], ref[:, 0])
    for s in range(2):
        assert col_sign_dev(basis[s][1:, :, n:], ref[s][1:, :, n:]) < 1e-9
    # both spins together span the left singular space: gauge-free projector check (<= 1e-10 Frobenius)
    o_basis, o_sigma, o_B, o_w = B.embBasis_proj(GRho, n, val)
    Bd = bcs.embBasis(L, GRho, only_return_bath=True)
    a, b = Bd.reshape(-1, Bd.shape[-1]), o_B.reshape(-1, o_B.shape[-1])
    assert np.sqrt(2.0) * np.linalg.norm(b - a @ (a.T @ b)) < 1e-10
    assert np.abs(a.T @ a - np.eye(a.shape[1])).max() < 1e-12
    assert bcs.get_emb_basis is bcs.embBasis
    with pytest.raises(NotImplementedError):
        bcs.embBasis(L, GRho, sites=[0])
    if name + "/basis_phsymm" in g:
        ph = bcs.embBasis(L, GRho, local=False)
        refp = g[name + "/basis_phsymm"]
        for s in range(2):
            a, b = ph[s].reshape(-1, 2 * n), refp[s].reshape(-1, 2 * n)
            assert np.abs(a @ a.T - b @ b.T).max() < 1e-9
            assert np.abs(a.T @ a - np.eye(2 * n)).max() < 1e-10
    else:
        with pytest.raises(Exception):
            bcs.embBasis(L, GRho, local=False)


@pytest.mark.parametrize("name", CASES)
def test_bcs_folds(ctx, golden, name):
    from libdmet_preview_amd.routine import bcs_helper as bh
    g = golden("G7_bcs.npz")
    mesh, FR, Fk, v, mu, val = _case(g, name)
    n = FR.shape[-1]
    L = _lattice(mesh, n, val)
    basis = g[name + "/basis_proj"]
    D_R = g[name + "/GFock_R"][:, :n, n:]
    H3 = np.asarray([FR[0], FR[1], D_R])
    todo = [("ti3", bh.transform_trans_inv, H3), ("ti2", bh.transform_trans_inv, FR), ("ti1", bh.transform_trans_inv, FR[0]),
            ("loc3", bh.transform_local, v), ("loc2", bh.transform_local, v[:2]), ("loc1", bh.transform_local, v[0]),
            ("imp3", bh.transform_imp, v), ("imp1", bh.transform_imp, v[0]),
            ("ie3", bh.transform_imp_env, H3), ("ie1", bh.transform_imp_env, FR[0])]
    for tag, fn, H in todo:
        (hA, hB), hD, e0 = fn(basis, L, H)
        assert np.abs(np.asarray([hA, hB, hD]) - g["%s/%s_H" % (name, tag)]).max() < 1e-10, tag
        assert abs(e0 - float(g["%s/%s_E0" % (name, tag)])) < 1e-10, tag
    VA, VB, UA, UB = bh.separate_basis(basis)
    assert np.abs(bh.contract_trans_inv(VA, UB, L, D_R) - B.contract_trans_inv(VA, UB, mesh, D_R)).max() < 1e-11
    assert np.abs(bh.contract_local(VB, UA, L, v[2]) - B.contract_local(VB, UA, mesh, v[2])).max() < 1e-12
    assert np.abs(bh.contract_imp_env(VA, VB, L, FR[1]) - B.contract_imp_env(VA, VB, mesh, FR[1])).max() < 1e-12
    dV = bh.get_dV_dparam(basis, L, _Vcor(v))
    assert np.abs(dV - g[name + "/dV_dparam"]).max() < 1e-12
    gA, gB, gD = bh.transform_local_grad(basis, L)
    assert np.abs(gD[0] - g[name + "/grad_D_A"]).max() < 1e-12
    assert np.abs(gD[1] - g[name + "/grad_D_D"]).max() < 1e-12
    assert np.abs(bh.contract_local_grad(VA, UB, L) - B.contract_local_grad(VA, UB)).max() < 1e-12
    assert np.abs(bh.contract_local_grad_DT(UB, VA, L) - B.contract_local_grad_DT(UB, VA)).max() < 1e-12
    # bookkeeping mirrors are bit-exact
    G0 = g[name + "/GRho"][0]
    assert np.array_equal(np.asarray(bh.extractRdm(G0)), g[name + "/extractRdm"])
    assert np.array_equal(np.asarray(bh.extractH1(G0)), g[name + "/extractH1"])
    assert np.array_equal(bh.swapSpin(G0), g[name + "/swapSpin"])
    assert np.array_equal(bh.basisToCanonical(basis), g[name + "/canonical"])
    assert np.array_equal(bh.basisToSpin(g[name + "/canonical"]), basis)


@pytest.mark.parametrize("tag", ["match", "match2"])
def test_basis_matching(ctx, golden, tag):
    from libdmet_preview_amd.dmet.HubPhSymm import basisMatching
    g = golden("G7_bcs.npz")
    out = basisMatching(g[tag + "/in"])
    ref = g[tag + "/out"]
    nb = ref.shape[-1]
    a, b = out.reshape(2, -1, nb), ref.reshape(2, -1, nb)
    for j in range(nb):
        sgn = np.sign(np.dot(a[0][:, j], b[0][:, j]))
        assert np.abs(a[0][:, j] - sgn * b[0][:, j]).max() < 1e-10
        assert np.abs(a[1][:, j] - sgn * b[1][:, j]).max() < 1e-10
    S = np.tensordot(out[0], out[1], axes=((0, 1), (0, 1)))
    assert np.abs(S - np.diag(np.diag(S))).max() < 1e-12
    assert (np.diff(np.diag(S)) <= 1e-14).all()


@pytest.mark.parametrize("n", [1, 2, 7, 33, 56, 84])
def test_svd_small(ctx, n):
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    if n >= 7:
        A[:, 3] = A[:, 1] * 1e-9 + A[:, 2]          # nearly dependent columns: tiny singular value
    dA = ctx.to_device(A)
    ds, dU, dVt = ctx.empty((n,), np.float64), ctx.empty((n, n), np.float64), ctx.empty((n, n), np.float64)
    ctx.check(lib.dmk_svd_small(ctx.h, n, dA.ptr, ds.ptr, dU.ptr, dVt.ptr))
    s, U, Vt = ds.get(), dU.get(), dVt.get()
    sref = np.linalg.svd(A, compute_uv=False)
    assert np.abs(s - sref).max() < 1e-13 * max(1.0, sref[0])
    assert np.abs((U * s) @ Vt - A).max() < 1e-12 * max(1.0, sref[0])
    assert np.abs(Vt @ Vt.T - np.eye(n)).max() < 1e-12
    assert np.abs(U.T @ U - np.eye(n)).max() < 1e-7    # columns of tiny sigma lose orthogonality like 1/sigma


def test_svd_small_rejects_big(ctx):
    from libdmet_preview_amd._lib import lib, DmkError
    d = ctx.zeros((85, 85), np.float64)
    with pytest.raises(DmkError):
        ctx.check(lib.dmk_svd_small(ctx.h, 85, d.ptr, d.ptr, d.ptr, d.ptr))


@pytest.mark.parametrize("M,N,K", [(5, 3, 7), (130, 64, 33), (400, 257, 1000), (56, 56, 43200)])
def test_dgemm_tn_acc_rect(ctx, M, N, K):
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(M + N + K)
    X, Y, C0 = rng.standard_normal((K, M)), rng.standard_normal((K, N)), rng.standard_normal((M, N))
    dX, dY, dC = ctx.to_device(X), ctx.to_device(Y), ctx.to_device(C0)
    ctx.check(lib.dmk_dgemm_tn_acc_rect(ctx.h, M, N, K, -0.5, dX.ptr, M, dY.ptr, N, dC.ptr, N))
    ref = C0 - 0.5 * X.T @ Y
    assert np.abs(dC.get() - ref).max() < 1e-12 * np.sqrt(K) * 10


def test_dgemm_nn_small(ctx):
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(3)
    A, Bm = rng.standard_normal((1000, 9)), rng.standard_normal((9, 6))
    dA, dB, dC = ctx.to_device(A), ctx.to_device(Bm), ctx.empty((1000, 6), np.float64)
    ctx.check(lib.dmk_dgemm_nn_small(ctx.h, 1000, 6, 9, dA.ptr, dB.ptr, 0, dC.ptr))
    assert np.abs(dC.get() - A @ Bm).max() < 1e-13
    dBt = ctx.to_device(np.ascontiguousarray(Bm.T))
    ctx.check(lib.dmk_dgemm_nn_small(ctx.h, 1000, 6, 9, dA.ptr, dBt.ptr, 1, dC.ptr))
    assert np.abs(dC.get() - A @ Bm).max() < 1e-13


def test_unit2emb(ctx, golden):
    from libdmet_preview_amd.routine import slater_helper as sh
    g = golden("G7_bcs.npz")
    neo = int(g["u2e/neo"])
    for k in ("4", "1", "8"):
        assert np.array_equal(sh.unit2emb(g["u2e/in" + k], neo), g["u2e/out" + k])
    x = g["u2e/in4"]
    assert np.array_equal(sh.reorder_spin_blocks(x), x[[0, 2, 1]])
    assert np.array_equal(sh.reorder_spin_blocks(x[:1]), x[:1])
    with pytest.raises(ValueError):
        sh.unit2emb(np.zeros((1, 2, 2, 2)), neo)
    with pytest.raises(ValueError):
        sh.init_H2(4, 2)
    box = {"ccdd": np.array(g["u2e/in4"])}
    out = sh.unit2emb(box, neo)
    assert out is box and np.array_equal(box["ccdd"], g["u2e/out4"])


def test_bcs_emb_basis_without_entanglement(ctx):
    """A generalised density whose env x imp block vanishes (one band, full at every k: found by tools/bcs_stress.py) or is rank
    deficient: the reference keeps all 2 nval left singular vectors and LAPACK completes the vanishing ones orthonormally
    (bcs.py:46, 84-103); the device factorisation has no direction for them, the host binding completes them -- the embedding basis
    must stay orthonormal, and the well-defined part must still be the oracle's."""
    from libdmet_preview_amd.routine import bcs
    mesh, n = (1, 1, 4), 1
    L = _lattice(mesh, n, [0])
    GRho = np.zeros((4, 2, 2))
    GRho[0] = np.diag([1.0, 0.0])
    b = bcs.embBasis(L, GRho)
    assert b.shape == (2, 4, 2, 2)
    for s in range(2):
        a = b[s].reshape(-1, b.shape[-1])
        assert np.abs(a.T @ a - np.eye(a.shape[1])).max() < 1e-12
    # rank one: a single entangled direction, the second column is a completion
    rng = np.random.default_rng(3)
    u = rng.standard_normal(6)
    u /= np.linalg.norm(u)
    A = 0.3 * np.outer(u, [0.8, 0.6])
    GRho2 = np.zeros((4, 2, 2))
    GRho2[0] = np.diag([0.6, 0.4])
    GRho2[1:] = A.reshape(3, 2, 2)
    Bd = bcs.embBasis(L, GRho2, only_return_bath=True)
    a = Bd.reshape(-1, Bd.shape[-1])
    assert np.abs(a.T @ a - np.eye(2)).max() < 1e-12
    assert abs(abs(a[:, 0] @ u) - 1.0) < 1e-12                     # the entangled direction itself


@pytest.mark.parametrize("name", ["c611", "c441"])
def test_bcs_embedding_hamiltonian(ctx, golden, name):
    """bcs.embHam (routine/bcs.py:137-318) on the branch the reference implements -- model lattice, local basis, bare bath on
    hcore, 'local' ERI -- against the reference's values (golden G28); the branches the reference refuses are refused."""
    from libdmet_preview_amd.routine import bcs
    from tests.test_oracle_bcs import BCS_HAM_RUNS
    g, g7 = golden("G28_bcs_embham.npz"), golden("G7_bcs.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    basis, v, mu = g7[name + "/basis_proj"], g7[name + "/vcor"], float(g7[name + "/mu"])
    n = v.shape[-1]
    L = _lattice(mesh, n, [int(x) for x in g7[name + "/val"]])
    L.set_Ham_lo(fock_lo_R=g[name + "/H3_R"], hcore_lo_R=g[name + "/H3_R"])
    L.set_H2_local(g[name + "/LatH2"])
    L.use_hcore_as_emb_ham = True
    vc = _Vcor(v)
    vc.islocal = lambda: True
    for tag, fitting, with_jk in BCS_HAM_RUNS:
        L.JK_imp = g[name + "/JK_imp"] if with_jk else None
        L.JK_core = "unset"
        Himp, (He, e0) = bcs.embHam(L, basis, vc, mu, fitting=fitting)
        k = "%s/%s" % (name, tag)
        assert L.JK_core is None and Himp.bogoliubov and not Himp.restricted and Himp.norb == int(g[k + "_shapes"][2])
        assert np.abs(Himp.H1["cd"] - g[k + "_cd"]).max() < 1e-11 and np.abs(Himp.H1["cc"] - g[k + "_cc"]).max() < 1e-11
        assert abs(Himp.H0 - float(g[k + "_H0"])) < 1e-11
        assert np.array_equal(Himp.H2["ccdd"], g[k + "_ccdd"])
        assert Himp.H2["cccd"].shape[0] == 2 and Himp.H2["cccc"].shape[0] == 1 and not np.any(Himp.H2["cccd"]) and not np.any(Himp.H2["cccc"])
        assert np.abs(He["cd"] - g[k + "_ecd"]).max() < 1e-11 and np.abs(He["cc"] - g[k + "_ecc"]).max() < 1e-11
        assert abs(e0 - float(g[k + "_eH0"])) < 1e-11
    L.JK_imp = None
    assert bcs.get_emb_Ham is bcs.embHam
    for kw in (dict(int_bath=True), dict(local=False), dict(sites=[0])):
        with pytest.raises(NotImplementedError):
            bcs.embHam(L, basis, vc, mu, **kw)
    L.use_hcore_as_emb_ham = False
    with pytest.raises(NotImplementedError):
        bcs.embHam(L, basis, vc, mu)


@pytest.mark.parametrize("name,n", [("c611", 2), ("c441", 4)])
def test_bcs_vcor_fit(ctx, golden, name, n):
    """bcs.FitVcorEmb (routine/bcs.py:356-530) on the device against the reference's closures and fits (golden G30): T = 0 and
    finite T around the fixed mu0 = 0, Fock or hcore as the embedding Hamiltonian, unrestricted and restricted potentials with
    pairing (the first through the all-parameters table of bcs_helper.get_dV_dparam, the second through the per-parameter folds)."""
    from libdmet_preview_amd.routine import bcs
    from libdmet_preview_amd.dmet import Hubbard
    from tests.test_oracle_bcs import BCS_FIT_RUNS
    g, g7, g28 = golden("G30_bcs_fit.npz"), golden("G7_bcs.npz"), golden("G28_bcs_embham.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    basis, mu, H3 = g7[name + "/basis_proj"], float(g7[name + "/mu"]), g28[name + "/H3_R"]
    L = _lattice(mesh, n, [int(x) for x in g7[name + "/val"]])
    L.set_Ham_lo(fock_lo_R=1.1 * H3, hcore_lo_R=H3)
    target = g[name + "/target"]
    for vtag, res in (("u", False), ("r", True)):
        for tag, beta, hcore in BCS_FIT_RUNS:
            L.use_hcore_as_emb_ham = hcore
            v = Hubbard.VcorLocal(res, True, n)
            v.update(np.zeros(v.length()))
            vfit, e0, e1 = bcs.FitVcorEmb(target, L, basis, v, mu, beta=beta, MaxIter=25)
            fit = bcs.FitVcorEmb.last_fit
            key = "%s/%s_%s" % (name, vtag, tag)
            for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
                assert abs(fit.errfunc(p) - e) < 1e-11, key
                assert np.abs(fit.gradfunc(p) - gr).max() < 1e-8 * max(1.0, np.abs(gr).max()), key
            pref, (r0, r1) = g[key + "/param"], g[key + "/err"]
            assert abs(e0 - r0) < 1e-11, key
            assert abs(e1 - r1) < 1e-6, (key, e1, r1)
            assert vfit is v and e1 <= e0
    L.use_hcore_as_emb_ham = False
    with pytest.raises(Exception):
        bcs.FitVcorEmb(target, L, basis, v, mu, imp_fit=True)
    v = Hubbard.VcorLocal(False, True, n)
    v.update(np.zeros(v.length()))
    vnew, err = bcs.FitVcorTwoStep(target, L, basis, v, mu, beta=np.inf, MaxIter1=25, MaxIter2=0)
    assert vnew is not v and np.abs(np.asarray(v.param)).max() == 0.0 and abs(err - g[name + "/u_t0/err"][1]) < 1e-6


@pytest.mark.parametrize("name", ["c611", "c441", "c222"])
def test_hfb_mean_field(ctx, golden, name):
    """mfd.HFB (routine/mfd.py:480-590) against the reference (golden G31): generalised density, particle number, energy, levels,
    k-space density and band edges at T = 0 and finite T, with and without the +-k symmetry, fixed / fitted half-filling level,
    Fock or hcore."""
    from libdmet_preview_amd.routine import mfd
    from tests.test_oracle_bcs import HFB_RUNS
    g, g7 = golden("G31_hfb.npz"), golden("G7_bcs.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    FR, v, mu = g7[name + "/Fock_R"], g7[name + "/vcor"], float(g7[name + "/mu"])
    n = FR.shape[-1]
    L = _lattice(mesh, n, [int(x) for x in g7[name + "/val"]])
    L.set_Ham_lo(fock_lo_R=FR, hcore_lo_R=0.7 * FR)
    L.H0 = 0.3
    vc = _Vcor(v)
    vc.islocal = lambda: True
    for tag, beta, kw in HFB_RUNS:
        GT, npart, E, res = mfd.HFB(L, vc, False, mu=mu, beta=beta, ires=True, **kw)
        key = "%s/%s" % (name, tag)
        assert np.abs(res["e"] - g[key + "/ew"]).max() < 1e-11
        assert np.abs(GT - g[key + "/GRhoT"]).max() < 1e-10 and np.abs(res["rho_k"] - g[key + "/rho_k"]).max() < 1e-10
        assert abs(npart - float(g[key + "/n"])) < 1e-10 and abs(E - float(g[key + "/E"])) < 1e-9
        assert np.abs(np.asarray([res["gap"], res["homo"], res["lumo"]]) - g[key + "/edges"]).max() < 1e-11
        for k in range(GT.shape[0]):                       # eigenpairs through their residual, never raw vectors
            ev, ew = res["coef"][k], res["e"][k]
            assert np.abs(ev.conj().T @ ev - np.eye(2 * n)).max() < 1e-10
        out3 = mfd.HFB(L, vc, False, mu=mu, beta=beta, **kw)
        assert len(out3) == 3 and np.array_equal(out3[0], GT)
    with pytest.raises(Exception):
        mfd.HFB(L, vc, True, mu=mu)


@pytest.mark.parametrize("name,n", [("c611", 2), ("c441", 4)])
def test_bcs_lattice_stage_fit(ctx, golden, name, n):
    """bcs.foldRho / FitVcorFull (routine/bcs.py:319-343, 532-562): the reference's objective at fixed parameters and its
    numerical-gradient fits (golden G31); the two-step wrapper with both stages."""
    from libdmet_preview_amd.routine import bcs
    from libdmet_preview_amd.dmet import Hubbard
    g, g7, g30 = golden("G31_hfb.npz"), golden("G7_bcs.npz"), golden("G30_bcs_fit.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    FR, mu, basis = g7[name + "/Fock_R"], float(g7[name + "/mu"]), g7[name + "/basis_proj"]
    L = _lattice(mesh, n, [int(x) for x in g7[name + "/val"]])
    L.set_Ham_lo(fock_lo_R=FR, hcore_lo_R=FR)
    assert np.abs(bcs.foldRho(g7[name + "/GRho"], L, basis) - g[name + "/foldRho"]).max() < 1e-11
    from libdmet_preview_amd.routine.bcs_helper import basisToCanonical
    assert np.abs(bcs.foldRho_k(g7[name + "/bdg_GRho_k"], basisToCanonical(basis).astype(complex)) - g[name + "/foldRho_k"]).max() < 1e-11
    target = g30[name + "/target"]
    for tag, beta in (("t0", np.inf), ("ft", 8.0)):
        key = "%s/full_%s" % (name, tag)
        v = Hubbard.VcorLocal(False, True, n)
        v.update(np.array(g[key + "/p0"]))
        vfit, e0, e1 = bcs.FitVcorFull(target, L, basis, v, mu, beta=beta, MaxIter=3)
        ef = bcs.FitVcorFull.last_errfunc
        for p, e in zip(g[key + "/probe"], g[key + "/probe_err"]):
            assert abs(ef(p) - e) < 1e-10
        r0, r1 = g[key + "/err"]
        assert abs(e0 - r0) < 1e-10 and abs(e1 - r1) < 1e-5 and e1 <= e0, (key, e0, e1, r0, r1)
    v = Hubbard.VcorLocal(False, True, n)
    v.update(np.zeros(v.length()))
    vnew, err = bcs.FitVcorTwoStep(target, L, basis, v, mu, beta=np.inf, MaxIter1=10, MaxIter2=1)
    assert vnew is not v and err <= g30[name + "/u_t0/err"][0]


def _nambu_levels(cd, cc):
    nb = cd.shape[-1]
    M = np.zeros((2 * nb, 2 * nb))
    M[:nb, :nb], M[nb:, nb:], M[:nb, nb:], M[nb:, :nb] = cd[0], -cd[1], cc[0], cc[0].T
    return np.linalg.eigvalsh(M)


@pytest.mark.parametrize("name,n", [("c611", 2), ("c441", 4)])
def test_bcs_driver_layer(ctx, golden, name, n):
    """dmet/HubbardBCS.py:9-112: HartreeFockBogoliubov with the chemical potential fitted to a filling (every mono_fit iterate is an
    HFB on the device), ConstructImpHam and apply_dmu, against the reference (golden G32).  Impurity Hamiltonians are compared
    through what does not depend on the gauge of the bath columns: the Nambu spectrum, H0, and the basis as a projector."""
    from libdmet_preview_amd.dmet import HubbardBCS as HB
    g, g7, g28 = golden("G32_bcs_driver.npz"), golden("G7_bcs.npz"), golden("G28_bcs_embham.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    FR, v = g7[name + "/Fock_R"], g7[name + "/vcor"]
    L = _lattice(mesh, n, [int(x) for x in g7[name + "/val"]])
    L.set_Ham_lo(fock_lo_R=FR, hcore_lo_R=FR)
    L.set_H2_local(g28[name + "/LatH2"])
    L.use_hcore_as_emb_ham = True
    vc = _Vcor(v)
    vc.islocal = lambda: True
    for tag, filling, beta, kw in (("fit_t0", 0.4, np.inf, dict()), ("fit_ft", 0.55, 10.0, dict(fix_mu=True)), ("nofit", None, np.inf, dict())):
        rho, mu, res = HB.HartreeFockBogoliubov(L, vc, filling, 0.2, beta=beta, full_return=True, **kw)
        key = "%s/%s" % (name, tag)
        assert abs(mu - float(g[key + "/mu"])) < 1e-9, (key, mu)
        assert np.abs(rho - g[key + "/GRho"]).max() < 1e-8 and abs(res["E"] - float(g[key + "/E"])) < 1e-8
        assert np.abs(res["e"] - g[key + "/ew"]).max() < 1e-8
        assert len(HB.HartreeFockBogoliubov(L, vc, None, 0.2)) == 2
    GRho, mu = g[name + "/fit_t0/GRho"], float(g[name + "/fit_t0/mu"])
    for tag, matching in (("match", True), ("nomatch", False)):
        ImpHam, (H1e, H0e), basis = HB.ConstructImpHam(L, GRho, vc, mu, matching=matching)
        key = "%s/imp_%s" % (name, tag)
        ref = g[key + "/basis"]
        assert basis.shape == ref.shape
        nb = basis.shape[-1]
        for s in range(2):                                  # impurity columns exact, bath columns as a subspace
            a, r = basis[s].reshape(-1, nb), ref[s].reshape(-1, nb)
            assert np.array_equal(a[:, :nb // 2], r[:, :nb // 2])
            assert np.linalg.norm(r[:, nb // 2:] - a[:, nb // 2:] @ (a[:, nb // 2:].T @ r[:, nb // 2:])) < 1e-8
        assert np.abs(_nambu_levels(ImpHam.H1["cd"], ImpHam.H1["cc"]) - _nambu_levels(g[key + "/cd"], g[key + "/cc"])).max() < 1e-8
        assert abs(ImpHam.H0 - float(g[key + "/H0"])) < 1e-8 and abs(H0e - float(g[key + "/eH0"])) < 1e-8
        if matching:
            ImpHam = HB.apply_dmu(L, ImpHam, basis, 0.13)
            assert np.abs(_nambu_levels(ImpHam.H1["cd"], ImpHam.H1["cc"]) - _nambu_levels(g[key + "/dmu_cd"], g[key + "/dmu_cc"])).max() < 1e-8
            assert abs(ImpHam.H0 - float(g[key + "/dmu_H0"])) < 1e-8


def test_bcs_kinetic_lattice_fit(ctx, golden):
    """bcs.FitVcorFullK (routine/bcs.py:564-619): cost and analytic gradient at fixed parameters, the constraint values and the fit
    of SciPy's quasi-Newton driver against the reference (golden G31); reached through FitVcorTwoStep(kinetic=True)."""
    from libdmet_preview_amd.routine import bcs
    from libdmet_preview_amd.dmet import Hubbard
    g, g7 = golden("G31_hfb.npz"), golden("G7_bcs.npz")
    name, n = "c611", 2
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    FR, mu, basis = g7[name + "/Fock_R"], float(g7[name + "/mu"]), g7[name + "/basis_proj"]
    L = _lattice(mesh, n, [0, 1])
    L.set_Ham_lo(fock_lo_R=FR, hcore_lo_R=FR)
    target = g[name + "/fullK/target"]
    v = Hubbard.VcorLocal(False, True, n)
    v.update(np.array(g[name + "/fullK/p0"]))
    vfit, c0, c1 = bcs.FitVcorFullK(target, L, basis, v, mu, 5)
    fitted = np.array(vfit.param)                           # (the cost closure updates the potential it is evaluated on)
    cost, grad = bcs.FitVcorFullK.last_cost
    for p, c, gr in zip(g[name + "/fullK/probe"], g[name + "/fullK/probe_cost"], g[name + "/fullK/probe_grad"]):
        assert abs(cost(p) - c) < 1e-9 * max(1.0, abs(c))
        assert np.abs(grad(p) - gr).max() < 1e-9
    r0, r1 = g[name + "/fullK/c"]
    assert abs(c0 - r0) < 1e-9 and abs(c1 - r1) < 1e-5
    assert np.abs(fitted - g[name + "/fullK/param"]).max() < 1e-4
    w = Hubbard.VcorLocal(False, True, n)
    w.update(np.array(g[name + "/fullK/p0"]))
    wnew, cend = bcs.FitVcorTwoStep(target, L, basis, w, mu, MaxIter1=0, MaxIter2=0, kinetic=True)
    assert wnew is not w and abs(cend - r1) < 1e-5
