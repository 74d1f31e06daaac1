"""
GPU parity tests (-m gpu) for the GSO ("spinless") twins, SURVEY.md section 8(f) rank 4: spinless.get_emb_basis and
get_emb_eri_gso through the C ABI against oracle/restate_gso.py and golden G12 (captured from the reference).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import restate as R
from oracle import restate_gso as G
from tests.test_oracle_gso import BATH, ERI, col_sign_dev, eri_inputs


@pytest.fixture(scope="module")
def ctx():
    from libdmet_preview_amd import _lib
    return _lib.get_ctx()


@pytest.mark.parametrize("name,n,val", BATH)
def test_gso_bath(ctx, golden, name, n, val):
    from libdmet_preview_amd.routine import spinless
    from libdmet_preview_amd.system.lattice import Lattice
    g7, g = golden("G7_bcs.npz"), golden("G12_gso.npz")
    mesh = tuple(int(x) for x in g7[name + "/mesh"])
    L = Lattice(n, mesh)
    L.val_idx = list(val)
    L.virt_idx = [i for i in range(n) if i > max(val)]
    L.core_idx = [i for i in range(n) if i < min(val)]
    GRho = g7[name + "/GRho"]
    nimp = 2 * (len(L.val_idx) + len(L.virt_idx))
    for key, vb in (("basis", True), ("basis_full", False)):
        b = spinless.get_emb_basis(L, GRho, valence_bath=vb)
        ref = g["bath/%s/%s" % (name, key)]
        assert b.shape == ref.shape
        assert np.array_equal(b[..., :nimp], ref[..., :nimp])
        assert col_sign_dev(b[..., nimp:], ref[..., nimp:]) < 1e-9
        a2, r2 = b.reshape(-1, b.shape[-1])[:, nimp:], ref.reshape(-1, ref.shape[-1])[:, nimp:]
        assert np.sqrt(2.0) * np.linalg.norm(r2 - a2 @ (a2.T @ r2)) < 1e-10        # projector, gauge free
    assert spinless.embBasis is spinless.get_emb_basis
    # the 'eig' and 'ph' flavours (routine/spinless.py:166-275, 351-423) against what the reference's own functions returned (G18)
    g18 = golden("G18_branches.npz")
    span = lambda a, r: np.abs(a.reshape(-1, a.shape[-1]) @ a.reshape(-1, a.shape[-1]).T
                               - r.reshape(-1, r.shape[-1]) @ r.reshape(-1, r.shape[-1]).T).max()
    for vb, tag in ((True, "val"), (False, "full")):
        be = spinless.get_emb_basis(L, GRho, kind="eig", valence_bath=vb)
        ref = g18["%s/gso_eig_%s" % (name, tag)]
        assert be.shape == ref.shape and np.array_equal(be[..., :nimp], ref[..., :nimp])
        assert span(be[..., nimp:], ref[..., nimp:]) < 1e-10
        bp = spinless.get_emb_basis(L, GRho, kind="ph", valence_bath=vb)
        refp = g18["%s/gso_ph_%s" % (name, tag)]
        assert bp.shape == refp.shape and span(bp, refp) < 1e-10
        Bm = bp.reshape(-1, bp.shape[-1])
        assert np.abs(Bm.T @ Bm - np.eye(Bm.shape[-1])).max() < 1e-10
    with pytest.raises(ValueError):
        spinless.get_emb_basis(L, GRho, kind="nope")
    # bath_opt (routine/spinless.py:44-54, 274-349) on the metallic matrices of G19: the rotated space against what the reference
    # returned (projector 1e-8: the root search stops at xtol = rtol = 1e-6 on both sides, on the same iterates), an integer electron
    # number, the keep_imp_identity variant, and the gapped matrix of G7 handed back unchanged
    g19 = golden("G19_bath_opt.npz")
    from oracle.restate import CellArith
    for tag in ("a", "b"):
        GT = g19["%s/%s/GRho" % (name, tag)]
        D = CellArith(mesh).expand(GT[None])[0]
        for vb, vtag in ((True, "val"), (False, "full")):
            key = "%s/%s/%s" % (name, tag, vtag)
            bo = spinless.get_emb_basis(L, GT, kind="svd", valence_bath=vb, bath_opt=True)
            ref = g19[key + "/basis_opt"]
            assert bo.shape == ref.shape and span(bo, ref) < 1e-8
            Bm = bo.reshape(-1, bo.shape[-1])
            ne = np.trace(Bm.T @ D @ Bm)
            assert abs(ne - round(ne)) < 1e-5 and np.abs(Bm.T @ Bm - np.eye(Bm.shape[-1])).max() < 1e-10
            bk = spinless.get_emb_basis_opt(L, GT, g19[key + "/basis_svd"], keep_imp_identity=True)
            refk = g19[key + "/basis_opt_keep"]
            assert span(bk, refk) < 1e-8 and np.array_equal(bk[..., :L.nimp], refk[..., :L.nimp])
    if min(val) == 0:
        b0 = spinless.get_emb_basis(L, GRho)
        assert np.array_equal(spinless.get_emb_basis(L, GRho, bath_opt=True), b0)


@pytest.mark.parametrize("name", ERI)
def test_gso_eri(ctx, golden, name):
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd.system import fourier
    from libdmet_preview_amd.system.lattice import _UnitCell
    g = golden("G12_gso.npz")
    mesh, ks, blocks, naux, nao, basis = eri_inputs(g, name)
    cell = _UnitCell(nao)
    kpts = cell.get_abs_kpts(fourier.make_kpts_scaled(mesh))
    nk = len(kpts)
    L = np.asarray([[blocks[(i, j)] for j in range(nk)] for i in range(nk)])
    mydf = et.GDFMemory(kpts, {(i, j): L[i, j] for i in range(nk) for j in range(nk)}, naux)
    for spin in (1, 2):
        st = "%s/s%d" % (name, spin)
        C = g[st + "/C_ao_lo"]
        for tr, key in ((True, "eri_tr"), (False, "eri_notr")):
            e = et.get_emb_eri_gso(cell, mydf, C_ao_lo=C, basis=basis, t_reversal_symm=tr)
            ref = g[st + "/" + key]
            assert e.shape == ref.shape
            assert np.abs(e - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())
        assert np.abs(et.get_emb_eri_gso(cell, mydf, C_ao_lo=C, basis=basis, symmetry=1) - g[st + "/eri_s1"]).max() < 1e-8
        assert np.abs(et.get_emb_eri_gso(cell, mydf, C_ao_lo=C, basis=basis, unit_eri=True) - g[st + "/eri_unit"]).max() < 1e-8
        # the same tensor resident in HBM (GDFResident): the GSO pipeline reads its groups in place, same bits
        res = et.make_df_resident(cell, mydf)
        assert np.array_equal(et.get_emb_eri_gso(cell, res, C_ao_lo=C, basis=basis), et.get_emb_eri_gso(cell, mydf, C_ao_lo=C, basis=basis))
        res.close()


@pytest.mark.parametrize("name", ["m311", "m221", "m231"])
def test_transform_gdf_to_lo_and_cderi_provider(ctx, golden, name, tmp_path):
    """The product's writer reproduces the datasets the reference writes (golden G13); an ERI computed from the
    container equals the ERI computed from the in-memory LO tensor."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd.system.lattice import _UnitCell
    from tests.test_oracle_cderi import inputs, golden_container
    g = golden("G13_cderi.npz")
    mesh, ks, kabs, blocks, naux, C = inputs(g, name)
    nk, nao, nlo = C.shape
    cell = _UnitCell(nao)
    kpts = cell.get_abs_kpts(ks)
    mydf = et.GDFMemory(kpts, {k: v for k, v in blocks.items()}, naux, cell=cell)
    for tr in (True, False):
        fn = str(tmp_path / ("lo_%d" % tr))
        prov = et.transform_gdf_to_lo(mydf, C, fname=fn, t_reversal_symm=tr)
        ref = golden_container(g, "%s/%s" % (name, "tr" if tr else "notr"))
        assert sorted(prov.feri.keys()) == sorted(ref.keys())
        for k in ref:
            assert prov.feri[k].shape == ref[k].shape and prov.feri[k].dtype.kind == ref[k].dtype.kind, k
            assert np.abs(prov.feri[k] - ref[k]).max() < 1e-11, k
        saved = np.load(fn + ".npz")
        assert sorted(saved.files) == sorted(ref.keys())
    # ERI through the container == ERI through memory blocks in the LO basis
    rng = np.random.default_rng(3)
    basis = rng.standard_normal((1, nk, nlo, 4))
    lo_blocks = {(i, j): prov.get_block(i, j) for i in range(nk) for j in range(nk)}
    cell_lo = _UnitCell(nlo)
    e_file = et.get_emb_eri(cell_lo, et.CderiProvider(dict(np.load(fn + ".npz")), kpts, nlo), basis=basis)
    e_mem = et.get_emb_eri(cell_lo, et.GDFMemory(kpts, lo_blocks, naux), basis=basis)
    e_ao = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis)
    assert np.abs(e_file - e_mem).max() == 0.0
    assert np.abs(e_file - e_ao).max() < 1e-9 * max(1.0, np.abs(e_ao).max())


@pytest.mark.parametrize("name,mesh,n,val", [("c611", (6, 1, 1), 2, [0, 1]), ("c441", (4, 4, 1), 4, [0, 1, 2, 3]), ("c222", (2, 2, 2), 5, [1, 2, 3])])
def test_scdm_localised_bcs_and_gso_baths(golden, name, mesh, n, val):
    """localize_bath='scdm' inside bcs.embBasis and spinless.get_emb_basis (svd and eig flavours) on the device against what the
    reference returned (golden G22: routine/bcs.py:84-88, routine/spinless.py:139-146, 248-255 over routine/localizer.py)."""
    from libdmet_preview_amd.routine import spinless, bcs
    from libdmet_preview_amd.system.lattice import Lattice
    from tests.test_oracle_golden import _match_columns
    g, g7 = golden("G22_scdm_bath.npz"), golden("G7_bcs.npz")
    L = Lattice(int(n), mesh)
    L.val_idx = list(val)
    L.virt_idx = [i for i in range(n) if i > max(val)]
    L.core_idx = [i for i in range(n) if i < min(val)]
    L.is_model = True
    GRho = g7[name + "/GRho"]
    b = bcs.embBasis(L, GRho, localize_bath="scdm")
    ref = g[name + "/bcs_scdm"]
    assert b.shape == ref.shape
    for s in range(2):
        assert _match_columns(ref[s].reshape(-1, ref.shape[-1]), b[s].reshape(-1, b.shape[-1]))[0] < 1e-8
    for kind in ("svd", "eig"):
        got = spinless.get_emb_basis(L, GRho, kind=kind, localize_bath="scdm")
        ref = g["%s/gso_%s_scdm" % (name, kind)]
        assert got.shape == ref.shape
        assert _match_columns(ref.reshape(-1, ref.shape[-1]), got.reshape(-1, got.shape[-1]))[0] < 1e-8
    with pytest.raises(NotImplementedError):
        spinless.get_emb_basis(L, GRho, localize_bath="pm")


# ---- round 6: the GSO embedding Hamiltonian (golden G27) -------------------------------------------------------------------

from tests.test_oracle_gso import GSO_HAM, gso_ham_inputs, gso_ham_runs  # noqa: E402


class _V3(object):
    def __init__(self, v):
        self.v = np.asarray(v)

    def islocal(self):
        return True

    is_local = islocal

    def get(self, i=0, kspace=True):
        return self.v


def _gso_lattice(g, name):
    from libdmet_preview_amd.system.lattice import Lattice
    mesh, basis, H2, H3, F3, S3, rk, v, mu = gso_ham_inputs(g, name)
    n = basis.shape[1] // 2
    val = [int(x) for x in g[name + "/val"]]
    L = Lattice(n, mesh)
    L.val_idx, L.virt_idx, L.core_idx = val, [i for i in range(n) if i > max(val)], [i for i in range(n) if i < min(val)]
    L.hcore_lo_k, L.fock_lo_k, L.fock_hf_lo_k, L.ovlp_lo_k = H3, F3, 0.9 * F3, S3
    L.rdm1_lo_k, L.H0 = rk, 0.75
    return L, mesh, basis, H2, F3, rk, v, mu


@pytest.mark.parametrize("name", GSO_HAM)
def test_gso_embedding_hamiltonian(ctx, golden, name):
    """spinless.get_emb_Ham (routine/spinless.py:431-725) in every bath regime with a given ERI, against the reference's values
    (golden G27) and the oracle; the folds, the ERI containers and the model branch on their own."""
    from libdmet_preview_amd.routine import spinless, spinless_helper as sh
    g = golden("G27_gso_embham.npz")
    L, mesh, basis, H2, F3, rk, v, mu = _gso_lattice(g, name)
    neo = basis.shape[-1]
    vc = _V3(v)
    for tag, kw in gso_ham_runs(g, name):
        kw = dict(kw)
        L.JK_imp = kw.pop("JK_imp", None)
        L.use_hcore_as_emb_ham = kw.pop("use_hcore_as_emb_ham", False)
        L.JK_core = "unset"
        if tag == "ib_add":
            kw["H0_add"] = 0.5
        Himp, none = spinless.get_emb_Ham(L, basis, vc, mu, H2_given=H2, **kw)
        assert none is None and Himp.norb == neo and Himp.restricted and not Himp.bogoliubov and Himp.H2["ccdd"] is H2
        assert Himp.H1["cd"].shape == (1, neo, neo)
        assert np.abs(Himp.H1["cd"] - g["%s/%s_H1" % (name, tag)]).max() < 1e-10, tag
        assert np.abs(Himp.ovlp - g["%s/%s_ovlp" % (name, tag)]).max() < 1e-12, tag
        assert abs(Himp.H0 - float(g["%s/%s_H0" % (name, tag)])) < 1e-14
        key = "%s/%s_JK_core" % (name, tag)
        assert (np.abs(L.JK_core - g[key]).max() < 1e-10) if key in g else L.JK_core is None
    L.JK_imp, L.use_hcore_as_emb_ham = None, False
    assert spinless.embHam is spinless.get_emb_Ham
    with pytest.raises(NotImplementedError):
        spinless.get_emb_Ham(L, basis, vc, mu, H2_given=H2, dft=True)
    # helpers
    bka, bkb = sh.separate_basis(L.R2k_basis(basis))
    bRa, bRb = sh.separate_basis(basis)
    assert np.abs(sh.transform_trans_inv_k(bka, bkb, F3) - g[name + "/ti_k3"]).max() < 1e-11
    assert np.abs(sh.transform_trans_inv_k(bka, bkb, F3[:2]) - g[name + "/ti_k2"]).max() < 1e-11
    for k3, k2, fn in (("loc3", "loc2", sh.transform_local), ("imp3", "imp2", sh.transform_imp)):
        assert np.abs(fn(bRa, bRb, v) - g[name + "/" + k3]).max() < 1e-12 and np.abs(fn(bRa, bRb, v[:2]) - g[name + "/" + k2]).max() < 1e-12
    assert np.abs(spinless.foldRho_k(rk, L.R2k_basis(basis)) - g[name + "/foldRho_k"]).max() < 1e-12
    unit = g[name + "/unit"]
    assert np.array_equal(sh.unit2emb(unit, neo), g[name + "/unit2emb"])
    n = basis.shape[1] // 2
    masks = sh.get_H2_mask(n, neo)
    assert np.array_equal(g[name + "/unit2emb"][masks[2]], unit[2]) and np.array_equal(g[name + "/unit2emb"][masks[3]], unit[2].T)
    if name + "/eri_local" in g:
        assert np.abs(sh.transform_eri_local(bRa, bRb, unit) - g[name + "/eri_local"]).max() < 1e-11
        assert np.abs(sh.transform_eri_local(bRa, bRb, unit) - G.transform_eri_local_gso(basis, unit)).max() < 1e-11
        L.set_H2_local(unit, H2_format="spin local")
        L.eri_symmetry = 4
        for tag, kw in (("model_ib", dict()), ("model_nib", dict(int_bath=False))):
            Himp, _ = spinless.get_emb_Ham(L, basis, vc, mu, **kw)
            assert np.abs(np.asarray(Himp.H2["ccdd"]) - g["%s/%s_H2" % (name, tag)]).max() < 1e-11, tag
            assert np.abs(Himp.H1["cd"] - g["%s/%s_H1" % (name, tag)]).max() < 1e-10, tag
            assert np.abs(L.JK_core - g["%s/%s_JK_core" % (name, tag)]).max() < 1e-10, tag
        L.set_H2_local(unit, H2_format="spin nearest")
        with pytest.raises(NotImplementedError):
            spinless.get_emb_Ham(L, basis, vc, mu)


# ---- round 6: the GSO vcor fit in the embedding space (golden G29) ---------------------------------------------------------

from tests.test_oracle_gso import GSO_FIT, GSO_FIT_RUNS  # noqa: E402


@pytest.mark.parametrize("name,n,val", GSO_FIT)
def test_gso_vcor_fit(ctx, golden, name, n, val):
    """spinless.get_dV_dparam / FitVcorEmb (routine/spinless.py:1090-1430) on the device against the reference's own closures and
    fits (golden G29): T = 0 and finite T, impurity block, diagonal, fixed mu, hcore as the embedding Hamiltonian."""
    from libdmet_preview_amd.routine import spinless
    from libdmet_preview_amd.dmet import Hubbard
    g, g27 = golden("G29_gso_fit.npz"), golden("G27_gso_embham.npz")
    L, mesh, basis, H2, F3, rk, vmat, mu = _gso_lattice(g27, name)
    v = Hubbard.VcorLocal(False, True, n)
    assert np.abs(spinless.get_dV_dparam(v, basis, None, L) - g[name + "/dV_compact"]).max() < 1e-13
    assert np.abs(spinless.get_dV_dparam(v, basis, None, L, compact=False) - g[name + "/dV_full"]).max() < 1e-13
    with pytest.raises(NotImplementedError):
        spinless.get_dV_dparam(v, basis, None, L, P_act=np.zeros(1))
    target = g[name + "/target"]
    for tag, beta, kw in GSO_FIT_RUNS:
        kw = dict(kw)
        L.use_hcore_as_emb_ham = kw.pop("hcore", False)
        v = Hubbard.VcorLocal(False, True, n)
        v.update(np.zeros(v.length()))
        vfit, e0, e1 = spinless.FitVcorEmb(target, L, basis, v, mu, beta=beta, MaxIter=30, **kw)
        fit = spinless.FitVcorEmb.last_fit
        key = "%s/%s" % (name, tag)
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-11, key
            assert np.abs(fit.gradfunc(p) - gr).max() < 1e-8 * max(1.0, np.abs(gr).max()), key
        pref, (r0, r1) = g[key + "/param"], g[key + "/err"]
        assert abs(e0 - r0) < 1e-11, key
        assert abs(e1 - r1) < 1e-6, (key, e1, r1)
        assert vfit is v and e1 <= e0
    L.use_hcore_as_emb_ham = False
    # the wrapper of the DMET loop: embedding stage on a COPY (spinless.py:2166-2231)
    v = Hubbard.VcorLocal(False, True, n)
    v.update(np.zeros(v.length()))
    vnew, err = spinless.FitVcorTwoStep(target, L, basis, v, mu, beta=np.inf, MaxIter1=30, MaxIter2=0)
    assert vnew is not v and np.abs(np.asarray(v.param)).max() == 0.0 and abs(err - g[name + "/t0/err"][1]) < 1e-6
    assert len(spinless.FitVcorTwoStep(target, L, basis, v, mu, MaxIter1=3, full_return=True)) == 4


# ---- round 6: generalised Hartree-Fock lattice mean field (golden G33) -----------------------------------------------------

@pytest.mark.parametrize("name", GSO_HAM)
def test_ghf_mean_field(ctx, golden, name):
    """mfd.GHF (routine/mfd.py:735-858) against the reference (golden G33): T = 0 and finite T, with and without the +-k symmetry,
    fixed level, hcore, another filling, frozen levels (`nfrac`), and the particle-hole entry on a plain two-spin Hamiltonian."""
    from libdmet_preview_amd.routine import mfd
    from libdmet_preview_amd.system.lattice import Lattice
    from tests.test_oracle_gso import GHF_RUNS
    g, g27, g7 = golden("G33_ghf.npz"), golden("G27_gso_embham.npz"), golden("G7_bcs.npz")
    mesh = tuple(int(x) for x in g27[name + "/mesh"])
    H3, F3, v = g27[name + "/H3_k"], g27[name + "/F3_k"], g27[name + "/vcor"]
    n = v.shape[-1]
    L = Lattice(n, mesh)
    L.hcore_lo_k, L.fock_lo_k, L.H0 = H3, F3, 0.3
    vc = _V3(v)
    for tag, beta, kw in GHF_RUNS:
        kw = dict(kw)
        filling = kw.pop("filling", 0.5)
        GT, npart, E, res = mfd.GHF(L, vc, False, filling=filling, mu=0.37, beta=beta, ires=True, **kw)
        key = "%s/%s" % (name, tag)
        assert np.abs(res["e"] - g[key + "/ew"]).max() < 1e-11 and np.abs(res["mo_occ"] - g[key + "/occ"]).max() < 1e-9
        assert np.abs(GT - g[key + "/GRhoT"]).max() < 1e-10 and np.abs(res["rho_k"] - g[key + "/rho_k"]).max() < 1e-10
        assert abs(npart - float(g[key + "/n"])) < 1e-10 and abs(E - float(g[key + "/E"])) < 1e-9
        edges = np.asarray([res["gap"], res["homo"], res["lumo"], res["mu_quasi"]])
        assert np.abs(edges - g[key + "/edges"][:4]).max() < 1e-9
        for k in range(GT.shape[0]):
            ev = res["coef"][k]
            assert np.abs(ev.conj().T @ ev - np.eye(2 * n)).max() < 1e-10
        assert len(mfd.GHF(L, vc, False, filling=filling, mu=0.37, beta=beta, **kw)) == 3
    FR = g7[name + "/Fock_R"]
    L.hcore_lo_k, L.fock_lo_k = R.R2k(0.7 * FR, mesh), R.R2k(FR, mesh)
    GT, npart, E, res = mfd.GHF(L, vc, False, mu=0.37, beta=np.inf, ires=True, ph_trans=True)
    assert np.abs(GT - g[name + "/ph/GRhoT"]).max() < 1e-10 and abs(E - float(g[name + "/ph/E"])) < 1e-9
    assert np.abs(res["e"] - g[name + "/ph/ew"]).max() < 1e-11 and abs(npart - float(g[name + "/ph/n"])) < 1e-10
    for bad in (dict(restricted=True), dict(restricted=False, scf=True)):
        with pytest.raises(NotImplementedError):
            mfd.GHF(L, vc, bad.pop("restricted"), mu=0.37, **bad)


# ---- round 6: the GSO driver layer (golden G34) ---------------------------------------------------------------------------

@pytest.mark.parametrize("name", GSO_HAM)
def test_gso_driver_layer(ctx, golden, name):
    """dmet/HubbardGSO.py:16-134: GHartreeFock with the chemical potential fitted to a filling (every iterate a GHF on the device),
    ConstructImpHam and the three forms of apply_dmu, against the reference (golden G34).  The impurity Hamiltonian is compared
    on the reference's own basis entry by entry (the bath columns have a gauge), the basis itself as a subspace."""
    from libdmet_preview_amd.dmet import HubbardGSO as HG
    from libdmet_preview_amd.routine import spinless
    g, g27 = golden("G34_gso_driver.npz"), golden("G27_gso_embham.npz")
    L, mesh, basis27, H2_27, F3, rk, v, mu27 = _gso_lattice(g27, name)
    vc = _V3(v)
    for tag, filling, beta in (("fit_t0", 0.45, np.inf), ("fit_ft", 0.55, 10.0), ("nofit", None, np.inf)):
        rho, mu, res = HG.GHartreeFock(L, vc, filling, 0.2, beta=beta, full_return=True)
        key = "%s/%s" % (name, tag)
        assert abs(mu - float(g[key + "/mu"])) < 1e-8, (key, mu)
        assert np.abs(rho - g[key + "/GRho"]).max() < 1e-7 and abs(res["E"] - float(g[key + "/E"])) < 1e-7
        assert np.abs(res["e"] - g[key + "/ew"]).max() < 1e-7
    assert len(HG.GHartreeFock(L, vc, None, 0.2)) == 2
    GRho, H2 = g[name + "/nofit/GRho"], g[name + "/imp/H2"]
    L.rdm1_lo_k = R.R2k(GRho, mesh)
    ImpHam, none, basis = HG.ConstructImpHam(L, GRho, vc, 0.2, H2_given=H2)
    ref = g[name + "/imp/basis"]
    assert none is None and basis.shape == ref.shape and abs(ImpHam.H0 - float(g[name + "/imp/H0"])) < 1e-12
    a, r = basis.reshape(-1, basis.shape[-1]), ref.reshape(-1, ref.shape[-1])       # same embedding space (a GIVEN ERI is not gauge covariant,
    assert np.linalg.norm(r - a @ (a.T @ r)) < 1e-8                                  # so the Hamiltonian is compared on the reference's basis)
    # on the reference's basis: the Hamiltonian and every shift entry by entry
    ImpHam, _ = spinless.embHam(L, ref, vc, 0.2, H2_given=H2)
    assert np.abs(ImpHam.H1["cd"] - g[name + "/imp/H1"]).max() < 1e-9
    ImpHam = HG.apply_dmu(L, ImpHam, ref, 0.11)
    assert np.abs(ImpHam.H1["cd"] - g[name + "/imp/dmu_H1"]).max() < 1e-9
    ImpHam = HG.apply_dmu(L, ImpHam, ref, 0.07, fit_ghf=True)
    assert np.abs(ImpHam.H1["cd"] - g[name + "/imp/dmu_ghf_H1"]).max() < 1e-9
    ImpHam = HG.apply_dmu(L, ImpHam, ref, -0.05, dmu_idx=[0])
    assert np.abs(ImpHam.H1["cd"] - g[name + "/imp/dmu_idx_H1"]).max() < 1e-9


# ---- round 6: the lattice stage of the GSO fit (golden G35) ----------------------------------------------------------------

@pytest.mark.parametrize("name,n,val", GSO_FIT)
def test_gso_lattice_stage_fit(ctx, golden, name, n, val):
    """spinless.get_dV_dparam_full / FitVcorFull (routine/spinless.py:1431-1769) on the device against the reference's closures and
    fits (golden G35): impurity block, diagonal, pairing blocks only, fixed quasiparticle level, the numerical-gradient T = 0 run;
    both stages through FitVcorTwoStep."""
    from libdmet_preview_amd.routine import spinless
    from libdmet_preview_amd.dmet import Hubbard
    from tests.test_oracle_gso import GSO_FULL_RUNS
    g, g27 = golden("G35_gso_full_fit.npz"), golden("G27_gso_embham.npz")
    L, mesh, basis, H2, F3, rk, vmat, mu = _gso_lattice(g27, name)
    target = g[name + "/target"]
    assert np.array_equal(spinless.get_dV_dparam_full(Hubbard.VcorLocal(False, True, n), L), g[name + "/dV_full"])
    for tag, beta, kw, iters in GSO_FULL_RUNS:
        key = "%s/%s" % (name, tag)
        v = Hubbard.VcorLocal(False, True, n)
        v.update(np.array(g[key + "/p0"]))
        vfit, e0, e1 = spinless.FitVcorFull(target, L, basis, v, mu, beta, None, MaxIter=iters, **kw)
        fit = spinless.FitVcorFull.last_fit
        for i, p in enumerate(g[key + "/probe"]):
            assert abs(fit.errfunc(p) - g[key + "/probe_err"][i]) < 1e-11, key
            if key + "/probe_grad" in g:
                gr = g[key + "/probe_grad"][i]
                assert np.abs(fit.gradfunc(p) - gr).max() < 1e-8 * max(1.0, np.abs(gr).max()), key
        r0, r1 = g[key + "/err"]
        assert abs(e0 - r0) < 1e-11 and abs(e1 - r1) < 1e-5 and e1 <= e0, (key, e0, e1, r0, r1)
    with pytest.raises(NotImplementedError):
        spinless.FitVcorFull(target, L, basis, v, mu, np.inf, None, MaxIter=2, imp_fit=True)
    v = Hubbard.VcorLocal(False, True, n)
    v.update(np.zeros(v.length()))
    emb_target = g29_target = golden("G29_gso_fit.npz")[name + "/target"]
    vnew, err = spinless.FitVcorTwoStep(emb_target, L, basis, v, mu, beta=12.0, MaxIter1=5, MaxIter2=0)
    assert vnew is not v
    vnew2, err2 = spinless.FitVcorTwoStep(target, L, basis, v, mu, beta=12.0, MaxIter1=0, MaxIter2=3, imp_fit=True)
    assert vnew2 is not v and err2 <= g[name + "/ft_imp/err"][0] + 1.0
    vnew3, err3 = spinless.FitVcorTwoStep(target, L, basis, v, mu, beta=12.0, MaxIter1=0, MaxIter2=2, filling=0.5, imp_fit=True)
    assert vnew3 is not v and np.isfinite(err3)


# ---- round 6: the lattice stage with the chemical potential re-fitted inside (golden G37) -----------------------------------

@pytest.mark.parametrize("name,n,val", GSO_FIT)
def test_gso_lattice_stage_fit_with_mu(ctx, golden, name, n, val):
    """spinless.FitVcorFull_mu (routine/spinless.py:1771-2164): fits, and objective / gradient at fixed parameters in the order the
    golden run took them (the inner chemical-potential search starts from the previous gradient evaluation's solution and stops at
    1e-6 in the electron number, hence the tolerances)."""
    from libdmet_preview_amd.routine import spinless
    from libdmet_preview_amd.dmet import Hubbard
    g, g27, g35 = golden("G37_gso_full_fit_mu.npz"), golden("G27_gso_embham.npz"), golden("G35_gso_full_fit.npz")
    L, mesh, basis, H2, F3, rk, vmat, mu = _gso_lattice(g27, name)
    target = g35[name + "/target"]
    for tag, beta, filling, kw in (("ft_imp", 12.0, 0.5, dict(imp_fit=True)), ("ft_det", 12.0, 0.45, dict(det=True)),
                                   ("ft_bogo", 12.0, 0.55, dict(imp_fit=True, bogo_only=True))):
        key = "%s/%s" % (name, tag)
        v = Hubbard.VcorLocal(False, True, n)
        v.update(np.array(g[key + "/p0"]))
        vfit, e0, e1 = spinless.FitVcorFull_mu(target, L, basis, v, mu, beta, filling, MaxIter=6, **kw)
        errfunc, gradfunc, mu_state = spinless.FitVcorFull_mu.last_fit
        r0, r1 = g[key + "/err"]
        assert abs(e0 - r0) < 1e-5 and abs(e1 - r1) < 1e-4 and e1 <= e0 + 1e-9, (key, e0, e1, r0, r1)
        for p, e in zip(g[key + "/probe"], g[key + "/probe_err"]):
            assert abs(errfunc(p) - e) < 1e-5, key
        for p, gr in zip(g[key + "/probe"], g[key + "/probe_grad"]):
            assert np.abs(gradfunc(p) - gr).max() < 1e-4 * max(1.0, np.abs(gr).max()), key
    for bad in (dict(use_cvx_frac=True, imp_fit=True), dict(), dict(imp_fit=True, scf=True)):
        with pytest.raises(NotImplementedError):
            spinless.FitVcorFull_mu(target, L, basis, v, mu, 12.0, 0.5, MaxIter=1, **bad)
    with pytest.raises(NotImplementedError):
        spinless.FitVcorFull_mu(target, L, basis, v, mu, np.inf, 0.5, MaxIter=1, imp_fit=True)
