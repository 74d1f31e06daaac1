"""
GPU parity tests (-m gpu) for SURVEY.md section 8(f) rank 1: the embedding one-body Hamiltonian and the
ERI x density contraction (dmk_jk_s4) that complete get_emb_Ham.  HIP path through the C ABI against
oracle/restate_ham.py and the golden fixture G8 captured from the reference.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import restate as R
from oracle import restate_ham as H
from tests.test_oracle_ham import AB, MODELS, RUNS, ab_inputs


@pytest.fixture(scope="module")
def ctx():
    from libdmet_preview_amd import _lib
    return _lib.get_ctx()


class _Vcor(object):
    def __init__(self, v):
        self.value = v

    def islocal(self):
        return True

    def get(self, i=0, kspace=True):
        return self.value if (kspace or i == 0) else np.zeros_like(self.value)


def _lattice(mesh, nlo, val):
    from libdmet_preview_amd.system.lattice import Lattice
    L = Lattice(int(nlo), mesh)
    L.val_idx = list(val)
    L.virt_idx = [i for i in range(nlo) if i > max(val)]
    L.core_idx = [i for i in range(nlo) if i < min(val)]
    return L


@pytest.mark.parametrize("n", [1, 2, 5, 12, 63, 64, 65, 136, 200])
def test_jk_s4_kernel(ctx, n):
    """General (non symmetric) density and an ERI block without pair-exchange symmetry: pins the index convention."""
    from libdmet_preview_amd.solver.scf import jk_dev
    rng = np.random.default_rng(n)
    npair = n * (n + 1) // 2
    E = rng.standard_normal((npair, npair))
    d1, d2, d3 = (rng.standard_normal((n, n)) for _ in range(3))
    dE = ctx.to_device(E)
    vj_row, vj_col, vk = jk_dev(ctx, n, dE, ctx.to_device(d1), ctx.to_device(d2), ctx.to_device(d3))
    ia, ib = np.tril_indices(n)
    x1 = np.where(ia == ib, d1[ia, ia], d1[ia, ib] + d1[ib, ia])
    x2 = np.where(ia == ib, d2[ia, ia], d2[ia, ib] + d2[ib, ia])
    def unpack(y):
        m = np.zeros((n, n))
        m[ia, ib] = y
        m[ib, ia] = y
        return m
    scale = max(1.0, np.sqrt(npair))
    assert np.abs(vj_row.get() - unpack(E @ x1)).max() < 1e-12 * scale * 10
    assert np.abs(vj_col.get() - unpack(E.T @ x2)).max() < 1e-12 * scale * 10
    if n <= 64:
        e1 = R.restore(1, E, n)
        kref = np.einsum('ijkl,il->jk', e1, d3)
        assert np.abs(vk.get() - kref).max() < 1e-12 * scale * 10
    else:
        # K[j,k] = sum_il (ij|kl) d[i,l] via the unpacked symmetric rows, in blocks to bound memory
        kref = np.zeros((n, n))
        for r in range(npair):
            i, j = ia[r], ib[r]
            M = unpack(E[r])
            kref[j] += M @ d3[i]
            if i != j:
                kref[i] += M @ d3[j]
        assert np.abs(vk.get() - kref).max() < 1e-12 * scale * 10
    # only-J and only-K calls leave the other outputs alone and agree with the fused call
    a, b, c = jk_dev(ctx, n, dE, ctx.to_device(d1), None, None)
    assert b is None and c is None and np.array_equal(a.get(), vj_row.get())
    a, b, c = jk_dev(ctx, n, dE, None, None, ctx.to_device(d3))
    assert a is None and b is None and np.array_equal(c.get(), vk.get())


def test_jk_s4_is_deterministic_and_rejects_bad_input(ctx):
    from libdmet_preview_amd._lib import lib, DmkError
    from libdmet_preview_amd.solver.scf import jk_dev
    rng = np.random.default_rng(0)
    n = 40
    npair = n * (n + 1) // 2
    dE, dd = ctx.to_device(rng.standard_normal((npair, npair))), ctx.to_device(rng.standard_normal((n, n)))
    r1 = [x.get() for x in jk_dev(ctx, n, dE, dd, dd, dd)]
    r2 = [x.get() for x in jk_dev(ctx, n, dE, dd, dd, dd)]
    assert all(np.array_equal(a, b) for a, b in zip(r1, r2))
    with pytest.raises(DmkError):
        ctx.check(lib.dmk_jk_s4(ctx.h, n, dE.ptr, npair - 1, dd.ptr, None, None, dd.ptr, None, None))
    with pytest.raises(DmkError):
        ctx.check(lib.dmk_jk_s4(ctx.h, n, dE.ptr, npair, dd.ptr, None, None, None, None, None))
    with pytest.raises(DmkError):
        ctx.check(lib.dmk_eri_to_s4(ctx.h, n, 4, dE.ptr, dE.ptr))


def test_jk_full_size_density_fitted(ctx):
    """BASELINE size (nemb = 256, one 8.7 GB block): E = X^T X built on the device, checked through the factor X."""
    from libdmet_preview_amd._lib import lib
    from libdmet_preview_amd.solver.scf import jk_dev
    n, naux = 256, 6
    npair = n * (n + 1) // 2
    rng = np.random.default_rng(11)
    X = rng.standard_normal((naux, npair)) / np.sqrt(naux)
    dX = ctx.to_device(X)
    dE = ctx.zeros((npair, npair), np.float64)
    ctx.check(lib.dmk_dgemm_tn_acc(ctx.h, npair, naux, 1.0, dX.ptr, dX.ptr, npair, dE.ptr, npair))
    dm = rng.standard_normal((n, n))
    dm = dm + dm.T
    d = ctx.to_device(dm)
    vj, vjc, vk = jk_dev(ctx, n, dE, d, d, d)
    ia, ib = np.tril_indices(n)
    xt = np.where(ia == ib, dm[ia, ia], 2.0 * dm[ia, ib])
    jp = X.T @ (X @ xt)
    jref = np.zeros((n, n))
    jref[ia, ib] = jp
    jref[ib, ia] = jp
    kref = np.zeros((n, n))
    for L in range(naux):
        XL = np.zeros((n, n))
        XL[ia, ib] = X[L]
        XL[ib, ia] = X[L]
        kref += XL.T @ dm @ XL
    sc = np.abs(jref).max()
    assert np.abs(vj.get() - jref).max() < 1e-11 * sc
    assert np.abs(vjc.get() - jref).max() < 1e-11 * sc         # E symmetric: both directions agree
    assert np.abs(vk.get() - kref).max() < 1e-11 * np.abs(kref).max()
    assert np.abs(vk.get() - vk.get().T).max() < 1e-11 * np.abs(kref).max()


@pytest.mark.parametrize("name", AB)
def test_get_jk_and_veff(ctx, golden, name):
    from libdmet_preview_amd.solver import scf
    from libdmet_preview_amd.routine import slater
    g = golden("G8_embham.npz")
    H2, dm = g[name + "/H2"], g[name + "/rdm1_emb"]
    nb = dm.shape[-1]
    for tag, eri in [("s4", H2), ("s1", np.asarray([R.restore(1, h, nb) for h in H2])), ("s8", R.restore(8, H2[0], nb)),
                     ("res", H2[:1])]:
        vj, vk = scf._get_jk(dm, eri)
        ref_j, ref_k = g["%s/jk_%s_vj" % (name, tag)], g["%s/jk_%s_vk" % (name, tag)]
        assert vj.shape == ref_j.shape and vk.shape == ref_k.shape, tag
        assert np.abs(vj - ref_j).max() < 1e-12, tag
        assert np.abs(vk - ref_k).max() < 1e-12, tag
    for hyb in (1.0, 0.0, 0.4):
        assert np.abs(slater.get_veff(dm, H2, hyb=hyb) - g["%s/veff_hyb%.1f" % (name, hyb)]).max() < 1e-12
    assert np.abs(slater.get_veff(dm[0], H2[:1]) - g[name + "/veff_dm2d"]).max() < 1e-12
    vj, vk = scf._get_jk(dm, H2, with_k=False)
    assert vk is None and np.abs(vj - g[name + "/jk_s4_vj"]).max() < 1e-12
    with pytest.raises(ValueError):
        scf._get_jk(dm, np.zeros((3, 7)))


@pytest.mark.parametrize("name", AB)
def test_get_emb_Ham(ctx, golden, name):
    from libdmet_preview_amd.routine import slater
    g = golden("G8_embham.npz")
    mesh, FR, Fk, Hk, Sk, v, rdm1_k, basis, H2 = ab_inputs(g, name)
    spin, nlo = basis.shape[0], FR.shape[-1]
    L = _lattice(mesh, nlo, [int(x) for x in g[name + "/val"]])
    sq = (lambda x: x[0]) if spin == 1 else (lambda x: x)
    L.fock_lo_k, L.hcore_lo_k, L.vhf_lo_k = sq(Fk), sq(Hk), sq(Fk - Hk)
    L.ovlp_lo_k, L.rdm1_lo_k, L.H0 = Sk, rdm1_k, 1.25
    vc = _Vcor(v)
    for tag, kw in RUNS:
        kw = dict(kw)
        L.JK_imp = g[name + "/" + kw.pop("JK_imp")] if "JK_imp" in kw else None
        L.use_hcore_as_emb_ham = kw.pop("use_hcore_as_emb_ham", False)
        L.JK_core = "unset"
        Himp, none = slater.get_emb_Ham(L, basis, vc, H2_given=H2, **kw)
        assert none is None and Himp.norb == basis.shape[-1] and Himp.restricted == (spin == 1)
        assert not Himp.bogoliubov and Himp.H0 == 1.25 and Himp.H2["ccdd"] is H2
        assert np.abs(Himp.H1["cd"] - g["%s/%s_H1" % (name, tag)]).max() < 1e-10, tag
        assert np.abs(Himp.ovlp - g["%s/%s_ovlp" % (name, tag)]).max() < 1e-12, tag
        key = "%s/%s_JK_core" % (name, tag)
        if key in g:
            assert np.abs(L.JK_core - g[key]).max() < 1e-10, tag
        else:
            assert L.JK_core is None
    assert slater.embHam is slater.get_emb_Ham
    with pytest.raises(NotImplementedError):
        slater.get_emb_Ham(L, basis, vc, H2_given=H2, dft=True)


@pytest.mark.parametrize("name", AB)
def test_one_body_folds(ctx, golden, name):
    from libdmet_preview_amd.routine import slater, slater_helper as sh
    g = golden("G8_embham.npz")
    mesh, FR, Fk, Hk, Sk, v, rdm1_k, basis, H2 = ab_inputs(g, name)
    spin, nlo = basis.shape[0], FR.shape[-1]
    L = _lattice(mesh, nlo, [int(x) for x in g[name + "/val"]])
    rng = np.random.default_rng(3)
    for s in range(spin):
        assert np.abs(sh.transform_trans_inv(basis[s], L, FR[s]) - g["%s/ti_sym_%d" % (name, s)]).max() < 1e-11
        assert np.abs(sh.transform_trans_inv(basis[s], L, FR[s], symmetric=False) - g["%s/ti_full_%d" % (name, s)]).max() < 1e-11
        assert np.abs(sh.transform_local(basis[s], L, v[s]) - g["%s/tloc_%d" % (name, s)]).max() < 1e-12
        assert np.abs(sh.transform_imp(basis[s], L, v[s]) - g["%s/timp_%d" % (name, s)]).max() < 1e-12
        assert np.abs(sh.transform_imp_env(basis[s], L, FR[s]) - g["%s/tie_%d" % (name, s)]).max() < 1e-12
        # non-Hermitian stripe: the reference's cell-ordering dependent symmetric=True form
        G = rng.standard_normal(FR[s].shape)
        assert np.abs(sh.transform_trans_inv(basis[s], L, G) - H.transform_trans_inv(basis[s], mesh, G)).max() < 1e-11
        assert np.abs(sh.transform_trans_inv(basis[s], L, G, symmetric=False)
                      - H.transform_trans_inv(basis[s], mesh, G, False)).max() < 1e-11
    basis_k = L.R2k_basis(basis)
    assert np.abs(slater.transform_h1(Hk if spin == 2 else Hk[0], basis_k) - g[name + "/h1_emb"]).max() < 1e-12
    assert np.abs(slater.foldRho_k(rdm1_k, basis_k) - g[name + "/rdm1_emb"]).max() < 1e-12
    rho_R = np.asarray([R.k2R(rdm1_k[s] if rdm1_k.ndim == 4 else rdm1_k, mesh) for s in range(spin)])
    assert np.abs(slater.foldRho(rho_R, L, basis) - g[name + "/rdm1_emb"]).max() < 1e-11


@pytest.mark.parametrize("name", MODELS)
def test_model_emb_Ham(ctx, golden, name):
    """C1 / C2 Hubbard lattices end to end: cell-local ERI transform, s1 J/K, interacting and non-interacting bath."""
    from libdmet_preview_amd.routine import slater, slater_helper as sh
    g = golden("G8_embham.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    H1R, v, basis, LatH2 = g[name + "/H1_R"], g[name + "/vcor"], g[name + "/basis"], g[name + "/LatH2"]
    n = H1R.shape[-1]
    L = _lattice(mesh, n, list(range(n)))
    L.set_Ham_lo(fock_lo_R=H1R, hcore_lo_R=H1R)
    L.set_H2_local(LatH2)
    L.rdm1_lo_k = g[name + "/rdm1_lo_k"]
    vc = _Vcor(v)
    Himp, _ = slater.get_emb_Ham(L, basis, vc)
    assert np.abs(Himp.H2["ccdd"] - g[name + "/H2"]).max() < 1e-12
    assert np.abs(Himp.H1["cd"] - g[name + "/H1"]).max() < 1e-10
    assert np.abs(L.JK_core - g[name + "/JK_core"]).max() < 1e-10
    Hn, _ = slater.get_emb_Ham(L, basis, vc, int_bath=False)
    assert np.array_equal(Hn.H2["ccdd"], g[name + "/nib_H2"])
    assert np.abs(Hn.H1["cd"] - g[name + "/nib_H1"]).max() < 1e-10
    b0 = basis[0, 1]
    rng = np.random.default_rng(1)
    vv = rng.standard_normal((n,) * 4)
    assert np.abs(sh.transform_4idx(vv, b0, b0, b0, b0) - H.transform_4idx(vv, b0, b0, b0, b0)).max() < 1e-12


@pytest.mark.parametrize("spin", [1, 2])
def test_pipeline_emb_ham_stage(ctx, spin):
    """Device-resident one-body stage of the pipeline against the oracle on the pipeline's own basis / ERI / density."""
    from libdmet_preview_amd import pipeline
    mesh, nlo, naux, nval = (3, 2, 1), 8, 6, 3
    sysm = pipeline.SyntheticSystem(ctx, mesh, nlo, naux, nval, spin, seed=21 + spin, name="t")
    out = pipeline.iteration(ctx, sysm)
    nemb = out["nemb"]
    basis = out["basis"].get().reshape(spin, sysm.nk, nlo, nemb)
    rhoR = out["rho_R"].get().reshape(spin, sysm.nk, nlo, nlo)
    eri = out["eri"].get()
    Fk = R.R2k(sysm.Fock_R, mesh)
    H2 = eri[[0, 2, 1]] if spin == 2 else eri
    rdm1_k = R.R2k(rhoR, mesh) * (2.0 if spin == 1 else 1.0)
    Sk = np.asarray([np.eye(nlo)] * sysm.nk)
    H1, _, JKc = H.embHam1e(mesh, basis, H2, 0.5 * Fk, Fk, Sk, rdm1_k)
    ham = out["emb_ham"]
    sc = max(1.0, np.abs(H1).max())
    assert np.abs(ham["H1"] - H1).max() < 1e-10 * sc
    assert np.abs(ham["JK_core"] - JKc).max() < 1e-10 * sc
    assert "emb_jk" in out["timers"] and "emb_h1" in out["timers"]


@pytest.mark.parametrize("name", ["rhf", "uhf"])
def test_set_Ham_and_update_Ham(ctx, golden, name):
    """Lattice.set_Ham / transform_obj_to_lo / update_Ham (AO -> LO in front of the path) against golden G11."""
    from libdmet_preview_amd.system.lattice import Lattice
    g = golden("G11_setham.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    C, S, hcore, vj, vk, rdm1, vhf = (g["%s/in_%s" % (name, k)] for k in ("C", "S", "hcore", "vj", "vk", "rdm1", "vhf"))
    L = Lattice(int(C.shape[-1]), mesh)
    L.set_Ham(None, None, C, eri_symmetry=4, ovlp=S, hcore=hcore, rdm1=rdm1, vj=vj, vk=vk, H0=0.5)    # vhf from vj, vk
    assert L.H0 == 0.5 and L.has_Ham and L.restricted == (C.ndim == 3)
    for k in ["hcore", "ovlp", "fock", "fock_hf", "veff", "vhf", "rdm1"]:
        assert np.abs(getattr(L, k + "_lo_k") - g["%s/%s_lo_k" % (name, k)]).max() < 1e-11, k
        assert np.abs(getattr(L, k + "_lo_R") - g["%s/%s_lo_R" % (name, k)].real).max() < 1e-11, k
    L.update_Ham(g[name + "/upd_rdm1_R"], vhf=g[name + "/upd_vhf"])
    for k in ("rdm1_ao_k", "fock_lo_k", "rdm1_lo_k", "fock_lo_R", "vhf_lo_R"):
        ref = g["%s/upd_%s" % (name, k)]
        ref = ref.real if k.endswith("_R") else ref
        assert np.abs(getattr(L, k) - ref).max() < 1e-11, k
    with pytest.raises(ValueError):
        Lattice(int(C.shape[-1]), mesh).set_Ham(None, None, C, ovlp=S, hcore=hcore)       # no rdm1, no kmf
    with pytest.raises(NotImplementedError):
        L.set_Ham(None, None, C, ovlp=S, hcore=hcore, rdm1=rdm1, vhf=vhf, vxc=vhf)


@pytest.mark.parametrize("name", ["n6", "n10"])
def test_veff_ghf(ctx, golden, name):
    """slater.get_veff(ghf=True) and scf._get_veff_ghf (slater.py:489-506, solver/scf.py:732-740) on dmk_jk_s4 against the
    reference's values (golden G25), from 1-, 4- and 8-fold ERI storage."""
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.solver import scf
    from libdmet_preview_amd.basis_transform.eri_transform import eri_restore
    from tests.test_oracle_ham import GHF_VEFF
    g = golden("G25_veff_ghf.npz")
    dm, e4 = g[name + "/dm"], g[name + "/eri_s4"]
    nso = dm.shape[-1]
    stores = {"s4": e4, "s1": eri_restore(e4[None], 1, nso)[0], "s8": eri_restore(e4[None], 8, nso)[0]}
    for fmt, e in stores.items():
        for tag, kw in GHF_VEFF:
            got = slater.get_veff(dm, e, ghf=True, **kw)
            assert got.shape == (nso, nso) and np.abs(got - g["%s/%s/%s" % (name, fmt, tag)]).max() < 1e-12, (fmt, tag)
    assert np.abs(scf._get_veff_ghf(dm, e4) - g[name + "/veff_ghf"]).max() < 1e-12
    with pytest.raises(AssertionError):
        slater.get_veff(dm[None], e4, ghf=True)


@pytest.mark.parametrize("name", ["C1", "C1u"])
def test_model_eri_formats_bare_bath(ctx, golden, name):
    """The model ERI formats other than 'local' with a non-interacting bath (slater.py:407-426): the impurity block of a
    neighbour list, of a full cell-resolved tensor, of per-spin blocks, zero-padded (dmk_pad_block_f64); an interacting bath
    raises like the reference.  Golden G26."""
    from libdmet_preview_amd.routine import slater
    g = golden("G26_embham_corners.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    H1R, v, basis = g[name + "/H1_R"], g[name + "/vcor"], g[name + "/basis"]
    n, nk = H1R.shape[-1], int(np.prod(mesh))
    L = _lattice(mesh, n, list(range(n)))
    L.set_Ham_lo(fock_lo_R=H1R, hcore_lo_R=H1R)
    L.rdm1_lo_k = g[name + "/rdm1_lo_k"]
    vc = _Vcor(v)
    for fmt in ("nearest", "full", "spin local"):
        tag = fmt.replace(" ", "_")
        if fmt == "full":
            LatH2 = np.zeros((nk, nk, nk) + (n,) * 4)
            LatH2[0, 0, 0], LatH2[1, 0, 2] = g["%s/%s_LatH2_000" % (name, tag)], g["%s/%s_LatH2_102" % (name, tag)]
        else:
            LatH2 = g["%s/%s_LatH2" % (name, tag)]
        L.set_H2_local(LatH2, H2_format=fmt)
        Hn, _ = slater.get_emb_Ham(L, basis, vc, int_bath=False)
        assert np.array_equal(Hn.H2["ccdd"], g["%s/%s_H2" % (name, tag)]), fmt
        assert np.abs(Hn.H1["cd"] - g["%s/%s_H1" % (name, tag)]).max() < 1e-10, fmt
        with pytest.raises(NotImplementedError):
            slater.get_emb_Ham(L, basis, vc)
    L.set_H2_local(LatH2, H2_format="diagonal")
    with pytest.raises(ValueError):
        slater.get_emb_Ham(L, basis, vc, int_bath=False)


@pytest.mark.parametrize("name", ["uhf_231", "rhf_411"])
def test_apply_dmu(ctx, golden, name):
    """dmet/Hubbard.py:82-102 apply_dmu on the embedding Hamiltonians of G8: all impurity orbitals, one chosen orbital (golden G36)."""
    from libdmet_preview_amd.dmet import Hubbard
    from libdmet_preview_amd.system import integral
    g, g8 = golden("G36_init_guess.npz"), golden("G8_embham.npz")
    mesh = tuple(int(x) for x in g8[name + "/mesh"])
    val = [int(x) for x in g8[name + "/val"]]
    basis = g8[name + "/basis"]
    spin, nlo, nb = basis.shape[0], basis.shape[2], basis.shape[-1]
    L = _lattice(mesh, nlo, val)
    for tag, kw in (("all", dict()), ("idx", dict(dmu_idx=[val[0]]))):
        Himp = integral.Integral(nb, spin == 1, False, 0.0, {"cd": np.array(g8[name + "/ib_H1"])}, {"ccdd": g8[name + "/H2"]})
        Himp = Hubbard.apply_dmu(L, Himp, basis, 0.17, **kw)
        assert np.abs(Himp.H1["cd"] - g["%s/dmu_%s" % (name, tag)]).max() < 1e-12


@pytest.mark.parametrize("name,make", [("C1", lambda lat: lat.ChainLattice(12, 2)), ("C2", lambda lat: lat.SquareLattice(12, 12, 2, 2))])
def test_model_lattice_constructors_drive_the_path(ctx, golden, name, make):
    """BASELINE configs 1 / 2 the way a libDMET script builds them -- ChainLattice / SquareLattice + HubbardHamiltonian + setHam
    (system/lattice.py:1085-1107, hamiltonian.py:118-165) -- through the lattice mean field, the bath and the embedding Hamiltonian:
    the values of golden G8 (captured from the reference on the same lattices)."""
    from libdmet_preview_amd.system import lattice, hamiltonian
    from libdmet_preview_amd.routine import mfd, slater
    g = golden("G8_embham.npz")
    L = make(lattice)
    Ham = hamiltonian.HubbardHamiltonian(L, 4.0)
    assert np.array_equal(Ham.getH1(), g[name + "/H1_R"]) and np.array_equal(Ham.getH2(), g[name + "/LatH2"])
    L.setHam(Ham)
    assert L.is_model and L.has_Ham and L.H2_format == "local" and L.use_hcore_as_emb_ham
    vc = _Vcor(g[name + "/vcor"])
    rhoT, mu, E, res = mfd.HF(L, vc, 0.5, True, beta=np.inf, ires=True)
    assert np.abs(res["rho_k"] * 2.0 - g[name + "/rdm1_lo_k"]).max() < 1e-10
    L.rdm1_lo_k = res["rho_k"] * 2.0
    L.use_hcore_as_emb_ham = False                              # G8's model runs fold the Fock (= hcore here) with the JK correction
    basis = g[name + "/basis"]
    Himp, _ = slater.get_emb_Ham(L, basis, vc)
    assert np.abs(Himp.H2["ccdd"] - g[name + "/H2"]).max() < 1e-12 and np.abs(Himp.H1["cd"] - g[name + "/H1"]).max() < 1e-10


@pytest.mark.parametrize("name,make", [("chain12_2", lambda lat: lat.ChainLattice(12, 2)), ("sq44_22", lambda lat: lat.SquareLattice(4, 4, 2, 2))])
def test_model_lattice_update_ham(ctx, golden, name, make):
    """LatticeModel.update_Ham (system/lattice.py:927-972): the Fock of a Hubbard lattice from a DMET density, J / K of the cell-0
    block on the device, restricted (spin-traced density) and unrestricted (golden G38)."""
    from libdmet_preview_amd.system import lattice, hamiltonian
    g = golden("G38_model_lattices.npz")
    for spin in (1, 2):
        L = make(lattice)
        L.setHam(hamiltonian.HubbardHamiltonian(L, 4.0))
        if spin == 2:
            L.hcore_lo_k = np.asarray([L.hcore_lo_k] * 2)
        stripe = g["%s/upd%d_rdm1" % (name, spin)]
        L.update_Ham(stripe * (2.0 if spin == 1 else 1.0))
        ref = g["%s/upd%d_fock_k" % (name, spin)]
        got = np.asarray(L.fock_lo_k)
        assert got.shape == ref.shape or got.squeeze().shape == ref.squeeze().shape
        assert np.abs(got.squeeze() - ref.squeeze()).max() < 1e-12
        assert np.abs(np.asarray(L.fock_lo_R).imag).max() < 1e-12 if np.iscomplexobj(L.fock_lo_R) else True
