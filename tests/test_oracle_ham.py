"""
Pins oracle/restate_ham.py (embedding one-body Hamiltonian + ERI x density)
against tests/golden/G8_embham.npz, captured from the reference's get_emb_Ham / _get_jk / get_veff /
transform_* under oracle/shim.py.  CPU only.
"""
import numpy as np
import pytest

from oracle import restate as R
from oracle import restate_ham as H

AB = ["uhf_231", "rhf_411", "uhf_222"]
MODELS = ["C1", "C2", "C1u"]
RUNS = [("ib", {}), ("ib_vcor", dict(add_vcor=True)), ("ib_vcor_fit", dict(add_vcor=True, fitting=True)),
        ("nib", dict(int_bath=False)), ("nib_jk2", dict(int_bath=False, JK_imp="JK_imp2")),
        ("nib_jk3", dict(int_bath=False, JK_imp="JK_imp3")),
        ("nib_hcore", dict(int_bath=False, use_hcore_as_emb_ham=True))]


def ab_inputs(g, name):
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    FR, HR, SR = g[name + "/Fock_R"], g[name + "/H1_R"], g[name + "/S_R"]
    Fk, Hk = R.R2k(FR, mesh), R.R2k(HR, mesh)
    Sk = R.R2k(SR, mesh)
    return mesh, FR, Fk, Hk, Sk, g[name + "/vcor"], g[name + "/rdm1_lo_k"], g[name + "/basis"], g[name + "/H2"]


@pytest.mark.parametrize("name", AB)
def test_G8_emb_ham(golden, name):
    g = golden("G8_embham.npz")
    mesh, FR, Fk, Hk, Sk, v, rdm1_k, basis, H2 = ab_inputs(g, name)
    spin = basis.shape[0]
    for tag, kw in RUNS:
        kw = dict(kw)
        if "JK_imp" in kw:
            kw["JK_imp"] = g[name + "/" + kw["JK_imp"]]
        H1, ov, JKc = H.embHam1e(mesh, basis, H2, Hk, Fk, Sk, rdm1_k, vcor_mat=v, **kw)
        assert np.abs(H1 - g["%s/%s_H1" % (name, tag)]).max() < 1e-11, tag
        assert np.abs(ov - g["%s/%s_ovlp" % (name, tag)]).max() < 1e-12, tag
        key = "%s/%s_JK_core" % (name, tag)
        if JKc is None:
            assert key not in g
        else:
            assert np.abs(JKc - g[key]).max() < 1e-11, tag
    assert float(g[name + "/ib_H0"]) == 1.25


@pytest.mark.parametrize("name", AB)
def test_G8_jk(golden, name):
    g = golden("G8_embham.npz")
    H2, dm = g[name + "/H2"], g[name + "/rdm1_emb"]
    nb = dm.shape[-1]
    mesh, FR, Fk, Hk, Sk, v, rdm1_k, basis, _ = ab_inputs(g, name)
    basis_k = np.asarray([R.R2k(basis[s], mesh) for s in range(basis.shape[0])])
    assert np.abs(H.foldRho_k(rdm1_k, basis_k) - dm).max() < 1e-12
    for tag, eri in [("s4", H2), ("s1", np.asarray([R.restore(1, h, nb) for h in H2])), ("s8", R.restore(8, H2[0], nb)),
                     ("res", H2[:1])]:
        vj, vk = H.get_jk(dm, eri)
        assert np.abs(vj - g["%s/jk_%s_vj" % (name, tag)]).max() < 1e-12, tag
        assert np.abs(vk - g["%s/jk_%s_vk" % (name, tag)]).max() < 1e-12, tag
    for hyb in (1.0, 0.0, 0.4):
        assert np.abs(H.get_veff(dm, H2, hyb=hyb) - g["%s/veff_hyb%.1f" % (name, hyb)]).max() < 1e-12
    assert np.abs(H.get_veff(dm[0], H2[:1]) - g[name + "/veff_dm2d"]).max() < 1e-12


@pytest.mark.parametrize("name", AB)
def test_G8_folds(golden, name):
    g = golden("G8_embham.npz")
    mesh, FR, Fk, Hk, Sk, v, rdm1_k, basis, H2 = ab_inputs(g, name)
    for s in range(basis.shape[0]):
        assert np.abs(H.transform_trans_inv(basis[s], mesh, FR[s]) - g["%s/ti_sym_%d" % (name, s)]).max() < 1e-12
        assert np.abs(H.transform_trans_inv(basis[s], mesh, FR[s], False) - g["%s/ti_full_%d" % (name, s)]).max() < 1e-12
        assert np.abs(H.transform_local(basis[s], v[s]) - g["%s/tloc_%d" % (name, s)]).max() < 1e-13
        assert np.abs(H.transform_imp(basis[s], v[s]) - g["%s/timp_%d" % (name, s)]).max() < 1e-13
        assert np.abs(H.transform_imp_env(basis[s], FR[s]) - g["%s/tie_%d" % (name, s)]).max() < 1e-13
    basis_k = np.asarray([R.R2k(basis[s], mesh) for s in range(basis.shape[0])])
    assert np.abs(H.transform_h1(Hk, basis_k) - g[name + "/h1_emb"]).max() < 1e-12


@pytest.mark.parametrize("name", MODELS)
def test_G8_model(golden, name):
    g = golden("G8_embham.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    H1R, v, basis, LatH2 = g[name + "/H1_R"], g[name + "/vcor"], g[name + "/basis"], g[name + "/LatH2"]
    spin, nb = basis.shape[0], basis.shape[-1]
    n = H1R.shape[-1]
    Hk = R.R2k(H1R, mesh)
    SR = np.zeros_like(H1R)
    SR[0] = np.eye(n)
    Sk = R.R2k(SR, mesh)
    H2 = H.transform_eri_local(basis, LatH2)
    assert np.abs(H2 - g[name + "/H2"]).max() < 1e-12
    H1, ov, JKc = H.embHam1e(mesh, basis, H2, Hk, Hk, Sk, g[name + "/rdm1_lo_k"], vcor_mat=v)
    assert np.abs(H1 - g[name + "/H1"]).max() < 1e-11
    assert np.abs(JKc - g[name + "/JK_core"]).max() < 1e-11
    H2n = H.unit2emb(np.asarray((LatH2,) * (spin * (spin + 1) // 2)), nb)
    assert np.array_equal(H2n, g[name + "/nib_H2"])
    H1n, _, _ = H.embHam1e(mesh, basis, H2n, Hk, Hk, Sk, g[name + "/rdm1_lo_k"], vcor_mat=v, int_bath=False)
    assert np.abs(H1n - g[name + "/nib_H1"]).max() < 1e-11


GHF_VEFF = [("hf", dict()), ("j", dict(hyb=0.0)), ("j07", dict(hyb=0.0, hyb_j=0.7)), ("hyb", dict(hyb=0.25)), ("hyb_j07", dict(hyb=0.25, hyb_j=0.7))]


@pytest.mark.parametrize("name", ["n6", "n10"])
def test_G25_veff_ghf(golden, name):
    """slater.get_veff(ghf=True) (slater.py:489-506): every branch, the three ERI storages agree."""
    g = golden("G25_veff_ghf.npz")
    dm, e4 = g[name + "/dm"], g[name + "/eri_s4"]
    for tag, kw in GHF_VEFF:
        ref = g["%s/s4/%s" % (name, tag)]
        assert np.abs(H.get_veff(dm, e4, ghf=True, **kw) - ref).max() < 1e-13
        assert np.abs(g["%s/s1/%s" % (name, tag)] - ref).max() < 1e-12 and np.abs(g["%s/s8/%s" % (name, tag)] - ref).max() < 1e-12
    assert np.abs(g[name + "/veff_ghf"] - g[name + "/s4/hf"]).max() < 1e-13
