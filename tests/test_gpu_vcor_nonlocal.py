"""
GPU parity tests (-m gpu) of the cell-resolved correlation potential and the fit it drives (reference routine/vcor.py:105-524
VcorNonLocal; slater.py:893-902 the non-local branch of get_dV_dparam; FitVcorEmb on top; mfd.py:369-392 the mean field under
a potential that differs from k to k).  HIP path against golden G23 (captured from the reference) and oracle/restate_fit.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import restate_fit as F
from tests.test_oracle_fit import NONLOCAL_LATTICES, NONLOCAL_MODES, NONLOCAL_FITS, fit_inputs, idx_sets


def _lattice(mesh, nlo, val, FR, Fk, spin):
    from libdmet_preview_amd.system.lattice import Lattice
    L = Lattice(int(nlo), mesh)
    L.val_idx = list(val)
    L.virt_idx = [i for i in range(nlo) if i > max(val)]
    L.core_idx = [i for i in range(nlo) if i < min(val)]
    if spin == 1:
        L.set_Ham_lo(fock_lo_R=FR[0], hcore_lo_R=FR[0])
    else:
        L.set_Ham_lo(fock_lo_R=FR, hcore_lo_R=FR)
    L.fock_lo_k = L.hcore_lo_k = Fk if spin == 2 else Fk[0]
    return L


@pytest.mark.parametrize("lat", NONLOCAL_LATTICES, ids=[x[0] for x in NONLOCAL_LATTICES])
@pytest.mark.parametrize("mode", NONLOCAL_MODES, ids=[x[0] for x in NONLOCAL_MODES])
def test_update_get_gradient_assign(golden, lat, mode):
    from libdmet_preview_amd.system.lattice import Lattice
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G23_vcor_nonlocal.npz")
    (lname, mesh, nlo, idx), (mname, res, bogo, bres) = lat, mode
    key = "tab/%s/%s" % (lname, mname)
    v = Hubbard.VcorNonLocal(res, bogo, Lattice(nlo, mesh), idx_range=idx, bogo_res=bres)
    v.update(g[key + "/param"])
    assert np.array_equal(v.value, g[key + "/value"])
    assert np.abs(v.value_k - g[key + "/value_k"]).max() < 1e-13
    assert np.abs(v.get(1, True) - g[key + "/get_k1"]).max() < 1e-13 and np.array_equal(v.get(1, False), g[key + "/get_R1"])
    assert v.get(return_all=True).shape == v.value_k.shape and v.get(kspace=False, return_all=True) is v.value
    gr = v.gradient()
    assert np.array_equal(np.asarray(np.nonzero(gr)), g[key + "/grad_nz"]) and np.array_equal(gr[np.nonzero(gr)], g[key + "/grad_val"])
    if lname == "m411":
        assert np.abs(v.grad_k - g[key + "/grad_k"]).max() < 1e-13
    v.assign(g[key + "/assign_in"])
    assert np.array_equal(v.param, g[key + "/assign_param"])
    with pytest.raises(Exception):
        v.update(np.zeros(v.length() + 1))


class _DenseOnly(object):
    """What the reference's own VcorNonLocal offers: a dense gradient(), no index table."""

    def __init__(self, v):
        self._v, self.grad, self.grad_k = v, None, None

    def is_local(self):
        return False

    def length(self):
        return self._v.length()

    def gradient(self):
        P, B, C, I, J = self._v.cell_entries()
        g = np.zeros((self._v.nparam, self._v.nblk, self._v.ncells, self._v.nscsites, self._v.nscsites))
        g[P, B, C, I, J] = 1
        return g


@pytest.mark.parametrize("name", NONLOCAL_FITS)
def test_dV_dparam_nonlocal(golden, name, monkeypatch):
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G23_vcor_nonlocal.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    L = _lattice(mesh, nlo, val, FR, Fk, spin)
    v = Hubbard.VcorNonLocal(spin == 1, False, L, idx_range=val)
    ref = g[name + "/dV_compact"]
    got = slater.get_dV_dparam(v, basis, None, L)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-13
    assert np.abs(got - F.get_dV_dparam(F.VcorNonLocal(spin == 1, False, mesh, nlo, val), basis)).max() < 1e-13
    assert np.abs(slater.get_dV_dparam(v, basis, None, L, compact=False) - g[name + "/dV_full"]).max() < 1e-13
    # an active-space projector is not read by this branch (slater.py:893-902)
    assert np.array_equal(slater.get_dV_dparam(v, basis, None, L, P_act=np.zeros((spin, basis.shape[1], nlo, nlo))), got)
    # the reference's own object (dense gradient only), and the lattice taken from the argument
    assert np.abs(slater.get_dV_dparam(_DenseOnly(v), basis, None, L) - ref).max() < 1e-13
    # shifted Gram matrices in several passes (one +-R pair per pass) and a rank's slice of the parameters
    monkeypatch.setattr(slater, "CELL_GRAM_DOUBLES", 1)
    assert np.array_equal(slater.get_dV_dparam(v, basis, None, L), got)
    from libdmet_preview_amd import _lib
    part = slater.get_dV_dparam_dev(_lib.get_ctx(), v, basis, rows=(3, 11)).get()
    assert np.array_equal(part, got[3:11])


@pytest.mark.parametrize("name", NONLOCAL_FITS)
def test_fit_nonlocal_vs_reference(golden, name):
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G23_vcor_nonlocal.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    L = _lattice(mesh, nlo, val, FR, Fk, spin)
    L.ovlp_lo_k = Sk if spin == 1 else np.asarray([Sk] * 2)
    for tag, beta, kw in [("t0", np.inf, {}), ("ft", 15.0, {}), ("imp_t0", np.inf, dict(imp_fit=True))]:
        key = "%s/%s" % (name, tag)
        v = Hubbard.VcorNonLocal(spin == 1, False, L, idx_range=val)
        v.update(np.zeros(v.length()))
        vfit, e0, e1 = slater.FitVcorEmb(target, L, basis, v, beta, MaxIter=30, **kw)
        fit = slater.FitVcorEmb.last_fit
        for p, e, gr in zip(g[key + "/probe"], g[key + "/probe_err"], g[key + "/probe_grad"]):
            assert abs(fit.errfunc(p) - e) < 1e-11, key
            assert np.abs(fit.gradfunc(p) - gr).max() < 1e-8 * max(1.0, np.abs(gr).max()), key
        pref, (r0, r1) = g[key + "/param"], g[key + "/err"]
        assert abs(e0 - r0) < 1e-11, key
        assert abs(e1 - r1) < 1e-6, (key, e1, r1)
        assert vfit is v and e1 <= e0
        assert np.abs(vfit.value - vfit.evaluate()).max() == 0.0 and vfit.value_k is not None
    # the two-step driver with no lattice stage (libdmet/test/test_vcor_nonlocal.py: MaxIter2 = 0) returns a fitted COPY
    v = Hubbard.VcorNonLocal(spin == 1, False, L, idx_range=val)
    v.update(np.zeros(v.length()))
    vnew, err = Hubbard.FitVcor(target, L, basis, v, np.inf, 0.5, MaxIter1=30, MaxIter2=0)
    assert vnew is not v and vnew.lattice is L and np.abs(np.asarray(v.param)).max() == 0.0
    assert abs(err - g[name + "/t0/err"][1]) < 1e-6
    with pytest.raises(NotImplementedError):
        slater.FitVcorFull(target, L, basis, v, 15.0, 0.5, MaxIter=2)


@pytest.mark.parametrize("name", NONLOCAL_FITS)
def test_mean_field_under_a_nonlocal_potential(golden, name):
    from libdmet_preview_amd.routine import mfd
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G23_vcor_nonlocal.npz")
    mesh, FR, Fk, Sk, basis, target, val, spin, nlo, nelec = fit_inputs(g, name)
    L = _lattice(mesh, nlo, val, FR, Fk, spin)
    v = Hubbard.VcorNonLocal(spin == 1, False, L, idx_range=val)
    v.update(g[name + "/hf_param"])
    for tag, beta in (("t0", np.inf), ("ft", 12.0)):
        rhoT, mu, E, res = mfd.HF(L, v, 0.5, spin == 1, beta=beta, ires=True)
        key = "%s/hf_%s" % (name, tag)
        assert np.abs(res["e"] - g[key + "/ew"]).max() < 1e-11
        assert np.abs(np.asarray(mu) - g[key + "/mu"]).max() < 1e-9
        assert np.abs(rhoT - g[key + "/rho"]).max() < 1e-10
        assert abs(E - float(g[key + "/E"])) < 1e-9


# ---- the k-point-resolved potential in the lattice-stage fit and the mean field (golden G24) --------------------------------

from tests.test_oracle_fit import KPTS_RUNS  # noqa: E402
from oracle import restate as R  # noqa: E402


def _kpts_inputs(g, name):
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    basis, FR, target = g[name + "/basis"], g[name + "/Fock_R"], g[name + "/target"]
    spin, nlo = basis.shape[0], FR.shape[-1]
    Fk = R.R2k(FR, mesh)
    L = _lattice(mesh, nlo, [int(x) for x in g[name + "/val"]], FR, Fk, spin)
    return mesh, basis, FR, Fk, target, spin, nlo, L


@pytest.mark.parametrize("name", NONLOCAL_FITS)
def test_full_fit_kpoints_vs_reference(golden, name):
    """FitVcorFull with a VcorKpoints potential (slater.py:1519-1628) against the reference's own closures and fits: finite-T
    analytic gradient (impurity block, diagonal, fixed mu) and the numerical-gradient T = 0 run; and against the oracle at a
    parameter vector of its own."""
    from libdmet_preview_amd.routine import slater
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G24_vcor_kpoints.npz")
    mesh, basis, FR, Fk, target, spin, nlo, L = _kpts_inputs(g, name)
    for tag, beta, kw in KPTS_RUNS:
        key = "%s/%s" % (name, tag)
        v = Hubbard.VcorKpoints(spin == 1, False, L)
        vfit, e0, e1 = slater.FitVcorFull(target, L, basis, v, beta, 0.5, MaxIter=3 if "num" in tag else 12, **kw)
        fit = slater.FitVcorFull.last_fit
        for i, p in enumerate(g[key + "/probe"]):
            assert abs(fit.errfunc(p) - g[key + "/probe_err"][i]) < 1e-11, key
            if key + "/probe_grad" in g:
                gr = g[key + "/probe_grad"][i]
                assert np.abs(fit.gradfunc(p) - gr).max() < 1e-8 * max(1.0, np.abs(gr).max()), key
        pref, (r0, r1) = g[key + "/param"], g[key + "/err"]
        assert abs(e0 - r0) < 1e-11, key
        assert abs(e1 - r1) < 1e-6, (key, e1, r1)
        assert vfit is v and e1 <= e0
    # the oracle at another point
    v = Hubbard.VcorKpoints(spin == 1, False, L)
    fit = slater.FullFitDevice(slater.get_ctx(), target, L, basis, v, 9.0, F.FullFit(target, mesh, basis, F.VcorKpoints(spin == 1, mesh, nlo), 9.0,
                               Fk if spin == 2 else Fk[0], 0.5, imp_idx=list(range(nlo)), det_idx=[]).nelec, list(range(nlo)), [], False)
    ofit = F.FullFit(target, mesh, basis, F.VcorKpoints(spin == 1, mesh, nlo), 9.0, Fk if spin == 2 else Fk[0], 0.5,
                     imp_idx=list(range(nlo)), det_idx=[])
    p = 0.2 * np.random.default_rng(3).standard_normal(v.length())
    assert abs(fit.errfunc(p) - ofit.errfunc(p)) < 1e-11
    assert np.abs(fit.gradfunc(p) - ofit.gradfunc_ft(p)).max() < 1e-9
    with pytest.raises(NotImplementedError):
        slater.FitVcorFull(target, L, basis, v, np.inf, 0.5, MaxIter=2, imp_fit=True)          # T = 0 needs num_grad (slater.py:1642-1645)
    with pytest.raises(NotImplementedError):
        nb = basis.shape[-1]                                                                    # imp + bath fit has no gradient (:1510-1512)
        slater.FitVcorFull(np.zeros((spin, nb, nb)), L, basis, v, 15.0, 0.5, MaxIter=2)


@pytest.mark.parametrize("name", NONLOCAL_FITS)
def test_mean_field_under_a_kpoints_potential(golden, name):
    from libdmet_preview_amd.routine import mfd
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G24_vcor_kpoints.npz")
    mesh, basis, FR, Fk, target, spin, nlo, L = _kpts_inputs(g, name)
    v = Hubbard.VcorKpoints(spin == 1, False, L)
    v.update(g[name + "/hf_param"])
    for tag, beta in (("t0", np.inf), ("ft", 12.0)):
        rhoT, mu, E, res = mfd.HF(L, v, 0.5, spin == 1, beta=beta, ires=True)
        key = "%s/hf_%s" % (name, tag)
        assert np.abs(res["e"] - g[key + "/ew"]).max() < 1e-11
        assert np.abs(np.asarray(mu) - g[key + "/mu"]).max() < 1e-9
        assert np.abs(rhoT - g[key + "/rho"]).max() < 1e-10
        assert abs(E - float(g[key + "/E"])) < 1e-9
