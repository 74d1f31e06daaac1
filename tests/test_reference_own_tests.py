"""
The reference's OWN known-answer tests for this path that need no PySCF, re-pointed at this package: the same
procedures, seeds, sizes and tolerances, written against `libdmet_preview_amd` instead of `libdmet`.

  routine/test/test_ft_system.py:7-71    test_ftsystem      finite-T response formulas against finite differences  (GPU)
  routine/test/test_ft_system.py:73-100  test_smearing_occ  broadcasting of mu in the smearing functions              (CPU)
  routine/test/test_bath_eig.py:13-65    test_deg           T = 0 occupations with degenerate levels: the embedding
                                                            density equals the projected lattice density to 1e-12     (GPU)
  routine/test/test_fit.py:9-24          test_fit           every optimiser driver on a convex function               (CPU)
  routine/test/test_vcor.py:7-30         test_vcor_local    the local vcor parametrisation                            (CPU)

(system/test/test_fourier.py and routine/test/test_mfd_mpi.py are in tests/test_host_abi.py / test_oracle_golden.py,
routine/test/test_slater.py in tests/test_gpu_parity.py.)
"""
import numpy as np
import pytest
import scipy.linalg as la


def _tril_to_sym(arr, n):
    m = np.zeros((n, n))
    m[np.tril_indices(n)] = arr
    return m + np.tril(m, -1).T


@pytest.mark.gpu
def test_ftsystem():
    from libdmet_preview_amd.routine import ftsystem as ft
    np.random.seed(1)
    norb, nelec, beta = 10, 4, 10.0
    h = ft.get_h_random_deg(norb, deg_orbs=[[0, 3], [1, 2], [4, 5, 6], [7]], deg_energy=[1.0, 0.1, 0.8, 3.0])
    assert np.abs(h - h.T).max() < 1e-13
    ew = la.eigvalsh(h)
    assert np.sum(np.abs(ew - 0.8) < 1e-10) == 3 and np.sum(np.abs(ew - 1.0) < 1e-10) == 2      # the degeneracies were planted
    fix_mu = False
    mo_energy, mo_coeff, mo_occ, mu = ft.kernel(h, nelec, beta)
    assert abs(mo_occ.sum() - nelec) < 1e-10
    grad_ana = ft.get_rho_grad(mo_energy, mo_coeff, mu, beta, fix_mu=fix_mu, compact=False)
    # d w / d v with w = sum rho^2, against forward differences over the lower triangle of h
    rho0 = ft.make_rdm1(mo_coeff, mo_occ)
    f0 = (rho0 * rho0).sum()
    dw_dv = ft.get_dw_dv(mo_energy[None], mo_coeff[None], rho0[None], mu, beta, fix_mu=fix_mu, compact=True)
    h_ref = h[np.tril_indices(norb)]
    dx = 1e-6
    grad_w = np.zeros_like(h_ref)
    grad_rho = np.zeros_like(grad_ana)
    for i in range(len(h_ref)):
        harr = h_ref.copy()
        harr[i] += dx
        e1, c1, o1, _ = ft.kernel(_tril_to_sym(harr, norb), nelec, beta, mu0=mu, fix_mu=fix_mu)
        rho = ft.make_rdm1(c1, o1)
        grad_w[i] = ((rho * rho).sum() - f0) / dx
        grad_rho[i] = (rho - rho0) / dx
    assert la.norm(grad_w - dw_dv) < 1e-4
    assert la.norm(grad_rho - grad_ana) < 1e-4
    # the compact form is the same tensor with the density index tril-packed too
    comp = ft.get_rho_grad(mo_energy, mo_coeff, mu, beta, fix_mu=fix_mu, compact=True)
    tl = np.tril_indices(norb)
    assert np.array_equal(comp, grad_ana[:, tl[0], tl[1]])
    # fixed chemical potential: the analytic response loses its mu term and matches differences taken at fixed mu
    ga_fix = ft.get_rho_grad(mo_energy, mo_coeff, mu, beta, fix_mu=True, compact=False)
    harr = h_ref.copy()
    harr[3] += dx
    e1, c1, o1, _ = ft.kernel(_tril_to_sym(harr, norb), nelec, beta, mu0=mu, fix_mu=True)
    assert la.norm((ft.make_rdm1(c1, o1) - rho0) / dx - ga_fix[3]) < 1e-4


@pytest.mark.parametrize("method", ["fermi", "gaussian"])
@pytest.mark.parametrize("mu", [-1.0, np.array(1.0), np.array([1.0]), [1.0, 2.0], np.arange(2)])
def test_smearing_occ(method, mu):
    from libdmet_preview_amd.routine.ftsystem import fermi_smearing_occ, gaussian_smearing_occ
    f_occ = fermi_smearing_occ if method == "fermi" else gaussian_smearing_occ
    beta = 50.0
    ew = np.arange(120).reshape(2, 3, 5, 4)
    occ = f_occ(mu, ew, beta)
    assert occ.shape == ew.shape and np.all((occ >= 0) & (occ <= 1))
    ew = np.arange(8).reshape(2, 4).astype(np.double)
    occ = f_occ(mu, ew, beta)
    assert occ.shape == ew.shape
    if np.array(mu).size == 2:                       # one chemical potential per spin sector
        assert np.array_equal(occ[0], f_occ(np.array(mu)[0], ew[0], beta)) and np.array_equal(occ[1], f_occ(np.array(mu)[1], ew[1], beta))
    if np.array(mu).size == 1:
        occ = f_occ(mu, np.arange(6), beta)
        assert occ.shape == (6,)


@pytest.mark.gpu
@pytest.mark.parametrize("nelec_lat", [8, 9])
def test_deg(nelec_lat):
    from libdmet_preview_amd.routine import mfd, ftsystem
    np.random.seed(1)
    norb = 16
    h = ftsystem.get_h_random_deg(norb, deg_orbs=[[0, 3, 8], [1, 2], [4, 5, 6], [7]], deg_energy=[1.0, 0.1, 0.8, 3.0])
    ew, ev = la.eigh(h)
    ewocc, mu, err = mfd.assignocc(ew, nelec_lat, beta=np.inf, mu0=0.0)
    assert abs(ewocc.sum() - nelec_lat) < 1e-12
    rdm1_full = (ev * ewocc) @ ev.T
    nimp = 2
    w, v = la.eigh(rdm1_full[nimp:, nimp:])
    bath = v[:, (np.abs(w) > 1e-6) & (np.abs(1 - w) > 1e-6)]
    nbath = bath.shape[-1]
    basis = np.zeros((norb, nimp + nbath))
    basis[:nimp, :nimp] = np.eye(nimp)
    basis[nimp:, nimp:] = bath
    h1_emb = basis.T @ h @ basis
    rdm1_emb = basis.T @ rdm1_full @ basis
    ew2, ev2 = la.eigh(h1_emb)
    nelec = int(np.round(rdm1_emb.trace()))
    occ2, mu2, err2 = mfd.assignocc(ew2, nelec=nelec, beta=np.inf, mu0=0.0)
    rdm1 = (ev2 * occ2) @ ev2.T
    assert np.abs(rdm1 - rdm1_emb).max() < 1e-12


@pytest.mark.parametrize("method", ["SD", "CG", "BFGS", "trust-NCG", "CIAH"])
def test_fit(method):
    from libdmet_preview_amd.routine.fit import minimize
    func = lambda x: x[0] ** 2 + x[1] ** 4 + 2 * x[1] ** 2 + 2 * x[0] + 2.0
    x0 = np.asarray([10.0, 20.0])
    kw = dict(MaxIter=3000, method=method, initial_trust_radius=1.0, max_trust_radius=1000.0, num_cg_steps=1, max_stepsize=100.0)
    if method == "CIAH":                             # needs pyscf.soscf.ciah: the one driver this package refuses
        with pytest.raises(NotImplementedError):
            minimize(func, x0, **kw)
        return
    x, y, pattern, _ = minimize(func, x0, **kw)
    assert y < func(x0) and y < 1.0 + 1e-3 and abs(x[0] + 1.0) < 1e-2 and abs(x[1]) < 0.05      # the minimum is f(-1, 0) = 1


def test_vcor_local():
    from libdmet_preview_amd.dmet.Hubbard import VcorLocal
    v = VcorLocal(True, False, 4)
    v.update(np.asarray([2, 1, 0, -1, 3, 4, 2, 1, 2, 3]))
    val, g = v.get(), v.gradient()
    assert val.shape == (2, 4, 4) and np.array_equal(val[0], val[1]) and np.array_equal(val[0], val[0].T) and g.shape == (10, 2, 4, 4)
    assert np.array_equal(val[0][0], [2, 1, 0, -1]) and np.array_equal(np.diag(val[0]), [2, 3, 1, 3])
    v = VcorLocal(False, False, 2)
    v.update(np.asarray([2, 1, 0, -1, 3, 4]))
    assert np.array_equal(v.get()[0], [[2, 1], [1, 0]]) and np.array_equal(v.get()[1], [[-1, 3], [3, 4]]) and v.gradient().shape == (6, 2, 2, 2)
    v = VcorLocal(False, True, 2)
    v.update(np.asarray([1, 2, 3, 4, 5, 6, 7, 8, 9, 10]))
    assert v.get().shape == (3, 2, 2) and np.array_equal(v.get()[2], [[7, 8], [9, 10]]) and v.gradient().shape == (10, 3, 2, 2)


@pytest.mark.gpu
def test_get_cderi_rhf_and_uhf():
    """utils/test/test_cholesky.py:8-91 with a seeded positive semi-definite 4-fold ERI in place of the PySCF molecule: the vectors
    of get_cderi_rhf rebuild the ERI (1e-7, the reference's bound), and get_cderi_uhf on (0.5 E, E, 0.05 E)... the reference's own
    cross-check: the spin-blocked routine equals modified_cholesky of the assembled [[aa, ab], [ab^T, bb]] matrix."""
    from libdmet_preview_amd.utils import cholesky
    from oracle import restate as R
    norb, rank = 7, 18
    npair = norb * (norb + 1) // 2
    rng = np.random.default_rng(8)
    Lm = rng.standard_normal((rank, npair)) * np.exp(-0.25 * np.arange(rank))[:, None]
    eri_s4 = Lm.T @ Lm
    cd = cholesky.get_cderi_rhf(eri_s4, norb, tol=1e-8)
    assert cd.shape[1:] == (norb, norb) and np.abs(cd - cd.transpose(0, 2, 1)).max() == 0.0
    packed = np.asarray([R.pack_tril(x) for x in cd])
    assert la.norm(packed.T @ packed - eri_s4) < 1e-7
    # a valid spin-dependent triple: aa, bb positive semi-definite and ab = A^T B of the same factors
    La = rng.standard_normal((rank, npair)) * np.exp(-0.3 * np.arange(rank))[:, None]
    Lb = 0.6 * La + 0.8 * rng.standard_normal((rank, npair)) * np.exp(-0.3 * np.arange(rank))[:, None]
    eri3 = [La.T @ La, Lb.T @ Lb, La.T @ Lb]
    cu = cholesky.get_cderi_uhf(eri3, norb, tol=1e-8)
    block = np.block([[eri3[0], eri3[2]], [eri3[2].T, eri3[1]]])
    ev = cholesky.modified_cholesky(block, max_error=1e-8)
    assert ev.shape[0] == cu.shape[1]
    def unpack(t):
        m = np.zeros((norb, norb))
        m[np.tril_indices(norb)] = t
        return m + np.tril(m, -1).T
    ref = np.asarray([[unpack(v[:npair]) for v in ev], [unpack(v[npair:]) for v in ev]])
    assert la.norm(cu[:, :-1] - ref[:, :-1]) < 1e-7                       # (the last vector of a rank-deficient matrix: rounding residue)
    pa, pb = (np.asarray([R.pack_tril(x) for x in cu[s]]) for s in (0, 1))
    assert max(la.norm(pa.T @ pa - eri3[0]), la.norm(pb.T @ pb - eri3[1]), la.norm(pa.T @ pb - eri3[2])) < 1e-7
    assert np.array_equal(cholesky.modified_cholesky_uhf(eri3, max_error=1e-8)[:, :npair], pa)
