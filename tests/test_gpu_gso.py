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
    with pytest.raises(NotImplementedError):
        spinless.get_emb_basis(L, GRho, kind="eig")
    with pytest.raises(ValueError):
        spinless.get_emb_basis(L, GRho, kind="nope")


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
