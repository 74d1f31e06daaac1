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
