"""
The oracle's own composition of the driver-layer chain (oracle/restate.py HF -> get_emb_basis -> oracle/restate_bcs.py
basisMatching -> restate.get_emb_eri_fast_gdf / unit ERI + unit2emb -> oracle/restate_ham.py embHam1e) against golden G16,
which the REFERENCE's drivers produced (dmet.HartreeFock -> dmet.ConstructImpHam -> embHam, ab-initio duck lattice, UHF,
interacting and bare bath).  CPU only: pins the restatement on the path tests/test_gpu_chain.py checks the HIP code on.
Gauge: the Schmidt basis is defined up to rotations inside degenerate singular subspaces; H1 / H2 are compared after the
rotation T = B_ref^T B into the golden basis (identity on the impurity block).
"""
import numpy as np
import pytest

from oracle import restate as R
from oracle import restate_bcs as RB
from oracle import restate_ham as RH
from oracle import restate_fit as RF
from libdmet_preview_amd import synth

CASES = ["uhf_221", "uhf_311"]


def _rot(basis, ref, nimp):
    spin, nemb = basis.shape[0], basis.shape[-1]
    T = np.zeros((spin, nemb, nemb))
    for s in range(spin):
        T[s] = basis[s].reshape(-1, nemb).T @ ref[s].reshape(-1, nemb)
        assert np.abs(T[s].T @ T[s] - np.eye(nemb)).max() < 1e-9
        assert np.abs(T[s][:nimp, :nimp] - np.eye(nimp)).max() < 1e-10
    return T


def _rot_h2(H2, T, nemb):
    out = []
    ia, ib = np.tril_indices(nemb)
    for b, (s1, s2) in enumerate([(0, 0), (1, 1), (0, 1)]):
        full = R.restore(1, H2[b], nemb)
        full = np.einsum("ijkl,ia,jb,kc,ld->abcd", full, T[s1], T[s1], T[s2], T[s2], optimize=True)
        out.append(full[ia, ib][:, ia, ib])
    return np.asarray(out)


@pytest.mark.parametrize("name", CASES)
def test_G16_oracle_chain(golden, name):
    g = golden("G16_chain.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    val = [int(x) for x in g[name + "/val"]]
    FR, HR, W0, C_ao_lo = g[name + "/Fock_R"], g[name + "/H1_R"], g[name + "/W0"], g[name + "/C_ao_lo"]
    nlo, nk, spin = FR.shape[-1], int(np.prod(mesh)), 2
    core = [i for i in range(nlo) if i < min(val)]
    virt = [i for i in range(nlo) if i > max(val)]
    imp = val + virt                                                   # system/lattice.py:118-120: imp_idx = val + virt
    nimp = len(imp)
    vc = RF.VcorLocal(False, False, nlo, idx_range=val)
    vc.param = np.asarray(g[name + "/vcor_param"])
    v = vc.evaluate()
    assert np.abs(v - g[name + "/vcor_value"]).max() < 1e-14
    Fk, Hk = R.R2k(FR, mesh), R.R2k(HR, mesh)
    rho, mu, E, res = R.HF(mesh, Fk, FR, HR, v, 0.5, False, ires=True)
    assert np.abs(rho - g[name + "/rho"]).max() < 1e-10 and np.abs(np.asarray(mu) - g[name + "/mu"]).max() < 1e-10
    ks = R.make_kpts_scaled(mesh)
    blocks = R.df_blocks_from_W0(W0, mesh, ks)
    naux = W0.shape[0]
    Sk = np.asarray([np.eye(nlo)] * nk)
    basis0 = R.get_emb_basis(mesh, nlo, rho, imp_idx=imp, val_idx=val)
    matched, _ = RB.basisMatching(basis0[:, :, :, nimp:])
    basis = basis0.copy()
    basis[:, :, :, nimp:] = matched
    nemb = basis.shape[-1]
    for tag in ("ib", "nib"):
        key = "%s/%s" % (name, tag)
        ref_basis = g[key + "/basis"]
        assert basis.shape == ref_basis.shape
        T = _rot(basis, ref_basis, nimp)
        if tag == "ib":
            eri = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: blocks[(i, j)], naux, nlo, C_ao_lo=C_ao_lo, basis=basis)
            H2 = RB.reorder_spin_blocks(eri)
        else:
            unit = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: blocks[(i, j)], naux, nlo, C_ao_lo=C_ao_lo, basis=basis, unit_eri=True)
            H2 = RB.unit2emb(RB.reorder_spin_blocks(unit), nemb)
        assert np.abs(_rot_h2(H2, T, nemb) - g[key + "/H2"]).max() < 1e-10
        H1, ovlp, JKc = RH.embHam1e(mesh, basis, H2, Hk, Fk, Sk, res["rho_k"], vcor_mat=v, int_bath=(tag == "ib"))
        H1r = np.einsum("sij,sia,sjb->sab", H1, T, T)
        assert np.abs(H1r - g[key + "/H1"]).max() < 1e-10, tag
