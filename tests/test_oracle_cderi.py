"""
Pins oracle/restate_cderi.py (on-disk DF layout) against tests/golden/G13_cderi.npz: the
datasets the reference's own transform_gdf_to_lo writes (captured through a dict-backed stand-in for h5py.File) and
its get_mask_kptij_lst tables.  Also the host-side reader of the product (CderiProvider) on that layout.  CPU only.
"""
import numpy as np
import pytest

from oracle import restate as R
from oracle import restate_cderi as Cd

CASES = ["m311", "m221", "m231"]


def inputs(g, name):
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    W0, C = g[name + "/W0"], g[name + "/C_ao_lo"]
    ks = R.make_kpts_scaled(mesh)
    blocks = R.df_blocks_from_W0(W0, mesh, ks)
    return mesh, ks, 2.0 * np.pi * ks, blocks, W0.shape[0], C


def golden_container(g, tag):
    pre = tag + "/data/"
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


@pytest.mark.parametrize("name", CASES)
def test_G13_writer_layout(golden, name):
    g = golden("G13_cderi.npz")
    mesh, ks, kabs, blocks, naux, C = inputs(g, name)
    for tr in (True, False):
        out, mask = Cd.transform_gdf_to_lo(lambda i, j: blocks[(i, j)], ks, kabs, naux, C, t_reversal_symm=tr)
        ref = golden_container(g, "%s/%s" % (name, "tr" if tr else "notr"))
        assert sorted(out.keys()) == sorted(ref.keys())
        for k in ref:
            assert out[k].shape == ref[k].shape and out[k].dtype.kind == ref[k].dtype.kind, k
            assert np.abs(out[k] - ref[k]).max() < 1e-12, k
        if tr:
            assert np.array_equal(mask, g[name + "/mask"])


@pytest.mark.parametrize("tag", ["4x1x1", "4x4x1", "2x2x2", "3x3x1"])
def test_G13_mask_tables(golden, tag):
    g = golden("G13_cderi.npz")
    mesh = tuple(int(x) for x in tag.split("x"))
    ks = R.make_kpts_scaled(mesh)
    assert np.array_equal(Cd.get_mask_kptij_lst(Cd.kptij_list(ks)), g["mask/" + tag])


@pytest.mark.parametrize("name", CASES)
def test_G13_reader_round_trip(golden, name):
    """Reading the reference-written container gives back C_i^H L^(ki,kj) C_j for EVERY ordered pair (swap, diagonal)."""
    from libdmet_preview_amd.basis_transform.eri_transform import CderiProvider
    g = golden("G13_cderi.npz")
    mesh, ks, kabs, blocks, naux, C = inputs(g, name)
    nk, nao, nlo = C.shape
    for tr in (True, False):
        feri = golden_container(g, "%s/%s" % (name, "tr" if tr else "notr"))
        prov = CderiProvider(feri, kabs, nlo)
        assert prov.naux == naux
        for i in range(nk):
            for j in range(nk):
                want = R.transform_ao_to_emb(blocks[(i, j)].reshape(naux, nao * nao), C[None], i, j)[0]
                assert np.abs(prov.get_block(i, j) - want).max() < 1e-12, (i, j)
                assert np.abs(Cd.load_block(feri, nk, nlo, i, j) - want).max() < 1e-12, (i, j)
                # host feed: ONE pass into the caller's (pinned) buffer, the block AS STORED; a pair kept as (j, i) is flagged
                # so that dmk_eri_push_block_host conjugate-transposes it on the device
                buf = np.full((naux, nlo, nlo), np.nan + 0j)
                swapped = prov.load_block_host(i, j, buf)
                got = buf.conj().transpose(0, 2, 1) if swapped else buf
                assert np.abs(got - want).max() < 1e-12, (i, j, swapped)
                assert swapped == ((i, j) not in prov.pair_of)
    # nested-dict containers (the form convert_eri_to_gdf returns without a file name) are read too
    nested = {"j3c-kptij": feri["j3c-kptij"], "j3c": {}}
    for k, v in feri.items():
        if k.startswith("j3c/"):
            _, p, s = k.split("/")
            nested["j3c"].setdefault(p, {})[s] = v
    assert np.abs(CderiProvider(nested, kabs, nlo).get_block(1, 0) - prov.get_block(1, 0)).max() == 0.0


@pytest.mark.parametrize("tag", ["4x1x1", "4x4x1", "2x2x2", "3x3x1"])
def test_product_mask_bit_exact(golden, tag):
    from libdmet_preview_amd.basis_transform.eri_transform import get_mask_kptij_lst
    from libdmet_preview_amd.system.lattice import _UnitCell
    g = golden("G13_cderi.npz")
    mesh = tuple(int(x) for x in tag.split("x"))
    cell = _UnitCell(2)
    kpts = cell.get_abs_kpts(R.make_kpts_scaled(mesh))
    kptij = np.asarray([(ki, kpts[j]) for i, ki in enumerate(kpts) for j in range(i + 1)])
    assert np.array_equal(get_mask_kptij_lst(cell, kptij), g["mask/" + tag])


# ---- golden G20 (gen_G20): convert_eri_to_gdf, the molecular-ERI -> Gamma-point container writer ---------------------------------
@pytest.mark.parametrize("name,norb", [("n4", 4), ("n6", 6), ("n9", 9)])
def test_G20_convert_eri_to_gdf(golden, name, norb):
    """eri_transform.py:1483-1535 over utils/cholesky.py:21-131 against what the reference returned: the vectors themselves (the
    pivot sequence is part of the result), the tolerance-dependent count, the spin-dependent triple, and the defining property
    sum_L cderi_L cderi_L^T = eri to the tolerance."""
    g = golden("G20_convert_eri.npz")
    eri = g[name + "/eri_s4"]
    for tol in (1e-8, 1e-4):
        ref = g["%s/cderi_tol%g" % (name, tol)]
        out = Cd.convert_eri_to_gdf(eri, norb, tol=tol)
        c = out["j3c"]["0"]["0"]
        assert c.shape == ref.shape and np.abs(c - ref).max() < 1e-12
        assert np.array_equal(out["j3c-kptij"], g[name + "/kptij"])
        cp = np.asarray([R.pack_tril(x) for x in c])
        assert np.abs(cp.T @ cp - eri).max() < 10 * tol
    e3 = g[name + "/eri3_s4"]
    c3 = Cd.convert_eri_to_gdf(e3, norb, tol=1e-8)["j3c"]["0"]["0"]
    ref3 = g[name + "/cderi3"]
    assert c3.shape == ref3.shape and np.abs(c3 - ref3).max() < 1e-12
    pa = np.asarray([R.pack_tril(x) for x in c3[0]])
    pb = np.asarray([R.pack_tril(x) for x in c3[1]])
    assert np.abs(pa.T @ pa - e3[0]).max() < 1e-7 and np.abs(pb.T @ pb - e3[1]).max() < 1e-7 and np.abs(pa.T @ pb - e3[2]).max() < 1e-7
