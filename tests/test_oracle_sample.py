"""
CPU tests of the C oracle (oracle/eri_sample.c, the checker of the production-geometry parity tests): pinned against
the numpy restatement oracle/restate.py -- itself pinned by the reference-generated G6 goldens and the Random123
known answers -- and against the reference-generated ERI golden directly.
"""
import numpy as np
import pytest

from oracle import restate as R
from oracle import eri_sample as ES


def _philox_rows_numpy(seed, ki, kj, nao, L0, nL):
    """Rows [L0, L0 + nL) with oracle/restate.py's vectorised Philox4x32-10 and the element recipe of df_block_philox."""
    e = np.arange(L0 * nao * nao, (L0 + nL) * nao * nao, dtype=np.uint64)
    ctr = e >> np.uint64(1)
    n = len(e)
    r = R.philox4x32_10(ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), np.full(n, ki, np.uint64), np.full(n, kj, np.uint64),
                        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    odd = (e & np.uint64(1)).astype(bool)
    sc = 1.0 / np.sqrt(float(nao))
    re = (np.where(odd, r[2], r[0]).astype(np.float64) * 2.0 ** -31 - 1.0) * sc
    im = (np.where(odd, r[3], r[1]).astype(np.float64) * 2.0 ** -31 - 1.0) * sc
    return (re + 1j * im).reshape(nL, nao, nao)


@pytest.mark.parametrize("nao,L0,nL", [(10, 0, 7), (7, 2, 3), (200, 797, 3), (13, 5, 1)])
def test_philox_rows_bit_exact(nao, L0, nL):
    if (L0 + nL) * nao * nao < 100000:
        assert np.array_equal(_philox_rows_numpy(20241223, 3, 2, nao, L0, nL), R.df_block_philox(20241223, 3, 2, L0 + nL, nao)[L0:])
    got = ES.philox_rows(20241223, 3, 2, nao, L0, nL)
    assert np.array_equal(got, _philox_rows_numpy(20241223, 3, 2, nao, L0, nL))


def test_philox_key_and_block_ids():
    a = ES.philox_rows((5 << 32) | 9, 1, 2, 6, 0, 2)
    assert np.array_equal(a, R.df_block_philox((5 << 32) | 9, 1, 2, 2, 6))
    assert not np.array_equal(a, ES.philox_rows((5 << 32) | 9, 2, 1, 6, 0, 2))


@pytest.mark.parametrize("mesh,spin", [((2, 2, 1), 2), ((3, 1, 1), 1), ((4, 1, 1), 2), ((2, 2, 2), 1)])
def test_eri_sample_equals_restatement(mesh, spin):
    nk = int(np.prod(mesh))
    nao, naux, nemb = 10, 6, 8
    rng = np.random.default_rng(nk + spin)
    Ce = rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))
    ks = R.make_kpts_scaled(mesh)
    ref = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: R.df_block_philox(11, i, j, naux, nao), naux, nao, C_ao_eo=Ce)
    A = [0, 3, 4, 7]
    kls = [k for k, w in enumerate(R.get_weights_t_reversal(ks)) if w > 0]
    e, idx, planes = ES.eri_sample(mesh, 11, Ce / nk ** 0.75, naux, A, kls)
    assert e.shape == (spin * (spin + 1) // 2, 10, 10)
    assert np.abs(e - ref[:, idx][:, :, idx]).max() < 1e-12 * np.abs(ref).max()
    # a shard of kL only (what one rank of the MPI twin computes)
    part = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: R.df_block_philox(11, i, j, naux, nao), naux, nao, C_ao_eo=Ce,
                                  kL_list=kls[:1])
    e1, idx1, _ = ES.eri_sample(mesh, 11, Ce / nk ** 0.75, naux, A, kls[:1])
    assert np.abs(e1 - part[:, idx1][:, :, idx1]).max() < 1e-12 * np.abs(ref).max()


def test_planes_sample_rows_and_golden_plan():
    """Sampled auxiliary rows; the 6x6x6 / 4x4x4 visiting plans come from the reference-recorded G1 golden."""
    w, by = ES.plan_records((6, 6, 6), {0, 1})
    assert len(by[0]) + len(by[1]) == 112 + 108 and w[1] == 2 and w[0] == 1
    w4, by4 = ES.plan_records((4, 4, 4))
    assert sum(len(v) for v in by4.values()) == 1184
    # restated loop == golden plan
    wr, plan = R.tr_block_plan(R.make_kpts_scaled((4, 4, 1)), True)
    _, byg = ES.plan_records((4, 4, 1))
    assert [(p[0], p[1], p[2], int(p[4])) for p in plan] == [(r[0], r[1], r[2], r[4]) for k in sorted(byg) for r in byg[k]]
    mesh, nao, naux, nemb, spin = (2, 1, 1), 8, 9, 6, 2
    rng = np.random.default_rng(3)
    Ce = rng.standard_normal((spin, 2, nao, nemb)) + 1j * rng.standard_normal((spin, 2, nao, nemb))
    recs = [(1, 0, 1, 1, 1), (1, 1, 0, 0, 0)]
    A = [1, 2, 5]
    S = ES.half_sample_kL(4, recs, Ce, naux, A, L_list=[0, 4, 8])
    ref = np.zeros((spin, naux, nemb, nemb), dtype=complex)
    for (_, i, j, _, sym) in recs:
        Lij = R.transform_ao_to_emb(R.df_block_philox(4, i, j, naux, nao).reshape(naux, -1), Ce, i, j)
        ref += Lij + Lij.transpose(0, 1, 3, 2) if sym else Lij
    assert np.abs(S - ref[:, [0, 4, 8]][:, :, A][:, :, :, A]).max() < 1e-12
    P, idx = ES.planes_sample(S, A)
    assert np.abs(P - R.pack_tril(ref)[:, [0, 4, 8]][:, :, idx]).max() < 1e-12


def test_eri_sample_vs_reference_golden(golden):
    """The reference-generated G6 ERI (shim-driven get_emb_eri_fast_gdf) restricted to sampled pair columns needs the
    W0-derived blocks, not Philox -- so here the C transform is checked through its Philox-independent part:
    restate.transform_ao_to_emb on a Philox block is what G6 pins, and orc_half_sample must agree with it."""
    nao, naux, nemb = 12, 5, 12
    rng = np.random.default_rng(1)
    Ce = rng.standard_normal((1, 3, nao, nemb)) + 1j * rng.standard_normal((1, 3, nao, nemb))
    A = list(range(nemb))
    S = ES.half_sample_kL(8, [(0, 2, 1, 0, 0)], Ce, naux, A)
    ref = R.transform_ao_to_emb(R.df_block_philox(8, 2, 1, naux, nao).reshape(naux, -1), Ce, 2, 1)
    assert np.abs(S - ref).max() < 1e-12 * np.abs(ref).max()
