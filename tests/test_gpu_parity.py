"""
GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the
oracle (oracle/restate.py) on the same seeded inputs and against the golden fixtures captured from the
reference.  Tolerances are the north star's: k bookkeeping bit-exact (tests/test_host_abi.py),
<= 1e-10 on density matrices / bath projectors, <= 1e-8 max-abs on the transformed ERI.
"""
import ctypes as C
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import restate as R                      # the checker
from libdmet_preview_amd import synth


@pytest.fixture(scope="module")
def ctx():
    from libdmet_preview_amd import _lib
    return _lib.get_ctx()


def _lattice(mesh, nlo, val=None, virt=None, core=None):
    from libdmet_preview_amd.system.lattice import Lattice
    L = Lattice(int(nlo), mesh)
    if val is not None:
        L.val_idx, L.virt_idx, L.core_idx = list(val), list(virt or []), list(core or [])
    return L


class _Vcor(object):
    def __init__(self, v):
        self.value = v

    def islocal(self):
        return True

    def get(self, i=0, kspace=True):
        return self.value if (kspace or i == 0) else np.zeros_like(self.value)


# ---------------------------------------------------------------------------------------------
# K7: real contraction
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("N,K", [(78, 28), (130, 37), (257, 64), (300, 800), (1000, 123), (2080, 160)])
def test_dgemm_tn_acc(ctx, N, K):
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(N * 1000 + K)
    X = rng.standard_normal((K, N))
    C0 = rng.standard_normal((N, N))
    dX, dC = ctx.to_device(X), ctx.to_device(C0)
    ctx.check(lib.dmk_dgemm_tn_acc(ctx.h, N, K, 2.0, dX.ptr, dX.ptr, N, dC.ptr, N))
    ref = C0 + 2.0 * X.T @ X
    err = np.abs(dC.get() - ref).max()
    assert err < 1e-11 * max(1.0, np.abs(ref).max()), err


@pytest.mark.parametrize("saddr", ["0", "1"])
@pytest.mark.parametrize("N,K,rect", [(300, 800, False), (2080, 160, False), (1000, 120, True), (1282, 64, True)])
def test_dgemm_tn_acc_both_dma_address_forms(ctx, N, K, rect, saddr):
    """The LDS-DMA contraction in BOTH source-address forms -- per-lane 64-bit pointers and scalar row pointer + per-lane byte
    offset (DMK_DGEMM_SADDR; the library picks by N, so small shapes would otherwise only ever see one of them) -- for the
    symmetric launch (lower tile triangle + mirrored store) and the rectangular one, ragged edge tiles included."""
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(N + K)
    X = rng.standard_normal((K, N))
    Y = rng.standard_normal((K, N)) if rect else X
    C0 = rng.standard_normal((N, N))
    dX, dC = ctx.to_device(X), ctx.to_device(C0)
    dY = ctx.to_device(Y) if rect else dX
    os.environ["DMK_DGEMM_SADDR"] = saddr
    try:
        ctx.check(lib.dmk_dgemm_tn_acc(ctx.h, N, K, 0.5, dX.ptr, dY.ptr, N, dC.ptr, N))
        got = dC.get()
    finally:
        del os.environ["DMK_DGEMM_SADDR"]
    ref = C0 + 0.5 * X.T @ Y
    assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def test_dgemm_tn_is_transpose_detecting(ctx):
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(5)
    K, N = 20, 150
    X, Y = rng.standard_normal((K, N)), rng.standard_normal((K, N))
    dX, dY, dC = ctx.to_device(X), ctx.to_device(Y), ctx.zeros((N, N), np.float64)
    from libdmet_preview_amd._lib import lib as L
    # X != Y path through the ERI pipeline is covered by the UHF cases below; here use planes API
    ctx.check(L.dmk_dgemm_tn_acc(ctx.h, N, K, 1.0, dX.ptr, dY.ptr, N, dC.ptr, N))
    assert np.abs(dC.get() - X.T @ Y).max() < 1e-12 * K


# ---------------------------------------------------------------------------------------------
# generic complex batched product / basis algebra (a9, a10)
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("opA,M,N,K,nb", [("C", 40, 24, 4320, 2), ("T", 72, 72, 8640, 1), ("C", 256, 256, 4096, 2), ("C", 16, 16, 4099, 3)])
def test_zgemm_batched_long_k_split(ctx, opA, M, N, K, nb):
    """The one-body folds (1/nk) sum_k B_k^H T_k are ONE product with K = nk * nlo (slater.py:682-704): few small output matrices
    and a very long K, which dmk_zgemm_batched cuts into equal K chunks (a batch of partial products summed in a fixed order).
    K = 4099 is prime: no equal split exists and the plain launch answers."""
    from libdmet_preview_amd.basis_transform.make_basis import _bgemm
    rng = np.random.default_rng(K + M)
    a = rng.standard_normal((nb, K, M)) + 1j * rng.standard_normal((nb, K, M))
    b = rng.standard_normal((nb, K, N)) + 1j * rng.standard_normal((nb, K, N))
    ref = np.einsum("bkm,bkn->bmn", a.conj() if opA == "C" else a, b)
    got = _bgemm(opA, "N", a, b)
    assert np.abs(got - ref).max() < 1e-12 * K


@pytest.mark.parametrize("opA", "NTC")
@pytest.mark.parametrize("opB", "NTC")
def test_zgemm_batched_ops(ctx, opA, opB):
    from libdmet_preview_amd.basis_transform.make_basis import _bgemm
    rng = np.random.default_rng(ord(opA) * 7 + ord(opB))
    nb, M, N, K = 3, 37, 21, 50
    a = rng.standard_normal((nb, M, K) if opA == "N" else (nb, K, M)) + 1j * rng.standard_normal((nb, M, K) if opA == "N" else (nb, K, M))
    b = rng.standard_normal((nb, K, N) if opB == "N" else (nb, N, K)) + 1j * rng.standard_normal((nb, K, N) if opB == "N" else (nb, N, K))
    op = {"N": lambda x: x, "T": lambda x: x.transpose(0, 2, 1), "C": lambda x: x.conj().transpose(0, 2, 1)}
    ref = np.einsum("bmk,bkn->bmn", op[opA](a), op[opB](b))
    got = _bgemm(opA, opB, a, b)
    assert np.abs(got - ref).max() < 1e-12 * K


def test_G5_basis_algebra(ctx, golden):
    from libdmet_preview_amd.basis_transform import make_basis as mb
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = golden("G5_basis.npz")
    bk = et.get_basis_k(g["basis"], g["phase_R2k"])
    assert np.abs(bk - g["basis_k"]).max() < 1e-12
    assert np.abs(mb.multiply_basis(g["C_ao_lo"], g["basis_k"]) - g["multiply_basis"]).max() < 1e-12
    r = mb.multiply_basis(g["C_ao_lo"][0], g["basis_k"][0])
    assert r.shape == g["multiply_basis_rhf"].shape and np.abs(r - g["multiply_basis_rhf"]).max() < 1e-12
    assert np.abs(mb.multiply_basis(g["C_ao_lo"][0], g["basis_k"]) - g["multiply_basis_mixed"]).max() < 1e-12
    assert np.abs(mb.transform_h1_to_lo(g["h_ao"], g["C_ao_lo"]) - g["h1_to_lo"]).max() < 1e-11
    assert np.abs(mb.transform_h1_to_lo(g["h_ao"][0], g["C_ao_lo"][0]) - g["h1_to_lo_rhf"]).max() < 1e-11
    assert np.abs(mb.transform_rdm1_to_lo(g["h_ao"], g["C_ao_lo"], g["S_ao"]) - g["rdm1_to_lo"]).max() < 1e-11
    assert np.abs(mb.transform_rdm1_to_ao(g["h1_to_lo"], g["C_ao_lo"]) - g["rdm1_to_ao"]).max() < 1e-11
    with pytest.raises(ValueError):
        mb.multiply_basis(np.zeros((2, 2)), np.zeros((2, 2, 2)))


# ---------------------------------------------------------------------------------------------
# K3: folds
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("tag", ["6x1x1", "4x4x1", "2x3x2", "6x6x6"])
def test_G2_folds(ctx, golden, tag):
    from libdmet_preview_amd.system import fourier
    from libdmet_preview_amd.utils import logger as log
    g = golden("G2_fourier.npz")
    mesh = tuple(int(x) for x in tag.split("x"))
    assert np.abs(fourier.FFTtoK(g[tag + "/A_R"], mesh) - g[tag + "/FFTtoK"]).max() < 1e-12
    assert np.abs(fourier.FFTtoT(g[tag + "/FFTtoK"], mesh) - g[tag + "/FFTtoT_of_FFTtoK"]).max() < 1e-13
    n0 = len(log.warnings_seen)
    out = fourier.FFTtoT(g[tag + "/Z_k"], mesh)          # generic complex input: imaginary part warned, real returned
    assert np.abs(out - g[tag + "/ifftn_full"].real).max() < 1e-13
    assert len(log.warnings_seen) == n0 + 1 and "imaginary" in log.warnings_seen[-1]
    assert np.abs(fourier.R2k(g[tag + "/S_R"], mesh) - g[tag + "/R2k_spin"]).max() < 1e-12
    assert np.abs(fourier.k2R(g[tag + "/R2k_spin"], mesh) - g[tag + "/k2R_spin"]).max() < 1e-13
    with pytest.raises(ValueError):
        fourier.R2k(np.zeros((2, 2)), mesh)


def test_fold_roundtrip_full_size(ctx):
    """Size-independent property at the C5 shape: k2R(R2k(x)) == x and Parseval."""
    from libdmet_preview_amd.system import fourier
    mesh, n = (6, 6, 6), 200
    rng = np.random.default_rng(1)
    x = rng.standard_normal((216, n * n))
    d = ctx.to_device(x)
    dk = fourier.fold_R2k_dev(d, mesh, 1, n * n)
    back = fourier.fold_k2R_dev(dk, mesh, 1, n * n).get().reshape(x.shape)
    assert np.abs(back - x).max() < 1e-12
    k = dk.get().reshape(216, -1)
    assert abs((np.abs(k) ** 2).sum() / 216 - (x ** 2).sum()) < 1e-8 * (x ** 2).sum()


def test_fold_k_subset_partial_sums(ctx):
    """Multi-GPU decomposition of k2R: partial folds over k shards add up to the full fold."""
    from libdmet_preview_amd.system import fourier
    mesh = (4, 4, 1)
    rng = np.random.default_rng(2)
    z = rng.standard_normal((16, 9)) + 1j * rng.standard_normal((16, 9))
    full = fourier.fold_k2R_dev(ctx.to_device(z), mesh, 1, 9).get()
    acc = np.zeros_like(full)
    for sub in ([0, 1, 2, 3, 12, 13, 14, 15], [4, 5, 6, 7, 8, 9, 10, 11]):
        acc += fourier.fold_k2R_dev(ctx.to_device(z[sub]), mesh, 1, 9, k_subset=sub).get()
    assert np.abs(acc - full).max() < 1e-14


# ---------------------------------------------------------------------------------------------
# K10: Philox generator
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("naux,nao,i,j", [(4, 5, 3, 7), (7, 3, 0, 215), (28, 10, 1, 2)])
def test_philox_block_bit_exact(ctx, naux, nao, i, j):
    from libdmet_preview_amd.basis_transform.eri_transform import GDFPhilox
    p = GDFPhilox(np.zeros((1, 3)), naux, nao, seed=0x1234567890ABCDEF)
    buf = ctx.empty((naux, nao, nao), np.complex128)
    p.load_block(ctx, i, j, buf)
    ref = R.df_block_philox(0x1234567890ABCDEF, i, j, naux, nao)
    assert np.array_equal(buf.get(), ref)


# ---------------------------------------------------------------------------------------------
# K6 + K7: ERI transform
# ---------------------------------------------------------------------------------------------

G6_CASES = [("m311", 1), ("m311", 2), ("m411", 1), ("m411", 2), ("m231", 1), ("m231", 2), ("m222", 1), ("m222", 2),
            ("mid411", 1), ("mid411", 2), ("mid221", 1)]


def _g6_setup(g, name, spin):
    from libdmet_preview_amd.system.lattice import _UnitCell
    from libdmet_preview_amd.basis_transform.eri_transform import GDFMemory
    from libdmet_preview_amd.system import fourier
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    W0 = g[name + "/W0"]
    naux, _, nao = W0.shape[:3]
    cell = _UnitCell(nao)
    kpts = cell.get_abs_kpts(np.pad(fourier.make_kpts_scaled(mesh), ((0, 0), (0, 0))))
    blocks = synth.df_blocks_from_W0(W0, mesh)
    mydf = GDFMemory(kpts, blocks, naux=naux)
    st = "%s/s%d" % (name, spin)
    return mesh, cell, mydf, st


@pytest.mark.parametrize("name,spin", G6_CASES)
def test_G6_eri_golden(ctx, golden, name, spin):
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = golden("G6_eri.npz")
    mesh, cell, mydf, st = _g6_setup(g, name, spin)
    C, basis = g[st + "/C_ao_lo"], g[st + "/basis"]
    scale = max(1.0, np.abs(g[st + "/eri_tr"]).max())
    tol = 1e-8          # max-abs, north star; the fixtures' |eri|max is O(1e2..1e4) so this is <= 1e-10 relative
    for tr in (True, False):
        e = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis, t_reversal_symm=tr)
        ref = g[st + "/eri_%s" % ("tr" if tr else "notr")]
        assert e.shape == ref.shape and e.dtype == np.float64
        assert np.abs(e - ref).max() < tol, (np.abs(e - ref).max(), scale)
        assert np.abs(e - g[st + "/eri_identity"]).max() < tol
    e1 = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis, symmetry=1)
    assert e1.shape == g[st + "/eri_s1"].shape and np.abs(e1 - g[st + "/eri_s1"]).max() < tol
    if spin == 1:
        e8 = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis, symmetry=8)
        assert e8.shape == g[st + "/eri_s8"].shape and np.abs(e8 - g[st + "/eri_s8"]).max() < tol
    else:
        with pytest.raises(ValueError):
            et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis, symmetry=8)
    eu = et.get_unit_eri(cell, mydf, C_ao_lo=C)
    assert np.abs(eu - g[st + "/eri_unit"]).max() < tol
    Ck = R.multiply_basis(C, R.get_basis_k(basis, R.get_phase_R2k(mesh, R.make_kpts_scaled(mesh))))
    ec = et.get_emb_eri_fast_gdf(cell, mydf, C_ao_eo=Ck)
    assert np.abs(ec - g[st + "/eri_C_ao_eo"]).max() < tol
    with pytest.raises(ValueError):
        et.get_emb_eri_fast_gdf(cell, mydf, C_ao_lo=C, C_ao_eo=Ck)
    with pytest.raises(ValueError):
        et.get_emb_eri(cell, object(), C_ao_lo=C, basis=basis)
    with pytest.raises(NotImplementedError):
        et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis, t_reversal_symm=False, incore=False)


def test_eri_outcore_matches_incore(ctx, golden, tmp_path):
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = golden("G6_eri.npz")
    mesh, cell, mydf, st = _g6_setup(g, "m231", 2)
    C, basis = g[st + "/C_ao_lo"], g[st + "/basis"]
    inc = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis)
    out = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis, incore=False, fout=str(tmp_path / "H2"))
    assert np.abs(np.asarray(out["ccdd"]) - inc[[0, 2, 1]]).max() < 1e-12      # (aa, bb, ab) on disk
    # slab by slab like the reference (eri_transform.py:486-521; its own test shrinks ERI_SLICE, test_eri_transform_gdf.py:62),
    # with a plane stack of two slots so that the slabs are accumulated over several flushes
    old = et.ERI_SLICE, et.OUTCORE_MAX_SLOTS
    try:
        et.ERI_SLICE, et.OUTCORE_MAX_SLOTS = 4, 2
        out2 = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis, incore=False, fout=str(tmp_path / "H2b.npy"))
        assert np.abs(np.asarray(out2["ccdd"]) - inc[[0, 2, 1]]).max() < 1e-12
        mesh1, cell1, mydf1, st1 = _g6_setup(g, "mid221", 1)
        inc1 = et.get_emb_eri(cell1, mydf1, C_ao_lo=g[st1 + "/C_ao_lo"], basis=g[st1 + "/basis"])
        out1 = et.get_emb_eri(cell1, mydf1, C_ao_lo=g[st1 + "/C_ao_lo"], basis=g[st1 + "/basis"], incore=False,
                              fout=str(tmp_path / "H2c"))
        assert np.asarray(out1["ccdd"]).shape == inc1.shape and np.abs(np.asarray(out1["ccdd"]) - inc1).max() < 1e-12 * max(1.0, np.abs(inc1).max())
    finally:
        et.ERI_SLICE, et.OUTCORE_MAX_SLOTS = old


def test_eri_sharded_sum(ctx, golden):
    """kL shards (eri_transform_mpi.py:151-157) accumulate into the same ERI; their sum is the serial result."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = golden("G6_eri.npz")
    mesh, cell, mydf, st = _g6_setup(g, "m222", 2)
    nao = cell.nao_nr()
    C_dev = et.make_C_ao_emb_dev(ctx, mesh, C_ao_lo=g[st + "/C_ao_lo"], basis=g[st + "/basis"])
    spin, _, _, nemb = C_dev.shape
    npair = nemb * (nemb + 1) // 2
    total = np.zeros((3, npair, npair))
    for kl in et.assign_workload(mesh, 3):
        eri_dev = ctx.zeros((3, npair, npair), np.float64)
        eng = et.EriEngine(ctx, mesh, nao, mydf.naux, nemb, spin, C_dev, eri_dev, True)
        eng.run(mydf, kL_list=kl)
        total += eri_dev.get()
        eng.close()
    assert np.abs(total - g[st + "/eri_tr"]).max() < 1e-8


def test_eri_philox_vs_oracle_midsize(ctx):
    """Procedural blocks (TR mode only) at a size the oracle finishes in seconds; UHF."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd.system.lattice import _UnitCell
    mesh, nao, naux, nemb, spin, seed = (2, 2, 2), 24, 40, 20, 2, 777
    nk = 8
    ks = R.make_kpts_scaled(mesh)
    cell = _UnitCell(nao)
    mydf = et.GDFPhilox(cell.get_abs_kpts(ks), naux, nao, seed=seed)
    rng = np.random.default_rng(9)
    C = synth.make_C_ao_lo(mesh, nao, nao, spin=spin, seed=4)
    basis = rng.standard_normal((spin, nk, nao, nemb))
    got = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis)
    ref = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: R.df_block_philox(seed, i, j, naux, nao), naux, nao,
                                 C_ao_lo=C, basis=basis)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("nemb,spin", [(40, 2), (56, 1), (33, 2)])
def test_eri_flat_step2_vs_oracle(ctx, nemb, spin):
    """General-nemb step 2 (flattened hot kernel + fold pass) against the oracle, and against the tiled kernel."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd.system.lattice import _UnitCell
    mesh, nao, naux, seed = (2, 2, 1), 24, 40, 4242
    nk = 4
    ks = R.make_kpts_scaled(mesh)
    cell = _UnitCell(nao)
    mydf = et.GDFPhilox(cell.get_abs_kpts(ks), naux, nao, seed=seed)
    rng = np.random.default_rng(nemb)
    C = synth.make_C_ao_lo(mesh, nao, nao, spin=spin, seed=4)
    basis = rng.standard_normal((spin, nk, nao, nemb))
    ref = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: R.df_block_philox(seed, i, j, naux, nao), naux, nao,
                                 C_ao_lo=C, basis=basis)
    tiled = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis)                  # default: tiled two-segment kernel
    os.environ["DMK_ERI_FLAT2"] = "1"                                            # opt-in variant
    try:
        got = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis)
        again = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis)
    finally:
        del os.environ["DMK_ERI_FLAT2"]
    assert np.abs(got - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())
    assert np.abs(tiled - ref).max() < 1e-8 * max(1.0, np.abs(ref).max())
    assert np.abs(got - tiled).max() < 1e-11 * max(1.0, np.abs(ref).max())
    assert np.array_equal(got, again)                        # one writer per plane element: bit-reproducible


def test_eri_properties_large(ctx):
    """Size-independent properties at a size beyond the oracle's reach (nao 104, naux 64, nemb 136 = C4 tile
    shapes, mesh 2x2x1): (ab|cd) = (cd|ab) for same-spin blocks and shard additivity."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    mesh, nao, naux, nemb, spin = (2, 2, 1), 104, 64, 136, 2
    ks = R.make_kpts_scaled(mesh)
    from libdmet_preview_amd.system.lattice import _UnitCell
    cell = _UnitCell(nao)
    mydf = et.GDFPhilox(cell.get_abs_kpts(ks), naux, nao, seed=11)
    rng = np.random.default_rng(10)
    C = synth.make_C_ao_lo(mesh, nao, nao, spin=spin, seed=6)
    basis = rng.standard_normal((spin, 4, nao, nemb)) / np.sqrt(nao)
    e = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=basis)
    assert np.abs(e[0] - e[0].T).max() < 1e-9 * np.abs(e[0]).max()
    assert np.abs(e[2] - e[2].T).max() < 1e-9 * np.abs(e[2]).max()
    # half transform of one block against the oracle's einsum
    blk = R.df_block_philox(11, 1, 3, naux, nao)
    Cemb = R.make_C_ao_emb(mesh, ks, C_ao_lo=C, basis=basis)
    ref = R.pack_tril(R.transform_ao_to_emb(blk.reshape(naux, -1), Cemb, 1, 3))
    C_dev = ctx.to_device(Cemb)
    npair = nemb * (nemb + 1) // 2
    eri_dev = ctx.zeros((3, npair, npair), np.float64)
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    from libdmet_preview_amd._lib import lib
    ctx.check(lib.dmk_eri_begin_kL(eng.h, 0))
    buf = ctx.to_device(blk)
    ctx.check(lib.dmk_eri_push_block(eng.h, 1, 3, 0, buf.ptr))
    planes = eng.planes().get()
    got = planes[:, 0] + 1j * planes[:, 1]
    assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
    ctx.check(lib.dmk_eri_end_kL(eng.h, 1))
    eng.close()


# ---------------------------------------------------------------------------------------------
# K1 / K2: eigensolver, density, HF
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("layout", ["tiles", "rows", "tiles_no_wy"])
@pytest.mark.parametrize("n,batch", [(65, 3), (112, 2), (113, 2), (136, 5), (199, 2), (200, 7)])
def test_eigh_resident_layouts(ctx, monkeypatch, n, batch, layout):
    """The two CU-resident Householder layouts of the complex eigensolver for 64 < n <= 200 -- 16 x 16 tiles updated on the matrix
    cores (default) and rows across the lanes (DMK_EIGH_TILES=0) -- against LAPACK, including a repeated level, a block-diagonal
    matrix (zero reflectors inside the sweep) and the shared real shift matrix."""
    from libdmet_preview_amd._lib import lib
    monkeypatch.delenv("DMK_EIGH_TILES", raising=False)
    monkeypatch.delenv("DMK_EIGH_WY", raising=False)
    if layout == "rows":
        monkeypatch.setenv("DMK_EIGH_TILES", "0")
    elif layout == "tiles_no_wy":
        monkeypatch.setenv("DMK_EIGH_WY", "0")              # reflectors one at a time (K1c) instead of the blocked WY form
    rng = np.random.default_rng(n * 10 + batch)
    A = rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))
    A = A + A.conj().transpose(0, 2, 1)
    A[0, :, 5] = A[0, :, 4]                                # a repeated row / column pair: one (near-)zero level split
    A[0, 5, :] = A[0, 4, :]
    A[0] = 0.5 * (A[0] + A[0].conj().T)
    A[1, n // 2:, : n // 2] = 0.0                          # block diagonal
    A[1, : n // 2, n // 2:] = 0.0
    add = rng.standard_normal((n, n))
    add = add + add.T
    dA, dadd = ctx.to_device(A, np.complex128), ctx.to_device(add[None])
    dw, dV = ctx.empty((batch, n), np.float64), ctx.empty((batch, n, n), np.complex128)
    ctx.check(lib.dmk_eigh_batched(ctx.h, n, batch, dA.ptr, dadd.ptr, batch, dw.ptr, dV.ptr))
    w, V = dw.get(), dV.get()
    for b in range(batch):
        M = A[b] + add
        wr = np.linalg.eigvalsh(M)
        scale = np.abs(wr).max()
        assert np.abs(w[b] - wr).max() < 1e-12 * scale
        Vb = V[b]                                           # rows = eigenvectors
        assert np.abs(Vb.conj() @ Vb.T - np.eye(n)).max() < 1e-12
        assert np.abs(Vb.conj() @ M @ Vb.T - np.diag(w[b])).max() < 1e-11 * scale


def test_eigh_resident_random_shapes(ctx):
    """Randomised shapes 65 <= n <= 200 through the CU-resident eigensolver path: generic, massively degenerate (low rank + shift),
    graded over twelve decades and banded matrices against LAPACK (tools/eigh_stress.py runs the longer version)."""
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(2026)
    for trial in range(24):
        n, batch, kind = int(rng.integers(65, 201)), int(rng.integers(1, 5)), trial % 4
        A = rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))
        A = A + A.conj().transpose(0, 2, 1)
        if kind == 1:
            u = rng.standard_normal((batch, n, 3)) + 1j * rng.standard_normal((batch, n, 3))
            A = u @ u.conj().transpose(0, 2, 1) + 2.0 * np.eye(n)
        elif kind == 2:
            s = np.logspace(0, -12, n)
            A = A * s[None, :, None] * s[None, None, :]
        elif kind == 3:
            A = np.triu(np.tril(A.real, 3), -3).astype(complex)
            A = A + A.conj().transpose(0, 2, 1)
        dA = ctx.to_device(A, np.complex128)
        dw, dV = ctx.empty((batch, n), np.float64), ctx.empty((batch, n, n), np.complex128)
        ctx.check(lib.dmk_eigh_batched(ctx.h, n, batch, dA.ptr, None, 0, dw.ptr, dV.ptr))
        w, V = dw.get(), dV.get()
        for b in range(batch):
            wr = np.linalg.eigvalsh(A[b])
            sc = np.abs(wr).max()
            assert np.abs(w[b] - wr).max() < 1e-12 * sc, (trial, n, kind)
            assert np.abs(V[b].conj() @ V[b].T - np.eye(n)).max() < 1e-11, (trial, n, kind)
            assert np.abs(V[b].conj() @ A[b] @ V[b].T - np.diag(w[b])).max() < 1e-11 * sc, (trial, n, kind)


@pytest.mark.parametrize("n,batch", [(1, 3), (2, 5), (3, 4), (10, 6), (33, 3), (64, 2), (65, 2), (200, 3)])
def test_eigh_batched_random(ctx, n, batch):
    from libdmet_preview_amd.routine import mfd
    rng = np.random.default_rng(n)
    A = rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))
    A = A + A.conj().transpose(0, 2, 1)
    dw, dVt = mfd.eigh_dev(ctx, ctx.to_device(A), n, batch)
    w, Vt = dw.get(), dVt.get()
    for b in range(batch):
        wr = np.linalg.eigvalsh(A[b])
        scale = max(1.0, np.abs(wr).max())
        assert np.abs(w[b] - wr).max() < 1e-12 * scale * max(1, n / 10)
        V = Vt[b].T                                              # columns = eigenvectors
        assert np.abs(V.conj().T @ V - np.eye(n)).max() < 1e-12 * max(1, n / 10)
        assert np.abs(A[b] @ V - V * w[b]).max() < 1e-11 * scale * max(1, n / 10)


def test_eigh_degenerate_and_lower_triangle(ctx):
    from libdmet_preview_amd.routine import mfd
    n = 12
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    lam = np.array([-1.0] * 4 + [0.5] * 5 + [2.0] * 3)
    A = (Q * lam) @ Q.conj().T
    A = 0.5 * (A + A.conj().T)
    junk = A + np.triu(rng.standard_normal((n, n)), 1)          # garbage above the diagonal must be ignored
    batch = np.stack([A, np.eye(n), junk, np.zeros((n, n))]).astype(np.complex128)
    dw, dVt = mfd.eigh_dev(ctx, ctx.to_device(batch), n, 4)
    w, Vt = dw.get(), dVt.get()
    assert np.abs(w[0] - lam).max() < 1e-13 and np.abs(w[2] - lam).max() < 1e-13
    assert np.abs(w[1] - 1.0).max() == 0.0 and np.abs(w[3]).max() == 0.0
    V = Vt[0].T
    P = V[:, :4] @ V[:, :4].conj().T
    assert np.abs(P - Q[:, :4] @ Q[:, :4].conj().T).max() < 1e-12


def _wilkinson(m):
    """W_{2m+1}^+ : diag |i - m|, unit couplings -- its upper eigenvalues come in pairs that agree to 1e-14."""
    d = np.abs(np.arange(2 * m + 1) - m).astype(float)
    return np.diag(d) + np.diag(np.ones(2 * m), 1) + np.diag(np.ones(2 * m), -1)


def _glued(blocks, glue):
    n = sum(b.shape[0] for b in blocks)
    T = np.zeros((n, n))
    o = 0
    for b in blocks:
        k = b.shape[0]
        T[o:o + k, o:o + k] = b
        if o + k < n:
            T[o + k - 1, o + k] = T[o + k, o + k - 1] = glue
        o += k
    return T


@pytest.mark.parametrize("inject", [0, 3])
def test_eigh_clustered_tridiagonal(ctx, inject, monkeypatch):
    """Exactly degenerate and tightly clustered spectra INSIDE one unreduced tridiagonal block (Wilkinson W21+, nine of
    them glued by 1e-9: clusters of nine eigenvalues within 1e-9), where inverse iteration from independent starts does
    not deliver orthogonal vectors by itself.  Every vector must pass the device's acceptance test (residual, norm);
    with DMK_EIGH_INJECT every third vector is declared failed and rebuilt by the dstein-style repair path."""
    from libdmet_preview_amd.routine import mfd
    if inject:
        monkeypatch.setenv("DMK_EIGH_INJECT", str(inject))
    mats = [_wilkinson(10), _glued([_wilkinson(10)] * 9, 1e-9), _glued([_wilkinson(10)] * 4, 1e-13),
            _glued([np.diag([1.0, 1.0, 1.0]) + 1e-7 * (np.eye(3, k=1) + np.eye(3, k=-1))] * 5, 1e-7)]
    for T in mats:
        n = T.shape[0]
        dw, dVt = mfd.eigh_dev(ctx, ctx.to_device(T[None].astype(np.complex128)), n, 1)
        w, V = dw.get()[0], dVt.get()[0].T
        wr = np.linalg.eigvalsh(T)
        scale = np.abs(wr).max()
        assert np.abs(w - wr).max() < 1e-13 * scale * max(1, n / 10)
        assert np.abs(V.conj().T @ V - np.eye(n)).max() < 1e-12 * max(1, n / 10)
        assert np.abs(T @ V - V * w).max() < 1e-12 * scale * max(1, n / 10)


def test_eigh_above_1024(ctx):
    """One real symmetric 1100 x 1100 matrix: the R = 32 column-slice variant behind the eig-flavoured bath of larger
    model lattices (routine/slater.py:224-318 has no size limit; here n <= 2000, one workgroup per matrix)."""
    from libdmet_preview_amd._lib import lib
    n = 1100
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    A = A + A.T
    dA = ctx.to_device(A[None])
    dw, dV = ctx.empty((1, n), np.float64), ctx.empty((1, n, n), np.float64)
    ctx.check(lib.dmk_eigh_batched_real(ctx.h, n, 1, dA.ptr, dw.ptr, dV.ptr))
    w, V = dw.get()[0], dV.get()[0].T
    wr = np.linalg.eigvalsh(A)
    assert np.abs(w - wr).max() < 1e-11 * np.abs(wr).max()
    assert np.abs(V.T @ V - np.eye(n)).max() < 1e-11
    assert np.abs(A @ V - V * w).max() < 1e-10 * np.abs(wr).max()


def test_eigh_reports_garbage(ctx):
    """A NaN in the input must surface as an error, not as a silently wrong basis."""
    from libdmet_preview_amd.routine import mfd
    from libdmet_preview_amd._lib import DmkError
    A = np.eye(8, dtype=np.complex128) + 0.1 * np.eye(8, k=1) + 0.1 * np.eye(8, k=-1)
    A[3, 3] = np.nan
    with pytest.raises(DmkError):
        mfd.eigh_dev(ctx, ctx.to_device(A[None]), 8, 1)


def test_Diag_wrappers_vs_oracle(ctx):
    from libdmet_preview_amd.routine import mfd
    mesh, nlo = (3, 2, 1), 5
    FR = synth.make_fock_R(mesh, nlo, spin=2, seed=3)
    Fk = R.R2k(FR, mesh)
    v = np.zeros((2, nlo, nlo))
    v[0], v[1] = np.diag(np.arange(nlo) * 0.1), np.diag(np.arange(nlo) * -0.05)
    L = _lattice(mesh, nlo)
    ew, ev = mfd.DiagRHF(Fk[0], _Vcor(v))
    ewr, _ = R.DiagRHF(Fk[0], v)
    assert ew.shape == ewr.shape and np.abs(ew - ewr).max() < 1e-12
    for k in range(6):
        assert np.abs((Fk[0, k] + v[0]) @ ev[k] - ev[k] * ew[k]).max() < 1e-11
    ew2, ev2 = mfd.DiagUHF(Fk, _Vcor(v))
    ewr2, _ = R.DiagUHF(Fk, v)
    assert ew2.shape == (2, 6, nlo) and np.abs(ew2 - ewr2).max() < 1e-12
    ew3, ev3 = mfd.DiagUHF_symm(Fk, _Vcor(v), L)
    ewr3, evr3 = R.DiagUHF_symm(Fk, v, mesh)
    assert np.abs(ew3 - ewr3).max() < 1e-12
    for i in range(6):
        ni = L.neg(i)
        if ni < i:
            assert np.array_equal(ev3[:, i], ev3[:, ni].conj())
    ew4, ev4 = mfd.DiagRHF_symm(Fk[0], None, L)
    assert np.abs(ew4 - R.DiagRHF_symm(Fk[0], None, mesh)[0]).max() < 1e-12
    ew5, _ = mfd.DiagUHF(Fk[0], None)                            # 3-d Fock promoted to two identical spins
    assert np.abs(ew5[0] - ew5[1]).max() == 0.0


class _VcorK(object):
    """k-dependent Hermitian correlation potential (routine/vcor.py:36-47: value.ndim == 4)."""
    def __init__(self, value):
        self.value, self.is_vcor_kpts = value, True

    def islocal(self):
        return False

    is_local = islocal

    def get(self, i=0, kspace=True):
        return self.value[i]


def test_k_dependent_complex_vcor(ctx):
    """vcor.get(k, True) differing from k to k and complex Hermitian (reference: mfd.py:42-45, 77-83 add it per block; the
    energy uses sum_k tr(v_k rho_k), mfd.py:372-392): added to the Fock batch before the upload."""
    from libdmet_preview_amd.routine import mfd
    mesh, nlo, spin = (3, 2, 1), 6, 2
    nk = 6
    FR = synth.make_fock_R(mesh, nlo, spin=spin, seed=31)
    Fk = R.R2k(FR, mesh)
    rng = np.random.default_rng(5)
    vR = 0.2 * synth.make_fock_R(mesh, nlo, spin=spin, seed=32)
    vk = R.R2k(vR, mesh).transpose(1, 0, 2, 3)                     # (nk, spin, n, n): Hermitian, TR symmetric, complex
    vc = _VcorK(np.ascontiguousarray(vk))
    ew, ev = mfd.DiagUHF(Fk, vc)
    for s in range(spin):
        for k in range(nk):
            H = Fk[s, k] + vk[k, s]
            assert np.abs(ew[s, k] - np.linalg.eigvalsh(H)).max() < 1e-12
            assert np.abs(H @ ev[s, k] - ev[s, k] * ew[s, k]).max() < 1e-11
    L = _lattice(mesh, nlo)
    L.set_Ham_lo(fock_lo_R=FR, hcore_lo_R=FR)
    ew2, _ = mfd.DiagUHF_symm(Fk, vc, L)
    assert np.abs(ew2 - ew).max() < 1e-12
    rhoT, mu, E, res = mfd.HF(L, vc, 0.5, False, ires=True)
    rho_k = res["rho_k"]
    E0 = 0.5 * np.sum((FR + FR) * rhoT)
    assert abs(res["E0"] - E0) < 1e-10
    assert abs(E - (E0 + 0.5 * np.einsum("kspq,skqp->", vk, rho_k)).real) < 1e-10
    for s in range(spin):
        for k in range(nk):
            assert np.abs(np.linalg.eigvalsh(rho_k[s, k]).round(8) % 1.0).max() < 1e-7       # projector at T = 0


G3_CASES = ["rhf_611", "rhf_661", "uhf_411", "uhf_222_T", "rhf_331_T", "uhf_231_sz", "rhf_444"]


@pytest.mark.parametrize("name", G3_CASES)
@pytest.mark.parametrize("symm", [False, True])
def test_G3_HF_golden(ctx, golden, name, symm):
    from libdmet_preview_amd.routine import mfd
    g = golden("G3_meanfield.npz")
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    FR, H1R, v = g[name + "/Fock_R"], g[name + "/H1_R"], g[name + "/vcor"]
    spin, nlo = FR.shape[0], FR.shape[-1]
    L = _lattice(mesh, nlo)
    if spin == 1:
        L.set_Ham_lo(fock_lo_R=FR[0], hcore_lo_R=H1R[0])
    else:
        L.set_Ham_lo(fock_lo_R=FR, hcore_lo_R=H1R)
    filling = g[name + "/filling"]
    filling = float(filling) if filling.ndim == 0 else tuple(filling)
    rhoT, mu, E, res = mfd.HF(L, _Vcor(v), filling, spin == 1, beta=float(g[name + "/beta"]), ires=True, symm=symm)
    t = name + ("_symm" if symm else "")
    assert rhoT.shape == g[t + "/rhoT"].shape
    assert np.abs(res["e"] - g[t + "/ew"]).max() < 1e-11
    assert np.abs(res["mo_occ"] - g[t + "/mo_occ"]).max() < 1e-9
    assert np.abs(np.asarray(mu) - g[t + "/mu"]).max() < 1e-9
    assert np.abs(res["rho_k"] - g[t + "/rho_k"]).max() < 1e-10
    assert np.abs(rhoT - g[t + "/rhoT"]).max() < 1e-10
    assert abs(E - float(g[t + "/E"])) < 1e-9
    for s in range(spin):                                       # eigenpairs: residual, never raw vectors
        for k in range(rhoT.shape[1]):
            Fk = L.getFock(True)[k] if spin == 1 else L.getFock(True)[s, k]
            ev = res["coef"][s, k]
            assert np.abs((Fk + v[s]) @ ev - ev * res["e"][s, k]).max() < 1e-10


def test_HF_c5_shape_vs_oracle(ctx):
    """Config-C5 matrix size (nlo = 200) on a small mesh: eigenvalues, occupations, rho_R vs the oracle."""
    from libdmet_preview_amd.routine import mfd
    mesh, nlo = (2, 2, 1), 200
    FR = synth.make_fock_R(mesh, nlo, spin=2, seed=21)
    L = _lattice(mesh, nlo)
    L.set_Ham_lo(fock_lo_R=FR)
    v = np.zeros((2, nlo, nlo))
    rhoT, mu, E, res = mfd.HF(L, _Vcor(v), 0.5, False, ires=True)
    Fk = R.R2k(FR, mesh)
    rr, mur, Er, resr = R.HF(mesh, Fk, FR, FR, v, 0.5, False, ires=True)
    assert np.abs(res["e"] - resr["e"]).max() < 1e-11
    assert np.array_equal(res["mo_occ"], resr["mo_occ"])
    assert np.abs(rhoT - rr).max() < 1e-10
    assert abs(E - Er) < 1e-8 * abs(Er)
    # idempotency of the zero-temperature density (libdmet/test/test_mfd.py:104-111 style)
    rk = res["rho_k"]
    assert np.abs(np.einsum("skij,skjl->skil", rk, rk) - rk).max() < 1e-10


# ---------------------------------------------------------------------------------------------
# K4: bath
# ---------------------------------------------------------------------------------------------

def _proj(b):
    b = b.reshape(b.shape[0], -1, b.shape[-1])
    return np.einsum("spa,sqa->spq", b, b)


def test_G4_bath_golden(ctx, golden):
    from libdmet_preview_amd.routine import slater
    g = golden("G4_bath.npz")
    rdm1_lo = np.load(os.path.join(os.path.dirname(__file__), "golden", "rdm1_lo.npy"))
    L = _lattice((1, 1, 3), 4, val=[0, 1], virt=[2, 3])
    b = slater.get_emb_basis(L, rdm1_lo)
    assert b.shape == g["hchain/basis_valbath"].shape
    assert np.linalg.norm(_proj(b) - _proj(g["hchain/basis_valbath"])) < 1e-10
    L2 = _lattice((1, 1, 3), 4, val=[0, 1, 2, 3], virt=[])
    b2 = slater.get_emb_basis(L2, rdm1_lo, nbath=2, valence_bath=False)
    assert np.linalg.norm(_proj(b2) - _proj(g["hchain/basis_full_nbath2"])) < 1e-10
    b3 = slater.embBasis(L2, np.array((rdm1_lo, rdm1_lo)), tol_bath=1e-7, valence_bath=False)
    assert b3.shape == g["hchain/basis_uhf_tol"].shape
    assert np.linalg.norm(_proj(b3) - _proj(g["hchain/basis_uhf_tol"])) < 1e-10
    assert np.linalg.norm(_proj(b) - _proj(b2)) < 1e-9            # span equality of test_slater.py:48-54
    for name in ("C1", "C2"):
        mesh = tuple(int(x) for x in g[name + "/mesh"])
        rho = g[name + "/rhoT"]
        nlo = rho.shape[-1]
        Lm = _lattice(mesh, nlo, val=list(range(nlo)))
        for kind in ("svd", "eig"):
            bb = slater.get_emb_basis(Lm, rho, kind=kind)
            ref = g[name + "/basis_" + kind]
            assert bb.shape == ref.shape, (name, kind, bb.shape, ref.shape)
            assert np.linalg.norm(_proj(bb) - _proj(ref)) < 1e-10
    rho = g["gen/rhoT"]
    Lg = _lattice((2, 2, 2), 7, val=[1, 2, 3], virt=[4, 5], core=[0])
    for key, extra in (("basis_svd", {}), ("basis_svd_noorth", {"orth": False}),
                       ("basis_svd_fullbath", {"valence_bath": False})):
        bb = slater.get_emb_basis(Lg, rho, **extra)
        ref = g["gen/" + key]
        assert bb.shape == ref.shape
        assert np.linalg.norm(_proj(bb) - _proj(ref)) < 1e-10
    with pytest.raises(ValueError):
        slater.get_emb_basis(Lg, rho, kind="nope")


def test_lowdin_module(ctx):
    """lo/lowdin.py twins (_lowdin, _vec_lowdin, vec_lowdin) against the oracle restatement of lowdin.py:83-134."""
    from libdmet_preview_amd.lo import lowdin
    rng = np.random.default_rng(21)
    n, m, nk = 9, 5, 4
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    S1 = A @ A.conj().T + n * np.eye(n)
    assert np.abs(lowdin._lowdin(S1) - R._lowdin(S1)).max() < 1e-12
    Sr = S1.real
    assert lowdin._lowdin(Sr).dtype == np.float64 and np.abs(lowdin._lowdin(Sr) - R._lowdin(Sr)).max() < 1e-12
    # rank-deficient metric: the near-null direction is dropped (tol), like the reference
    v = rng.standard_normal((n, 3))
    assert np.abs(lowdin._lowdin(v @ v.T, tol=1e-10) - R._lowdin(v @ v.T, tol=1e-10)).max() < 1e-10
    c = rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))
    assert np.abs(lowdin._vec_lowdin(c, S1) - R.vec_lowdin(c, S1)).max() < 1e-12
    assert np.abs(lowdin._vec_lowdin(c.real) - R.vec_lowdin(c.real)).max() < 1e-12
    f = rng.uniform(0.5, 1.5, m)
    ref = (c * f) @ R._lowdin(c.conj().T @ S1 @ c)
    assert np.abs(lowdin._vec_lowdin(c, S1, f) - ref).max() < 1e-12
    # (spin, k) batches
    Sk = np.array([S1 + k * np.eye(n) for k in range(nk)])
    Ck = rng.standard_normal((2, nk, n, m)) + 1j * rng.standard_normal((2, nk, n, m))
    out = lowdin.vec_lowdin(Ck, Sk)
    for s in range(2):
        for k in range(nk):
            assert np.abs(out[s, k] - R.vec_lowdin(Ck[s, k], Sk[k])).max() < 1e-12
            assert np.abs(out[s, k].conj().T @ Sk[k] @ out[s, k] - np.eye(m)).max() < 1e-12
    assert np.abs(lowdin.vec_lowdin(Ck[0], Sk) - out[0]).max() < 1e-14
    out2 = lowdin.vec_lowdin_k(Ck[:, 0], S1)
    assert np.abs(out2[1] - R.vec_lowdin(Ck[1, 0], S1)).max() < 1e-12
    with pytest.raises(NotImplementedError):
        lowdin.vec_lowdin(Ck, Sk, f=np.ones((2, nk, m)))


def test_bath_c5_shape_vs_oracle(ctx):
    """Config-C5 bath shape: stripe (216, 200, 200), 56 valence orbitals -> A = 43144 x 56 per spin."""
    from libdmet_preview_amd.routine import slater
    mesh, nlo, nval = (6, 6, 6), 200, 56
    rng = np.random.default_rng(8)
    # a physical-looking rdm1: projector-like stripe built from a random low-rank real-space vector set
    nocc = 40
    V = rng.standard_normal((216 * nlo, nocc)) * np.exp(-0.02 * np.arange(216 * nlo))[:, None]
    V, _ = np.linalg.qr(V)
    rdm1 = (V @ V[:nlo].T).reshape(216, nlo, nlo)               # columns of cell 0 of a projector
    L = _lattice(mesh, nlo, val=list(range(nval)), virt=list(range(nval, nlo)))
    b = slater.get_emb_basis(L, rdm1)
    ref, info = R.get_emb_basis(mesh, nlo, rdm1, imp_idx=list(range(nlo)), val_idx=list(range(nval)), return_info=True)
    assert b.shape == ref.shape
    # embedding basis is orthonormal; for two orthonormal bases of equal rank
    # ||B1 B1^T - B2 B2^T||_F = sqrt(2) ||(1 - B2 B2^T) B1||_F  (avoids the 43200^2 projector)
    B = b.reshape(-1, b.shape[-1])
    Bref = ref.reshape(-1, ref.shape[-1])
    assert np.abs(B.T @ B - np.eye(B.shape[1])).max() < 1e-12
    assert np.abs(Bref.T @ Bref - np.eye(B.shape[1])).max() < 1e-12
    resid = B - Bref @ (Bref.T @ B)
    assert np.sqrt(2.0) * np.linalg.norm(resid) < 1e-10, np.sqrt(2.0) * np.linalg.norm(resid)
    assert info["nbath_s"][0] == b.shape[-1] - nlo


# ---------------------------------------------------------------------------------------------
# hot kernels (zhot.hip, dgemm_big): production tile shapes, nemb = 256
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("nao,naux,spin", [(40, 24, 1), (200, 8, 2), (104, 19, 1), (16, 160, 1), (24, 80, 2), (16, 130, 2)])
def test_hot_half_transform_planes(ctx, nao, naux, spin):
    """nemb = 256 triggers the LDS-DMA ring kernels; checked block-by-block against the oracle's r_e2 restatement,
    with and without the time-reversal partner term, accumulating over two pushes."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd._lib import lib
    mesh, nemb = (2, 2, 1), 256
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(nao)
    Cemb = (rng.standard_normal((spin, 4, nao, nemb)) + 1j * rng.standard_normal((spin, 4, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Cemb)
    eri_dev = ctx.zeros((spin * (spin + 1) // 2, 8, 8), np.float64)      # never contracted in this test
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    ctx.check(lib.dmk_eri_begin_kL(eng.h, 1))
    ref = np.zeros((spin, naux, npair), dtype=np.complex128)
    # naux >= 80: more work items than resident workgroups -> the tail of the step-2 launch is cut into slot ranges
    # accumulated through the partial buffers; 11 pushes = one full group of 8 queued blocks + a group of 3
    pushes = [(1, 0, 1), (3, 2, 0), (0, 1, 1)]
    if naux >= 80:
        pushes = pushes + [(2, 3, 1), (0, 0, 0), (1, 1, 1), (2, 0, 0), (3, 1, 1), (0, 2, 1), (1, 3, 0), (2, 2, 1)]
    for (i, j, sym) in pushes:
        blk = R.df_block_philox(5, i, j, naux, nao)
        d_blk = ctx.to_device(blk)
        ctx.check(lib.dmk_eri_push_block(eng.h, i, j, sym, d_blk.ptr))
        Lij = R.transform_ao_to_emb(blk.reshape(naux, -1), Cemb, i, j)
        if sym:
            Lij = Lij + Lij.transpose(0, 1, 3, 2)
        ref += R.pack_tril(Lij)
    planes = eng.planes().get()
    got = planes[:, 0] + 1j * planes[:, 1]
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(got - ref).max() < 1e-11 * scale, np.abs(got - ref).max()
    # leave the engine in a clean state without running the 8.6 GB contraction
    eng.close()


@pytest.mark.parametrize("nao,naux,spin", [(24, 80, 2), (200, 8, 1), (16, 130, 2)])
def test_hot_half_transform_all_symmetrised_groups(ctx, nao, naux, spin):
    """nemb = 256 with EVERY queued block carrying the time-reversal partner term (the usual case: 12 040 of the 12 152 C5
    blocks): the diagonal 16 x 16 blocks then run one segment and are completed as P + P^T in the epilogue."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd._lib import lib
    mesh, nemb = (2, 2, 1), 256
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(nao + 1)
    Cemb = (rng.standard_normal((spin, 4, nao, nemb)) + 1j * rng.standard_normal((spin, 4, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Cemb)
    eri_dev = ctx.zeros((spin * (spin + 1) // 2, 8, 8), np.float64)
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    ctx.check(lib.dmk_eri_begin_kL(eng.h, 1))
    ref = np.zeros((spin, naux, npair), dtype=np.complex128)
    for (i, j) in [(1, 0), (3, 2), (0, 1), (2, 3), (0, 0), (1, 1), (2, 0), (3, 1), (0, 2), (1, 3), (2, 2)]:
        blk = R.df_block_philox(6, i, j, naux, nao)
        d_blk = ctx.to_device(blk)                       # must outlive the call: the pipeline reads it asynchronously
        ctx.check(lib.dmk_eri_push_block(eng.h, i, j, 1, d_blk.ptr))
        Lij = R.transform_ao_to_emb(blk.reshape(naux, -1), Cemb, i, j)
        ref += R.pack_tril(Lij + Lij.transpose(0, 1, 3, 2))
    planes = eng.planes().get()
    got = planes[:, 0] + 1j * planes[:, 1]
    assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()
    eng.close()


@pytest.mark.parametrize("nao,naux,nemb,spin,env", [
    (104, 19, 136, 1, {}), (40, 24, 200, 2, {}), (16, 40, 272, 2, {}), (24, 30, 40, 1, {}),
    (32, 20, 100, 2, {}), (200, 8, 136, 2, {}), (48, 16, 137, 1, {}), (16, 64, 33, 2, {}),
    # larger embedding spaces: four and more segments, large row / column offsets of the items
    (24, 24, 400, 2, {}), (16, 33, 520, 1, {}), (16, 32, 1040, 1, {}),
    # both occupancy points forced on both item kinds (segment items at 3 workgroups / CU, wide items at 2)
    (16, 40, 272, 2, {"DMK_ERI_TAB_OCC": "3"}), (24, 24, 400, 1, {"DMK_ERI_TAB_OCC": "3"}), (32, 20, 100, 2, {"DMK_ERI_TAB_OCC": "2"}),
    # the grouped kernel declines: eri_flush falls back to the generic step 2 for the queued blocks
    (40, 16, 200, 2, {"DMK_ERI_TAB_DECLINE": "1"})])
def test_tab_half_transform_planes(ctx, nao, naux, nemb, spin, env, monkeypatch):
    """General embedding dimension: the table-driven step-2 kernel (zhot_tab.hip) behind the same block queue as the
    nemb = 256 kernels -- directly pushed blocks (step 1 per block, step 2 per group) and the ring feed (both steps per
    group) -- against the oracle's r_e2 restatement, with and without the time-reversal partner term, 11 pushes =
    one full group of 8 queued blocks + a group of 3.  nemb not a multiple of 16 exercises the clamped panels."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd._lib import lib
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    mesh = (2, 2, 1)
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(nao + nemb)
    Cemb = (rng.standard_normal((spin, 4, nao, nemb)) + 1j * rng.standard_normal((spin, 4, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Cemb)
    mixed = [(1, 0, 1), (3, 2, 0), (0, 1, 1), (2, 3, 1), (0, 0, 0), (1, 1, 1), (2, 0, 0), (3, 1, 1), (0, 2, 1), (1, 3, 0), (2, 2, 1)]
    # second pass: every block symmetrised (the usual case) -> the diagonal blocks are folded as P + P^T in the epilogue
    for pushes in (mixed, [(i, j, 1) for (i, j, _) in mixed]):
        ref = np.zeros((spin, naux, npair), dtype=np.complex128)
        blocks = {}
        for (i, j, sym) in pushes:
            blocks[(i, j)] = R.df_block_philox(5, i, j, naux, nao)
            Lij = R.transform_ao_to_emb(blocks[(i, j)].reshape(naux, -1), Cemb, i, j)
            ref += R.pack_tril(Lij + Lij.transpose(0, 1, 3, 2) if sym else Lij)
        scale = max(1.0, np.abs(ref).max())
        for feed in ("push", "ring"):
            eri_dev = ctx.zeros((spin * (spin + 1) // 2, 8, 8), np.float64)      # never contracted in this test
            eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
            assert eng.ring_slots == (8 if nemb == 256 else 16)                  # the grouped hot path is active for this shape
            ctx.check(lib.dmk_eri_begin_kL(eng.h, 1))
            for n, (i, j, sym) in enumerate(pushes):
                if feed == "push":
                    d_blk = ctx.to_device(blocks[(i, j)])
                    ctx.check(lib.dmk_eri_push_block(eng.h, i, j, sym, d_blk.ptr))
                else:
                    eng.ring[n % eng.ring_slots].set(blocks[(i, j)])
                    ctx.check(lib.dmk_eri_push_ring_slot(eng.h, i, j, sym))
            planes = eng.planes().get()
            got = planes[:, 0] + 1j * planes[:, 1]
            assert np.abs(got - ref).max() < 1e-11 * scale, (feed, np.abs(got - ref).max())
            eng.close()


@pytest.mark.parametrize("nao,naux,nemb,spin", [(40, 24, 256, 2), (10, 7, 12, 1), (24, 16, 40, 2)])
def test_host_block_feed_matches_device_feed(ctx, nao, naux, nemb, spin):
    """dmk_eri_push_block_host (two pinned buffers, copy stream overlapped with the transform) against
    dmk_eri_push_block on device-resident blocks: same kernels in the same order, so the planes and the contracted
    ERI agree bit for bit; hot (nemb = 256) and generic kernels."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd._lib import lib, PinnedArray
    mesh, nk = (3, 2, 1), 6
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(nao + nemb)
    Cemb = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Cemb)
    blocks = {}

    def block(i, j):
        if (i, j) not in blocks:
            blocks[(i, j)] = R.df_block_philox(9, i, j, naux, nao)
        return blocks[(i, j)]
    out = []
    for feed in ("device", "host"):
        eri_dev = ctx.zeros((spin * (spin + 1) // 2, npair, npair), np.float64)
        eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
        prov = et.GDFMemory(np.zeros((nk, 3)), block, naux=naux)
        if feed == "device":
            prov = type("DevOnly", (), {"load_block": lambda self, c, i, j, o: o.set(block(i, j))})()
        eng.run(prov)
        ctx.sync()
        out.append(eri_dev.get())
        eng.close()
    assert np.array_equal(out[0], out[1])
    assert np.abs(out[0]).max() > 0
    # raw C ABI: pageable host memory is accepted too (synchronous copy), and a bad slot is an error
    eri_dev = ctx.zeros((spin * (spin + 1) // 2, npair, npair), np.float64)
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    kL = eng.irreducible_kL()[1]
    ctx.check(lib.dmk_eri_begin_kL(eng.h, kL))
    for n, r in enumerate(eng.by_kL[kL]):
        i, j, sym = int(r[1]), int(r[2]), int(r[4])
        blk = np.ascontiguousarray(block(i, j))
        ctx.check(lib.dmk_eri_push_block_host(eng.h, i, j, sym, blk.ctypes.data, n & 1))
        ctx.check(lib.dmk_eri_host_slot_wait(eng.h, n & 1))
    assert lib.dmk_eri_push_block_host(eng.h, 0, 0, 0, blk.ctypes.data, 2) != 0
    ctx.check(lib.dmk_eri_end_kL(eng.h, int(eng.weights[kL])))
    ctx.sync()
    eng.close()
    p = PinnedArray(ctx, (4, 3), np.complex128)
    p.a[...] = 1 + 2j
    assert p.a.sum() == 12 * (1 + 2j)
    p.free()


@pytest.mark.parametrize("spin", [1, 2])
def test_reference_named_eri_pieces(ctx, spin):
    """transform_ao_to_emb / _Lij_s4_to_eri / sr_loop / get_naoaux with the reference's numpy signatures
    (eri_transform.py:159-227, 403-521) against the oracle restatement."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    rng = np.random.default_rng(40 + spin)
    nk, nao, nemb, nL = 3, 7, 5, 11
    npair = nemb * (nemb + 1) // 2
    Cemb = rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))
    Lpq = rng.standard_normal((nL, nao * nao)) + 1j * rng.standard_normal((nL, nao * nao))
    got = et.transform_ao_to_emb(Lpq, Cemb, 2, 1)
    ref = R.transform_ao_to_emb(Lpq, Cemb, 2, 1).reshape(spin, nL, nemb * nemb)
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-12 * np.abs(ref).max()
    if spin == 2:
        Lb = rng.standard_normal((nL, nao * nao)) + 1j * rng.standard_normal((nL, nao * nao))
        got = et.transform_ao_to_emb(Lpq, Cemb, 0, 1, Lpq_beta=Lb)
        refb = R.transform_ao_to_emb(Lb, Cemb, 0, 1).reshape(spin, nL, -1)
        refa = R.transform_ao_to_emb(Lpq, Cemb, 0, 1).reshape(spin, nL, -1)
        assert np.abs(got[0] - refa[0]).max() < 1e-11 and np.abs(got[1] - refb[1]).max() < 1e-11
    assert np.abs(et.transform_ao_to_emb(Lpq, Cemb[0], 1, 1)[0] - R.transform_ao_to_emb(Lpq, Cemb[:1], 1, 1)[0].reshape(nL, -1)).max() < 1e-11
    Lij = rng.standard_normal((spin, nL, npair)) + 1j * rng.standard_normal((spin, nL, npair))
    nb = spin * (spin + 1) // 2
    for w, tr in ((1, True), (2, True), (1, False)):
        e0 = rng.standard_normal((nb, npair, npair)).astype(np.float64 if tr else np.complex128)
        e_ref, e_got = e0.copy(), e0.copy()
        R.Lij_s4_to_eri(Lij, e_ref, w, tr)
        et._Lij_s4_to_eri(Lij if spin == 2 else Lij[0], e_got, w, tr)
        assert np.abs(e_got - e_ref).max() < 1e-11 * np.abs(e_ref).max(), (w, tr)
    # out-of-core form: {"ccdd": array} in (aa, bb, ab) order
    h = {"ccdd": np.zeros((nb, npair, npair))}
    et._Lij_s4_to_eri(Lij, h, 2, True)
    e_ref = np.zeros((nb, npair, npair))
    R.Lij_s4_to_eri(Lij, e_ref, 2, True)
    order = [0] if spin == 1 else [0, 2, 1]
    assert np.abs(h["ccdd"] - e_ref[order]).max() < 1e-11 * np.abs(e_ref).max()
    with pytest.raises(ValueError):
        et._Lij_s4_to_eri(Lij, e_ref, 3, True)
    # cderi readers
    kpts = np.array([[0.0, 0, 0], [0.5, 0, 0]])
    naux = 6
    L00 = rng.standard_normal((naux, nao, nao))
    L00 = L00 + L00.transpose(0, 2, 1)
    L10 = rng.standard_normal((naux - 1, nao, nao)) + 1j * rng.standard_normal((naux - 1, nao, nao))
    L11 = rng.standard_normal((naux, nao, nao)) + 1j * rng.standard_normal((naux, nao, nao))
    L11 = L11 + L11.conj().transpose(0, 2, 1)
    il = np.tril_indices(nao)
    feri = {"j3c-kptij": np.array([[kpts[0], kpts[0]], [kpts[1], kpts[0]], [kpts[1], kpts[1]]]),
            "j3c/0/0": L00[:, il[0], il[1]], "j3c/1/0": L10[:4].reshape(4, -1), "j3c/1/1": L10[4:].reshape(naux - 5, -1),
            "j3c/2/0": L11[:, il[0], il[1]]}
    cell = type("Cell", (), {"nao_nr": lambda self: nao})()
    gdf = type("GDF", (), {})()
    gdf._cderi, gdf.kpts, gdf.cell, gdf.blockdim = feri, kpts, cell, 4
    assert et.get_naoaux(gdf) == naux
    got = np.concatenate(list(et.sr_loop(gdf, kpts[[1, 0]], compact=False, blksize=2)))
    assert np.abs(got[:naux - 1] - L10.reshape(naux - 1, -1)).max() == 0 and np.abs(got[naux - 1:]).max() == 0
    got = np.concatenate(list(et.sr_loop(gdf, kpts[[0, 1]], compact=False)))            # swapped pair: conjugate transpose
    assert np.abs(got[:naux - 1] - L10.conj().transpose(0, 2, 1).reshape(naux - 1, -1)).max() == 0
    got = np.concatenate(list(et.sr_loop(gdf, kpts[[1, 1]], compact=True)))
    assert got.shape == (naux, nao * (nao + 1) // 2) and np.abs(got - L11[:, il[0], il[1]]).max() == 0
    got = np.concatenate(list(et.sr_loop(gdf, kpts[[0, 0]], compact=False)))
    assert np.abs(got - L00.reshape(naux, -1)).max() == 0


def test_hot_contraction_big(ctx):
    """dgemm_big (LDS-DMA ring, 256 x 128 tiles) on stacked Re/Im planes incl. masked edge tiles."""
    from libdmet_preview_amd._lib import lib
    rng = np.random.default_rng(77)
    for N, K in [(2080, 160), (1538, 64), (4096, 1600)]:
        X = rng.standard_normal((K, N))
        Y = rng.standard_normal((K, N))
        C0 = rng.standard_normal((N, N))
        dX, dY, dC = ctx.to_device(X), ctx.to_device(Y), ctx.to_device(C0)
        ctx.check(lib.dmk_dgemm_tn_acc(ctx.h, N, K, 2.0, dX.ptr, dY.ptr, N, dC.ptr, N))
        ref = C0 + 2.0 * X.T @ Y
        assert np.abs(dC.get() - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize("case", ["perm", "shift", "shiftperm"])
@pytest.mark.parametrize("spin", [1, 2])
def test_G15_eri_general_k_lists(ctx, golden, case, spin):
    """get_emb_eri_fast_gdf on k lists that are not the np.fft-ordered Gamma-centred mesh -- a permuted list and a shifted
    Monkhorst-Pack mesh with kscaled_center (eri_transform.py:262-266) -- against the reference-generated golden, with and
    without time reversal, and the `ERI imaginary` diagnostic of the non-TR branch (eri_transform.py:385-394)."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd.system.lattice import _UnitCell
    g = golden("G15_eri_kopts.npz")
    mesh = tuple(int(x) for x in g["mesh"])
    W0 = g["W0"]
    naux, nao = W0.shape[0], W0.shape[2]
    ks = g[case + "/kpts_scaled"]
    center = None if case == "perm" else g["shift"]
    blocks = R.df_blocks_from_W0(W0, mesh, ks)
    cell = _UnitCell(nao)
    mydf = et.GDFMemory(cell.get_abs_kpts(ks), blocks, naux=naux, cell=cell)
    st = "%s/s%d" % (case, spin)
    for tr in (True, False):
        e = et.get_emb_eri_fast_gdf(cell, mydf, C_ao_lo=g[st + "/C_ao_lo"], basis=g[st + "/basis"], t_reversal_symm=tr,
                                    kscaled_center=center)
        ref = g[st + "/eri_%s" % ("tr" if tr else "notr")]
        # |eri| reaches ~1e6 in this fixture, so 1e-10 RELATIVE is ~1e-4 absolute -- all that f64 accumulation in a different
        # order can give at that magnitude; the north star's 1e-8 is an ABSOLUTE bound on O(1) integrals, checked below on
        # the same problem rescaled to max |eri| = 1 (the transform is quadratic in the DF tensor)
        assert e.shape == ref.shape and np.abs(e - ref).max() < 1e-10 * np.abs(ref).max(), (tr, np.abs(e - ref).max())
        if not tr:
            im_ref = float(g[st + "/imag_norm"])
            assert abs(et.get_emb_eri_fast_gdf.last_imag_norm - im_ref) < 1e-10 * im_ref
        if tr:
            top = np.abs(ref).max()
            unit_df = et.GDFMemory(cell.get_abs_kpts(ks), {ij: b / np.sqrt(top) for ij, b in blocks.items()}, naux=naux, cell=cell)
            e1 = et.get_emb_eri_fast_gdf(cell, unit_df, C_ao_lo=g[st + "/C_ao_lo"], basis=g[st + "/basis"], t_reversal_symm=True,
                                         kscaled_center=center)
            assert np.abs(e1 - ref / top).max() < 1e-8
    # the canonical mesh keeps its (zero) imaginary part: physical blocks, TR-symmetric orbitals
    assert np.array_equal(et.get_weights_t_reversal(cell, cell.get_abs_kpts(ks)), R.get_weights_t_reversal(ks))


def test_bath_with_fixed_nbath_beyond_the_rank(ctx):
    """`nbath` fixed by the caller beyond the rank of the env x imp block (slater.py:177-186: the reference warns "Zero singular value
    exists" and keeps LAPACK's orthonormal completion): the basis must stay orthonormal and its entangled part must be the oracle's."""
    from libdmet_preview_amd.routine import slater
    mesh, nlo = (1, 1, 3), 2
    rng = np.random.default_rng(12)
    u = rng.standard_normal(4)
    u /= np.linalg.norm(u)
    rdm1 = np.zeros((3, nlo, nlo))
    rdm1[0] = np.diag([0.7, 0.3])
    rdm1[1:] = (0.25 * np.outer(u, [0.6, 0.8])).reshape(2, nlo, nlo)           # rank one
    L = _lattice(mesh, nlo, val=[0, 1], virt=[])
    b = slater.get_emb_basis(L, rdm1, nbath=2, valence_bath=False)
    assert b.shape[-1] == nlo + 2
    B = b[0].reshape(-1, b.shape[-1])
    assert np.abs(B.T @ B - np.eye(B.shape[1])).max() < 1e-12
    assert abs(abs(B[nlo:, nlo] @ u) - 1.0) < 1e-12
    ref = R.get_emb_basis(mesh, nlo, rdm1, imp_idx=[0, 1], val_idx=[0, 1], nbath=2, valence_bath=False)
    Bref = ref[0].reshape(-1, ref.shape[-1])
    assert abs(abs(Bref[nlo:, nlo] @ u) - 1.0) < 1e-12


def test_G22_scdm_bath_localisation(ctx, golden):
    """routine/localizer.localize_bath('scdm') on the device (dmk_cpqr_pivots + Loewdin + one product) against the reference's
    output (golden G22), directly and through slater.get_emb_basis(localize_bath='scdm'); 'pm' needs PySCF's optimiser and raises."""
    from libdmet_preview_amd.routine import slater, localizer
    from tests.test_oracle_golden import _match_columns
    g = golden("G22_scdm_bath.npz")
    for name in ("a", "b", "c"):
        B, ref = g[name + "/B"], g[name + "/B_scdm"]
        got = localizer.localize_bath(B, "scdm")
        assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-10
        # pivots agree with the column-pivoted QR of LAPACK on these (generic) orbitals
        import scipy.linalg as sla
        _, _, perm = sla.qr(B.T, pivoting=True)
        import ctypes as C
        from libdmet_preview_amd._lib import lib
        piv = np.zeros(B.shape[1], dtype=np.int32)
        d_B = ctx.to_device(np.ascontiguousarray(B))
        ctx.check(lib.dmk_cpqr_pivots(ctx.h, B.shape[0], B.shape[1], d_B.ptr, B.shape[1], piv.ctypes.data_as(C.c_void_p)))
        assert list(piv) == [int(x) for x in perm[:B.shape[1]]]
    rho = g["gen/rhoT"]
    Lg = _lattice((2, 2, 2), 7, val=[1, 2, 3], virt=[4, 5], core=[0])
    Lg.is_model = True
    for key, extra in (("basis_svd_scdm", {}), ("basis_svd_scdm_fullbath", {"valence_bath": False})):
        bb = slater.get_emb_basis(Lg, rho, localize_bath="scdm", **extra)
        ref = g["gen/" + key]
        assert bb.shape == ref.shape and np.linalg.norm(_proj(bb) - _proj(ref)) < 1e-10
        for s in range(bb.shape[0]):
            err, _ = _match_columns(ref[s].reshape(-1, ref.shape[-1]), bb[s].reshape(-1, bb.shape[-1]))
            assert err < 1e-8
    with pytest.raises(NotImplementedError):
        slater.get_emb_basis(Lg, rho, localize_bath="pm")
    with pytest.raises(ValueError):
        slater.get_emb_basis(Lg, rho, localize_bath="nope")
