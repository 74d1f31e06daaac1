"""
GPU parity at the PRODUCTION launch geometry (run with -m gpu on an MI355X): whole momentum transfers kL of the
BASELINE configs C5 (6x6x6, nao 200, naux 800, nemb 256, UHF: 108 / 112 AO blocks per kL through the block ring,
8-slot groups, both spins per launch, symmetric contraction at N = 32896 with K = 1600 and K = 800) and C4 (4x4x4,
nao 104, naux 416, nemb 136: the complete transform, all 1184 blocks) against the sampled oracle
(oracle/eri_sample.py: exact entries of Lij_s4 and of the ERI for a sample of embedding-orbital pairs, ALL auxiliary
rows, reference visiting plan from the G1 golden).  Tolerance: the north star's 1e-8 max-abs on the ERI (and 1e-11
relative, which is what f64 accumulation in a different order gives).
reference: basis_transform/eri_transform.py:338-382 (loop), 403-434 (half transform), 436-485 (contraction).
"""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import eri_sample as ES                  # the checker


@pytest.fixture(scope="module")
def ctx():
    from libdmet_preview_amd import _lib
    return _lib.get_ctx()


def _eri_rows(eri_dev, blk, npair, rows):
    return np.stack([eri_dev.offset((blk * npair + int(r)) * npair, (npair,)).get() for r in rows])


def _run_and_check(ctx, mesh, nao, naux, nemb, spin, kL_list, A, seed, check_planes=True, want_ring=False):
    from libdmet_preview_amd.basis_transform import eri_transform as et
    nk = int(np.prod(mesh))
    npair = nemb * (nemb + 1) // 2
    nblk = spin * (spin + 1) // 2
    rng = np.random.default_rng(seed)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Ce)
    eri_dev = ctx.zeros((nblk, npair, npair), np.float64)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=seed + 1)
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    if want_ring:
        assert eng.ring_slots > 0, "the grouped hot path declined this shape"
    ref_eri, idx, ref_planes = ES.eri_sample(mesh, seed + 1, Ce, naux, A, kL_list)
    # Freivalds probe: the contraction is checked on ALL pair rows and columns, not only on the sampled ones
    d_x = ctx.to_device(rng.uniform(-1.0, 1.0, npair))
    d_yref = ctx.zeros((nblk, npair), np.float64)
    eng.set_probe(d_x, d_yref)
    worst_p = 0.0
    try:
        for kL in kL_list:
            n = eng.run_kL(kL, df)
            assert n == len(ES.plan_records(mesh, {kL})[1][kL])
            if check_planes:
                pl = eng.planes().get()                                  # (spin, 2, naux, npair) of this kL
                got = pl[:, 0][:, :, idx] + 1j * pl[:, 1][:, :, idx]
                ref = ref_planes[kL]
                if int(eng.weights[kL]) == 1:
                    # a kL that is its own time-reversal partner: only Re of its planes is ever read (eri_transform.py:453-455) and
                    # only Re is computed (real-part-only step 2); the Im half stays zero
                    assert np.abs(pl[:, 1]).max() == 0.0 or os.environ.get("DMK_ERI_RE_ONLY") == "0"
                    got, ref = got.real, ref.real
                err = np.abs(got - ref).max()
                worst_p = max(worst_p, err / np.abs(ref).max())
                assert err < 1e-11 * np.abs(ref).max(), (kL, err, np.abs(ref).max())
                # columns outside the sample are not silently zero
                assert np.abs(pl[:, 0]).min(axis=(0, 1)).max() > 0
        ctx.sync()
        for b in range(nblk):
            got = _eri_rows(eri_dev, b, npair, idx)[:, idx]
            err = np.abs(got - ref_eri[b]).max()
            assert err < 1e-8 and err < 1e-11 * np.abs(ref_eri).max(), (b, err, np.abs(ref_eri).max())
        y, yref = et.eri_times_vector_dev(ctx, eri_dev, nblk, npair, d_x).get(), d_yref.get()
        assert np.abs(yref).max() > 0
        assert np.abs(y - yref).max() <= 1e-10 * max(1.0, np.abs(yref).max()), (np.abs(y - yref).max(), np.abs(yref).max())
        if nblk == 3:
            # aa and bb come from the symmetric (lower-tile-triangle + mirrored store) contraction: both halves present
            for b in (0, 2):
                got = _eri_rows(eri_dev, b, npair, idx)[:, idx]
                assert np.abs(got - got.T).max() <= 1e-12 * np.abs(got).max()
    finally:
        eng.close()
    return worst_p


def test_c5_full_kL_production_geometry(ctx):
    """C5, two whole kL (w = 2: 108 blocks, K = 1600; w = 1: 112 blocks, K = 800) through the ring feed."""
    A = [0, 17, 127, 128, 191, 192, 255]          # every workgroup type of step 2 and both triangle halves
    _run_and_check(ctx, (6, 6, 6), 200, 800, 256, 2, [1, 0], A, seed=101)


def test_c4_complete_transform(ctx):
    """C4 at full size: 4x4x4 mesh, nao 104, naux 416, nemb 136, RHF, all 1184 AO blocks / 36 irreducible kL."""
    w, by = ES.plan_records((4, 4, 4))
    kls = sorted(by)
    assert sum(len(by[k]) for k in kls) == 1184
    A = [0, 15, 16, 63, 64, 127, 128, 135]
    _run_and_check(ctx, (4, 4, 4), 104, 416, 136, 1, kls, A, seed=202, check_planes=False)


@pytest.mark.parametrize("spin", [1, 2])
def test_c4_one_kL_planes(ctx, spin):
    """C4 shape, one w = 2 kL, planes and ERI, RHF and UHF."""
    A = [0, 15, 16, 63, 64, 127, 128, 135]
    _run_and_check(ctx, (4, 4, 4), 104, 416, 136, spin, [1], A, seed=303 + spin)


def test_c5_meanfield_bath_full_size(ctx):
    """The mean-field -> bath -> C_ao_emb chain of C5 at FULL size (432 x eigh(200), 86 400 occupations, rho_R, two
    43144 x 56 bath SVDs, 432 C_ao_emb blocks) against an independent computation by the oracle on the same seeded inputs
    (oracle/stage_check.py): eigenvalues <= 1e-10, bit-equal occupations, mu, rho_R <= 1e-10, equal nbath and the
    gauge-invariant projector distance <= 1e-10 Frobenius, C_ao_emb <= 1e-12.  reference: routine/mfd.py:235-360,
    routine/slater.py:117-220, eri_transform.py:118-126, 289-292."""
    from libdmet_preview_amd import pipeline
    from oracle import stage_check as SC
    sysm = pipeline.SyntheticSystem.from_workload(ctx, "C5")
    d_rhoR, mf = pipeline.mean_field_stage(ctx, sysm)
    d_basis, nemb, sigmas = pipeline.bath_stage(ctx, sysm, d_rhoR)
    d_C = pipeline.c_ao_emb_stage(ctx, sysm, d_basis, nemb)
    n, nk, spin = sysm.nlo, sysm.nk, sysm.spin
    got = {"ew": mf["ew"].get().reshape(spin, nk, n), "occ": mf["occ"].get().reshape(spin, nk, n), "mu": mf["mu"],
           "rho_R": d_rhoR.get().reshape(spin, nk, n, n), "basis": d_basis.get().reshape(spin, nk, n, nemb), "sigma": sigmas,
           "C_ao_emb": d_C.get().reshape(spin, nk, sysm.nao, nemb)}
    res = SC.compare(sysm.mesh, sysm.Fock_R, sysm.vcor, sysm.filling, sysm.restricted, sysm.imp_idx, sysm.val_idx, sysm.C_ao_lo, got)
    assert nemb == 256
    assert res["parity_ew_maxabs"] <= 1e-10, res
    assert res["parity_occ_equal"] and res["parity_mu_abs"] <= 1e-10, res
    assert res["parity_rho_maxabs"] <= 1e-10, res
    assert res["parity_nbath_equal"] and res["parity_bath_frob"] <= 1e-10 and res["parity_basis_orth"] <= 1e-12, res
    assert res["parity_c_ao_emb_maxabs"] <= 1e-12, res
    assert res["parity_stages_ok"]


@pytest.mark.parametrize("nao,naux,nemb,spin,nslots,pad", [(16, 40, 256, 2, 3, 1), (24, 16, 40, 1, 2, 1), (10, 7, 12, 2, 4, 1), (10, 7, 12, 2, 4, 0),
                                                           (104, 24, 136, 1, 16, 1), (27, 21, 41, 2, 3, 1), (27, 21, 41, 2, 3, 0)])
def test_plane_stack_and_banded_contraction(ctx, nao, naux, nemb, spin, nslots, pad, monkeypatch):
    """Deferred, K-stacked contraction (dmk_eri_stack): the planes of several kL stay resident -- weight-2 slots from the front,
    weight-1 slots (Re halves only) from the back -- and one segmented-K GEMM per weight class and spin block contracts them;
    a full stack flushes by itself; the final contraction run band by band (dmk_eri_contract) gives the same ERI and leaves
    every finished band of rows complete.  Against the per-kL contraction of the same engine and the sampled oracle; hot
    (nemb 256 / 136) and generic half-transform kernels.  Auxiliary dimensions off the K tile of the contraction kernel and odd
    pair counts (naux 7, 21; nemb 41 -> 861 pairs): with the padded plane geometry of round 6 (pad = 1: zero rows up to a multiple
    of 8, one zero column) they run on the LDS-DMA kernel with its symmetric launch; pad = 0 (DMK_ERI_PLANE_PAD=0) is the
    unpadded layout of rounds 1 - 5 on the register-staged kernel with per-segment launches."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    if not pad:
        monkeypatch.setenv("DMK_ERI_PLANE_PAD", "0")
    mesh = (3, 2, 1)                                 # weights 1, 2, 0 mixed: 2 weight-1 and 2 weight-2 irreducible kL
    nk = 6
    npair = nemb * (nemb + 1) // 2
    nblk = spin * (spin + 1) // 2
    rng = np.random.default_rng(nao + nemb)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Ce)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=77)

    d_x = ctx.to_device(rng.uniform(-1.0, 1.0, npair))

    def run(stack, banded):
        eri_dev = ctx.zeros((nblk, npair, npair), np.float64)
        eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
        snap = None
        d_yref = ctx.zeros((nblk, npair), np.float64)
        try:
            kls = eng.irreducible_kL()
            assert sorted(int(eng.weights[k]) for k in kls) == [1, 1, 2, 2]
            if stack:
                assert eng.set_stack(nslots=stack) == stack
            eng.set_probe(d_x, d_yref)                   # every contraction path (per kL, stacked, banded, self-flushing stack)
            for kL in kls:
                eng.run_kL(kL, df)
            if banded:
                nb, rows = eng.nbands()
                assert nb == (npair + rows - 1) // rows
                for b in range(nb):
                    eng.contract(b, b + 1, done=(b == nb - 1))
                    if b == 0 and nb > 1:
                        ctx.sync()
                        snap = eri_dev.get()[:, :rows].copy()          # rows of band 0 must already be final
            else:
                eng.contract()
            ctx.sync()
            y, yref = et.eri_times_vector_dev(ctx, eri_dev, nblk, npair, d_x).get(), d_yref.get()
            assert np.abs(y - yref).max() <= 1e-10 * max(1.0, np.abs(yref).max()), (stack, banded, np.abs(y - yref).max())
            return eri_dev.get(), snap
        finally:
            eng.close()

    ref, _ = run(0, False)
    scale = np.abs(ref).max()
    for stack, banded in ((nslots, False), (nslots, True), (2, True)):      # 2 slots < 4 kL: the stack flushes on its own in between
        got, snap = run(stack, banded)
        assert np.abs(got - ref).max() < 1e-11 * scale, (stack, banded, np.abs(got - ref).max(), scale)
        if snap is not None:
            assert np.abs(snap - ref[:, :snap.shape[1]]).max() < 1e-11 * scale
    A = [0, 1, nemb // 2, nemb - 1]
    want, idx, _ = ES.eri_sample(mesh, 77, Ce, naux, A, [int(k) for k in range(nk) if ES.plan_records(mesh)[0][k] > 0])
    for b in range(nblk):
        assert np.abs(ref[b][np.ix_(idx, idx)] - want[b]).max() < 1e-8


def test_jk_row_ranges_add_up(ctx):
    """dmk_jk_s4_rows: the J (both directions) and K of disjoint ranges of packed rows add up to the whole-ERI result -- the
    embedding-Hamiltonian stage of a row-sharded ERI sums exactly these partial matrices over ranks."""
    from libdmet_preview_amd.solver import scf
    n = 72
    npair = n * (n + 1) // 2
    rng = np.random.default_rng(4)
    E = rng.standard_normal((npair, npair))
    dm = rng.standard_normal((2, n, n))
    dE, d_dm = ctx.to_device(E), ctx.to_device(dm)
    da, db = d_dm.offset(0, (n, n)), d_dm.offset(n * n, (n, n))
    full = [x.get() for x in scf.jk_dev(ctx, n, dE, da, db, da)]
    cuts = [0, 128, 1024, 1056, npair]
    acc = [np.zeros((n, n)) for _ in range(3)]
    for own in (0, 1):
        ranges = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if i % 2 == own]
        part = scf.jk_dev(ctx, n, dE, da, db, da, row_ranges=ranges)
        for a_, p_ in zip(acc, part):
            a_ += p_.get()
    for a_, f_ in zip(acc, full):
        assert np.abs(a_ - f_).max() < 1e-11 * np.abs(f_).max()
    empty = scf.jk_dev(ctx, n, dE, da, None, da, row_ranges=[])
    assert np.abs(empty[0].get()).max() == 0.0 and empty[1] is None


def test_contraction_probe_detects_a_misplaced_tile(ctx):
    """The Freivalds probe (dmk_eri_probe) is a real detector: after a correct contraction the residual is at rounding level;
    moving ONE 128 x 128 tile of the ERI to a neighbouring position, zeroing one, or breaking the symmetry of one mirrored tile
    makes it jump by many orders of magnitude, whichever tile row is hit.  reference: eri_transform.py:451-478."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    mesh, nk, nao, naux, nemb, spin = (2, 2, 1), 4, 16, 24, 64, 2
    npair = nemb * (nemb + 1) // 2                          # 2080: 17 tile rows
    rng = np.random.default_rng(5)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Ce)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=9)
    eri_dev = ctx.zeros((3, npair, npair), np.float64)
    d_x = ctx.to_device(rng.uniform(-1.0, 1.0, npair))
    d_yref = ctx.zeros((3, npair), np.float64)
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    try:
        eng.set_stack(nslots=4)
        eng.set_probe(d_x, d_yref)
        for kL in eng.irreducible_kL():
            eng.run_kL(kL, df)
        eng.contract()
        ctx.sync()
    finally:
        eng.close()
    yref = d_yref.get()
    resid = lambda: np.abs(et.eri_times_vector_dev(ctx, eri_dev, 3, npair, d_x).get() - yref).max()
    clean = resid()
    assert clean <= 1e-10 * max(1.0, np.abs(yref).max())
    good = eri_dev.get()
    scale = np.abs(good).max()
    for blk, tr, tc in ((0, 0, 0), (1, 7, 3), (2, 16, 16), (0, 16, 0), (2, 5, 11)):
        r0, c0 = 128 * tr, 128 * tc
        r1, c1 = min(npair, r0 + 128), min(npair, c0 + 128)
        bad = good.copy()
        bad[blk, r0:r1, c0:c1] = 0.0                       # a tile the GEMM never wrote
        eri_dev.set(bad)
        assert resid() > 1e6 * max(clean, 1e-18) and resid() > 1e-6 * scale, (blk, tr, tc, resid(), clean)
        bad = good.copy()
        bad[blk, r0:r1, c0:c1] = good[blk, r0:r1, c0:c1][:, ::-1]      # right entries, wrong places inside the tile
        eri_dev.set(bad)
        assert resid() > 1e6 * max(clean, 1e-18), (blk, tr, tc)
    eri_dev.set(good)
    assert resid() == clean


def test_rows_only_pipeline_refuses_to_contract_into_an_internal_eri(ctx):
    """A pipeline opened without an ERI of its own (dmk_eri_begin flag 4, the out-of-core driver) must never launch a GEMM into
    one: a full stack at begin_kL, an explicit contract and the probe are refused with an error, and finish() -- also on the
    way out of an exception -- drops the resident planes instead of contracting them (round-3 advisor finding)."""
    from libdmet_preview_amd import _lib
    from libdmet_preview_amd.basis_transform import eri_transform as et
    mesh, nk, nao, naux, nemb, spin = (3, 2, 1), 6, 10, 8, 12, 2
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(1)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Ce)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=3)
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, None, True, rows_only=True)
    try:
        assert eng.set_stack(nslots=2) == 2
        kls = eng.irreducible_kL()
        eng.run_kL(kls[0], df)
        eng.run_kL(kls[1], df)
        assert eng.stack_free() == 0
        with pytest.raises(_lib.DmkError):
            eng.run_kL(kls[2], df)                         # full stack: would have contracted into the (absent) ERI
        with pytest.raises(_lib.DmkError):
            eng.contract()
        # the slab interface still works and agrees with an ordinary pipeline
        d_slab = ctx.zeros((3, npair, npair), np.float64)
        eng.contract_rows_into(0, npair, d_slab)
        ctx.sync()
        got = d_slab.get()
    finally:
        eng.close()                                        # planes still resident: dropped, no launch, no error
    eri_dev = ctx.zeros((3, npair, npair), np.float64)
    eng2 = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    try:
        for kL in kls[:2]:
            eng2.run_kL(kL, df)
        ctx.sync()
    finally:
        eng2.close()
    ref = eri_dev.get()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_jk_row_ranges_must_end_on_a_block(ctx):
    """dmk_jk_s4_rows: the J kernel works on whole 32-row blocks, so a range that ends inside a block (and not at the last row)
    would make two owners count the same rows; the library rejects it instead of returning a silently wrong J."""
    from libdmet_preview_amd import _lib
    from libdmet_preview_amd.solver import scf
    n = 24
    npair = n * (n + 1) // 2                               # 300
    rng = np.random.default_rng(2)
    dE, d_dm = ctx.to_device(rng.standard_normal((npair, npair))), ctx.to_device(rng.standard_normal((n, n)))
    with pytest.raises(_lib.DmkError):
        scf.jk_dev(ctx, n, dE, d_dm, None, d_dm, row_ranges=[(0, 100)])
    ok = scf.jk_dev(ctx, n, dE, d_dm, None, d_dm, row_ranges=[(0, 96), (96, npair)])
    full = scf.jk_dev(ctx, n, dE, d_dm, None, d_dm)
    assert np.abs(ok[0].get() - full[0].get()).max() <= 1e-12 * np.abs(full[0].get()).max()


@pytest.mark.parametrize("nao,naux,nemb,spin", [(104, 24, 136, 1), (16, 40, 256, 2)])
def test_producer_stream_ring_is_bitwise_the_single_stream_result(ctx, nao, naux, nemb, spin):
    """Device-side block producers on the pipeline's second stream (dmk_eri_ring_slot: double-buffered ring, the generator of group
    g + 1 overlapping the transform of group g, ordered by events; opt-in with DMK_ERI_GEN_STREAM=1) against the default (single-buffered
    ring on the compute stream): the same blocks in the same order, so the ERI must agree BIT FOR BIT -- a missing event edge (a block transformed before
    it was generated, or overwritten before it was consumed) would show up as a different number.  Several kL, several groups per kL,
    an explicit dmk_eri_flush in between (equal-length launches)."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    mesh, nk = (4, 3, 1), 12
    npair = nemb * (nemb + 1) // 2
    nblk = spin * (spin + 1) // 2
    rng = np.random.default_rng(nao + naux)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Ce)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=41)
    assert hasattr(df, "load_block_on")

    def run(gen_stream):
        os.environ["DMK_ERI_GEN_STREAM"] = "1" if gen_stream else "0"
        try:
            eri_dev = ctx.zeros((nblk, npair, npair), np.float64)
            eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
            try:
                assert eng.ring_slots > 0
                kls = eng.irreducible_kL()
                eng.set_stack(nslots=len(kls))
                for rep in range(2):                       # the second pass reuses both halves of the ring
                    for kL in kls:
                        eng.run_kL(kL, df)
                    eng.contract()
                ctx.sync()
                return eri_dev.get()
            finally:
                eng.close()
        finally:
            os.environ.pop("DMK_ERI_GEN_STREAM", None)

    a = run(True)
    b = run(False)
    assert np.abs(a).max() > 0
    assert np.array_equal(a, b)
    # and against the oracle on a sample, so that "equal" is not "equally wrong"
    A = [0, 1, nemb // 2, nemb - 1]
    want, idx, _ = ES.eri_sample(mesh, 41, Ce, naux, A, [int(k) for k in range(nk) if ES.plan_records(mesh)[0][k] > 0])
    for blk in range(nblk):
        assert np.abs(a[blk][np.ix_(idx, idx)] - 2.0 * want[blk]).max() < 1e-8


@pytest.mark.parametrize("mesh,nao,naux,nemb,spin", [((3, 2, 1), 27, 40, 256, 2), ((2, 2, 2), 101, 24, 136, 1), ((3, 1, 1), 50, 33, 72, 2),
                                                     ((2, 2, 1), 203, 16, 256, 1), ((2, 2, 1), 17, 64, 40, 2), ((2, 2, 1), 19, 33, 33, 2),
                                                     ((3, 1, 1), 31, 37, 257, 1)])
def test_off_tile_ao_dimension_takes_the_hot_path(ctx, mesh, nao, naux, nemb, spin):
    """AO dimensions that are NOT a multiple of the K tile of the hot kernels (8): real basis sets rarely are (13 functions per
    carbon atom in GTH-DZVP, 5 per hydrogen in cc-pVDZ).  Up to round 5 such a system fell to the generic kernels at about half the
    rate; now the hot kernels loop over hot_kdim(nao) against the pipeline's zero-padded copy of C_ao_emb (step 1 clamps its A rows
    inside the block, step 2 reads the padding rows of its own initialised Ut buffer against zeros).  Planes, ERI and the Freivalds
    probe against the oracle, the block ring in use, both step-2 kernels (nemb = 256 and table-driven), one and two spins."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    nk = int(np.prod(mesh))
    w, by = ES.plan_records(mesh)
    kls = sorted(by)
    A = sorted(set([0, 1, nemb // 3, nemb // 2, nemb - 2, nemb - 1]))
    worst = _run_and_check(ctx, mesh, nao, naux, nemb, spin, kls, A, seed=500 + nao, want_ring=True)
    assert worst < 1e-11


@pytest.mark.parametrize("nao,naux,nemb,spin,rows", [(40, 100, 256, 2, 32), (27, 90, 72, 1, 37), (104, 48, 136, 1, 16)])
def test_step1_in_ranges_of_L_is_bitwise_the_single_launch(ctx, nao, naux, nemb, spin, rows):
    """An AO block of 4 GiB or more (naux nao^2 >= 2^28: nao 500 with naux 1100) exceeds the 32-bit lane offsets of the step-1
    kernel's LDS-DMA addressing; up to round 5 such a system fell to the generic kernels.  Step 1 now runs in ranges of L
    (half1_hot_max_rows).  DMK_ERI_HOT_LCHUNK forces the cut on small shapes: the ERI must be BIT-identical to the single launch
    (every output row is computed by the same instructions either way), on and off the K tile, both step-2 kernels."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    mesh, nk = (3, 1, 1), 3
    npair = nemb * (nemb + 1) // 2
    nblk = spin * (spin + 1) // 2
    rng = np.random.default_rng(nao * naux)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=5)

    def run(cut):
        if cut:
            os.environ["DMK_ERI_HOT_LCHUNK"] = str(rows)
        try:
            eri_dev = ctx.zeros((nblk, npair, npair), np.float64)
            eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, ctx.to_device(Ce), eri_dev, True)
            try:
                assert eng.ring_slots > 0
                for kL in eng.irreducible_kL():
                    eng.run_kL(kL, df)
                ctx.sync()
                return eri_dev.get()
            finally:
                eng.close()
        finally:
            os.environ.pop("DMK_ERI_HOT_LCHUNK", None)
    a, b = run(False), run(True)
    assert np.abs(a).max() > 0 and np.array_equal(a, b)
    A = [0, nemb // 2, nemb - 1]
    want, idx, _ = ES.eri_sample(mesh, 5, Ce, naux, A, [int(k) for k in range(nk) if ES.plan_records(mesh)[0][k] > 0])
    for blk in range(nblk):
        assert np.abs(b[blk][np.ix_(idx, idx)] - want[blk]).max() < 1e-8


def test_ao_block_beyond_4_GiB_takes_the_hot_path(ctx):
    """The real thing: nao 500, naux 1100 -- AO blocks of 4.4 GB, a ring of 70 GB -- two kL of a 2 x 1 x 1 mesh through the block
    ring against the sampled oracle.  Skipped when the device has less than 120 GB free."""
    free, _ = ctx.mem_info()
    if free < 120 * (1 << 30):
        pytest.skip("needs 120 GB of free device memory")
    _run_and_check(ctx, (2, 1, 1), 500, 1100, 40, 1, [0, 1], [0, 19, 39], seed=77, check_planes=True, want_ring=True)


@pytest.mark.parametrize("nao,naux,nemb,spin", [(40, 48, 256, 2), (27, 45, 72, 1), (104, 24, 136, 2)])
def test_real_part_only_step2_of_weight_one_kL(ctx, nao, naux, nemb, spin, monkeypatch):
    """A momentum transfer kL that is its own time-reversal partner (weight 1) contributes Re(Lij)^T Re(Lij) only
    (eri_transform.py:453-455, 464-467), so its blocks run a real-part-only step 2 (Re S = Ur Cr - Ui Ci: two real MFMAs per complex
    block step instead of 3M's three).  The Re planes -- and with them the ERI -- are BIT-identical to the full complex step 2
    (DMK_ERI_RE_ONLY=0: the same two accumulators, the same order), the Im planes of those kL are zero, weight-2 kL are untouched; on
    a 2 x 2 x 2 mesh EVERY kL has weight 1.  Both step-2 kernels, on and off the K tile, and against the sampled oracle."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    for mesh in ((2, 2, 2), (3, 2, 1)):
        nk = int(np.prod(mesh))
        npair = nemb * (nemb + 1) // 2
        nblk = spin * (spin + 1) // 2
        rng = np.random.default_rng(nao + nk)
        Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
        df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=9)

        def run(re_only):
            if not re_only:
                monkeypatch.setenv("DMK_ERI_RE_ONLY", "0")
            else:
                monkeypatch.delenv("DMK_ERI_RE_ONLY", raising=False)
            eri_dev = ctx.zeros((nblk, npair, npair), np.float64)
            eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, ctx.to_device(Ce), eri_dev, True)
            planes = {}
            try:
                assert eng.ring_slots > 0
                for kL in eng.irreducible_kL():
                    eng.run_kL(kL, df)
                    planes[kL] = eng.planes().get()
                ctx.sync()
                return eri_dev.get(), planes, {k: int(eng.weights[k]) for k in planes}
            finally:
                eng.close()
        e_re, p_re, w = run(True)
        e_full, p_full, _ = run(False)
        assert np.array_equal(e_re, e_full) and np.abs(e_re).max() > 0
        assert 1 in w.values() and (mesh != (2, 2, 2) or set(w.values()) == {1})
        for kL, wk in w.items():
            assert np.array_equal(p_re[kL][:, 0], p_full[kL][:, 0])
            if wk == 1:
                assert np.abs(p_re[kL][:, 1]).max() == 0.0 and np.abs(p_full[kL][:, 1]).max() > 0.0
            else:
                assert np.array_equal(p_re[kL][:, 1], p_full[kL][:, 1])
        A = [0, nemb // 2, nemb - 1]
        want, idx, _ = ES.eri_sample(mesh, 9, Ce, naux, A, sorted(w))
        for blk in range(nblk):
            assert np.abs(e_re[blk][np.ix_(idx, idx)] - want[blk]).max() < 1e-8


@pytest.mark.parametrize("gen_stream", ["0", "1"])
def test_resident_push_refuses_an_outstanding_ring_slot(ctx, gen_stream):
    """dmk_eri_push_resident while a ring slot handed out by dmk_eri_ring_slot has not been pushed: the resident launch resets the
    ring's producer state (half, pending event), so the later dmk_eri_push_ring_slot would read the wrong half without waiting for
    its generator -- the library returns DMK_ERR_STATE instead; pushing the slot first (or a new kL) clears the reservation."""
    import ctypes as C
    from libdmet_preview_amd import _lib
    from libdmet_preview_amd.basis_transform import eri_transform as et
    mesh, nk, nao, naux, nemb, spin = (2, 2, 1), 4, 16, 40, 256, 1
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(3)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    os.environ["DMK_ERI_GEN_STREAM"] = gen_stream
    try:
        eri_dev = ctx.zeros((1, npair, npair), np.float64)
        eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, ctx.to_device(Ce), eri_dev, True)
    finally:
        os.environ.pop("DMK_ERI_GEN_STREAM", None)
    try:
        assert eng.ring_slots > 0
        lib = _lib.lib
        kL = eng.irreducible_kL()[0]
        blocks = ctx.zeros((1, naux, nao, nao), np.complex128)
        one = np.zeros(1, dtype=np.int32)
        args = (eng.h, C.c_void_p(blocks.address), 1, one.ctypes.data_as(C.c_void_p),
                one.ctypes.data_as(C.c_void_p), one.ctypes.data_as(C.c_void_p))
        ctx.check(lib.dmk_eri_begin_kL(eng.h, int(kL)))
        ptr, stream = C.c_void_p(), C.c_void_p()
        ctx.check(lib.dmk_eri_ring_slot(eng.h, 0, C.byref(ptr), C.byref(stream)))
        with pytest.raises(_lib.DmkError):
            ctx.check(lib.dmk_eri_push_resident(*args))
        ctx.check(lib.dmk_eri_push_ring_slot(eng.h, 0, 0, 0))          # (the slot's content is whatever the ring held: only the state matters)
        ctx.check(lib.dmk_eri_push_resident(*args))                    # flushes the pushed slot, then its own launch
        ctx.check(lib.dmk_eri_end_kL(eng.h, int(eng.weights[kL])))
        ctx.sync()
    finally:
        eng.close()


def test_randomised_eri_campaign_short():
    """tools/eri_stress.py with a fixed seed: 30 random systems (meshes with odd axes, dimensions on and off the tile sizes, one and two
    spin channels, with and without time reversal, 4-fold and 1-fold results) through get_emb_eri against the oracle at 1e-8.  The long
    campaign (850 systems, worst 1.1e-15) is profiles/r04_e_eri_stress.txt."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEED="20261002", STRESS_TRIALS="30", GRAFT_REPO_ROOT=root)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "eri_stress.py")], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "eri stress ok: 30 systems" in run.stdout


def test_randomised_mean_field_and_bath_campaign_short():
    """tools/meanfield_stress.py with a fixed seed: 60 random lattices (odd and even mesh axes, 2 .. 72 orbitals per cell, restricted and
    unrestricted, T = 0 and T > 0, random valence / virtual splits) through mfd.HF and slater.get_emb_basis against the oracle
    (eigenvalues 1e-10, T = 0 occupations equal, rho 1e-9, bath projector 1e-9 where the singular values are clear of the cut-off).
    The long campaign (1200 lattices) is profiles/r04_e_meanfield_stress.txt."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEED="20261002", STRESS_TRIALS="60", GRAFT_REPO_ROOT=root)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "meanfield_stress.py")], env=env, capture_output=True, text=True,
                         timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "mean-field stress ok: 60 lattices" in run.stdout


def test_randomised_fit_campaign_short():
    """tools/fit_stress.py with a fixed seed: 30 random embedding problems (random meshes, 3 .. 40 orbitals per cell, random valence counts,
    T = 0 and T > 0, full / impurity-only / impurity + bath-diagonal index sets, remove_diag_grad) through EmbFitDevice against
    oracle/restate_fit.py: dV_dparam 1e-12, objective 1e-9, analytic gradient 1e-7; then 15 lattice-fit problems (FullFitDevice against
    FullFit).  The long campaign (800 + 400 problems, worst 1e-12) is
    profiles/r04_e_fit_stress.txt."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEED="20261002", STRESS_TRIALS="30", GRAFT_REPO_ROOT=root)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "fit_stress.py")], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "fit stress ok: 30 embedding problems" in run.stdout and "lattice-fit stress ok: 15 lattice problems" in run.stdout


def test_randomised_iteration_campaign_short():
    """tools/iteration_stress.py with a fixed seed: 25 random systems through the WHOLE device-resident iteration (diag -> bath -> C_ao_emb
    -> DF transform -> J / K -> H1_emb); ERI (1e-8), H1_emb / JK_core (1e-10), C_ao_emb (1e-12) against the oracle on the pipeline's own
    intermediate products, the Freivalds probe, and a second pass bit-identical to the first.  The long campaign (600 systems) is
    profiles/r04_e_iteration_stress.txt."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEED="20261002", STRESS_TRIALS="25", GRAFT_REPO_ROOT=root)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "iteration_stress.py")], env=env, capture_output=True, text=True,
                         timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "iteration stress ok: 25 systems" in run.stdout


def test_randomised_bcs_campaign_short():
    """tools/bcs_stress.py with a fixed seed: 60 random lattices through the BCS / Nambu twin (DiagBdG, bcs.embBasis, bcs_helper folds and
    dV_dparam) against oracle/restate_bcs.py.  The long campaign (1000 lattices) is profiles/r04_e_bcs_stress.txt."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEED="20261002", STRESS_TRIALS="60", GRAFT_REPO_ROOT=root)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "bcs_stress.py")], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "bcs stress ok: 60 lattices" in run.stdout


@pytest.mark.parametrize("tool,trials,needle", [("algebra_stress.py", 100, "algebra stress ok: 100 rounds"), ("ham_stress.py", 60, "ham stress ok: 60 lattices"),
                                                ("cderi_stress.py", 40, "cderi stress ok: 40 DF tensors"), ("gso_stress.py", 50, "gso stress ok: 50 rounds")])
def test_randomised_algebra_and_ham_campaigns_short(tool, trials, needle):
    """tools/algebra_stress.py (folds on meshes with axes 1 .. 7, basis algebra with every spin-dimension combination, eri_restore) and
    tools/ham_stress.py (J / K on every ERI format, one-body folds, get_emb_Ham with the reference's option combinations) with a fixed seed
    against the oracle; tools/cderi_stress.py (writer and reader of the cderi layout, ERI fed from the container through the host-feed
    path); tools/gso_stress.py (the GSO twins: spinless.get_emb_basis, get_emb_eri_gso).  The long campaigns (2000 rounds / 1500 lattices /
    800 DF tensors / 1000 rounds, worst 6e-14) are profiles/r04_e_algebra_ham_stress.txt, r04_e_cderi_stress.txt, r04_e_gso_stress.txt."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEED="20261002", STRESS_TRIALS=str(trials), GRAFT_REPO_ROOT=root)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", tool)], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert needle in run.stdout


def test_randomised_hot_kernel_campaign_short():
    """tools/hot_stress.py with a fixed seed: 15 random production-like shapes (nao 16 .. 208, naux 32 .. 640, nemb 32 .. 320, one and two spin
    channels) through the hot kernels -- whole kL through the block ring, planes and ERI on sampled orbital pairs against the sampled C
    oracle, Freivalds probe on every pair row.  The long campaign (480 shapes) is profiles/r04_e_hot_stress.txt."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEED="20261002", STRESS_TRIALS="15", GRAFT_REPO_ROOT=root)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "hot_stress.py")], env=env, capture_output=True, text=True, timeout=1200)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "hot stress ok: 15 shapes" in run.stdout


def test_engines_with_padded_geometry_give_their_memory_back(ctx):
    """Sixty pipelines in a row on a shape off every tile (padded copy of C_ao_emb, padded planes, the compact copy dmk_eri_planes hands
    out): the free device memory after the 10th and after the 60th is the same -- dmk_eri_finish frees what dmk_eri_begin and
    dmk_eri_planes allocated (found in round 6: the compact plane copy was not freed)."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    mesh, nk, nao, naux, nemb, spin = (2, 2, 1), 4, 27, 45, 41, 2
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(0)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=3)
    mark = None
    for it in range(60):
        eri = ctx.zeros((3, npair, npair), np.float64)
        d_C = ctx.to_device(Ce)
        eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, d_C, eri, True)
        kls = eng.irreducible_kL()
        eng.run_kL(kls[0], df)
        pl = eng.planes().get()
        assert pl.shape == (spin, 2, naux, npair)
        eng.close()
        eri.free()
        d_C.free()
        ctx.sync()
        ctx.trim()
        free, _ = ctx.mem_info()
        if it == 9:
            mark = free
    assert mark - free < (1 << 20), (mark, free)             # (it may grow: blocks parked by earlier tests are handed back over time)


def test_randomised_twins_campaign_short():
    """tools/twins_stress.py with a fixed seed: 60 random lattices through mfd.HFB / GHF, spinless.get_emb_Ham, the GSO embedding and
    lattice fits, the cell-resolved potential's dV/dparam and the k-resolved lattice fit against the oracle.  The long campaign (1600
    lattices, worst 1e-11) is profiles/r06_t_twins_stress_1600_trials.txt."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_SEED="20261003", STRESS_TRIALS="60", GRAFT_REPO_ROOT=root)
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "twins_stress.py")], env=env, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    out = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert out["trials"] == 60 and len(out["worst"]) >= 12 and max(out["worst"].values()) < 1e-8
