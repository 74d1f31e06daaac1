"""
GPU parity of the drop-in boundary with the argument the REFERENCE passes: a pyscf.pbc.df.GDF-shaped
object -- only `_cderi`, `kpts`, `cell`, `blockdim`, `max_memory`; no `load_block` -- goes through the PATCHED reference name
`libdmet.routine.slater.get_emb_eri` (routine/slater.py:32-33 binds it at import; :451 calls it with `lattice.df`) and must
reproduce the ERIs the reference's own get_emb_eri_fast_gdf produced for the same DF tensor (golden G6) to the north star's
1e-8 max-abs.  The `_cderi` container is written in the layout PySCF / the reference's transform_gdf_to_lo use (pinned by golden
G13): as an open mapping, as a path string opened through a lazily imported h5py (a stand-in module whose File returns the
mapping: there is no h5py on this box), as the reference-written LO container of G13 itself, and through `feri`.
Reference: basis_transform/eri_transform.py:68-94 (dispatch), :159-227 (readers), :260-261 (feri); eri_transform_mpi.py:57-62.
"""
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import restate as R                        # the checker
from tests.df_duck import DuckGDF, ao_container, fake_h5py, patched_reference

CASES = [("m311", 1), ("m411", 2), ("m231", 2), ("m222", 1), ("mid411", 2), ("mid221", 1)]
TOL = 1e-8


@pytest.fixture(scope="module")
def ctx():
    from libdmet_preview_amd import _lib
    return _lib.get_ctx()


def _setup(g, name):
    from libdmet_preview_amd.system.lattice import _UnitCell
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    W0 = g[name + "/W0"]
    naux, nao = W0.shape[0], W0.shape[2]
    ks = R.make_kpts_scaled(mesh)
    cell = _UnitCell(nao)
    kabs = cell.get_abs_kpts(ks)
    blocks = R.df_blocks_from_W0(W0, mesh, ks)
    return mesh, cell, kabs, ao_container(blocks, ks, kabs, naux, nao), naux


@pytest.mark.parametrize("name,spin", CASES)
def test_gdf_object_through_the_patched_reference_name(ctx, golden, name, spin):
    g = golden("G6_eri.npz")
    mesh, cell, kabs, cont, naux = _setup(g, name)
    st = "%s/s%d" % (name, spin)
    C, basis = g[st + "/C_ao_lo"], g[st + "/basis"]
    duck = DuckGDF(cell, kabs, cont, blockdim=max(1, naux // 2 + 1))
    with patched_reference() as rs:
        for tr in (True, False):
            e = rs.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis, t_reversal_symm=tr)
            ref = g[st + "/eri_%s" % ("tr" if tr else "notr")]
            assert e.shape == ref.shape and np.abs(e - ref).max() < TOL, (tr, np.abs(e - ref).max())
        e1 = rs.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis, symmetry=1)
        assert np.abs(e1 - g[st + "/eri_s1"]).max() < TOL
        eu = rs.get_unit_eri(cell, duck, C_ao_lo=C)
        assert np.abs(eu - g[st + "/eri_unit"]).max() < TOL
        import libdmet.basis_transform.eri_transform as ret          # the defining module's copy is the same function
        Ck = R.multiply_basis(C, R.get_basis_k(basis, R.get_phase_R2k(mesh, R.make_kpts_scaled(mesh))))
        ec = ret.get_emb_eri_fast_gdf(cell, duck, C_ao_eo=Ck)
        assert np.abs(ec - g[st + "/eri_C_ao_eo"]).max() < TOL
    assert duck._cderi is cont and not hasattr(duck, "load_block")    # the caller's object is left as it was


def test_cderi_path_string_and_feri(ctx, golden, monkeypatch, tmp_path):
    g = golden("G6_eri.npz")
    name, spin = "m231", 2
    mesh, cell, kabs, cont, naux = _setup(g, name)
    st = "%s/s%d" % (name, spin)
    C, basis, ref = g[st + "/C_ao_lo"], g[st + "/basis"], g[st + "/eri_tr"]
    h5 = fake_h5py({"/scratch/gdf_ints.h5": cont})
    monkeypatch.setitem(sys.modules, "h5py", h5)
    with patched_reference() as rs:
        duck = DuckGDF(cell, kabs, "/scratch/gdf_ints.h5")
        e = rs.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis)
        assert np.abs(e - ref).max() < TOL
        assert h5.opened == ["/scratch/gdf_ints.h5"] and h5.closed == h5.opened      # opened once, closed when done
        # feri stands in while _cderi is None and is stored on the object (eri_transform.py:260-261)
        duck2 = DuckGDF(cell, kabs, None)
        e = rs.get_emb_eri(cell, duck2, C_ao_lo=C, basis=basis, feri="/scratch/gdf_ints.h5")
        assert np.abs(e - ref).max() < TOL and duck2._cderi == "/scratch/gdf_ints.h5"
        # a set _cderi wins over feri
        e = rs.get_emb_eri(cell, DuckGDF(cell, kabs, cont), C_ao_lo=C, basis=basis, feri="/nowhere.h5")
        assert np.abs(e - ref).max() < TOL
        # out of core with the object: same numbers in the reference's (aa, bb, ab) file order
        out = rs.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis, incore=False, fout=str(tmp_path / "H2"))
        assert np.abs(np.asarray(out["ccdd"])[[0, 2, 1]] - ref).max() < TOL
    # the .npz a box without h5py writes (transform_gdf_to_lo(fname=...)) is read back through the same branch
    monkeypatch.setitem(sys.modules, "h5py", None)
    from libdmet_preview_amd.basis_transform import eri_transform as et
    fn = str(tmp_path / "cderi.npz")
    np.savez(fn, **cont)
    e = et.get_emb_eri(cell, DuckGDF(cell, kabs, fn), C_ao_lo=C, basis=basis)
    assert np.abs(e - ref).max() < TOL
    with pytest.raises(NotImplementedError, match="h5py"):
        et.get_emb_eri(cell, DuckGDF(cell, kabs, "/scratch/gdf_ints.h5"), C_ao_lo=C, basis=basis)


@pytest.mark.parametrize("name", ["m311", "m221", "m231"])
def test_reference_written_lo_container_as_cderi(ctx, golden, name):
    """The container the REFERENCE's transform_gdf_to_lo wrote (golden G13, both with and without its time-reversal mask) as
    the `_cderi` of a GDF-shaped object in the LO basis: the ERI equals the one from the AO tensor + C_ao_lo."""
    from libdmet_preview_amd.system.lattice import _UnitCell
    from tests.test_oracle_cderi import inputs, golden_container
    from oracle import restate_cderi as Cd
    g = golden("G13_cderi.npz")
    mesh, ks, kabs, blocks, naux, C = inputs(g, name)
    nk, nao, nlo = C.shape
    rng = np.random.default_rng(11)
    basis = rng.standard_normal((1, nk, nlo, 5))
    # oracle: reader restatement on the golden container -> LO blocks -> restated transform
    with patched_reference() as rs:
        for tag in ("tr", "notr"):
            feri = golden_container(g, "%s/%s" % (name, tag))
            lo_blocks = {(i, j): Cd.load_block(feri, nk, nlo, i, j) for i in range(nk) for j in range(nk)}
            ref = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: lo_blocks[(i, j)], naux, nlo, basis=basis)
            cell_lo = _UnitCell(nlo)
            e = rs.get_emb_eri(cell_lo, DuckGDF(cell_lo, cell_lo.get_abs_kpts(ks), feri), basis=basis)
            assert e.shape == ref.shape and np.abs(e - ref).max() < TOL * max(1.0, np.abs(ref).max()), tag


def test_use_mpi_takes_the_reference_route(ctx, golden, tmp_path):
    """get_emb_eri(..., use_mpi=True) hands `mydf._cderi` and `mydf.kpts` to the MPI twin (eri_transform.py:76-87,
    eri_transform_mpi.py:57-62): one rank, gloo (the exchange itself is covered by tests/test_gpu_dist.py)."""
    import torch.distributed as td
    g = golden("G6_eri.npz")
    name, spin = "m222", 1
    mesh, cell, kabs, cont, naux = _setup(g, name)
    st = "%s/s%d" % (name, spin)
    C, basis, ref = g[st + "/C_ao_lo"], g[st + "/basis"], g[st + "/eri_tr"]
    duck = DuckGDF(cell, kabs, cont)
    assert not td.is_initialized()
    td.init_process_group("gloo", init_method="file://" + str(tmp_path / "pg"), rank=0, world_size=1)
    try:
        with patched_reference() as rs:
            e = rs.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis, use_mpi=True)
            assert np.abs(e - ref).max() < TOL
            from libdmet_preview_amd.basis_transform import eri_transform_mpi as etm
            e = etm.get_emb_eri_fast_gdf(cell, cont, kabs, C_ao_lo=C, basis=basis)
            assert np.abs(e - ref).max() < TOL
    finally:
        td.destroy_process_group()


def test_resident_df_blocks_bit_identical_to_the_ring():
    """et.GDFResident (dmk_eri_push_resident): the blocks of a kL shard kept in HBM and read in place give the SAME bits as the
    same blocks fed through the block ring, for the whole config and for a sub-shard, both spin counts; a kL outside the
    resident shard falls back to the generator."""
    import numpy as np
    from libdmet_preview_amd import _lib, pipeline
    ctx = _lib.get_ctx()
    for over in (dict(mesh=(3, 2, 2), nlo=40, naux=48, nval=24, spin=2), dict(mesh=(4, 4, 1), nlo=104, naux=64, nval=32, spin=1),
                 dict(mesh=(3, 2, 1), nlo=42, naux=48, nval=23, spin=2)):             # (the last: an AO dimension off the K tile)
        sysm = pipeline.SyntheticSystem.from_workload(ctx, "C4", **over)
        ref = pipeline.iteration(ctx, sysm, emb_ham=False)
        eri_ref = ref["eri"].get()
        assert sysm.make_df_resident(None, 0.45) > 0 and sysm.df_resident.nblocks == ref["nblocks"]
        got = pipeline.iteration(ctx, sysm, emb_ham=False)
        assert got["nblocks"] == ref["nblocks"] and np.array_equal(got["eri"].get(), eri_ref)
        # a shard: the resident half of the kL list from HBM, the rest from the generator -- still the same sum
        from libdmet_preview_amd.basis_transform import eri_transform as et
        w, _ = et.eri_plan(sysm.mesh, True)
        irr = [k for k in range(len(w)) if w[k] > 0]
        sysm.make_df_resident(irr[: len(irr) // 2], 0.45)
        mixed = pipeline.iteration(ctx, sysm, emb_ham=False)
        assert np.abs(mixed["eri"].get() - eri_ref).max() <= 1e-15 * np.abs(eri_ref).max()
        sysm.df_resident.free()
        sysm.df_resident = None


@pytest.mark.parametrize("name,spin", [("m231", 2), ("mid411", 2), ("m222", 1)])
def test_make_df_resident_through_the_entry_points(ctx, golden, name, spin):
    """et.make_df_resident(cell, <the reference's GDF-shaped object>) -> a provider whose blocks sit in HBM; get_emb_eri /
    get_unit_eri with it reproduce golden G6 (1e-8) and the ERIs of the file-backed object to rounding -- Gamma-centred meshes
    (integer plan) and shifted ones (general plan), with and without time reversal."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = golden("G6_eri.npz")
    mesh, cell, kabs, cont, naux = _setup(g, name)
    st = "%s/s%d" % (name, spin)
    C, basis = g[st + "/C_ao_lo"], g[st + "/basis"]
    duck = DuckGDF(cell, kabs, cont, blockdim=max(1, naux // 2 + 1))
    for tr in (True, False):
        res = et.make_df_resident(cell, duck, t_reversal_symm=tr)
        assert res.nblocks > 0 and not hasattr(duck, "load_block")
        e = et.get_emb_eri(cell, res, C_ao_lo=C, basis=basis, t_reversal_symm=tr)
        ref = g[st + "/eri_%s" % ("tr" if tr else "notr")]
        assert e.shape == ref.shape and np.abs(e - ref).max() < TOL
        assert not getattr(res, "_warned_plan", False)          # the blocks were read in place: the transform's plan is the stored one
        e_file = et.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis, t_reversal_symm=tr)
        assert np.abs(e - e_file).max() <= 1e-14 * max(1.0, np.abs(ref).max())
        if tr:
            eu = et.get_unit_eri(cell, res, C_ao_lo=C)
            assert np.abs(eu - g[st + "/eri_unit"]).max() < TOL
        res.close()
    with pytest.raises(MemoryError):
        et.make_df_resident(cell, duck, max_fraction_of_free=1e-12)


@pytest.mark.parametrize("name,spin", [("m231", 2), ("m222", 1)])
def test_resident_blocks_of_another_plan_are_not_read_in_place(ctx, golden, name, spin):
    """A GDFResident laid out under one plan (time reversal on) handed to a transform with ANOTHER plan (time reversal off: about
    twice the records per kL) and the other way round: group_ptr is a bare offset into the stored order, so the engine compares its
    (ki, kj) list per kL with the stored one (GDFResident.matches) and reads the source instead -- the ERI is still golden G6."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = golden("G6_eri.npz")
    mesh, cell, kabs, cont, naux = _setup(g, name)
    st = "%s/s%d" % (name, spin)
    C, basis = g[st + "/C_ao_lo"], g[st + "/basis"]
    duck = DuckGDF(cell, kabs, cont, blockdim=max(1, naux // 2 + 1))
    for tr_stored in (True, False):
        res = et.make_df_resident(cell, duck, t_reversal_symm=tr_stored)
        for tr in (not tr_stored, tr_stored):
            e = et.get_emb_eri(cell, res, C_ao_lo=C, basis=basis, t_reversal_symm=tr)
            ref = g[st + "/eri_%s" % ("tr" if tr else "notr")]
            assert np.abs(e - ref).max() < TOL, (tr_stored, tr)
        assert getattr(res, "_warned_plan", False)
        kL = max(res.pairs, key=lambda k: len(res.pairs[k]))
        mine = res.pairs[kL]
        assert res.matches(kL, mine) and not res.matches(kL, mine[:-1]) and not res.matches(kL, np.roll(mine, 1, axis=0) + 1)
        res.close()


def test_partial_residency_streams_the_rest(ctx, golden):
    """make_df_resident(partial=True) with a budget that holds only some kL: the held ones are read in place, the others from the
    source (here the file-backed GDF-shaped object, host-fed) -- the ERI is the one of the fully streamed transform, bit for bit."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = golden("G6_eri.npz")
    name, spin = "m222", 1
    mesh, cell, kabs, cont, naux = _setup(g, name)
    st = "%s/s%d" % (name, spin)
    C, basis = g[st + "/C_ao_lo"], g[st + "/basis"]
    duck = DuckGDF(cell, kabs, cont, blockdim=max(1, naux // 2 + 1))
    full = et.make_df_resident(cell, duck)
    free, _ = ctx.mem_info()
    blk = full.block_bytes
    part = et.make_df_resident(cell, duck, max_fraction_of_free=(0.5 * full.nblocks * blk + blk) / free, partial=True)
    assert 0 < part.nblocks < full.nblocks == part.nblocks_shard and len(part.offset) < len(full.offset)
    e_full = et.get_emb_eri(cell, full, C_ao_lo=C, basis=basis)
    e_part = et.get_emb_eri(cell, part, C_ao_lo=C, basis=basis)
    e_file = et.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis)
    ref = g[st + "/eri_tr"]
    assert np.abs(e_part - ref).max() < TOL and np.array_equal(e_part, e_full) and np.array_equal(e_part, e_file)
    full.close()
    part.close()


def test_resident_df_without_touching_the_script(ctx, golden, monkeypatch):
    """patch.install(resident_df=True): the reference-named get_emb_eri called twice with the SAME GDF-shaped object loads its blocks
    into HBM once (one GDFResident built), both calls reproduce golden G6; a different time-reversal flag is a different plan and
    gets its own copy; the copies go away with drop_resident() and the switch is undone by uninstall()."""
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = golden("G6_eri.npz")
    name, spin = "m231", 2
    mesh, cell, kabs, cont, naux = _setup(g, name)
    st = "%s/s%d" % (name, spin)
    C, basis = g[st + "/C_ao_lo"], g[st + "/basis"]
    duck = DuckGDF(cell, kabs, cont, blockdim=max(1, naux // 2 + 1))
    built = []
    real_init = et.GDFResident.__init__

    def counting_init(self, *a, **k):
        built.append(1)
        return real_init(self, *a, **k)
    monkeypatch.setattr(et.GDFResident, "__init__", counting_init)
    assert et.RESIDENT_DF is False
    with patched_reference(resident_df=True) as rs:
        assert et.RESIDENT_DF is True
        e1 = rs.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis)
        e2 = rs.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis)
        assert len(built) == 1 and np.array_equal(e1, e2) and np.abs(e1 - g[st + "/eri_tr"]).max() < TOL
        e3 = rs.get_emb_eri(cell, duck, C_ao_lo=C, basis=basis, t_reversal_symm=False)
        assert len(built) == 2 and np.abs(e3 - g[st + "/eri_notr"]).max() < TOL
        eu = rs.get_unit_eri(cell, duck, C_ao_lo=C)
        assert len(built) == 2 and np.abs(eu - g[st + "/eri_unit"]).max() < TOL
        assert len(et._resident_cache) == 2
        et.drop_resident()
        assert len(et._resident_cache) == 0
    assert et.RESIDENT_DF is False and not hasattr(duck, "load_block")
