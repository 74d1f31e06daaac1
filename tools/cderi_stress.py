"""Randomised differential campaign of the on-disk DF tensor layout: transform_gdf_to_lo (the writer of the
PySCF `cderi` layout: pairs i >= j only, packed triangles at ki == kj, real at Gamma, time-reversed partners as conjugates), the reader
(CderiProvider.get_block: swap -> conjugate transpose, Hermitian unpack) and the ERI transform fed from such a container through the
host-feed path (pinned buffers, copy stream, swapped pairs conjugate-transposed on the device), against oracle/restate_cderi.py and
oracle/restate.py (reference: basis_transform/eri_transform.py:195-227, 1312-1427) on random physical DF tensors.
    STRESS_SEED=1 STRESS_TRIALS=30 python tools/cderi_stress.py          (test infrastructure: imports the oracle)"""
import os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from oracle import restate_cderi as Cd
from libdmet_preview_amd import synth
from libdmet_preview_amd.basis_transform import eri_transform as et
from libdmet_preview_amd.system.lattice import _UnitCell

rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "30"))
worst = {"writer": 0.0, "reader": 0.0, "eri": 0.0}
t0 = time.time()
tmp = tempfile.mkdtemp(prefix="cderi_stress_")
for trial in range(trials):
    mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.5, 0.3, 0.12, 0.08]))
    nk = mesh[0] * mesh[1] * mesh[2]
    if nk < 2 or nk > 12:
        mesh, nk = (2, 2, 1), 4
    nao, naux = int(rng.integers(2, 9)), int(rng.integers(2, 12))
    nlo = int(rng.integers(max(1, nao - 2), nao + 1))
    # a physical DF tensor: real, symmetric under (R1 p) <-> (R2 s), decaying with the cell index
    W0 = rng.standard_normal((naux, nk, nao, nk, nao)) * np.exp(-0.4 * np.arange(nk))[None, :, None, None, None] \
        * np.exp(-0.4 * np.arange(nk))[None, None, None, :, None]
    W0 = W0 + W0.transpose(0, 3, 4, 1, 2)
    ks = R.make_kpts_scaled(mesh)
    blocks = R.df_blocks_from_W0(W0, mesh, ks)
    C = synth.make_C_ao_lo(mesh, nao, nlo, spin=1, seed=int(rng.integers(1, 1000)))[0]
    cell = _UnitCell(nao)
    kpts = cell.get_abs_kpts(ks)
    mydf = et.GDFMemory(kpts, dict(blocks), naux, cell=cell)
    for tr in (True, False):
        fn = os.path.join(tmp, "lo_%d_%d" % (trial, tr))
        prov = et.transform_gdf_to_lo(mydf, C, fname=fn, t_reversal_symm=tr)
        ref, mask = Cd.transform_gdf_to_lo(lambda i, j: blocks[(i, j)], ks, 2.0 * np.pi * ks, naux, C, t_reversal_symm=tr)
        assert sorted(prov.feri.keys()) == sorted(ref.keys()), (trial, mesh, tr, "datasets")
        for k in ref:
            assert prov.feri[k].shape == ref[k].shape and prov.feri[k].dtype.kind == ref[k].dtype.kind, (trial, k)
            if k != "j3c-kptij":
                e = float(np.abs(prov.feri[k] - ref[k]).max())
                assert e < 1e-11, (trial, mesh, tr, k, e)
                worst["writer"] = max(worst["writer"], e)
        # the reader on the REFERENCE's container
        rp = et.CderiProvider(ref, kpts, nlo)
        for i in range(nk):
            for j in range(nk):
                e = float(np.abs(rp.get_block(i, j) - Cd.load_block(ref, nk, nlo, i, j)).max())
                assert e < 1e-12, (trial, mesh, tr, i, j, e)
                worst["reader"] = max(worst["reader"], e)
    # the ERI transform fed from the container (host-feed path) against the oracle on the oracle's reading of it
    nemb = int(rng.choice([int(rng.integers(2, 12)), 32 + int(rng.integers(0, 9))]))
    spin = int(rng.integers(1, 3))
    basis = rng.standard_normal((spin, nk, nlo, nemb)) / np.sqrt(nlo)
    cell_lo = _UnitCell(nlo)
    got = et.get_emb_eri(cell_lo, et.CderiProvider(dict(np.load(fn + ".npz")), kpts, nlo), basis=basis)
    want = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: Cd.load_block(ref, nk, nlo, i, j), naux, nlo, C_ao_lo=None, basis=basis)
    e = float(np.abs(got - np.asarray(want).reshape(got.shape)).max()) / max(1.0, float(np.abs(want).max()))
    assert e < 1e-8, (trial, mesh, nao, nlo, naux, nemb, spin, e)
    worst["eri"] = max(worst["eri"], e)
print("cderi stress ok: %d DF tensors in %.0f s, worst: writer %.1e, reader %.1e, ERI through the container %.1e"
      % (trials, time.time() - t0, worst["writer"], worst["reader"], worst["eri"]))
