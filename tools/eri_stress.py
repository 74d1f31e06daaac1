"""Randomised differential campaign of the DF AO -> embedding ERI transform: the HIP path (get_emb_eri through the C ABI) against the
oracle's restatement of get_emb_eri_fast_gdf (oracle/restate.py) on random small systems -- meshes with odd and even axes, AO /
auxiliary / embedding dimensions on and off the tile sizes (every kernel family: generic zgemm, flattened step 1, table-driven and
nemb = 256-free step 2, symmetric and rectangular contraction), one and two spin channels, 4-fold and 1-fold results.
    STRESS_SEED=1 STRESS_TRIALS=100 python tools/eri_stress.py           (test infrastructure: imports the oracle)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from libdmet_preview_amd import _lib, synth
from libdmet_preview_amd.basis_transform import eri_transform as et
from libdmet_preview_amd.system.lattice import _UnitCell

ctx = _lib.get_ctx()
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "60"))
budget = float(os.environ.get("STRESS_ORACLE_MFLOP", "4e4"))       # keep one oracle run in seconds
worst, t0, done = 0.0, time.time(), 0
for trial in range(trials):
    while True:
        mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.45, 0.3, 0.15, 0.1]))
        nk = mesh[0] * mesh[1] * mesh[2]
        nao = int(rng.choice([int(rng.integers(3, 41)), 8 * int(rng.integers(1, 6))]))
        naux = int(rng.integers(3, 72))
        nemb = int(rng.choice([int(rng.integers(3, 30)), int(rng.integers(32, 90)), 16 * int(rng.integers(2, 6))]))
        spin = int(rng.integers(1, 3))
        npair = nemb * (nemb + 1) // 2
        cost = nk * nk * naux * nao * nemb * (nao + nemb) * 8e-6 * spin + nk * naux * npair * npair * 2e-6 * 3
        if nk <= 27 and cost <= budget and npair * npair * 8 * 3 < 2e9:
            break
    seed = int(rng.integers(1, 1 << 40))
    sym = int(rng.choice([4, 4, 1])) if nemb <= 24 else 4
    ks = R.make_kpts_scaled(mesh)
    cell = _UnitCell(nao)
    mydf = et.GDFPhilox(cell.get_abs_kpts(ks), naux, nao, seed=seed)
    C = synth.make_C_ao_lo(mesh, nao, nao, spin=spin, seed=int(rng.integers(1, 1000)))
    basis = rng.standard_normal((spin, nk, nao, nemb)) / np.sqrt(nao)
    tr = bool(rng.random() < 0.8)
    unit = bool(rng.random() < 0.1) and nemb == nao
    kw = dict(symmetry=sym, t_reversal_symm=tr, unit_eri=unit)
    got = et.get_emb_eri(cell, mydf, C_ao_lo=C, basis=None if unit else basis, **kw)
    ref = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: R.df_block_philox(seed, i, j, naux, nao), naux, nao, C_ao_lo=C,
                                 basis=None if unit else basis, **kw)
    assert got.shape == ref.shape, (trial, mesh, nao, naux, nemb, spin, kw, got.shape, ref.shape)
    err = float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max()))
    worst = max(worst, err)
    assert err < 1e-8, (trial, mesh, nao, naux, nemb, spin, kw, err)
    done += 1
print("eri stress ok: %d systems in %.0f s, worst max|device - oracle| / max(1, |ref|) = %.2e" % (done, time.time() - t0, worst))
