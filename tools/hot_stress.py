"""Randomised campaign of the HOT kernels at production-like tile geometry (K6 step 1 wide / narrow tiles, K6 step 2 for nemb = 256, K6b the
table-driven step 2 for any other embedding dimension, K7 the symmetric / rectangular contraction with its band cuts and mirrored
stores): whole momentum transfers kL through the block ring on random shapes -- nao 16 .. 208 (two in three OFF the K tile of 8: the
zero-padded K loop of round 6), naux 32 .. 640,
nemb 32 .. 320 (256 one time in three), one and two spin channels, meshes of 3 .. 8 k-points -- checked as
tests/test_gpu_production.py checks C5 / C4: the Lij_s4 planes and the ERI on a sample of embedding-orbital pairs against the sampled
C oracle (oracle/eri_sample.py: exact entries, all auxiliary rows), the Freivalds probe of the contraction on EVERY pair row, and the
symmetry of the same-spin blocks.
    STRESS_SEED=1 STRESS_TRIALS=12 python tools/hot_stress.py          (test infrastructure: imports the oracle and a test helper)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import eri_sample as ES
from libdmet_preview_amd import _lib
from tests.test_gpu_production import _run_and_check

ctx = _lib.get_ctx()
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "12"))
t0, worst, log = time.time(), 0.0, []
for trial in range(trials):
    while True:
        mesh = [(2, 2, 1), (3, 1, 1), (2, 2, 2), (3, 2, 1), (4, 1, 1), (5, 1, 1)][int(rng.integers(0, 6))]
        big = os.environ.get('STRESS_BIG') == '1'
        nao = 8 * int(rng.integers(2, 33 if big else 27))
        if rng.random() < 0.67:
            nao = max(16, nao - int(rng.integers(1, 8)))          # off the K tile
        naux = int(rng.integers(32, 1201 if big else 641))
        nemb = 256 if rng.random() < 0.33 else int(rng.integers(32, 449 if big else 321))
        spin = int(rng.integers(1, 3))
        npair = nemb * (nemb + 1) // 2
        nk = mesh[0] * mesh[1] * mesh[2]
        eri_gb = spin * (spin + 1) // 2 * npair * npair * 8 / 2 ** 30
        half_gflop = nk * naux * nao * nemb * (nao + nemb) * 8e-9 * spin * 2
        if eri_gb < (120 if big else 40) and half_gflop < (4e4 if big else 6e3) and naux * nao >= 1024:
            break
    w, by = ES.plan_records(mesh)
    kls = sorted(by)
    w2 = [k for k in kls if w[k] == 2]
    w1 = [k for k in kls if w[k] == 1]
    pick = ([int(rng.choice(w2))] if w2 else []) + ([int(rng.choice(w1))] if w1 else [])
    edges = sorted({0, nemb - 1, min(nemb - 1, 16 * int(rng.integers(0, max(1, nemb // 16)))), min(nemb - 1, 127), min(nemb - 1, 128),
                    int(rng.integers(0, nemb)), int(rng.integers(0, nemb))})
    seed = int(rng.integers(1, 1 << 30))
    t1 = time.time()
    wp = _run_and_check(ctx, mesh, nao, naux, nemb, spin, pick, edges, seed=seed)
    worst = max(worst, wp)
    log.append("   mesh %s nao %3d naux %3d nemb %3d spin %d kL %s orbitals %s: planes %.1e  (%.1f s)" % (mesh, nao, naux, nemb, spin, pick, edges, wp, time.time() - t1))
    print(log[-1], flush=True)
print("hot stress ok: %d shapes in %.0f s, worst relative plane error %.1e (ERI <= 1e-11 relative on the sample, Freivalds <= 1e-10 on every pair row)"
      % (trials, time.time() - t0, worst))
