"""Lab: what does an AO dimension that is NOT a multiple of the K tile cost?  One w = 2 kL of a C5- / C4-like system through
et.EriEngine with the Philox block generator, hot shapes next to their off-tile neighbours (executed TF from the library's flop
accounting / wall time)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdmet_preview_amd import _lib
from libdmet_preview_amd.basis_transform import eri_transform as et

ctx = _lib.get_ctx()


def run(mesh, nao, naux, nemb, spin, kL=1, reps=2):
    nk = int(np.prod(mesh))
    npair = nemb * (nemb + 1) // 2
    nblk = spin * (spin + 1) // 2
    rng = np.random.default_rng(7)
    Ce = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / np.sqrt(nao)
    C_dev = ctx.to_device(Ce)
    eri_dev = ctx.zeros((nblk, npair, npair), np.float64)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=3)
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    try:
        best = 1e30
        for r in range(reps + 1):
            ctx.sync()
            t0 = time.perf_counter()
            n = eng.run_kL(kL, df)
            ctx.sync()
            dt = time.perf_counter() - t0
            if r > 0:
                best = min(best, dt)
        half = spin * (8.0 * naux * nao * nao * nemb + 8.0 * naux * nao * nemb * nemb) * n
        contr = nblk * 2 * 2.0 * naux * npair * npair
        print("mesh %s nao %4d naux %4d nemb %4d spin %d ring %2d: %3d blocks %8.2f ms  %6.2f TF algorithmic (half transform + contraction)"
              % (mesh, nao, naux, nemb, spin, eng.ring_slots, n, best * 1e3, (half + contr) / best / 1e12), flush=True)
    finally:
        eng.close()
        eri_dev.free()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c5", "c4"]
    if "c5" in which:
        for nao in (200, 203, 196, 201):
            run((6, 6, 6), nao, 800, 256, 2)
        run((6, 6, 6), 200, 800, 250, 2)
    if "c4" in which:
        for nao in (104, 100, 101, 107):
            run((4, 4, 4), nao, 416, 136, 1)
        run((4, 4, 4), 104, 411, 131, 1)
