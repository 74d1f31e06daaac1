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
        best, fam = 1e30, ""
        for r in range(reps + 1):
            ctx.sync()
            ctx.profile(True)
            ctx.profile_read(reset=True)
            ctx.profile_read_flops(reset=True)
            t0 = time.perf_counter()
            n = eng.run_kL(kL, df)
            ctx.sync()
            dt = time.perf_counter() - t0
            ms, fl = ctx.profile_read(reset=True), ctx.profile_read_flops(reset=True)
            ctx.profile(False)
            if r > 0 and dt < best:
                best = dt
                # executed fraction of the FP64 matrix peak per kernel family (flop issued to the pipe / HIP-event time)
                fam = "  ".join("%s %.2f" % (k.replace("zgemm_", ""), fl[k] / (ms[k][0] * 1e-3) / 78.6e12)
                                for k in ("zgemm_half1", "zgemm_half2", "dgemm") if ms.get(k, (0, 0))[0] > 0)
        half = spin * (8.0 * naux * nao * nao * nemb + 8.0 * naux * nao * nemb * nemb) * n
        contr = nblk * 2 * 2.0 * naux * npair * npair
        print("mesh %s nao %4d naux %4d nemb %4d spin %d ring %2d: %3d blocks %8.2f ms  %6.2f TF algorithmic   executed / peak: %s"
              % (mesh, nao, naux, nemb, spin, eng.ring_slots, n, best * 1e3, (half + contr) / best / 1e12, fam), flush=True)
    finally:
        eng.close()
        eri_dev.free()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c5", "c4"]
    if "c5" in which:
        for nao in (200, 203, 196, 201):
            run((6, 6, 6), nao, 800, 256, 2)
        run((6, 6, 6), 200, 800, 250, 2)
    if "scan" in which:
        # a scan over shapes a real basis set produces (13 / 26 / 52 / 104 functions per cell and neighbours, naux ~ 4 nao)
        for nao, naux, nemb, spin in ((26, 110, 40, 1), (52, 230, 72, 2), (78, 330, 100, 1), (104, 416, 136, 2), (130, 560, 180, 1),
                                     (150, 640, 200, 2), (203, 811, 250, 2), (203, 811, 300, 1), (260, 1040, 330, 1), (300, 1210, 400, 1)):
            run((4, 4, 2) if nao <= 104 else (3, 3, 2), nao, naux, nemb, spin)
    if "tab" in which:
        # the table-driven step 2 over embedding dimensions (DMK_ERI_TAB_SEG / DMK_ERI_TAB256 are read when the table is built:
        # one process per setting)
        for nemb in (200, 208, 224, 240, 250, 256, 272, 300, 330, 400):
            run((2, 2, 2), 200, 800, nemb, 1)
    if "c4" in which:
        for nao in (104, 100, 101, 107):
            run((4, 4, 4), nao, 416, 136, 1)
        run((4, 4, 4), 104, 411, 131, 1)
