"""Micro-benchmark of dmk_jk_s4 at the C5 block size (nemb 256, one 8.66 GB 4-fold block): GB/s per pass."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdmet_preview_amd import _lib
from libdmet_preview_amd._lib import lib
from libdmet_preview_amd.solver.scf import jk_dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _lib.get_ctx()
npair = n * (n + 1) // 2
rng = np.random.default_rng(1)
X = rng.standard_normal((8, npair))
dX = ctx.to_device(X)
dE = ctx.zeros((npair, npair), np.float64)
ctx.check(lib.dmk_dgemm_tn_acc(ctx.h, npair, 8, 1.0, dX.ptr, dX.ptr, npair, dE.ptr, npair))
d = ctx.to_device(rng.standard_normal((n, n)))
gb = npair * npair * 8 / 1e9
for name, args in [("J rows", (d, None, None)), ("J rows+cols", (d, d, None)), ("K", (None, None, d)), ("J+K", (d, None, d))]:
    jk_dev(ctx, n, dE, *args)
    ctx.sync()
    t = time.perf_counter()
    for _ in range(reps):
        jk_dev(ctx, n, dE, *args)
    ctx.sync()
    dt = (time.perf_counter() - t) / reps
    passes = (1 if (args[0] is not None or args[1] is not None) else 0) + (1 if args[2] is not None else 0)
    print("%-12s %8.3f ms   %7.1f GB/s (%.2f GB x %d pass)" % (name, dt * 1e3, gb * passes / dt, gb, passes))
