"""eigh time vs batch: python tools/eigh_scan.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdmet_preview_amd import _lib
from libdmet_preview_amd._lib import lib
ctx = _lib.get_ctx()
rng = np.random.default_rng(0)
for n in (136, 200):
    for batch in (1, 8, 64, 128, 256, 432, 512):
        A = rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))
        A = A + A.conj().transpose(0, 2, 1)
        dA = ctx.to_device(A, np.complex128)
        dw = ctx.empty((batch, n), np.float64)
        dV = ctx.empty((batch, n, n), np.complex128)
        run = lambda: ctx.check(lib.dmk_eigh_batched(ctx.h, n, batch, dA.ptr, None, 0, dw.ptr, dV.ptr))
        run(); ctx.sync()
        t = time.perf_counter()
        for _ in range(3):
            run()
        ctx.sync()
        print("n=%d batch=%d: %.2f ms" % (n, batch, (time.perf_counter() - t) / 3 * 1e3), flush=True)
