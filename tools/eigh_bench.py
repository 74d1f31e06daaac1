"""eigh latency / throughput probe: python tools/eigh_bench.py  (set DMK_EIGH_TIMING=1 for the phase clocks)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdmet_preview_amd import _lib
from libdmet_preview_amd._lib import lib
ctx = _lib.get_ctx()
rng = np.random.default_rng(0)
for n, batch, real in [(256, 2, True), (256, 2, False), (200, 432, False), (136, 64, False), (64, 2, True)]:
    A = rng.standard_normal((batch, n, n)) + (0 if real else 1j * rng.standard_normal((batch, n, n)))
    A = A + A.conj().transpose(0, 2, 1)
    dA = ctx.to_device(A, np.float64 if real else np.complex128)
    dw = ctx.empty((batch, n), np.float64)
    dV = ctx.empty((batch, n, n), np.float64 if real else np.complex128)
    def run():
        if real:
            ctx.check(lib.dmk_eigh_batched_real(ctx.h, n, batch, dA.ptr, dw.ptr, dV.ptr))
        else:
            ctx.check(lib.dmk_eigh_batched(ctx.h, n, batch, dA.ptr, None, 0, dw.ptr, dV.ptr))
    run(); ctx.sync()
    t = time.perf_counter()
    for _ in range(3):
        run()
    ctx.sync()
    dt = (time.perf_counter() - t) / 3
    w = dw.get()
    wr = np.linalg.eigvalsh(A[0])
    V = dV.get()                                             # row m = eigenvector m
    res = orth = 0.0
    for bi in (0, batch - 1):
        X = V[bi].T                                          # row m of Vt = eigenvector m -> columns
        res = max(res, np.abs(A[bi] @ X - X * w[bi]).max() / np.abs(w[bi]).max())
        orth = max(orth, np.abs(X.conj().T @ X - np.eye(n)).max())
    print("n=%d batch=%d %s: %.2f ms   (max |dw| %.1e, residual / |w| %.1e, orthogonality %.1e)"
          % (n, batch, "real" if real else "c128", dt * 1e3, np.abs(w[0] - wr).max(), res, orth), flush=True)
