"""Jacobi eigensolver probe: accuracy and latency, cold and warm start."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdmet_preview_amd import _lib
from libdmet_preview_amd._lib import lib
ctx = _lib.get_ctx()
rng = np.random.default_rng(0)
for n, batch in [(5, 1), (33, 2), (64, 3), (100, 2), (200, 2), (256, 2), (300, 1)]:
    A = rng.standard_normal((batch, n, n)); A = A + A.transpose(0, 2, 1)
    dA = ctx.to_device(A); dw = ctx.empty((batch, n), np.float64); dV = ctx.empty((batch, n, n), np.float64)
    sw = C.c_int()
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA.ptr, None, dw.ptr, dV.ptr, C.byref(sw)))
    ctx.sync(); t = time.perf_counter()
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA.ptr, None, dw.ptr, dV.ptr, C.byref(sw)))
    ctx.sync(); t_cold = time.perf_counter() - t
    w, V = dw.get(), dV.get()
    wr = np.linalg.eigvalsh(A)
    res = max(np.abs(V[b] @ A[b] @ V[b].T - np.diag(w[b])).max() for b in range(batch))
    orth = max(np.abs(V[b] @ V[b].T - np.eye(n)).max() for b in range(batch))
    # warm start on a perturbed matrix
    P = 1e-3 * rng.standard_normal((batch, n, n)); A2 = A + P + P.transpose(0, 2, 1)
    dA2 = ctx.to_device(A2); dw2 = ctx.empty((batch, n), np.float64); dV2 = ctx.empty((batch, n, n), np.float64)
    sw2 = C.c_int()
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA2.ptr, dV.ptr, dw2.ptr, dV2.ptr, C.byref(sw2)))
    ctx.sync(); t = time.perf_counter()
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA2.ptr, dV.ptr, dw2.ptr, dV2.ptr, C.byref(sw2)))
    ctx.sync(); t_warm = time.perf_counter() - t
    w2 = dw2.get()
    print("n=%3d batch=%d: cold %6.2f ms (%2d sweeps) |dw| %.1e resid %.1e orth %.1e | warm %6.2f ms (%2d sweeps) |dw| %.1e"
          % (n, batch, t_cold * 1e3, sw.value, np.abs(w - wr).max(), res, orth, t_warm * 1e3, sw2.value,
             np.abs(w2 - np.linalg.eigvalsh(A2)).max()), flush=True)
