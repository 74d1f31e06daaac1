"""Warm-started symmetric eigensolver (dmk_eigh_jacobi_real): refinement fast path against the Jacobi sweeps.
For perturbations of several sizes: time per call, passes / sweeps, residual and orthogonality of the result.

    python tools/refine_lab.py [n] [batch]          (DMK_EIGH_REFINE=0: sweeps only)
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdmet_preview_amd._lib import lib, get_ctx      # noqa: E402


def run(ctx, A, A2, reps=20):
    batch, n, _ = A.shape
    dA, dw, dV = ctx.to_device(A), ctx.empty((batch, n), np.float64), ctx.empty((batch, n, n), np.float64)
    sw = C.c_int()
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA.ptr, None, dw.ptr, dV.ptr, C.byref(sw)))
    V0 = dV.get()
    dA2 = ctx.to_device(A2)
    dV0, dV1 = ctx.to_device(V0), ctx.empty((batch, n, n), np.float64)
    ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA2.ptr, dV0.ptr, dw.ptr, dV1.ptr, C.byref(sw)))
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.check(lib.dmk_eigh_jacobi_real(ctx.h, n, batch, dA2.ptr, dV0.ptr, dw.ptr, dV1.ptr, C.byref(sw)))
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    w, V = dw.get(), dV1.get()
    scale = max(1.0, np.abs(A2).max() * n ** 0.5)
    res = max(np.abs(V[b] @ A2[b] @ V[b].T - np.diag(w[b])).max() for b in range(batch)) / scale
    orth = max(np.abs(V[b] @ V[b].T - np.eye(n)).max() for b in range(batch))
    dw_ = max(np.abs(w[b] - np.linalg.eigvalsh(A2[b])).max() for b in range(batch)) / scale
    return dt * 1e3, sw.value, res, orth, dw_


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ctx = get_ctx()
    rng = np.random.default_rng(5)
    A = rng.standard_normal((batch, n, n))
    A = A + A.transpose(0, 2, 1)
    cases = [("random", A)]
    # exactly degenerate spectrum (pairs) and a tight cluster
    Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
    lam = np.repeat(np.linspace(-3, 3, (n + 1) // 2), 2)[:n]
    D = np.stack([Q @ np.diag(lam) @ Q.T] * batch)
    cases.append(("degenerate pairs", 0.5 * (D + D.transpose(0, 2, 1))))
    lam2 = np.linspace(-3, 3, n)
    lam2[: n // 4] = -3 + 1e-7 * np.arange(n // 4)
    D2 = np.stack([Q @ np.diag(lam2) @ Q.T] * batch)
    cases.append(("tight cluster 1e-7", 0.5 * (D2 + D2.transpose(0, 2, 1))))
    for name, A0 in cases:
        for eps in (0.0, 1e-8, 1e-5, 1e-3, 1e-2, 1e-1):
            P = eps * rng.standard_normal((batch, n, n))
            A2 = A0 + P + P.transpose(0, 2, 1)
            out = run(ctx, A0, A2)
            print("refine=%s %-20s eps %.0e | %.3f ms sweeps %d res %.1e orth %.1e dw %.1e"
                  % ((os.environ.get("DMK_EIGH_REFINE", "1"), name, eps) + out), flush=True)


if __name__ == "__main__":
    main()
