"""Schmidt-bath latency probe at the C5 / C4 / C2 shapes:  python tools/bath_bench.py
(run under `rocprofv3 --kernel-trace --stats` for the per-kernel split; DMK_BATH_TSQR=0 restores the column-at-a-time QR)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdmet_preview_amd import _lib
from libdmet_preview_amd.routine import slater
from libdmet_preview_amd.system.lattice import Lattice

ctx = _lib.get_ctx()
rng = np.random.default_rng(8)
for mesh, nlo, nval in [((6, 6, 6), 200, 56), ((4, 4, 4), 104, 32), ((6, 6, 1), 4, 4)]:
    nk = int(np.prod(mesh))
    nocc = max(2, nlo // 5)
    V = rng.standard_normal((nk * nlo, nocc)) * np.exp(-0.02 * np.arange(nk * nlo))[:, None]
    V, _ = np.linalg.qr(V)
    rdm1 = (V @ V[:nlo].T).reshape(nk, nlo, nlo)
    val = list(range(nval))
    bath_set = set(val)
    env = np.asarray([i for i in range(nk * nlo) if i not in bath_set], dtype=np.int32)
    virt = np.asarray([i < nlo for i in env], dtype=np.int32)
    d_env, d_virt = ctx.to_device(env), ctx.to_device(virt)
    d_col, d_imp = ctx.to_device(np.asarray(val, dtype=np.int32)), ctx.to_device(np.arange(nlo, dtype=np.int32))
    d_rdm = ctx.to_device(rdm1)
    nenv = len(env)

    def run():
        d_sigma, d_U = slater.bath_svd_dev(ctx, mesh, nlo, d_rdm, d_env, nenv, d_col, nval)
        sig = d_sigma.get()
        nbath = int((sig >= 1e-9).sum())
        d_basis = ctx.empty((nk * nlo, nlo + nbath), np.float64)
        slater.bath_assemble_dev(ctx, d_U, nenv, nval, nbath, d_virt, True, d_env, d_imp, nlo, nk * nlo, nlo + nbath, d_basis)
        return sig, d_U, d_basis

    sig, d_U, d_basis = run()
    ctx.sync()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        run()
    ctx.sync()
    dt = (time.perf_counter() - t) / reps
    A = rdm1.reshape(nk * nlo, nlo)[env][:, val]
    sref = np.linalg.svd(A, compute_uv=False)
    U = d_U.get()
    alg = 8.0 * nk * nlo * (nval + nlo + nval)
    print("mesh %s nlo %d nb %d (A %d x %d): %.3f ms per spin, %.1f GB/s algorithmic; max|dsigma| %.1e, |U^T U - 1| %.1e, |U S V^T - A| col-space %.1e"
          % (mesh, nlo, nval, nenv, nval, dt * 1e3, alg / dt / 1e9, np.abs(sig - sref).max(), np.abs(U.T @ U - np.eye(nval)).max(),
             np.abs(U @ (U.T @ A) - A).max()), flush=True)
