import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from libdmet_preview_amd import _lib
from libdmet_preview_amd._lib import lib
ctx = _lib.get_ctx()
rng = np.random.default_rng(int(os.environ.get('STRESS_SEED', '2026')))
NMAX = int(os.environ.get('STRESS_NMAX', '200')); BMAX = int(os.environ.get('STRESS_BMAX', '5'))
worst = [0, 0, 0]
for trial in range(int(os.environ.get('STRESS_TRIALS', '60'))):
    n = int(rng.integers(2, NMAX + 1)); batch = int(rng.integers(1, BMAX + 1))
    kind = trial % 4
    A = rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))
    A = A + A.conj().transpose(0, 2, 1)
    if kind == 1:      # low rank + identity: massive degeneracy
        u = rng.standard_normal((batch, n, 3)) + 1j * rng.standard_normal((batch, n, 3))
        A = u @ u.conj().transpose(0, 2, 1) + 2.0 * np.eye(n)
    elif kind == 2:    # graded
        s = np.logspace(0, -12, n)
        A = A * s[None, :, None] * s[None, None, :]
    elif kind == 3:    # real symmetric stored as complex, banded
        A = np.triu(np.tril(A.real, 3), -3).astype(complex)
        A = A + A.conj().transpose(0, 2, 1)
    dA = ctx.to_device(A, np.complex128)
    dw, dV = ctx.empty((batch, n), np.float64), ctx.empty((batch, n, n), np.complex128)
    ctx.check(lib.dmk_eigh_batched(ctx.h, n, batch, dA.ptr, None, 0, dw.ptr, dV.ptr))
    w, V = dw.get(), dV.get()
    for b in range(batch):
        wr = np.linalg.eigvalsh(A[b]); sc = max(np.abs(wr).max(), 1e-300)
        e0 = np.abs(w[b] - wr).max() / sc
        e1 = np.abs(V[b].conj() @ V[b].T - np.eye(n)).max()
        e2 = np.abs(V[b].conj() @ A[b] @ V[b].T - np.diag(w[b])).max() / sc
        worst = [max(worst[0], e0), max(worst[1], e1), max(worst[2], e2)]
        assert e0 < 1e-12 and e1 < 1e-11 and e2 < 1e-11, (trial, n, batch, kind, e0, e1, e2)
print("stress ok: worst |dw|/|w| %.1e  orth %.1e  resid %.1e" % tuple(worst))
