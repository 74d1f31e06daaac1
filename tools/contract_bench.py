"""Contraction kernel alone at C5 size: python tools/contract_bench.py [N] [K] -- TF executed for the symmetric (aa/bb)
and the rectangular (ab) launch, per tile-order setting (DMK_DGEMM_SUPER)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32896
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
from libdmet_preview_amd import _lib
from libdmet_preview_amd._lib import lib
for sb in (os.environ.get("SUPERS", "8,1,4,16").split(",")):
    os.environ["DMK_DGEMM_SUPER"] = sb
    ctx = _lib.Context(0)
    rng = np.random.default_rng(0)
    X = ctx.to_device(rng.standard_normal((K, N)))
    Y = ctx.to_device(rng.standard_normal((K, N)))
    Cd = ctx.zeros((N, N), np.float64)
    for name, A, B in (("symm", X, X), ("rect", X, Y)):
        ctx.check(lib.dmk_dgemm_tn_acc(ctx.h, N, K, 1.0, A.ptr, B.ptr, N, Cd.ptr, N))
        ctx.sync()
        ctx.profile(True); ctx.profile_read(True); ctx.profile_read_flops(True)
        reps = 5
        for _ in range(reps):
            ctx.check(lib.dmk_dgemm_tn_acc(ctx.h, N, K, 1.0, A.ptr, B.ptr, N, Cd.ptr, N))
        ms, n = ctx.profile_read(True)["dgemm"]
        fl = ctx.profile_read_flops(True)["dgemm"]
        ctx.profile(False)
        print("super %s %s N %d K %d: %.3f ms/launch, executed %.1f TF, algorithmic %.1f TF" %
              (sb, name, N, K, ms / n, fl / (ms * 1e-3) / 1e12, 2.0 * K * N * N * n / (ms * 1e-3) / 1e12), flush=True)
    del X, Y, Cd
    ctx.close()
