"""Where the `diag` stage of a C5 iteration spends its wall-clock: python tools/diag_stage_probe.py"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdmet_preview_amd import _lib, pipeline
from libdmet_preview_amd._lib import lib
from libdmet_preview_amd.routine import mfd

ctx = _lib.get_ctx()
sysm = pipeline.SyntheticSystem.from_workload(ctx, sys.argv[1] if len(sys.argv) > 1 else "C5")
n, nk, spin = sysm.nlo, sysm.nk, sysm.spin
d_F = sysm.d_Fock_k.reshape(spin * nk, n, n)
for rep in range(3):
    ctx.sync(); t0 = time.perf_counter()
    d_w = ctx.empty((spin * nk, n), np.float64); d_Vt = ctx.empty((spin * nk, n, n), np.complex128)
    ctx.sync(); t1 = time.perf_counter()
    ms = C.c_double()
    lib.dmk_timer_start(ctx.h)
    ctx.check(lib.dmk_eigh_batched(ctx.h, n, spin * nk, d_F.ptr, sysm.d_vcor.ptr if sysm.d_vcor is not None else None, nk, d_w.ptr, d_Vt.ptr))
    t2 = time.perf_counter()
    lib.dmk_timer_stop(ctx.h, C.byref(ms))
    t3 = time.perf_counter()
    print("rep %d: alloc %.3f ms, eigh call returns after %.3f ms, done after %.3f ms (HIP events %.3f ms)" %
          (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t1), ms.value))
    del d_w, d_Vt
timers = {}
for rep in range(3):
    timers = {}
    pipeline.mean_field_stage(ctx, sysm, timers)
    print({k: round(1e3 * v, 3) for k, v in timers.items()})
