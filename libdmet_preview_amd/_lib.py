"""
ctypes binding of libdmetk.so (include/libdmetk.h) and the device-array helper.

The product path has NO CPU fallback: if the HIP library is missing this module
raises at import, and creating a context without a GPU raises RuntimeError.
"""
import ctypes as C
import os
import threading
import weakref
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("LIBDMETK_SO", os.path.join(_HERE, "libdmetk.so"))

if not os.path.exists(_LIBPATH):
    raise ImportError(
        "libdmetk.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C libdmet_preview_amd/csrc); there is no CPU fallback." % _LIBPATH)

lib = C.CDLL(_LIBPATH)

c_int, c_i64, c_dbl, c_vp, c_sz = C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_size_t
P = C.POINTER
_int3 = c_int * 3

# name -> (restype, argtypes); every symbol declared in include/libdmetk.h
PROTOTYPES = {
    "dmk_init": (c_int, [c_int, c_vp, P(c_vp)]),
    "dmk_destroy": (c_int, [c_vp]),
    "dmk_set_stream": (c_int, [c_vp, c_vp]),
    "dmk_sync": (c_int, [c_vp]),
    "dmk_set_oom_hook": (c_int, [c_vp, c_vp, c_vp]),
    "dmk_last_error": (C.c_char_p, [c_vp]),
    "dmk_version": (C.c_char_p, []),
    "dmk_mem_info": (c_int, [c_vp, P(c_sz), P(c_sz)]),
    "dmk_malloc": (c_int, [c_vp, c_sz, P(c_vp)]),
    "dmk_free": (c_int, [c_vp, c_vp]),
    "dmk_memset": (c_int, [c_vp, c_vp, c_int, c_sz]),
    "dmk_memcpy_h2d": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "dmk_memcpy_d2h": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "dmk_memcpy_d2d": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "dmk_timer_start": (c_int, [c_vp]),
    "dmk_timer_stop": (c_int, [c_vp, P(c_dbl)]),
    "dmk_profile": (c_int, [c_vp, c_int]),
    "dmk_profile_read": (c_int, [c_vp, P(c_dbl), P(c_i64), c_int]),
    "dmk_profile_read_flops": (c_int, [c_vp, P(c_dbl), c_int]),
    "dmk_kmesh_tables": (c_int, [_int3, c_vp, c_vp, c_vp]),
    "dmk_kconserv_table": (c_int, [_int3, c_vp]),
    "dmk_cell_add_table": (c_int, [_int3, c_int, c_vp]),
    "dmk_kpts_scaled": (c_int, [_int3, c_vp]),
    "dmk_kpt_member": (c_int, [_int3, P(c_dbl), c_dbl]),
    "dmk_eri_plan": (c_int, [_int3, c_int, c_vp, c_i64, P(c_i64)]),
    "dmk_kptij_mask": (c_int, [_int3, c_int, c_vp, c_vp]),
    "dmk_assign_workload": (c_int, [_int3, c_int, c_int, c_int, c_vp, P(c_int)]),
    "dmk_fold_R2k": (c_int, [c_vp, _int3, c_i64, c_int, c_vp, c_int, c_vp]),
    "dmk_fold_k2R": (c_int, [c_vp, _int3, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_int]),
    "dmk_fold_k2R_complex": (c_int, [c_vp, _int3, c_i64, c_int, c_vp, c_vp]),
    "dmk_eigh_batched": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "dmk_eigh_batched_real": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "dmk_eigh_jacobi_real": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, P(c_int)]),
    "dmk_occ_density": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "dmk_assign_occ": (c_int, [c_vp, c_i64, c_vp, c_dbl, c_dbl, c_dbl, c_int, c_dbl, c_dbl, c_vp, P(c_dbl)]),
    "dmk_transpose_c128": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp]),
    "dmk_bath_svd": (c_int, [c_vp, _int3, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "dmk_bath_svd_batched": (c_int, [c_vp, _int3, c_int, c_int, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_vp, c_vp]),
    "dmk_bath_assemble": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "dmk_zgemm_batched": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_dbl, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64]),
    "dmk_eri_begin": (c_int, [c_vp, _int3, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, P(c_vp)]),
    "dmk_eri_begin_kL": (c_int, [c_vp, c_int]),
    "dmk_eri_push_block": (c_int, [c_vp, c_int, c_int, c_int, c_vp]),
    "dmk_eri_block_ring": (c_int, [c_vp, P(c_vp), P(c_int)]),
    "dmk_eri_push_ring_slot": (c_int, [c_vp, c_int, c_int, c_int]),
    "dmk_eri_flush": (c_int, [c_vp]),
    "dmk_eri_ring_slot": (c_int, [c_vp, c_int, P(c_vp), P(c_vp)]),
    "dmk_eri_push_resident": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "dmk_df_block_philox_on": (c_int, [c_vp, c_vp, C.c_uint64, c_int, c_int, c_int, c_int, c_vp]),
    "dmk_eri_push_block_host": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int]),
    "dmk_eri_host_slot_wait": (c_int, [c_vp, c_int]),
    "dmk_host_alloc": (c_int, [c_vp, c_sz, P(c_vp)]),
    "dmk_host_free": (c_int, [c_vp, c_vp]),
    "dmk_eri_end_kL": (c_int, [c_vp, c_int]),
    "dmk_eri_end_kL_gso": (c_int, [c_vp, c_int]),
    "dmk_eri_stack": (c_int, [c_vp, c_int, P(c_int)]),
    "dmk_eri_begin_kL_weighted": (c_int, [c_vp, c_int, c_int]),
    "dmk_eri_contract": (c_int, [c_vp, c_int, c_int, c_int]),
    "dmk_eri_probe": (c_int, [c_vp, c_vp, c_vp]),
    "dmk_eri_bands": (c_int, [c_vp, P(c_int), P(c_int)]),
    "dmk_eri_contract_rows": (c_int, [c_vp, c_i64, c_i64, c_vp]),
    "dmk_eri_stack_clear": (c_int, [c_vp]),
    "dmk_eri_stack_free_slots": (c_int, [c_vp, P(c_int)]),
    "dmk_eri_planes": (c_int, [c_vp, P(c_vp), P(c_i64)]),
    "dmk_eri_finish": (c_int, [c_vp]),
    "dmk_eri_flops": (c_int, [c_vp, P(c_dbl)]),
    "dmk_eri_imag_norm": (c_int, [c_vp, P(c_dbl)]),
    "dmk_eri_imag_buffer": (c_int, [c_vp, P(c_vp), P(c_i64)]),
    "dmk_df_block_philox": (c_int, [c_vp, C.c_uint64, c_int, c_int, c_int, c_int, c_vp]),
    "dmk_eri_restore": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "dmk_dgemm_tn_acc": (c_int, [c_vp, c_int, c_int, c_dbl, c_vp, c_vp, c_i64, c_vp, c_i64]),
    "dmk_dgemm_tn_acc_rect": (c_int, [c_vp, c_int, c_int, c_int, c_dbl, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64]),
    "dmk_svd_small": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "dmk_dgemm_nn_small": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "dmk_bcs_weight": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "dmk_bcs_assemble": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "dmk_pad_block_f64": (c_int, [c_vp, c_int, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp]),
    "dmk_copy_rows_f64": (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_int]),
    "dmk_jk_s4": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmk_jk_s4_rows": (c_int, [c_vp, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmk_eri_to_s4": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "dmk_sym_norm_bound": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "dmk_modified_cholesky": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_dbl, c_int, c_vp, P(c_int), P(c_int)]),
    "dmk_cpqr_pivots": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    "dmk_dgemv2": (c_int, [c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "dmk_dgemm_batched": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_dbl, c_vp, c_i64, c_i64, c_vp, c_i64,
                                  c_i64, c_dbl, c_vp, c_i64, c_i64]),
    "dmk_sym_fold": (c_int, [c_vp, c_int, c_int, c_vp, c_vp]),
    "dmk_sym_unpack": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "dmk_gather2d_f64": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dmk_ewise_mul": (c_int, [c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "dmk_sub_sumsq": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "dmk_fit_kmat": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_dbl, c_int, c_vp]),
    "dmk_scatter2d_add_f64": (c_int, [c_vp, c_int, c_vp, c_vp, c_dbl, c_vp, c_i64, c_int]),
    "dmk_axpy_f64": (c_int, [c_vp, c_i64, c_dbl, c_vp, c_vp]),
    "dmk_vcor_dV_dparam": (c_int, [c_vp, c_int, c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dmk_df_blocks_philox_on": (c_int, [c_vp, c_vp, C.c_uint64, c_int, c_vp, c_int, c_int, c_vp, c_i64]),
    "dmk_fit_objective": (c_int, [c_vp, c_vp, P(c_dbl), P(c_int), P(c_int)]),
    "dmk_small_meanfield": (c_int, [c_vp, P(c_int), c_int, c_int, c_vp, c_vp, c_int, c_dbl, c_dbl, c_dbl, c_int, c_dbl, c_dbl, c_vp,
                                    c_vp, c_vp, c_vp, c_vp, c_vp, P(c_int)]),
    "dmk_small_bath": (c_int, [c_vp, P(c_int), c_int, c_int, c_vp, c_i64, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int,
                               c_dbl, c_vp, c_vp, c_vp, c_vp, P(c_int)]),
}


class FitArgs(C.Structure):
    """dmk_fit_args of include/libdmetk.h (the fused T = 0 objective of FitVcorEmb)."""
    _fields_ = [("nb", c_int), ("spin", c_int), ("nidx", c_int), ("npass", c_int), ("has_mu0", c_int),
                ("t", c_dbl), ("tol_deg", c_dbl),
                ("v0", c_vp), ("v1", c_vp), ("H1", c_vp),
                ("H", c_vp), ("Vp", c_vp), ("w", c_vp), ("occ", c_vp),
                ("nelec", c_vp), ("mu0", c_vp),
                ("fit_idx", c_vp), ("W", c_vp), ("target", c_vp),
                ("drho", c_vp), ("work", c_vp), ("slot", c_vp), ("ray_norm", c_vp)]

for _name, (_res, _args) in PROTOTYPES.items():
    _f = getattr(lib, _name)          # AttributeError here = header / library mismatch: fail loudly
    _f.restype = _res
    _f.argtypes = _args

FAMILIES = ["dgemm", "zgemm_half1", "zgemm_half2", "philox", "fold", "eigh", "bath", "zgemm_small", "misc", "jk", "fit"]


_OOM_HOOK = C.CFUNCTYPE(None, c_vp)


class DmkError(RuntimeError):
    pass


def mesh3(kmesh):
    m = [int(x) for x in kmesh]
    if len(m) > 3 or len(m) < 1 or any(x <= 0 for x in m):
        raise ValueError("kmesh must have 1-3 positive entries, got %r" % (kmesh,))
    m = m + [1] * (3 - len(m))
    return _int3(*m)


class Context(object):
    """One libdmetk context (one GPU, one stream).  Not shared between threads."""

    def __init__(self, device=0, stream=None):
        h = c_vp()
        rc = lib.dmk_init(int(device), c_vp(stream) if stream else None, C.byref(h))
        if rc != 0 or not h:
            raise RuntimeError("libdmetk: dmk_init(device=%d) failed with %d -- an MI355X (HIP device) is required; "
                               "there is no CPU fallback" % (device, rc))
        self.h = h
        self.device = int(device)
        self.default_stream = not stream          # the legacy null stream: ordered with torch's default stream
        self.stream_ptr = int(stream) if stream else 0
        # Freed device blocks are parked here by size and handed out again instead of going through hipFree / hipMalloc
        # (each a device synchronisation plus ~1 ms per few hundred MB): an embedding-construction iteration allocates the
        # same two dozen temporaries every time.  One stream per context, so reuse is stream ordered.
        self._pool = {}
        self._pool_bytes = 0
        self._pool_limit = int(float(os.environ.get("DMK_POOL_GB", "24")) * (1 << 30))
        # the library cannot see the parked blocks: before one of ITS allocations (ERI planes, Ut, rings, eigensolver scratch)
        # is reported as out of memory it calls back here, and retries once
        me = weakref.ref(self)
        self._oom_cb = _OOM_HOOK(lambda _user: me() is not None and me().trim())
        self.check(lib.dmk_set_oom_hook(self.h, C.cast(self._oom_cb, c_vp), None))

    def _alloc(self, nbytes):
        """(pointer, capacity) of a device block of at least nbytes."""
        cap = (max(int(nbytes), 16) + 255) & ~255
        stack = self._pool.get(cap)
        if stack:
            self._pool_bytes -= cap
            return stack.pop(), cap
        p = c_vp()
        self.check(lib.dmk_malloc(self.h, cap, C.byref(p)))     # the library retries through the hook after trim()
        return p, cap

    def _release(self, ptr, cap):
        if not self.h:
            return
        if cap <= self._pool_limit - self._pool_bytes:
            self._pool.setdefault(cap, []).append(ptr)
            self._pool_bytes += cap
        else:
            lib.dmk_free(self.h, ptr)

    def trim(self):
        """Return every parked block to the driver."""
        for stack in self._pool.values():
            for ptr in stack:
                lib.dmk_free(self.h, ptr)
        self._pool = {}
        self._pool_bytes = 0

    def check(self, rc):
        if rc != 0:
            msg = lib.dmk_last_error(self.h)
            raise DmkError("libdmetk error %d: %s" % (rc, msg.decode() if msg else "?"))

    def close(self):
        if getattr(self, "h", None):
            self.trim()
            lib.dmk_set_oom_hook(self.h, None, None)
            lib.dmk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self.check(lib.dmk_sync(self.h))

    def mem_info(self):
        """(free, total) bytes of HBM; blocks parked in this context's pool count as free."""
        f, t = c_sz(), c_sz()
        self.check(lib.dmk_mem_info(self.h, C.byref(f), C.byref(t)))
        return int(f.value) + self._pool_bytes, int(t.value)

    def set_stream(self, stream):
        # parked blocks were released in the order of the OLD stream: drain it and hand them back to the driver, so that
        # none of them can be given out again (and written on the new stream) while work on the old one still reads it
        self.sync()
        self.trim()
        self.check(lib.dmk_set_stream(self.h, c_vp(stream) if stream else None))
        self.default_stream = not stream
        self.stream_ptr = int(stream) if stream else 0

    # ---- device arrays ---------------------------------------------------------------
    def empty(self, shape, dtype):
        return DevArray(self, shape, dtype)

    def zeros(self, shape, dtype):
        a = DevArray(self, shape, dtype)
        a.zero_()
        return a

    def to_device(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        d = DevArray(self, arr.shape, arr.dtype)
        if arr.nbytes:
            self.check(lib.dmk_memcpy_h2d(self.h, d.ptr, arr.ctypes.data_as(c_vp), arr.nbytes))
        return d

    def wrap(self, ptr, shape, dtype, keepalive=None):
        """Borrow device memory owned by someone else (e.g. a torch tensor)."""
        return DevArray(self, shape, dtype, ptr=ptr, keepalive=keepalive)

    # ---- timing ----------------------------------------------------------------------
    def timer_start(self):
        self.check(lib.dmk_timer_start(self.h))

    def timer_stop(self):
        ms = c_dbl()
        self.check(lib.dmk_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self, enable):
        self.check(lib.dmk_profile(self.h, 1 if enable else 0))

    def profile_read(self, reset=True):
        ms = (c_dbl * len(FAMILIES))()
        n = (c_i64 * len(FAMILIES))()
        self.check(lib.dmk_profile_read(self.h, ms, n, 1 if reset else 0))
        return {FAMILIES[i]: (ms[i], int(n[i])) for i in range(len(FAMILIES))}

    def profile_read_flops(self, reset=True):
        """Flop issued to the f64 matrix pipe per kernel family since the last reset (dmk_profile_read_flops)."""
        f = (c_dbl * len(FAMILIES))()
        self.check(lib.dmk_profile_read_flops(self.h, f, 1 if reset else 0))
        return {FAMILIES[i]: float(f[i]) for i in range(len(FAMILIES))}


class DevArray(object):
    """A typed, shaped block of HBM.  Owns its memory unless constructed with ptr=."""

    def __init__(self, ctx, shape, dtype, ptr=None, keepalive=None):
        self.ctx = ctx
        self.shape = tuple(int(x) for x in (shape if hasattr(shape, "__iter__") else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if len(self.shape) else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._keep = keepalive
        if ptr is None:
            self.ptr, self._cap = ctx._alloc(self.nbytes)
            self._own = True
        else:
            self.ptr = c_vp(int(ptr))
            self._own = False

    @property
    def address(self):
        return self.ptr.value or 0

    def offset(self, nelem, shape):
        """A borrowed view starting `nelem` elements into this array."""
        return DevArray(self.ctx, shape, self.dtype, ptr=self.address + int(nelem) * self.dtype.itemsize,
                        keepalive=self)

    def reshape(self, *shape):
        if len(shape) == 1 and hasattr(shape[0], "__iter__"):
            shape = tuple(shape[0])
        assert int(np.prod(shape)) == self.size
        return DevArray(self.ctx, shape, self.dtype, ptr=self.address, keepalive=self)

    def zero_(self):
        if self.nbytes:
            self.ctx.check(lib.dmk_memset(self.ctx.h, self.ptr, 0, self.nbytes))
        return self

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            self.ctx.check(lib.dmk_memcpy_d2h(self.ctx.h, out.ctypes.data_as(c_vp), self.ptr, self.nbytes))
        return out

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.size == self.size
        if self.nbytes:
            self.ctx.check(lib.dmk_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data_as(c_vp), self.nbytes))
        return self

    def free(self):
        if self._own and self.ptr and self.ctx.h:
            self.ctx._release(self.ptr, self._cap)
        self.ptr = None
        self._own = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray(object):
    """Page-locked host memory (dmk_host_alloc) exposed as a numpy array `.a`; the feed buffers of the host block path."""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        shape = tuple(int(x) for x in (shape if hasattr(shape, "__iter__") else (shape,)))
        nbytes = int(np.prod(shape)) * self.dtype.itemsize
        p = c_vp()
        ctx.check(lib.dmk_host_alloc(ctx.h, max(nbytes, 16), C.byref(p)))
        self.ptr = p
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        self.a = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self.ptr and self.ctx.h:
            self.a = None
            lib.dmk_host_free(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_tls = threading.local()


def get_ctx():
    """The calling thread's default context on the current device (LOCAL_RANK aware)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None:
        dev = int(os.environ.get("DMK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        ctx = Context(dev)
        _tls.ctx = ctx
    return ctx


def set_ctx(ctx):
    _tls.ctx = ctx


def host_i32(n):
    return np.empty(n, dtype=np.int32)


def as_ptr(a):
    """Pointer of a host numpy array, or of a DevArray, or None."""
    if a is None:
        return None
    if isinstance(a, DevArray):
        return a.ptr
    return a.ctypes.data_as(c_vp)
