"""
k-mesh bookkeeping and k <-> R Fourier folds with the signatures of the reference's
libdmet/system/fourier.py, computed by libdmetk (HIP) instead of scipy.fft.

  make_kpts_scaled   fourier.py:46-53     -> dmk_kpts_scaled (bit-identical table)
  round_to_FBZ       fourier.py:55-65     host, elementwise on (nk,3) arrays
  kpt_member         fourier.py:73-81     host; mesh queries go through dmk_kpt_member
  FFTtoK / FFTtoT    fourier.py:160-177   -> dmk_fold_R2k / dmk_fold_k2R (DFT as MFMA GEMM)
  R2k / k2R          fourier.py:129-158

`*_dev` variants take and return device arrays (no PCIe traffic).
"""
import ctypes as C
import numpy as np

from libdmet_preview_amd import _lib
from libdmet_preview_amd._lib import lib, mesh3, get_ctx, DevArray
from libdmet_preview_amd.settings import IMAG_DISCARD_TOL, KPT_DIFF_TOL
from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.utils.misc import max_abs


def _check_host(rc, what):
    if rc != 0:
        raise ValueError("%s failed (%d)" % (what, rc))


def make_kpts_scaled(kmesh):
    """Scaled k-points in np.fft ordering; shape (nk, len(kmesh)) like the reference."""
    m = mesh3(kmesh)
    nk = m[0] * m[1] * m[2]
    out = np.empty((nk, 3))
    _check_host(lib.dmk_kpts_scaled(m, out.ctypes.data_as(C.c_void_p)), "dmk_kpts_scaled")
    return out[:, :len(kmesh)].copy()


def make_cells(kmesh):
    m = mesh3(kmesh)
    nk = m[0] * m[1] * m[2]
    out = np.empty((nk, 3), dtype=np.int32)
    _check_host(lib.dmk_kmesh_tables(m, out.ctypes.data_as(C.c_void_p), None, None), "dmk_kmesh_tables")
    return out.astype(np.int64)


def kmesh_tables(kmesh):
    """(mesh integers, index of -k, time-reversal weights) as int arrays."""
    m = mesh3(kmesh)
    nk = m[0] * m[1] * m[2]
    kint = np.empty((nk, 3), dtype=np.int32)
    mk = np.empty(nk, dtype=np.int32)
    w = np.empty(nk, dtype=np.int32)
    _check_host(lib.dmk_kmesh_tables(m, kint.ctypes.data_as(C.c_void_p), mk.ctypes.data_as(C.c_void_p),
                                     w.ctypes.data_as(C.c_void_p)), "dmk_kmesh_tables")
    return kint, mk, w


def round_to_FBZ(kpts, tol=1e-10, wrap_around=True):
    kpts = np.asarray(kpts, dtype=float)
    kr = kpts - np.floor(kpts)
    if wrap_around:
        kr[kr > (0.5 - tol)] -= 1.0
    else:
        kr[kr > (1.0 - tol)] = 0.0
    return kr


def round_to_FUC(coords, tol=1e-10, wrap_around=False):
    return round_to_FBZ(coords, tol=tol, wrap_around=wrap_around)


def kpt_member(kpt, kpts, tol=KPT_DIFF_TOL):
    kpt = np.asarray(kpt, dtype=float)
    kpts = np.reshape(kpts, (len(kpts), kpt.size))
    dk = kpts - kpt.ravel()
    dk = dk - np.round(dk)
    dk = np.sqrt((dk * dk).sum(axis=-1))
    return np.where(dk < tol)[0]


def kpt_member_mesh(kpt, kmesh, tol=KPT_DIFF_TOL):
    """Index of `kpt` in the fftfreq mesh (or -1); integer-mesh twin of kpt_member."""
    k = (C.c_double * 3)(*([float(x) for x in kpt] + [0.0] * (3 - len(kpt))))
    return int(lib.dmk_kpt_member(mesh3(kmesh), k, float(tol)))


def get_R_vec(cell, kmesh):
    R_rel = make_cells(kmesh)[:, :len(kmesh)].astype(float)
    latt = np.asarray(cell.lattice_vectors())[:len(kmesh)]
    return np.dot(R_rel, latt)


def get_phase_R2k(cell, kpts, kmesh=None):
    """exp(-i R.k), shape (ncells, nkpts) (fourier.py:112-121); small host table."""
    R = get_R_vec(cell, kmesh)
    return np.exp(-1.0j * np.einsum("Ru,ku->Rk", R, np.asarray(kpts)))


def get_phase_k2R(cell, kpts, kmesh=None):
    return get_phase_R2k(cell, kpts, kmesh=kmesh).conj().T / len(kpts)


# ---------------------------------------------------------------------------------------------
# device-resident folds
# ---------------------------------------------------------------------------------------------

def fold_R2k_dev(in_R, kmesh, batch, ncol, out=None):
    ctx = in_R.ctx
    m = mesh3(kmesh)
    nk = m[0] * m[1] * m[2]
    if out is None:
        out = ctx.empty((batch, nk, ncol), np.complex128)
    is_c = 1 if in_R.dtype == np.complex128 else 0
    ctx.check(lib.dmk_fold_R2k(ctx.h, m, int(ncol), int(batch), in_R.ptr, is_c, out.ptr))
    return out


def fold_k2R_dev(in_k, kmesh, batch, ncol, out=None, imag_max=None, k_subset=None):
    ctx = in_k.ctx
    m = mesh3(kmesh)
    nk = m[0] * m[1] * m[2]
    if out is None:
        out = ctx.empty((batch, nk, ncol), np.float64)
    sub = None
    nsub = 0
    if k_subset is not None:
        sub = np.ascontiguousarray(k_subset, dtype=np.int32)
        nsub = len(sub)
    ctx.check(lib.dmk_fold_k2R(ctx.h, m, int(ncol), int(batch), in_k.ptr, out.ptr,
                               imag_max.ptr if imag_max is not None else None,
                               sub.ctypes.data_as(C.c_void_p) if sub is not None else None, nsub))
    return out


# ---------------------------------------------------------------------------------------------
# reference-signature (numpy in / numpy out) entry points
# ---------------------------------------------------------------------------------------------

def FFTtoK(A, kmesh):
    A = np.asarray(A)
    nk = int(np.prod(kmesh))
    assert A.shape[-3] == nk, "first index must be the cell"
    ctx = get_ctx()
    ncol = int(A.shape[-2] * A.shape[-1])
    if np.iscomplexobj(A):
        d = ctx.to_device(A, np.complex128)
    else:
        d = ctx.to_device(A, np.float64)
    out = fold_R2k_dev(d, kmesh, 1, ncol)
    return out.get().reshape(A.shape)


def FFTtoT(B, kmesh, tol=IMAG_DISCARD_TOL):
    B = np.asarray(B)
    nk = int(np.prod(kmesh))
    assert B.shape[-3] == nk
    ctx = get_ctx()
    ncol = int(B.shape[-2] * B.shape[-1])
    d = ctx.to_device(B, np.complex128)
    imax = ctx.zeros((1,), np.float64)
    out = fold_k2R_dev(d, kmesh, 1, ncol, imag_max=imax)
    A = out.get().reshape(B.shape)
    im = float(imax.get()[0])
    if im > tol:
        log.warn("k2R: non-zero imaginary part: %15.8g", im)
    return A


def R2k(dm_R, kmesh):
    dm_R = np.asarray(dm_R)
    if dm_R.ndim == 3:
        return FFTtoK(dm_R, kmesh)
    elif dm_R.ndim == 4:
        ctx = get_ctx()
        spin, nk = dm_R.shape[:2]
        ncol = int(dm_R.shape[-2] * dm_R.shape[-1])
        d = ctx.to_device(dm_R, np.complex128 if np.iscomplexobj(dm_R) else np.float64)
        return fold_R2k_dev(d, kmesh, spin, ncol).get().reshape(dm_R.shape)
    raise ValueError("unknown shape of dm_R: %s" % str(dm_R.shape))


def k2R(dm_k, kmesh, tol=IMAG_DISCARD_TOL):
    dm_k = np.asarray(dm_k)
    if dm_k.ndim == 3:
        return FFTtoT(dm_k, kmesh, tol=tol)
    elif dm_k.ndim == 4:
        ctx = get_ctx()
        spin = dm_k.shape[0]
        ncol = int(dm_k.shape[-2] * dm_k.shape[-1])
        d = ctx.to_device(dm_k, np.complex128)
        imax = ctx.zeros((1,), np.float64)
        out = fold_k2R_dev(d, kmesh, spin, ncol, imag_max=imax).get().reshape(dm_k.shape)
        im = float(imax.get()[0])
        if im > tol:
            log.warn("k2R: non-zero imaginary part: %15.8g", im)
        return out
    raise ValueError("unknown shape of dm_k: %s" % str(dm_k.shape))
