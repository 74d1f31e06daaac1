"""
Basis algebra with the reference's signatures (libdmet/basis_transform/make_basis.py),
every product running through the batched complex MFMA GEMM of libdmetk:

  multiply_basis         make_basis.py:923-962   (utils/misc.py:49-59 kdot)
  transform_h1_to_lo     make_basis.py:524-558
  transform_rdm1_to_lo   make_basis.py:590-618
  transform_rdm1_to_ao   make_basis.py:620-644
"""
import numpy as np

from libdmet_preview_amd._lib import lib, get_ctx
from libdmet_preview_amd.utils.misc import add_spin_dim, get_spin_dim

_OP = {"N": 0, "T": 1, "C": 2}


def bgemm_dev(ctx, opA, opB, M, N, K, batch, A, strideA, B, strideB, C=None, strideC=None, alpha=1.0):
    """C[b] = alpha op(A[b]) op(B[b]) on device arrays (c128)."""
    if C is None:
        C = ctx.empty((batch, M, N), np.complex128)
    if strideC is None:
        strideC = M * N
    ctx.check(lib.dmk_zgemm_batched(ctx.h, _OP[opA], _OP[opB], int(M), int(N), int(K), int(batch), float(alpha),
                                    A.ptr, int(strideA), B.ptr, int(strideB), C.ptr, int(strideC)))
    return C


def _bgemm(opA, opB, a, b):
    """Host convenience: batched op(a[k]) @ op(b[k]) for (nk, r, c) stacks -> numpy c128."""
    ctx = get_ctx()
    a = np.ascontiguousarray(a, dtype=np.complex128)
    b = np.ascontiguousarray(b, dtype=np.complex128)
    nk = a.shape[0]
    M, K = (a.shape[1], a.shape[2]) if opA == "N" else (a.shape[2], a.shape[1])
    K2, N = (b.shape[1], b.shape[2]) if opB == "N" else (b.shape[2], b.shape[1])
    assert K == K2 and b.shape[0] == nk
    da, db = ctx.to_device(a), ctx.to_device(b)
    return bgemm_dev(ctx, opA, opB, M, N, K, nk, da, a.shape[1] * a.shape[2], db, b.shape[1] * b.shape[2]).get()


def _result(x, *inputs):
    """Reference dtype rule: np.result_type of the inputs (real only if everything is real)."""
    rt = np.result_type(*[np.asarray(i).dtype for i in inputs])
    if rt.kind != "c":
        return np.ascontiguousarray(x.real)
    return x


def _spin_stacks(*stacks):
    """Bring (nk, ., .) / (spin, nk, ., .) stacks to a common leading spin axis (a single channel is repeated).
    Returns the flattened (spin * nk, ., .) views, the spin count and whether ANY input carried a spin axis."""
    stacks = [np.asarray(x) for x in stacks]
    if any(x.ndim not in (3, 4) for x in stacks):
        raise ValueError("expected (nkpts, n, m) or (spin, nkpts, n, m) arrays, got shapes %s" % ([x.shape for x in stacks],))
    spin = get_spin_dim(stacks)
    flat = []
    for x in stacks:
        x = add_spin_dim(x, spin)
        flat.append(x.reshape((spin * x.shape[1],) + x.shape[2:]))
    return flat, spin, any(x.ndim == 4 for x in stacks)


def _unflatten(out, spin, keep_spin):
    return out.reshape((spin, out.shape[0] // spin) + out.shape[1:]) if keep_spin else out


def multiply_basis(C_ao_lo, C_lo_eo):
    """C_ao_eo[s, k] = C_ao_lo[s, k] C_lo_eo[s, k] (make_basis.py:923-962); either factor may omit the spin axis."""
    try:
        (A, B), spin, keep = _spin_stacks(C_ao_lo, C_lo_eo)
    except ValueError:
        raise ValueError("invalid shape for multiply_basis: C_ao_lo shape %s, C_lo_eo shape: %s"
                         % (np.shape(C_ao_lo), np.shape(C_lo_eo)))
    return _result(_unflatten(_bgemm("N", "N", A, B), spin, keep), C_ao_lo, C_lo_eo)


def _triple(opL, L, Mid, opR, Rt):
    """op(L[k]) @ Mid[k] @ op(R[k]) for flat (nk, ., .) stacks, intermediates stay on the device."""
    ctx = get_ctx()
    L = np.ascontiguousarray(L, dtype=np.complex128)
    Mid = np.ascontiguousarray(Mid, dtype=np.complex128)
    Rt = np.ascontiguousarray(Rt, dtype=np.complex128)
    nk = L.shape[0]
    dL, dM, dR = ctx.to_device(L), ctx.to_device(Mid), ctx.to_device(Rt)
    m1, k1 = (L.shape[1], L.shape[2]) if opL == "N" else (L.shape[2], L.shape[1])
    n1 = Mid.shape[2]
    assert Mid.shape[1] == k1
    T = bgemm_dev(ctx, opL, "N", m1, n1, k1, nk, dL, L.shape[1] * L.shape[2], dM, Mid.shape[1] * Mid.shape[2])
    k2, n2 = (Rt.shape[1], Rt.shape[2]) if opR == "N" else (Rt.shape[2], Rt.shape[1])
    assert k2 == n1
    out = bgemm_dev(ctx, "N", opR, m1, n2, k2, nk, T, m1 * n1, dR, Rt.shape[1] * Rt.shape[2])
    return out.get()


def transform_h1_to_lo(h_ao_ao, C_ao_lo):
    r"""h^{LO}[s, k] = C^{\dagger} h^{AO} C (make_basis.py:524-558).  A scalar (or one scalar per spin) stands for that
    multiple of the all-ones matrix and is only broadcast."""
    h_ao_ao = np.asarray(h_ao_ao)
    C_ao_lo = np.asarray(C_ao_lo)
    nkpts, nlo = C_ao_lo.shape[-3], C_ao_lo.shape[-1]
    if h_ao_ao.ndim <= 1:
        ones = np.ones((nkpts, nlo, nlo), dtype=np.result_type(h_ao_ao.dtype, C_ao_lo.dtype))
        return ones * h_ao_ao if h_ao_ao.ndim == 0 else np.asarray([ones * x for x in h_ao_ao])
    (H, Cf), spin, keep = _spin_stacks(h_ao_ao, C_ao_lo)
    return _result(_unflatten(_triple("C", Cf, H, "N", Cf), spin, keep), h_ao_ao, C_ao_lo)


def transform_rdm1_to_lo(dm_ao_ao, C_ao_lo, S_ao_ao):
    r"""\gamma^{LO} = C^{-1} \gamma^{AO} (C^{-1})^{\dagger} with C^{-1} = C^{\dagger} S (make_basis.py:590-618); the overlap has
    no spin axis."""
    (D, Cf, Sf), spin, _ = _spin_stacks(dm_ao_ao, C_ao_lo, S_ao_ao)
    keep = np.ndim(dm_ao_ao) == 4 or np.ndim(C_ao_lo) == 4
    Cinv = _bgemm("C", "N", Cf, Sf)
    return _result(_unflatten(_triple("N", Cinv, D, "C", Cinv), spin, keep), dm_ao_ao, C_ao_lo, S_ao_ao)


def transform_rdm1_to_ao(dm_lo_lo, C_ao_lo):
    r"""\gamma^{AO} = C \gamma^{LO} C^{\dagger} (make_basis.py:620-644)."""
    (D, Cf), spin, keep = _spin_stacks(dm_lo_lo, C_ao_lo)
    return _result(_unflatten(_triple("N", Cf, D, "C", Cf), spin, keep), dm_lo_lo, C_ao_lo)

