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


def multiply_basis(C_ao_lo, C_lo_eo):
    """C_ao_eo = C_ao_lo * C_lo_eo per (spin,) k-point."""
    C_ao_lo = np.asarray(C_ao_lo)
    C_lo_eo = np.asarray(C_lo_eo)
    nkpts, nlo, neo = C_lo_eo.shape[-3:]
    nao = C_ao_lo.shape[-2]
    if C_ao_lo.ndim == 3 and C_lo_eo.ndim == 3:
        return _result(_bgemm("N", "N", C_ao_lo, C_lo_eo), C_ao_lo, C_lo_eo)
    if C_ao_lo.ndim == 3 and C_lo_eo.ndim == 4:
        spin = C_lo_eo.shape[0]
        C_ao_lo = add_spin_dim(C_ao_lo, spin)
    elif C_ao_lo.ndim == 4 and C_lo_eo.ndim == 3:
        spin = C_ao_lo.shape[0]
        C_lo_eo = add_spin_dim(C_lo_eo, spin)
    elif C_ao_lo.ndim == 4 and C_lo_eo.ndim == 4:
        spin = max(C_ao_lo.shape[0], C_lo_eo.shape[0])
        C_ao_lo = add_spin_dim(C_ao_lo, spin)
        C_lo_eo = add_spin_dim(C_lo_eo, spin)
    else:
        raise ValueError("invalid shape for multiply_basis: C_ao_lo shape %s, C_lo_eo shape: %s"
                         % (C_ao_lo.shape, C_lo_eo.shape))
    out = _bgemm("N", "N", C_ao_lo.reshape(spin * nkpts, nao, nlo), C_lo_eo.reshape(spin * nkpts, nlo, neo))
    return _result(out.reshape(spin, nkpts, nao, neo), C_ao_lo, C_lo_eo)


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
    r"""h^{LO} = C^{\dagger} h^{AO} C, with kpts."""
    h_ao_ao = np.asarray(h_ao_ao)
    C_ao_lo = np.asarray(C_ao_lo)
    nkpts = C_ao_lo.shape[-3]
    nlo = C_ao_lo.shape[-1]
    res_type = np.result_type(h_ao_ao.dtype, C_ao_lo.dtype)
    if h_ao_ao.ndim == 0:
        return np.ones((nkpts, nlo, nlo), dtype=res_type) * h_ao_ao
    elif h_ao_ao.ndim == 1:
        spin = len(h_ao_ao)
        h_lo_lo = np.ones((spin, nkpts, nlo, nlo), dtype=res_type)
        for s in range(spin):
            h_lo_lo[s] *= h_ao_ao[s]
        return h_lo_lo
    if C_ao_lo.ndim == 3 and h_ao_ao.ndim == 3:
        return _result(_triple("C", C_ao_lo, h_ao_ao, "N", C_ao_lo), h_ao_ao, C_ao_lo)
    spin = get_spin_dim((h_ao_ao, C_ao_lo))
    h_ao_ao = add_spin_dim(h_ao_ao, spin)
    C_ao_lo = add_spin_dim(C_ao_lo, spin)
    assert h_ao_ao.ndim == C_ao_lo.ndim
    nao = C_ao_lo.shape[-2]
    Cf = C_ao_lo.reshape(spin * nkpts, nao, nlo)
    out = _triple("C", Cf, h_ao_ao.reshape(spin * nkpts, nao, nao), "N", Cf)
    return _result(out.reshape(spin, nkpts, nlo, nlo), h_ao_ao, C_ao_lo)


def transform_rdm1_to_lo(dm_ao_ao, C_ao_lo, S_ao_ao):
    r"""\gamma^{LO} = C^{-1} \gamma^{AO} (C^{-1})^{\dagger},  C^{-1} = C^{\dagger} S."""
    dm_ao_ao = np.asarray(dm_ao_ao)
    C_ao_lo = np.asarray(C_ao_lo)
    S_ao_ao = np.asarray(S_ao_ao)
    nkpts = C_ao_lo.shape[-3]
    nlo = C_ao_lo.shape[-1]
    nao = C_ao_lo.shape[-2]
    if C_ao_lo.ndim == 3 and dm_ao_ao.ndim == 3:
        Cinv = _bgemm("C", "N", C_ao_lo, S_ao_ao)
        return _result(_triple("N", Cinv, dm_ao_ao, "C", Cinv), dm_ao_ao, C_ao_lo, S_ao_ao)
    spin = get_spin_dim((dm_ao_ao, C_ao_lo))
    dm_ao_ao = add_spin_dim(dm_ao_ao, spin)
    C_ao_lo = add_spin_dim(C_ao_lo, spin)
    assert dm_ao_ao.ndim == C_ao_lo.ndim
    Cf = C_ao_lo.reshape(spin * nkpts, nao, nlo)
    Sf = np.ascontiguousarray(np.broadcast_to(S_ao_ao[None], (spin,) + S_ao_ao.shape)).reshape(spin * nkpts, nao, nao)
    Cinv = _bgemm("C", "N", Cf, Sf)
    out = _triple("N", Cinv, dm_ao_ao.reshape(spin * nkpts, nao, nao), "C", Cinv)
    return _result(out.reshape(spin, nkpts, nlo, nlo), dm_ao_ao, C_ao_lo, S_ao_ao)


def transform_rdm1_to_ao(dm_lo_lo, C_ao_lo):
    r"""\gamma^{AO} = C \gamma^{LO} C^{\dagger}."""
    dm_lo_lo = np.asarray(dm_lo_lo)
    C_ao_lo = np.asarray(C_ao_lo)
    nkpts = C_ao_lo.shape[-3]
    nao = C_ao_lo.shape[-2]
    nlo = C_ao_lo.shape[-1]
    if C_ao_lo.ndim == 3 and dm_lo_lo.ndim == 3:
        return _result(_triple("N", C_ao_lo, dm_lo_lo, "C", C_ao_lo), dm_lo_lo, C_ao_lo)
    spin = get_spin_dim((dm_lo_lo, C_ao_lo))
    dm_lo_lo = add_spin_dim(dm_lo_lo, spin)
    C_ao_lo = add_spin_dim(C_ao_lo, spin)
    assert dm_lo_lo.ndim == C_ao_lo.ndim
    Cf = C_ao_lo.reshape(spin * nkpts, nao, nlo)
    out = _triple("N", Cf, dm_lo_lo.reshape(spin * nkpts, nlo, nlo), "C", Cf)
    return _result(out.reshape(spin, nkpts, nao, nao), dm_lo_lo, C_ao_lo)
