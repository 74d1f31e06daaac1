// Shared pieces of the hot half-transform kernels (zhot.hip, zhot_tab.hip): the complex 16 x 16 x 4 tile step on the
// real f64 MFMA and the tril-pack accumulate epilogue.
#pragma once
#include "common.h"

namespace {

constexpr int HNT = 256;

// Karatsuba "3M" complex product: T1 += Ar Br, T2 += Ai Bi, T3 += (Ar + Ai)(Br + Bi); Re = T1 - T2, Im = T3 - T1 - T2.
// 25 % fewer MFMAs than the four-product form for 1.5x the accumulator registers; with LDS-DMA there are no staging
// registers in these kernels, so up to 9 accumulator tiles (216 VGPRs) of a wave fit two waves per SIMD without spilling.
// Normwise backward stable; the parity tests hold it to the same 1e-8 / 1e-10 budgets (measured 2e-15).
struct cfrag { double2 v; double s; };      // operand fragment and re + im
struct cacc { d4_t p, q, t; };              // T1, T2, T3
__device__ __forceinline__ void cacc_zero(cacc &c) {
    c.p = d4_t{0.0, 0.0, 0.0, 0.0};
    c.q = d4_t{0.0, 0.0, 0.0, 0.0};
    c.t = d4_t{0.0, 0.0, 0.0, 0.0};
}
__device__ __forceinline__ double2 lds_frag(const double2 *p) { return *p; }
__device__ __forceinline__ cfrag cfrag_of(double2 v) {
    cfrag f;
    f.v = v;
    f.s = v.x + v.y;
    return f;
}
__device__ __forceinline__ void cmfma(cacc &c, const cfrag &a, const cfrag &b) {
    c.p = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v.x, b.v.x, c.p, 0, 0, 0);
    c.q = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v.y, b.v.y, c.q, 0, 0, 0);
    c.t = __builtin_amdgcn_mfma_f64_16x16x4f64(a.s, b.s, c.t, 0, 0, 0);
}
__device__ __forceinline__ double cacc_re(const cacc &c, int r) { return c.p[r] - c.q[r]; }
__device__ __forceinline__ double cacc_im(const cacc &c, int r) { return (c.t[r] - c.p[r]) - c.q[r]; }

// RE = true: only the REAL part of the product is wanted -- a momentum transfer kL that is its own time-reversal partner
// (weight 1) contributes Re(Lij)^T Re(Lij) only (eri_transform.py:453-455, 464-467: the imaginary part of its planes is never read),
// so step 2 runs Re S = Ur Cr - Ui Ci: two real MFMAs per complex block step instead of the three of 3M, no operand sums, no T3
// accumulator, one plane atomic per element.  RE = false is the code above, instruction for instruction.
template <bool RE> __device__ __forceinline__ cfrag cfrag_of_t(double2 v) {
    if constexpr (RE) { cfrag f; f.v = v; f.s = 0.0; return f; }
    else return cfrag_of(v);
}
template <bool RE> __device__ __forceinline__ void cmfma_t(cacc &c, const cfrag &a, const cfrag &b) {
    if constexpr (RE) {
        c.p = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v.x, b.v.x, c.p, 0, 0, 0);
        c.q = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v.y, b.v.y, c.q, 0, 0, 0);
    } else {
        cmfma(c, a, b);
    }
}

// planes[(ri * naux + L) * npair + row (row + 1) / 2 + col] += value for row >= col, row < nrows.
// Single writer per address per launch -> deterministic; fire-and-forget atomics: no load latency in the epilogue.
__device__ __forceinline__ void pack_acc(double *planes, long long naux, long long npair, int L, int row, int col,
                                         double vr, double vi, int nrows = 0x7fffffff) {
    if (row >= col && row < nrows) {
        const long long idx = (long long)row * (row + 1) / 2 + col;
        unsafeAtomicAdd(planes + (long long)L * npair + idx, vr);
        unsafeAtomicAdd(planes + (naux + (long long)L) * npair + idx, vi);
    }
}

template <bool RE> __device__ __forceinline__ void pack_acc_t(double *planes, long long naux, long long npair, int L, int row, int col,
                                                             const cacc &c, int r, int nrows = 0x7fffffff) {
    if constexpr (RE) {
        if (row >= col && row < nrows)
            unsafeAtomicAdd(planes + (long long)L * npair + (long long)row * (row + 1) / 2 + col, cacc_re(c, r));
    } else {
        pack_acc(planes, naux, npair, L, row, col, cacc_re(c, r), cacc_im(c, r), nrows);
    }
}

}  // namespace
