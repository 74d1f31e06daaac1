// Freivalds probe of the ERI contraction (a self-check of K7, dgemm_tn.hip, that covers EVERY tile row and column).
//
//   eri[b] = sum_kL w (Re_a^T Re_b + Im_a^T Im_b)          basis_transform/eri_transform.py:451-478 (_Lij_s4_to_eri)
//
// For a probe vector x of the pair space the same sum applied to x is two matrix-VECTOR passes over the resident planes,
//   t_b = X_b x   (one dot product per auxiliary row),        yref[b] += w X_a^T t_b   (column sums weighted by t),
// written here as two plain streaming kernels that share NOTHING with the tiled GEMM: no tile table, no super-block order,
// no mirrored store, no band cuts.  The caller compares  eri[b] x  (dmk_dgemv2 on the finished ERI) with yref[b]: a tile the
// GEMM dropped, wrote to the wrong place or mirrored wrongly changes a whole 128-entry stretch of eri x by O(|eri| |x|),
// twelve orders of magnitude above the rounding noise of the comparison.  HBM-bound and cheap: five passes over one kL's
// planes (2.1 GB at C5) per kL against the 3.5 TFLOP of its contraction.
#include "common.h"

namespace {

constexpr int PB_NT = 256;
constexpr int PB_ROWCH = 64;        // auxiliary rows per workgroup of the column-sum kernel

// t[r] = sum_q X[r][q] x[q]: one workgroup per row, fixed reduction order
__global__ __launch_bounds__(PB_NT) void probe_rowdot_kernel(const double *__restrict__ X, long long ld, long long n,
                                                             const double *__restrict__ x, double *__restrict__ t) {
    __shared__ double red[PB_NT / 64];
    const double *row = X + (long long)blockIdx.x * ld;
    double s = 0.0;
    for (long long q = threadIdx.x; q < n; q += PB_NT) s = fma(row[q], x[q], s);
    s = dmk_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) t[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// y[q] += w sum_r X[r][q] t[r] over a chunk of PB_ROWCH rows; thread <-> column (coalesced rows)
__global__ __launch_bounds__(PB_NT) void probe_colsum_kernel(const double *__restrict__ X, long long ld, int nrows, long long n,
                                                             const double *__restrict__ t, double w, double *__restrict__ y) {
    __shared__ double ts[PB_ROWCH];
    const int r0 = blockIdx.y * PB_ROWCH;
    const int nr = nrows - r0 < PB_ROWCH ? nrows - r0 : PB_ROWCH;
    if ((int)threadIdx.x < nr) ts[threadIdx.x] = t[r0 + threadIdx.x];
    __syncthreads();
    const long long q = (long long)blockIdx.x * PB_NT + threadIdx.x;
    if (q >= n) return;
    const double *col = X + (long long)r0 * ld + q;
    double s = 0.0;
    for (int r = 0; r < nr; ++r) s = fma(col[(long long)r * ld], ts[r], s);
    unsafeAtomicAdd(y + q, w * s);
}

}  // namespace

// yref[b] += w X_a^T (X_b x) for the spin blocks b = (aa) or (aa, ab, bb) of ONE plane slot: X0 / X1 point at the slot's planes of
// spin 0 / 1 (`nrows` rows of length npair, `ld` apart: both halves of a weight-2 kL, the Re half of a weight-1 kL; padding rows
// of the plane layout are zero and may be included).  `twork`: 2 * nrows doubles of scratch.
int launch_eri_probe_slot(dmk_ctx *ctx, const double *X0, const double *X1, int nrows, long long npair, long long ld, double w,
                          const double *x, double *yref, double *twork) {
    FamScope fs(ctx, DMK_FAM_MISC);
    const dim3 cgrid((unsigned)((npair + PB_NT - 1) / PB_NT), (unsigned)((nrows + PB_ROWCH - 1) / PB_ROWCH));
    double *t0 = twork, *t1 = twork + nrows;
    hipLaunchKernelGGL(probe_rowdot_kernel, dim3(nrows), dim3(PB_NT), 0, ctx->stream, X0, ld, npair, x, t0);
    hipLaunchKernelGGL(probe_colsum_kernel, cgrid, dim3(PB_NT), 0, ctx->stream, X0, ld, nrows, npair, t0, w, yref);
    if (X1) {
        hipLaunchKernelGGL(probe_rowdot_kernel, dim3(nrows), dim3(PB_NT), 0, ctx->stream, X1, ld, npair, x, t1);
        // (ab): eri[1] = X0^T X1  ->  eri[1] x = X0^T (X1 x)
        hipLaunchKernelGGL(probe_colsum_kernel, cgrid, dim3(PB_NT), 0, ctx->stream, X0, ld, nrows, npair, t1, w, yref + npair);
        hipLaunchKernelGGL(probe_colsum_kernel, cgrid, dim3(PB_NT), 0, ctx->stream, X1, ld, nrows, npair, t1, w, yref + 2 * npair);
    }
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}
