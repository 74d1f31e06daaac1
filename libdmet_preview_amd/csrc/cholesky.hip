// Modified (pivoted, incomplete) Cholesky decomposition of a 4-fold ERI matrix -- the arithmetic behind convert_eri_to_gdf
// (basis_transform/eri_transform.py:1483-1535: a molecular ERI rewritten as a Gamma-point cderi container), i.e.
// utils/cholesky.py:21-52 `modified_cholesky` (one spin block) and :54-105 `modified_cholesky_uhf` (aa, bb, ab blocks sharing
// one pivot sequence over the 2 n diagonal entries).
//
// The reference's loop, kept step for step because the PIVOT SEQUENCE is the result (vector i is the residual row of the i-th
// pivot scaled by 1 / sqrt(residual diagonal)):
//     v_0 = M[p_0] / sqrt(d[p_0]),  p_0 = argmax d                      (d = diag M)
//     repeat:  A += v_i * v_i;  p = argmax |d - A| (first maximum);  delta = |d - A|[p];
//              R = sum_{j <= i} v_j[p] * v_j   (j ascending);  v_{i+1} = (M[p] - R) / sqrt(delta);  stop once delta < max_error
// -- the vector of the step that meets the tolerance is still appended (numpy does so too).  Every product and sum is rounded on
// its own (__dmul_rn / __dadd_rn / __ddiv_rn: what numpy's element-wise `vec[idx] * vec`, `R += ...`, `/` do), in the same order,
// so pivots and vectors agree with the host loop bit for bit on a given input; ties go to the first index like np.argmax.
//
// One workgroup: the decomposition is a chain of n_chol dependent steps (argmax -> row update), each O(n i) -- 2.5 GFLOP for a
// 100-orbital molecule (n = 5050 pairs, ~1000 vectors), ~20 ms on one CU against the seconds of the Python loop; a converter that
// runs once per system, not a hot kernel.  Thread <-> pair column; the coefficients v_j[p] of a step are staged in LDS.
#include "common.h"
#include <cmath>

namespace {

constexpr int CH_NT = 1024;
constexpr int CH_COEF = 4096;                  // coefficients staged per chunk of previous vectors

struct CholArgs {
    int n, uhf, max_vecs;
    const double *m0, *m1, *m2;                // rhf: m0 (n x n); uhf: aa, bb, ab (each n x n, row-major; ab[i][j] = (aa-pair i | bb-pair j))
    double max_error;
    double *vecs;                              // rhf: [max_vecs][n]; uhf: [2][max_vecs][n]
    double *work;                              // diag [N] | approx [N],  N = n (rhf) or 2 n (uhf)
    int *out;                                  // [0] number of vectors, [1] 1 = the loop ran out of cycles (reference: a warning)
};

// np.argmax order: a NaN is the maximum (the FIRST NaN wins), otherwise the first largest value.  An all-zero or non-positive pair
// diagonal makes the first vector 0 / 0 = NaN; with a plain `>` no candidate would ever be taken and the pivot would stay at its
// out-of-range start value -- the reference returns NaN vectors with a warning there, and so does this kernel.
__device__ __forceinline__ bool argmax_takes(double ov, int oi, double v, int idx) {
    if (ov != ov) return !(v != v) || oi < idx;
    if (v != v) return false;
    return ov > v || (ov == v && oi < idx);
}

// (value, index) of the first maximum over the workgroup; every thread receives it
__device__ void block_argmax(double &v, int &idx, double *shv, int *shi) {
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (argmax_takes(ov, oi, v, idx)) { v = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { shv[wave] = v; shi[wave] = idx; }
    __syncthreads();
    v = shv[0]; idx = shi[0];
    for (int w = 1; w < CH_NT / 64; ++w)
        if (argmax_takes(shv[w], shi[w], v, idx)) { v = shv[w]; idx = shi[w]; }
}

__global__ __launch_bounds__(CH_NT) void modified_cholesky_kernel(const CholArgs g) {
    __shared__ double shv[CH_NT / 64];
    __shared__ int shi[CH_NT / 64];
    __shared__ double coef[CH_COEF];
    const int n = g.n, tid = threadIdx.x;
    const int N = g.uhf ? 2 * n : n;
    double *diag = g.work, *approx = g.work + N;
    const long long vs = (long long)g.max_vecs * n;            // distance between the alpha and beta vector sets (uhf)
    // row `idx` of the stacked matrix restricted to the alpha (part 0) / beta (part 1) pair columns
    auto mrow = [&](int idx, int part, int c) -> double {
        if (!g.uhf) return g.m0[(long long)idx * n + c];
        if (idx < n) return part == 0 ? g.m0[(long long)idx * n + c] : g.m2[(long long)idx * n + c];             // mat[0][idx], mat[2][idx]
        return part == 0 ? g.m2[(long long)c * n + (idx - n)] : g.m1[(long long)(idx - n) * n + c];              // mat[2].T[idx - n], mat[1][idx - n]
    };
    for (int t = tid; t < N; t += CH_NT) {
        diag[t] = !g.uhf ? g.m0[(long long)t * n + t] : (t < n ? g.m0[(long long)t * n + t] : g.m1[(long long)(t - n) * n + (t - n)]);
        approx[t] = 0.0;
    }
    __syncthreads();
    double best = -INFINITY;
    int idx = 0x7fffffff;
    for (int t = tid; t < N; t += CH_NT)
        if (argmax_takes(diag[t], t, best, idx)) { best = diag[t]; idx = t; }  // (ascending t per thread: the first maximum)
    block_argmax(best, idx, shv, shi);
    idx = min(idx, N - 1);
    double delta_max = best;
    {
        const double sq = __dsqrt_rn(delta_max);
        for (int part = 0; part <= g.uhf; ++part)
            for (int c = tid; c < n; c += CH_NT) g.vecs[part * vs + c] = __ddiv_rn(mrow(idx, part, c), sq);
    }
    __syncthreads();
    int nvec = 1, exhausted = 1;
    const int max_cycle = 2 * n + 1;
    for (int i = 0; i < max_cycle; ++i) {
        // approx += v_i * v_i; delta = diag - approx; pivot = first maximum of |delta|
        best = -INFINITY;
        idx = 0x7fffffff;
        for (int t = tid; t < N; t += CH_NT) {
            const double v = t < n ? g.vecs[(long long)i * n + t] : g.vecs[vs + (long long)i * n + (t - n)];
            const double a = __dadd_rn(approx[t], __dmul_rn(v, v));
            approx[t] = a;
            const double d = fabs(__dsub_rn(diag[t], a));
            if (argmax_takes(d, t, best, idx)) { best = d; idx = t; }
        }
        block_argmax(best, idx, shv, shi);
        idx = min(idx, N - 1);
        delta_max = best;
        // R = sum_j v_j[idx] * v_j (j ascending), then the new vector; the coefficients v_j[idx] in chunks through LDS
        const int ipart = (g.uhf && idx >= n) ? 1 : 0, icol = ipart ? idx - n : idx;
        const double sq = __dsqrt_rn(delta_max);
        for (int part = 0; part <= g.uhf; ++part) {
            // per-thread running sums for its columns: at most ceil(n / 1024) columns -- kept in the output row while chunks pass
            double *dst = g.vecs + part * vs + (long long)(i + 1) * n;
            for (int c = tid; c < n; c += CH_NT) dst[c] = 0.0;
            for (int j0 = 0; j0 <= i; j0 += CH_COEF) {
                const int jn = min(CH_COEF, i + 1 - j0);
                __syncthreads();
                for (int j = tid; j < jn; j += CH_NT) coef[j] = g.vecs[ipart * vs + (long long)(j0 + j) * n + icol];
                __syncthreads();
                for (int c = tid; c < n; c += CH_NT) {
                    double r = dst[c];
                    const double *col = g.vecs + part * vs + (long long)j0 * n + c;
                    for (int j = 0; j < jn; ++j) r = __dadd_rn(r, __dmul_rn(coef[j], col[(long long)j * n]));
                    dst[c] = r;
                }
            }
            for (int c = tid; c < n; c += CH_NT) dst[c] = __ddiv_rn(__dsub_rn(mrow(idx, part, c), dst[c]), sq);
        }
        nvec = i + 2;
        __syncthreads();
        if (delta_max < g.max_error) { exhausted = 0; break; }
    }
    if (tid == 0) { g.out[0] = nvec; g.out[1] = exhausted; }
}

// Column-pivoted QR, pivots only (Businger-Golub: at every step the column with the largest residual norm) -- the column
// selection of SCDM (lo/scdm.py:116-150 scdm_model: `la.qr(psiT, pivoting=True)` of the bath orbitals, of which only perm[:nmo] is
// used; LAPACK dgeqp3 picks by the same rule, first index on ties).  The "columns" are the ROWS of the caller's (ncol x vlen)
// array (psiT = B^T: column j of psiT is site j of the bath orbitals B).  Residuals by modified Gram-Schmidt against the chosen
// directions, their norms recomputed exactly at every step (no downdating: ncol vlen npiv flop, 0.5 GFLOP at C5 sizes, one CU).
__global__ __launch_bounds__(CH_NT) void cpqr_pivots_kernel(int ncol, int vlen, const double *__restrict__ cols, int npiv, double *R,
                                                            double *nrm, int *piv) {
    __shared__ double shv[CH_NT / 64];
    __shared__ int shi[CH_NT / 64];
    extern __shared__ double q[];                       // the current direction, vlen doubles
    const int tid = threadIdx.x;
    for (int j = tid; j < ncol; j += CH_NT) {
        double s = 0.0;
        for (int k = 0; k < vlen; ++k) {
            const double v = cols[(long long)j * vlen + k];
            R[(long long)j * vlen + k] = v;
            s += v * v;
        }
        nrm[j] = s;
    }
    __syncthreads();
    for (int t = 0; t < npiv; ++t) {
        double best = -INFINITY;
        int idx = 0x7fffffff;
        for (int j = tid; j < ncol; j += CH_NT)
            if (argmax_takes(nrm[j], j, best, idx)) { best = nrm[j]; idx = j; }     // chosen columns carry -1
        block_argmax(best, idx, shv, shi);
        idx = min(idx, ncol - 1);
        if (tid == 0) piv[t] = idx;
        const double inv = best > 0.0 ? 1.0 / sqrt(best) : 0.0;
        for (int k = tid; k < vlen; k += CH_NT) q[k] = R[(long long)idx * vlen + k] * inv;
        __syncthreads();
        for (int j = tid; j < ncol; j += CH_NT) {
            double *r = R + (long long)j * vlen;
            if (j == idx) { nrm[j] = -1.0; continue; }
            if (nrm[j] < 0.0) continue;
            double dot = 0.0;
            for (int k = 0; k < vlen; ++k) dot += q[k] * r[k];
            double s = 0.0;
            for (int k = 0; k < vlen; ++k) {
                const double v = r[k] - dot * q[k];
                r[k] = v;
                s += v * v;
            }
            nrm[j] = s;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" {

int dmk_cpqr_pivots(dmk_ctx *ctx, int ncol, int vlen, const double *cols, int npiv, int32_t *piv_out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (ncol < 1 || vlen < 1 || vlen > 4096 || npiv < 0 || npiv > ncol || !cols || (npiv > 0 && !piv_out))
        return dmk_fail(ctx, DMK_ERR_INVALID, "cpqr_pivots: bad arguments (1 <= vector length <= 4096, npiv <= ncol)");
    if (npiv == 0) return DMK_OK;
    void *ws = nullptr;
    const size_t rbytes = (size_t)ncol * vlen * sizeof(double), nbytes = (size_t)ncol * sizeof(double);
    int rc = dmk_scratch(ctx, rbytes + nbytes + (size_t)npiv * sizeof(int) + 256, &ws);
    if (rc) return rc;
    double *R = static_cast<double *>(ws), *nrm = R + (size_t)ncol * vlen;
    int *piv = reinterpret_cast<int *>(nrm + ncol);
    {
        FamScope fs(ctx, DMK_FAM_MISC);
        hipLaunchKernelGGL(cpqr_pivots_kernel, dim3(1), dim3(CH_NT), (size_t)vlen * sizeof(double), ctx->stream, ncol, vlen, cols, npiv, R,
                           nrm, piv);
        DMK_CHECK_LAUNCH(ctx);
    }
    DMK_HIP(ctx, hipMemcpyAsync(piv_out, piv, (size_t)npiv * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DMK_OK;
}

int dmk_modified_cholesky(dmk_ctx *ctx, int n, int uhf, const double *m_aa, const double *m_bb, const double *m_ab, double max_error,
                          int max_vecs, double *vecs, int32_t *nvec_out, int32_t *exhausted_out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n < 1 || !m_aa || (uhf && (!m_bb || !m_ab)) || !vecs || !nvec_out || !(max_error > 0.0))
        return dmk_fail(ctx, DMK_ERR_INVALID, "modified_cholesky: bad arguments");
    if (max_vecs < 2 * n + 2)
        return dmk_fail(ctx, DMK_ERR_INVALID, "modified_cholesky: room for %d vectors needed (2 n + 2: the reference's cycle limit)", 2 * n + 2);
    const int N = uhf ? 2 * n : n;
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, (size_t)2 * N * sizeof(double) + 256, &ws);
    if (rc) return rc;
    CholArgs g;
    g.n = n; g.uhf = uhf ? 1 : 0; g.max_vecs = max_vecs; g.m0 = m_aa; g.m1 = m_bb; g.m2 = m_ab; g.max_error = max_error;
    g.vecs = vecs; g.work = static_cast<double *>(ws);
    g.out = reinterpret_cast<int *>(static_cast<char *>(ws) + (size_t)2 * N * sizeof(double));
    {
        FamScope fs(ctx, DMK_FAM_MISC);
        hipLaunchKernelGGL(modified_cholesky_kernel, dim3(1), dim3(CH_NT), 0, ctx->stream, g);
        DMK_CHECK_LAUNCH(ctx);
    }
    int out[2] = {0, 0};
    DMK_HIP(ctx, hipMemcpyAsync(out, g.out, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *nvec_out = out[0];
    if (exhausted_out) *exhausted_out = out[1];
    return DMK_OK;
}

}  // extern "C"
