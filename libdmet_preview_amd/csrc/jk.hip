// K10 -- ERI x density: Coulomb and exchange matrices from the 4-fold packed embedding ERI resident in HBM.
//
// Replaces pyscf.scf.hf.dot_eri_dm as the reference calls it from solver/scf.py:255-335 (_get_jk, used by
// _get_veff :337-352 and routine/slater.py:477-523 get_veff -> JK_emb of __embHam1e :599-606):
//     J: (ij|kl), kl -> ij         K: (ij|kl), il -> jk
// on E[pair(i,j)][pair(k,l)], pair(a,b) = a(a+1)/2 + b (a >= b), npair x npair f64 (8.7 GB per spin block at C5).
//
// Bound: HBM.  Every kernel streams E exactly once with coalesced 512 B wave loads:
//   jk_j_kernel   32 packed rows per workgroup; row dots E x~ (x~ = pair-folded density) reduced in the wave,
//                 column sums E^T x~' (the alpha-beta block needs both directions, solver/scf.py:321-327) as
//                 per-row-block partials that a second kernel adds in a fixed order (no atomics: bit-reproducible)
//   jk_k_kernel   one workgroup per packed row (i,j): the row is the packed symmetric matrix M[k][l] = (ij|kl);
//                 each wave owns rows k of M, lanes run along l, so that y = M x for x = dm[i,:] and dm[j,:]
//                 needs one wave reduction per k (y[k] += M[k][l] x[l], l < k) and private per-lane sums for the
//                 mirrored half (y[l] += M[k][l] x[k]).  K[j,:] += M dm[i,:] and, for i != j, K[i,:] += M dm[j,:]
//                 are written as per-row vectors and gathered by jk_k_reduce_kernel in a fixed order.
// Algorithmic bytes: 8 * npair^2 per block and pass (J and K are separate passes over aa / bb; ab needs J only).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int NW = NT / 64;
constexpr int JRB = 32;          // packed rows per workgroup in the J kernel

__device__ __forceinline__ double wave_sum(double v) { return dmk_wave_sum(v); }

// x~[pair(k,l)] = dm[k][l] + dm[l][k] (k > l), dm[k][k]
__global__ void jk_fold_dm_kernel(int n, const double *__restrict__ dm, double *__restrict__ xt) {
    const long long npair = (long long)n * (n + 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npair;
         t += (long long)gridDim.x * blockDim.x) {
        int k = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((long long)(k + 1) * (k + 2) / 2 <= t) ++k;
        while ((long long)k * (k + 1) / 2 > t) --k;
        const int l = (int)(t - (long long)k * (k + 1) / 2);
        xt[t] = (k == l) ? dm[(long long)k * n + k] : dm[(long long)k * n + l] + dm[(long long)l * n + k];
    }
}

// v[i][j] = y[pair(max, min)]
__global__ void jk_unpack_kernel(int n, const double *__restrict__ y, double *__restrict__ v) {
    const long long total = (long long)n * n;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(t / n), j = (int)(t % n);
        const int a = i > j ? i : j, b = i > j ? j : i;
        v[t] = y[(long long)a * (a + 1) / 2 + b];
    }
}

// yrow[r] = sum_c E[r][c] xrow[c]   (xrow != nullptr);   part[blk][c] = sum_{r in blk} E[r][c] xcol[r]   (xcol != nullptr)
__global__ __launch_bounds__(NT) void jk_j_kernel(long long npair, const double *__restrict__ E, long long ld,
                                                  const double *__restrict__ xrow, const double *__restrict__ xcol,
                                                  double *__restrict__ yrow, double *__restrict__ part, int blk0) {
    __shared__ double red[NW][JRB];
    __shared__ double xc[JRB];
    const long long rblk = (long long)blk0 + blockIdx.x;        // row block (blk0 > 0: a row range of a sharded ERI)
    const long long r0 = rblk * JRB;
    const int nr = (int)((npair - r0) < JRB ? (npair - r0) : JRB);
    if (threadIdx.x < JRB) xc[threadIdx.x] = (xcol && threadIdx.x < nr) ? xcol[r0 + threadIdx.x] : 0.0;
    __syncthreads();
    double racc[JRB];
#pragma unroll
    for (int rr = 0; rr < JRB; ++rr) racc[rr] = 0.0;
    const double *Eb = E + r0 * ld;
    for (long long c = threadIdx.x; c < npair; c += NT) {
        const double x1 = xrow ? xrow[c] : 0.0;
        double v[JRB];
#pragma unroll
        for (int rr = 0; rr < JRB; ++rr) v[rr] = (rr < nr) ? Eb[rr * ld + c] : 0.0;
        double cacc = 0.0;
#pragma unroll
        for (int rr = 0; rr < JRB; ++rr) {
            racc[rr] = fma(v[rr], x1, racc[rr]);
            cacc = fma(v[rr], xc[rr], cacc);
        }
        if (part) part[rblk * npair + c] = cacc;
    }
    if (yrow) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int rr = 0; rr < JRB; ++rr) {
            const double s = wave_sum(racc[rr]);
            if (lane == 0) red[wave][rr] = s;
        }
        __syncthreads();
        if (threadIdx.x < nr) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[w][threadIdx.x];
            yrow[r0 + threadIdx.x] = s;
        }
    }
}

// ycol[c] = sum_b part[b][c]: 64 columns per workgroup, wave w adds the blocks b = w, w+4, ... (coalesced 512 B
// rows, four independent chains), combined in a fixed order
__global__ __launch_bounds__(NT) void jk_colsum_kernel(long long npair, int nblk, const double *__restrict__ part,
                                                       double *__restrict__ ycol) {
    __shared__ double red[NW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c = (long long)blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < npair) {
        int b = wave;
        for (; b + 3 * NW < nblk; b += 4 * NW) {
            s0 += part[(long long)b * npair + c];
            s1 += part[(long long)(b + NW) * npair + c];
            s2 += part[(long long)(b + 2 * NW) * npair + c];
            s3 += part[(long long)(b + 3 * NW) * npair + c];
        }
        for (; b < nblk; b += NW) s0 += part[(long long)b * npair + c];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < npair) ycol[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// One workgroup per packed row r = pair(i, j).  Yi[r][:] = M dm[i,:],  Yj[r][:] = M dm[j,:],  M[k][l] = E[r][pair(k,l)].
template <int NM>
__global__ __launch_bounds__(NT) void jk_k_kernel(int n, const double *__restrict__ E, long long ld,
                                                  const double *__restrict__ dm, double *__restrict__ Yi,
                                                  double *__restrict__ Yj, long long row0) {
    extern __shared__ double sh[];
    double *xi = sh, *xj = sh + n;                 // dm[i,:], dm[j,:]
    double *ai = sh + 2 * n, *aj = sh + 3 * n;     // row parts y[k] (one writer per k)
    double *red = sh + 4 * n;                      // [NW][2][n] mirrored parts per wave
    const long long r = row0 + blockIdx.x;
    int i = (int)((sqrt(8.0 * (double)r + 1.0) - 1.0) * 0.5);
    while ((long long)(i + 1) * (i + 2) / 2 <= r) ++i;
    while ((long long)i * (i + 1) / 2 > r) --i;
    const int j = (int)(r - (long long)i * (i + 1) / 2);
    for (int t = threadIdx.x; t < n; t += NT) {
        xi[t] = dm[(long long)i * n + t];
        xj[t] = dm[(long long)j * n + t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double xli[NM], xlj[NM], yi[NM], yj[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int l = lane + 64 * m;
        xli[m] = l < n ? xi[l] : 0.0;
        xlj[m] = l < n ? xj[l] : 0.0;
        yi[m] = 0.0;
        yj[m] = 0.0;
    }
    const double *Er = E + r * ld;
    // rows k and n-1-k of the triangle hold n+1 elements together: every wave iteration moves the same number of
    // bytes and has both rows' loads in flight before the first use
    const int nhalf = (n + 1) / 2;
    double v0[NM], v1[NM], w0n[NM], w1n[NM];
    auto load_pair = [&](int p, double (&a0)[NM], double (&a1)[NM]) {
        const int k0 = p, k1 = n - 1 - p;
        const double *row0 = Er + (long long)k0 * (k0 + 1) / 2;
        const double *row1 = Er + (long long)k1 * (k1 + 1) / 2;
        const bool two = k1 != k0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int l = lane + 64 * m;
            a0[m] = (p < nhalf && l <= k0) ? row0[l] : 0.0;
            a1[m] = (p < nhalf && two && l <= k1) ? row1[l] : 0.0;
        }
    };
    load_pair(wave, v0, v1);
    for (int p = wave; p < nhalf; p += NW) {
        const int k0 = p, k1 = n - 1 - p;
        const bool two = k1 != k0;
        load_pair(p + NW, w0n, w1n);                       // next pair's rows in flight while this one is reduced
        const double xi0 = xi[k0], xj0 = xj[k0], xi1 = xi[k1], xj1 = xj[k1];
        double pa0 = 0.0, pb0 = 0.0, pa1 = 0.0, pb1 = 0.0;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int l = lane + 64 * m;
            yi[m] = fma(v0[m], xi0, yi[m]);                // y[l] += M[k][l] x[k]   (includes the diagonal once)
            yj[m] = fma(v0[m], xj0, yj[m]);
            yi[m] = fma(v1[m], xi1, yi[m]);
            yj[m] = fma(v1[m], xj1, yj[m]);
            const double w0 = (l < k0) ? v0[m] : 0.0, w1 = (l < k1) ? v1[m] : 0.0;
            pa0 = fma(w0, xli[m], pa0);                    // y[k] += M[k][l] x[l],  l < k
            pb0 = fma(w0, xlj[m], pb0);
            pa1 = fma(w1, xli[m], pa1);
            pb1 = fma(w1, xlj[m], pb1);
        }
        pa0 = wave_sum(pa0);
        pb0 = wave_sum(pb0);
        pa1 = wave_sum(pa1);
        pb1 = wave_sum(pb1);
        if (lane == 0) {
            ai[k0] = pa0;
            aj[k0] = pb0;
            if (two) {
                ai[k1] = pa1;
                aj[k1] = pb1;
            }
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            v0[m] = w0n[m];
            v1[m] = w1n[m];
        }
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int l = lane + 64 * m;
        if (l < n) {
            red[(wave * 2 + 0) * n + l] = yi[m];
            red[(wave * 2 + 1) * n + l] = yj[m];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += NT) {
        double si = ai[t], sj = aj[t];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            si += red[(w * 2 + 0) * n + t];
            sj += red[(w * 2 + 1) * n + t];
        }
        Yi[r * n + t] = si;
        Yj[r * n + t] = sj;
    }
}

// K[a][k] = sum_{i >= a} Yi[pair(i,a)][k] + sum_{j < a} Yj[pair(a,j)][k]
__global__ __launch_bounds__(NT) void jk_k_reduce_kernel(int n, const double *__restrict__ Yi, const double *__restrict__ Yj,
                                                         double *__restrict__ K) {
    const int a = blockIdx.x;
    for (int k = threadIdx.x; k < n; k += NT) {
        double s = 0.0;
        for (int i = a; i < n; ++i) s += Yi[((long long)i * (i + 1) / 2 + a) * n + k];
        for (int j = 0; j < a; ++j) s += Yj[((long long)a * (a + 1) / 2 + j) * n + k];
        K[(long long)a * n + k] = s;
    }
}

// s1 (n^4) or s8 (tril of npair x npair) -> s4 (npair x npair): pure gather (pyscf ao2mo.restore(4, .))
__global__ void eri_to_s4_kernel(int n, int from_sym, const double *__restrict__ in, double *__restrict__ out) {
    const long long npair = (long long)n * (n + 1) / 2;
    const long long total = npair * npair;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long p = t / npair, q = t % npair;
        if (from_sym == 8) {
            const long long a = p > q ? p : q, b = p > q ? q : p;
            out[t] = in[a * (a + 1) / 2 + b];
        } else {
            int i = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
            while ((long long)(i + 1) * (i + 2) / 2 <= p) ++i;
            while ((long long)i * (i + 1) / 2 > p) --i;
            const int j = (int)(p - (long long)i * (i + 1) / 2);
            int k = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
            while ((long long)(k + 1) * (k + 2) / 2 <= q) ++k;
            while ((long long)k * (k + 1) / 2 > q) --k;
            const int l = (int)(q - (long long)k * (k + 1) / 2);
            out[t] = in[(((long long)i * n + j) * n + k) * n + l];
        }
    }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

// Rows of the packed ERI that take part: nranges pairs [lo, hi) (lo a multiple of 32), or nranges = 0 for all of them.
// With ranges the outputs are the PARTIAL J / K of those rows (a row-sharded ERI: every rank runs its own rows and the
// small n x n results are summed over ranks).
static int jk_s4_impl(dmk_ctx *ctx, int n, const double *eri, int64_t ld, int nranges, const int64_t *ranges, const double *dm_row,
                      const double *dm_col, const double *dm_k, double *vj_row, double *vj_col, double *vk) {
    if (!ctx) return DMK_ERR_INVALID;
    const long long npair = (long long)n * (n + 1) / 2;
    if (n <= 0 || !eri || ld < npair) return dmk_fail(ctx, DMK_ERR_INVALID, "jk_s4: bad arguments");
    if ((dm_row && !vj_row) || (dm_col && !vj_col) || (dm_k && !vk))
        return dmk_fail(ctx, DMK_ERR_INVALID, "jk_s4: a density was given without its output matrix");
    if (dm_k && n > 512) return dmk_fail(ctx, DMK_ERR_INVALID, "jk_s4: exchange supports n <= 512 (got %d)", n);
    const int64_t whole[2] = {0, npair};
    const bool sharded = nranges > 0;
    if (!sharded) { nranges = 1; ranges = whole; }
    for (int q = 0; q < nranges; ++q)
        // the J kernel works on whole JRB-row blocks: a range must start on one and end on one (or at npair), else the rows
        // between `hi` and the end of its last block would be counted here AND by the owner of the next range
        if (ranges[2 * q] < 0 || ranges[2 * q + 1] > npair || ranges[2 * q] > ranges[2 * q + 1] || (ranges[2 * q] % JRB) != 0 ||
            ((ranges[2 * q + 1] % JRB) != 0 && ranges[2 * q + 1] != npair))
            return dmk_fail(ctx, DMK_ERR_INVALID, "jk_s4: row range %d = [%lld, %lld) must lie in [0, %lld), start on a multiple of %d and "
                            "end on one (or at the last row)", q, (long long)ranges[2 * q], (long long)ranges[2 * q + 1], npair, JRB);
    FamScope fs(ctx, DMK_FAM_JK);
    const int nblk = (int)((npair + JRB - 1) / JRB);
    // workspace carve
    const size_t b_vec = align256((size_t)npair * 8);
    const size_t b_part = dm_col ? align256((size_t)nblk * npair * 8) : 0;
    const size_t b_Y = dm_k ? align256((size_t)npair * n * 8) : 0;
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, 4 * b_vec + b_part + 2 * b_Y, &ws);
    if (rc) return rc;
    char *p = static_cast<char *>(ws);
    double *xrow = reinterpret_cast<double *>(p); p += b_vec;
    double *xcol = reinterpret_cast<double *>(p); p += b_vec;
    double *yrow = reinterpret_cast<double *>(p); p += b_vec;
    double *ycol = reinterpret_cast<double *>(p); p += b_vec;
    double *part = reinterpret_cast<double *>(p); p += b_part;
    double *Yi = reinterpret_cast<double *>(p); p += b_Y;
    double *Yj = reinterpret_cast<double *>(p);
    const int gsmall = (int)std::min<long long>((npair + 255) / 256, 4096);
    const int gsq = (int)std::min<long long>(((long long)n * n + 255) / 256, 4096);
    if (sharded) {       // rows outside the ranges contribute zeros to the fixed-order reductions below
        if (dm_row) DMK_HIP(ctx, hipMemsetAsync(yrow, 0, (size_t)npair * 8, ctx->stream));
        if (dm_col) DMK_HIP(ctx, hipMemsetAsync(part, 0, (size_t)nblk * npair * 8, ctx->stream));
        if (dm_k) DMK_HIP(ctx, hipMemsetAsync(Yi, 0, 2 * b_Y, ctx->stream));
    }
    if (dm_row || dm_col) {
        if (dm_row) hipLaunchKernelGGL(jk_fold_dm_kernel, dim3(gsmall), dim3(256), 0, ctx->stream, n, dm_row, xrow);
        if (dm_col) hipLaunchKernelGGL(jk_fold_dm_kernel, dim3(gsmall), dim3(256), 0, ctx->stream, n, dm_col, xcol);
        for (int q = 0; q < nranges; ++q) {
            const int b0 = (int)(ranges[2 * q] / JRB), b1 = (int)((ranges[2 * q + 1] + JRB - 1) / JRB);
            if (b1 <= b0) continue;
            hipLaunchKernelGGL(jk_j_kernel, dim3(b1 - b0), dim3(NT), 0, ctx->stream, npair, eri, (long long)ld,
                               dm_row ? xrow : (const double *)nullptr, dm_col ? xcol : (const double *)nullptr,
                               dm_row ? yrow : (double *)nullptr, dm_col ? part : (double *)nullptr, b0);
        }
        DMK_CHECK_LAUNCH(ctx);
        if (dm_row) hipLaunchKernelGGL(jk_unpack_kernel, dim3(gsq), dim3(256), 0, ctx->stream, n, yrow, vj_row);
        if (dm_col) {
            hipLaunchKernelGGL(jk_colsum_kernel, dim3((unsigned)((npair + 63) / 64)), dim3(NT), 0, ctx->stream, npair,
                               nblk, part, ycol);
            hipLaunchKernelGGL(jk_unpack_kernel, dim3(gsq), dim3(256), 0, ctx->stream, n, ycol, vj_col);
        }
        DMK_CHECK_LAUNCH(ctx);
    }
    if (dm_k) {
        const size_t lds = (size_t)(4 + 2 * NW) * n * sizeof(double);
        const int nm = (n + 63) / 64;
#define JK_K_LAUNCH(NM)                                                                                          \
    do {                                                                                                         \
        if (lds > 48 * 1024)                                                                                     \
            DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(jk_k_kernel<NM>),                    \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        for (int q = 0; q < nranges; ++q)                                                                        \
            if (ranges[2 * q + 1] > ranges[2 * q])                                                               \
                hipLaunchKernelGGL(jk_k_kernel<NM>, dim3((unsigned)(ranges[2 * q + 1] - ranges[2 * q])), dim3(NT), lds, \
                                   ctx->stream, n, eri, (long long)ld, dm_k, Yi, Yj, (long long)ranges[2 * q]);  \
    } while (0)
        if (nm <= 1) JK_K_LAUNCH(1);
        else if (nm <= 2) JK_K_LAUNCH(2);
        else if (nm <= 4) JK_K_LAUNCH(4);
        else JK_K_LAUNCH(8);
#undef JK_K_LAUNCH
        DMK_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(jk_k_reduce_kernel, dim3(n), dim3(NT), 0, ctx->stream, n, Yi, Yj, vk);
        DMK_CHECK_LAUNCH(ctx);
    }
    return DMK_OK;
}

int dmk_jk_s4(dmk_ctx *ctx, int n, const double *eri, int64_t ld, const double *dm_row, const double *dm_col,
              const double *dm_k, double *vj_row, double *vj_col, double *vk) {
    return jk_s4_impl(ctx, n, eri, ld, 0, nullptr, dm_row, dm_col, dm_k, vj_row, vj_col, vk);
}

int dmk_jk_s4_rows(dmk_ctx *ctx, int n, const double *eri, int64_t ld, int nranges, const int64_t *ranges_host, const double *dm_row,
                   const double *dm_col, const double *dm_k, double *vj_row, double *vj_col, double *vk) {
    if (ctx && (nranges <= 0 || !ranges_host)) return dmk_fail(ctx, DMK_ERR_INVALID, "jk_s4_rows: needs at least one row range");
    return jk_s4_impl(ctx, n, eri, ld, nranges, ranges_host, dm_row, dm_col, dm_k, vj_row, vj_col, vk);
}

int dmk_eri_to_s4(dmk_ctx *ctx, int n, int from_symmetry, const double *in, double *out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n <= 0 || !in || !out || (from_symmetry != 1 && from_symmetry != 8))
        return dmk_fail(ctx, DMK_ERR_INVALID, "eri_to_s4: bad arguments (from_symmetry must be 1 or 8)");
    FamScope fs(ctx, DMK_FAM_MISC);
    const long long npair = (long long)n * (n + 1) / 2;
    const int g = (int)std::min<long long>((npair * npair + 255) / 256, 65536);
    hipLaunchKernelGGL(eri_to_s4_kernel, dim3(g), dim3(256), 0, ctx->stream, n, from_symmetry, in, out);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

}  // extern "C"
