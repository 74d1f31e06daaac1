// K12 -- device pieces of the correlation-potential least-squares fit.
//
// routine/slater.py:909-1329 (FitVcorEmb): every objective / gradient evaluation is
//     V_emb = sum_p param[p] dV_dparam[p]          (slater.py:1059-1071, np.tensordot over nparam)
//     ew, ev = eigh(embH1 + V_emb, ovlp_emb)       (K1, eigh.hip)
//     rho = (ev occ) ev^T, drho, |drho|            (small dense algebra, nemb x nemb)
//     dw_dV = ev [(C^T drho C) o K] ev^T           (slater.py:1131-1141 / ftsystem.py:151-213)
//     grad[p] = <dV_dparam[p], dw_dV>              (slater.py:1141, np.tensordot over spin * npair)
// dV_dparam is (nparam, spin, npair) f64 = 1.7 GB at C5 (3192 x 2 x 32896): the two contractions with it are
// HBM-bound streaming passes and share ONE kernel (gemv2: row dots and column sums of a row-major matrix in the
// same pass, 2-D grid, per-block partials reduced in a fixed order -- bit-reproducible, no atomics).
// The nemb x nemb algebra runs through a plain LDS-tiled real GEMM (64 x 64 tile, f64 FMA): at <= 256^3 it is
// launch-latency bound, not worth the matrix pipe.
// dV_dparam itself (slater.py:851-907 with transform_local_sparseH, slater_helper.py:91-100) is a gather from the
// cell Gram matrix G[(i,p),(j,q)] = sum_c B[c,i,p] B[c,j,q] (one symmetric MFMA GEMM, dgemm_tn.hip).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int NW = NT / 64;
constexpr int RB = 32;            // rows per workgroup (gemv2)
constexpr int CI = 8;             // column iterations per workgroup (gemv2): 2048 columns

__device__ __forceinline__ double wave_sum(double v) { return dmk_wave_sum(v); }

__device__ __forceinline__ void tril_rc(long long t, int &k, int &l) {
    k = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((long long)(k + 1) * (k + 2) / 2 <= t) ++k;
    while ((long long)k * (k + 1) / 2 > t) --k;
    l = (int)(t - (long long)k * (k + 1) / 2);
}

// rowpart[cc][r] = sum_{c in chunk cc} A[r][c] xrow[c];   colpart[rb][c] = sum_{r in block rb} A[r][c] xcol[r]
__global__ __launch_bounds__(NT) void gemv2_kernel(long long M, long long N, const double *__restrict__ A, long long lda,
                                                   const double *__restrict__ xrow, const double *__restrict__ xcol,
                                                   double *__restrict__ rowpart, double *__restrict__ colpart) {
    __shared__ double red[NW][RB];
    __shared__ double xc[RB];
    const long long r0 = (long long)blockIdx.x * RB;
    const long long c0 = (long long)blockIdx.y * NT * CI;
    const int nr = (int)((M - r0) < RB ? (M - r0) : RB);
    if (threadIdx.x < RB) xc[threadIdx.x] = (xcol && threadIdx.x < nr) ? xcol[r0 + threadIdx.x] : 0.0;
    __syncthreads();
    double racc[RB];
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) racc[rr] = 0.0;
    const double *Ab = A + r0 * lda;
#pragma unroll 1
    for (int it = 0; it < CI; ++it) {
        const long long c = c0 + (long long)it * NT + threadIdx.x;
        if (c < N) {
            const double x1 = xrow ? xrow[c] : 0.0;
            double v[RB];
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) v[rr] = (rr < nr) ? Ab[rr * lda + c] : 0.0;
            double cacc = 0.0;
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                racc[rr] = fma(v[rr], x1, racc[rr]);
                cacc = fma(v[rr], xc[rr], cacc);
            }
            if (colpart) colpart[(long long)blockIdx.x * N + c] = cacc;
        }
    }
    if (rowpart) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const double s = wave_sum(racc[rr]);
            if (lane == 0) red[wave][rr] = s;
        }
        __syncthreads();
        if (threadIdx.x < nr) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red[w][threadIdx.x];
            rowpart[(long long)blockIdx.y * M + r0 + threadIdx.x] = s;
        }
    }
}

// y[c] = sum_b part[b][c]  (64 columns per workgroup, wave w takes b = w, w+4, ...; fixed combination order)
__global__ __launch_bounds__(NT) void partsum_kernel(long long n, int nblk, const double *__restrict__ part,
                                                     double *__restrict__ y) {
    __shared__ double red[NW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long c = (long long)blockIdx.x * 64 + lane;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < n) {
        int b = wave;
        for (; b + 3 * NW < nblk; b += 4 * NW) {
            s0 += part[(long long)b * n + c];
            s1 += part[(long long)(b + NW) * n + c];
            s2 += part[(long long)(b + 2 * NW) * n + c];
            s3 += part[(long long)(b + 3 * NW) * n + c];
        }
        for (; b < nblk; b += NW) s0 += part[(long long)b * n + c];
    }
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && c < n) y[c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// C[b] = alpha op(A[b]) op(B[b]) + beta C[b], row-major with leading dimensions, on v_mfma_f64_16x16x4_f64.
// These are the small dense products of the fit (n = nemb, a few hundred) and of the basis rotations: a few MFLOP each, so
// what counts is latency and the number of CUs reached, not operand reuse -- one 16 x 16 tile of C per wave (2 x 2 waves per
// workgroup), operands straight from global memory (they are L2 resident), no LDS, no barrier.  The k index is permuted
// identically in both operands: lane (x, q) owns k0 + 16 q .. + 15 of a 64-wide chunk and the t-th MFMA of the chunk contracts
// element t of the four q groups, so a K-contiguous operand is read as 128 contiguous bytes per lane (8 x b128 when the rows
// are 16-byte aligned) and the other kind as 16 row segments shared by 16 lanes.  The next chunk is in flight during the MFMAs.
template <int TA, int TB>
__global__ __launch_bounds__(NT) void dgemm_small_kernel(int M, int N, int K, double alpha, const double *__restrict__ A,
                                                         long long lda, long long sA, const double *__restrict__ B,
                                                         long long ldb, long long sB, double beta, double *__restrict__ C,
                                                         long long ldc, long long sC) {
    const int b = blockIdx.z;
    A += (long long)b * sA;
    B += (long long)b * sB;
    C += (long long)b * sC;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 32 + (wave >> 1) * 16, n0 = blockIdx.x * 32 + (wave & 1) * 16;
    if (m0 >= M || n0 >= N) return;
    const int am = m0 + x, bn = n0 + x;
    const bool a_ok = am < M, b_ok = bn < N;
    // K-contiguous operands: A when TA == 0 (row am), B when TB == 1 (row bn)
    const bool a_vec = TA == 0 && ((lda & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_vec = TB == 1 && ((ldb & 1) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    const double *ap = TA ? A + (a_ok ? am : 0) : A + (long long)(a_ok ? am : 0) * lda;
    const double *bp = TB ? B + (long long)(b_ok ? bn : 0) * ldb : B + (b_ok ? bn : 0);

    auto load_contig = [&](const double *row, bool ok, bool vec, int k0, double (&r)[16]) {
        if (ok && vec && k0 + 16 <= K) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double2 v = *reinterpret_cast<const double2 *>(row + k0 + 2 * t);
                r[2 * t] = v.x;
                r[2 * t + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) r[t] = (ok && k0 + t < K) ? row[k0 + t] : 0.0;
        }
    };
    auto load_strided = [&](const double *col, long long ld, bool ok, int k0, double (&r)[16]) {
#pragma unroll
        for (int t = 0; t < 16; ++t) r[t] = (ok && k0 + t < K) ? col[(long long)(k0 + t) * ld] : 0.0;
    };
    auto load_a = [&](int k0, double (&r)[16]) {
        if (TA) load_strided(ap, lda, a_ok, k0, r); else load_contig(ap, a_ok, a_vec, k0, r);
    };
    auto load_b = [&](int k0, double (&r)[16]) {
        if (TB) load_contig(bp, b_ok, b_vec, k0, r); else load_strided(bp, ldb, b_ok, k0, r);
    };

    d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
    double a[2][16], bb[2][16];
    const int nchunk = (K + 63) / 64;
    if (nchunk > 0) {
        load_a(16 * q, a[0]);
        load_b(16 * q, bb[0]);
    }
#pragma unroll 1
    for (int c = 0; c < nchunk; c += 2) {
        if (c + 1 < nchunk) {
            load_a((c + 1) * 64 + 16 * q, a[1]);
            load_b((c + 1) * 64 + 16 * q, bb[1]);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][t], bb[0][t], acc, 0, 0, 0);
        if (c + 1 < nchunk) {
            if (c + 2 < nchunk) {
                load_a((c + 2) * 64 + 16 * q, a[0]);
                load_b((c + 2) * 64 + 16 * q, bb[0]);
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][t], bb[1][t], acc, 0, 0, 0);
        }
    }
    // D layout: row = (lane >> 4) + 4 r, col = lane & 15
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + q + 4 * r;
        if (row < M && bn < N) {
            double *c = C + (long long)row * ldc + bn;
            *c = (beta == 0.0) ? alpha * acc[r] : alpha * acc[r] + beta * (*c);
        }
    }
}

// packed lower triangle <-> full symmetric, batched
__global__ void sym_fold_kernel(int n, int batch, const double *__restrict__ full, double *__restrict__ tril) {
    const long long npair = (long long)n * (n + 1) / 2, total = npair * batch;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / npair;
        int k, l;
        tril_rc(t % npair, k, l);
        const double *f = full + b * n * n;
        tril[t] = (k == l) ? f[(long long)k * n + k] : f[(long long)k * n + l] + f[(long long)l * n + k];
    }
}
__global__ void sym_unpack_kernel(int n, int batch, const double *__restrict__ tril, const double *__restrict__ add_tril,
                                  double *__restrict__ full) {
    const long long npair = (long long)n * (n + 1) / 2, total = (long long)n * n * batch;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / ((long long)n * n), e = t % ((long long)n * n);
        const int i = (int)(e / n), j = (int)(e % n);
        const int a = i > j ? i : j, c = i > j ? j : i;
        const long long p = b * npair + (long long)a * (a + 1) / 2 + c;
        full[t] = tril[p] + (add_tril ? add_tril[p] : 0.0);
    }
}

// out[r][c] = in[ridx ? ridx[r] : r][cidx ? cidx[c] : c]
__global__ void gather2d_kernel(int nr, int nc, const int *__restrict__ ridx, const int *__restrict__ cidx,
                                const double *__restrict__ in, long long ld_in, double *__restrict__ out) {
    const long long total = (long long)nr * nc;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(t / nc), c = (int)(t % nc);
        out[t] = in[(long long)(ridx ? ridx[r] : r) * ld_in + (cidx ? cidx[c] : c)];
    }
}

// out = A o B   (mode 0);  out[r][c] = A[r][c] * s[r]   (mode 1, B = s);  mode 2: A, out complex (nc counts doubles), B real
__global__ void ewise_kernel(int mode, long long nr, long long nc, const double *__restrict__ A, const double *__restrict__ B,
                             double *__restrict__ out) {
    const long long total = nr * nc;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x)
        out[t] = A[t] * (mode == 0 ? B[t] : (mode == 1 ? B[t / nc] : B[t >> 1]));
}

// diff = a - b, sumsq = sum diff^2 ; one workgroup, fixed order
// diff = a - b and its sum of squares in two launches: up to SS_BLOCKS workgroups leave one partial sum each (contiguous
// slices, fixed partition), a single wave adds them in index order -- reproducible, and the 2 * nidx^2 residual entries of the
// fit (131 072 at C5) no longer trickle through one workgroup (0.22 ms -> a few microseconds)
constexpr int SS_BLOCKS = 64;
__global__ __launch_bounds__(NT) void sub_sumsq_kernel(long long n, long long per_block, const double *__restrict__ a,
                                                       const double *__restrict__ b, double *__restrict__ diff,
                                                       double *__restrict__ part) {
    __shared__ double red[NW];
    const long long t0 = (long long)blockIdx.x * per_block, t1 = (t0 + per_block < n) ? t0 + per_block : n;
    double s = 0.0;
    for (long long t = t0 + threadIdx.x; t < t1; t += NT) {
        const double d = a[t] - b[t];
        if (diff) diff[t] = d;
        s = fma(d, d, s);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void sumsq_final_kernel(int nblk, const double *__restrict__ part, double *__restrict__ sumsq) {
    const double v = (int)threadIdx.x < nblk ? part[threadIdx.x] : 0.0;
    const double s = wave_sum(v);
    if (threadIdx.x == 0) sumsq[0] = s;
}

// dV[ip][pair(p,q)] = sum over the nonzeros (i, j, val) of parameter-spin entry ip of  val * G[(i,p),(j,q)]
__global__ void dv_dparam_kernel(int nent, int nb, long long ldg, const double *__restrict__ G, const int *__restrict__ ptr,
                                 const int *__restrict__ zi, const int *__restrict__ zj, const double *__restrict__ zv,
                                 double *__restrict__ dV, const long long *__restrict__ out_off) {
    const long long npair = (long long)nb * (nb + 1) / 2;
    const int ent = blockIdx.y;
    if (ent >= nent) return;
    const int z0 = ptr[ent], z1 = ptr[ent + 1];
    double *o = dV + out_off[ent];
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < npair;
         t += (long long)gridDim.x * blockDim.x) {
        int p, q;
        tril_rc(t, p, q);
        double s = 0.0;
        for (int z = z0; z < z1; ++z)
            s = fma(zv[z], G[((long long)zi[z] * nb + p) * ldg + (long long)zj[z] * nb + q], s);
        o[t] = s;
    }
}

// K[b][p][q] = (f_p - f_q) / (e_p - e_q), and -beta f_p (1 - f_q) where |e_p - e_q| < 1e-10 (routine/ftsystem.py:170-181);
// beta <= 0 selects the T = 0 form of routine/slater.py:1126-1134: K = 1 / (e_occ - e_virt) on the (virt, occ) and
// (occ, virt) blocks split at index nocc, zero elsewhere
__global__ void fit_kmat_kernel(int n, int batch, const double *__restrict__ ew, const double *__restrict__ f, double beta, int nocc,
                                double *__restrict__ K) {
    const long long total = (long long)batch * n * n;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(t / ((long long)n * n)), p = (int)((t / n) % n), q = (int)(t % n);
        const double ep = ew[(long long)b * n + p], eq = ew[(long long)b * n + q];
        double v;
        if (beta <= 0.0) {
            if (p >= nocc && q < nocc) v = 1.0 / (eq - ep);
            else if (p < nocc && q >= nocc) v = 1.0 / (ep - eq);
            else v = 0.0;
        } else {
            const double fp = f[(long long)b * n + p], fq = f[(long long)b * n + q];
            const double de = ep - eq;
            v = (fabs(de) < 1e-10) ? -beta * fp * (1.0 - fq) : (fp - fq) / de;
        }
        K[t] = v;
    }
}

// dst[(idx[a] * ld + idx[b]) * estride] += alpha src[a][b]; repeated indices accumulate (f64 atomics)
__global__ void scatter2d_kernel(int m, const int *__restrict__ idx, const double *__restrict__ src, double alpha,
                                 double *__restrict__ dst, long long ld, int estride) {
    const long long total = (long long)m * m;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int a = (int)(t / m), b = (int)(t % m);
        atomicAdd(dst + ((long long)idx[a] * ld + idx[b]) * estride, alpha * src[t]);
    }
}

__global__ void axpy_kernel(long long n, double alpha, const double *__restrict__ x, double *__restrict__ y) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
        y[t] = fma(alpha, x[t], y[t]);
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
int grid_for(long long total, int cap = 8192) { return (int)std::min<long long>((total + 255) / 256, cap); }

// ---- fused T = 0 objective of the embedding fit (dmk_fit_objective) --------------------------------------------------------
// full[b] = unpack(v0[b] + t v1[b] + H1[b]): the ray form of the embedding potential (V_emb is linear in the parameters) and the
// fixed one-body part in one pass -- replaces a device copy, an axpy and sym_unpack
// ... and, on a line-search ray whose two norm bounds the caller computed once (ray_norm), the refinement's bound on |H| together
// with its cleared state words and arrival counters: |H0 + t V1| <= |H0| + |t| |V1| -- what rf_norm_kernel measured from the
// unpacked matrix in a launch of its own (10.6 us) in front of every trial step.
__global__ void fit_ray_unpack_kernel(int n, int batch, const double *__restrict__ v0, const double *__restrict__ v1, double t,
                                      const double *__restrict__ h1, double *__restrict__ full, const double *__restrict__ ray_norm,
                                      double *__restrict__ anorm, int *__restrict__ state, unsigned *__restrict__ arrive) {
    if (ray_norm && blockIdx.x == 0 && threadIdx.x < batch) {
        const int m = threadIdx.x;
        const double bound = ray_norm[m] + fabs(t) * ray_norm[batch + m];
        anorm[m] = bound > 0.0 ? 1.015625 * bound : 1.0;          // (the safety factor of rf_norm_kernel)
        state[m] = 0;
        state[batch + m] = -1;
        arrive[m] = 0;
    }
    const long long npair = (long long)n * (n + 1) / 2, total = (long long)n * n * batch;
    for (long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; e0 < total; e0 += (long long)gridDim.x * blockDim.x) {
        const long long b = e0 / ((long long)n * n), e = e0 % ((long long)n * n);
        const int i = (int)(e / n), j = (int)(e % n);
        const int a = i > j ? i : j, c = i > j ? j : i;
        const long long p = b * npair + (long long)a * (a + 1) / 2 + c;
        const double v = v1 ? fma(t, v1[p], v0[p]) : v0[p];
        full[e0] = v + h1[p];
    }
}

// drho[s][a][b] = W[s][a][b] * sum_m occ[s][m] Vt[s][m][fit[a]] Vt[s][m][fit[b]] - target[s][a][b]  and per-workgroup partial
// sums of drho^2: the density of the fitted block straight from the eigenvector rows (rho = ev occ ev^T restricted to the fitted
// indices), the mask, the residual and its norm in ONE launch -- replaces ewise (scale), dgemm, 2 x gather, ewise (mask),
// sub_sumsq.  One 16 x 16 tile per WORKGROUP on v_mfma_f64_16x16x4_f64; the four waves split the sum over the orbitals m
// (wave w takes the 64-wide chunks w, w + 4, ...: one L2 round trip each at nb = 256) and their partial tiles are added through
// LDS in a fixed order.  part[(s * gy + by) * gx + bx] = sum of squares over the workgroup's tile.
__global__ __launch_bounds__(NT) void fit_rho_diff_kernel(int nb, int nidx, const double *__restrict__ Vt, const double *__restrict__ occ,
                                                          const int *__restrict__ fit, const double *__restrict__ W,
                                                          const double *__restrict__ target, double *__restrict__ drho,
                                                          double *__restrict__ part) {
    __shared__ double red[3][4][64];
    const int s = blockIdx.z;
    const double *V = Vt + (long long)s * nb * nb;
    const double *oc = occ + (long long)s * nb;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
    const int ia = m0 + x, ib = n0 + x;
    const bool a_ok = ia < nidx, b_ok = ib < nidx;
    const int ca = fit[a_ok ? ia : 0], cb = fit[b_ok ? ib : 0];
    d4_t acc = d4_t{0.0, 0.0, 0.0, 0.0};
    const int nchunk = (nb + 63) / 64;
#pragma unroll 1
    for (int c = wave; c < nchunk; c += 4) {
        double a[16], bb[16];
        const int k0 = c * 64 + 16 * q;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int m = k0 + t;
            const bool ok = m < nb;
            const double *row = V + (long long)(ok ? m : 0) * nb;
            a[t] = (ok && a_ok) ? row[ca] * oc[m] : 0.0;
            bb[t] = (ok && b_ok) ? row[cb] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bb[t], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave != 0) return;
    double ss = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = (acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane]);
        const int row = m0 + q + 4 * r;
        if (row < nidx && ib < nidx) {
            const long long o = ((long long)s * nidx + row) * nidx + ib;
            const double d = v * W[o] - target[o];
            drho[o] = d;
            ss = fma(d, d, ss);
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) part[((long long)s * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = ss;
}

// The end of the chain: the partial sums in index order (fixed order: reproducible) -> sum of squares, on the device for the
// gradient and, with the eigensolver's verdicts and the occupation status words, in a PINNED host record the caller polls:
//   slot: [0] sum of squares, [1 .. 1 + 2 batch) verdicts (state, settling pass) as doubles, then batch occupation status words,
//   last the sequence number, written LAST with system-scope release -- the host sees a complete record or the old sequence number.
struct FitSlot { double f2; double verdict[8]; double occ_status[4]; double occ_spread[4]; unsigned long long seq; };
__global__ __launch_bounds__(256) void fit_final_kernel(int npart, const double *__restrict__ part, double *__restrict__ sumsq_dev,
                                                        int batch, const int *__restrict__ verdict, const double *__restrict__ occ_info,
                                                        FitSlot *__restrict__ slot, unsigned long long seq) {
    __shared__ double red[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < npart; i += 256) v += part[i];          // fixed assignment of terms to threads: reproducible
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double f2 = (red[0] + red[1]) + (red[2] + red[3]);
        sumsq_dev[0] = f2;
        slot->f2 = f2;
        for (int i = 0; i < 2 * batch && i < 8; ++i) slot->verdict[i] = (double)verdict[i];
        for (int i = 0; i < batch && i < 4; ++i) {
            slot->occ_status[i] = occ_info[8 * i + 4];
            slot->occ_spread[i] = occ_info[8 * i + 2];
        }
        __threadfence_system();
        __hip_atomic_store(&slot->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

// C[b] = A[b] B[b] (row-major, contiguous) for the other translation units
int launch_dgemm_small_nn(dmk_ctx *ctx, int M, int N, int K, int batch, const double *A, const double *B, double *C) {
    const dim3 grid((N + 31) / 32, (M + 31) / 32, batch);
    hipLaunchKernelGGL((dgemm_small_kernel<0, 0>), grid, dim3(NT), 0, ctx->stream, M, N, K, 1.0, A, (long long)K, (long long)M * K, B,
                       (long long)N, (long long)K * N, 0.0, C, (long long)N, (long long)M * N);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

extern "C" {

int dmk_dgemv2(dmk_ctx *ctx, int64_t M, int64_t N, const double *A, int64_t lda, const double *xrow, const double *xcol,
               double *yrow, double *ycol) {
    if (!ctx) return DMK_ERR_INVALID;
    if (M <= 0 || N <= 0 || !A || lda < N || (xrow && !yrow) || (xcol && !ycol) || (!xrow && !xcol))
        return dmk_fail(ctx, DMK_ERR_INVALID, "dgemv2: bad arguments");
    FamScope fs(ctx, DMK_FAM_FIT);
    const int nrb = (int)((M + RB - 1) / RB), ncc = (int)((N + NT * CI - 1) / (NT * CI));
    const size_t b_row = xrow ? align256((size_t)ncc * M * 8) : 0, b_col = xcol ? align256((size_t)nrb * N * 8) : 0;
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, b_row + b_col + 256, &ws);
    if (rc) return rc;
    double *rowpart = reinterpret_cast<double *>(ws);
    double *colpart = reinterpret_cast<double *>(static_cast<char *>(ws) + b_row);
    hipLaunchKernelGGL(gemv2_kernel, dim3(nrb, ncc), dim3(NT), 0, ctx->stream, (long long)M, (long long)N, A, (long long)lda,
                       xrow, xcol, xrow ? rowpart : (double *)nullptr, xcol ? colpart : (double *)nullptr);
    DMK_CHECK_LAUNCH(ctx);
    if (xrow) hipLaunchKernelGGL(partsum_kernel, dim3((unsigned)((M + 63) / 64)), dim3(NT), 0, ctx->stream, (long long)M, ncc,
                                 rowpart, yrow);
    if (xcol) hipLaunchKernelGGL(partsum_kernel, dim3((unsigned)((N + 63) / 64)), dim3(NT), 0, ctx->stream, (long long)N, nrb,
                                 colpart, ycol);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_dgemm_batched(dmk_ctx *ctx, int opA, int opB, int M, int N, int K, int batch, double alpha, const double *A,
                      int64_t lda, int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double beta, double *C,
                      int64_t ldc, int64_t strideC) {
    if (!ctx) return DMK_ERR_INVALID;
    if (opA < 0 || opA > 1 || opB < 0 || opB > 1 || M < 0 || N < 0 || K < 0 || batch < 0 || !A || !B || !C)
        return dmk_fail(ctx, DMK_ERR_INVALID, "dgemm_batched: bad arguments");
    if (M == 0 || N == 0 || batch == 0) return DMK_OK;
    FamScope fs(ctx, DMK_FAM_FIT);
    const dim3 grid((N + 31) / 32, (M + 31) / 32, batch);
#define DG_LAUNCH(TA, TB)                                                                                             \
    hipLaunchKernelGGL((dgemm_small_kernel<TA, TB>), grid, dim3(NT), 0, ctx->stream, M, N, K, alpha, A, (long long)lda,    \
                       (long long)strideA, B, (long long)ldb, (long long)strideB, beta, C, (long long)ldc, (long long)strideC)
    if (opA == 0 && opB == 0) DG_LAUNCH(0, 0);
    else if (opA == 0) DG_LAUNCH(0, 1);
    else if (opB == 0) DG_LAUNCH(1, 0);
    else DG_LAUNCH(1, 1);
#undef DG_LAUNCH
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_sym_fold(dmk_ctx *ctx, int n, int batch, const double *full, double *tril) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n <= 0 || batch <= 0 || !full || !tril) return dmk_fail(ctx, DMK_ERR_INVALID, "sym_fold: bad arguments");
    FamScope fs(ctx, DMK_FAM_FIT);
    hipLaunchKernelGGL(sym_fold_kernel, dim3(grid_for((long long)n * (n + 1) / 2 * batch)), dim3(256), 0, ctx->stream, n, batch,
                       full, tril);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_sym_unpack(dmk_ctx *ctx, int n, int batch, const double *tril, const double *add_tril, double *full) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n <= 0 || batch <= 0 || !full || !tril) return dmk_fail(ctx, DMK_ERR_INVALID, "sym_unpack: bad arguments");
    FamScope fs(ctx, DMK_FAM_FIT);
    hipLaunchKernelGGL(sym_unpack_kernel, dim3(grid_for((long long)n * n * batch)), dim3(256), 0, ctx->stream, n, batch, tril,
                       add_tril, full);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_gather2d_f64(dmk_ctx *ctx, int nrow, int ncol, const int32_t *row_idx, const int32_t *col_idx, const double *in,
                     int64_t ld_in, double *out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (nrow < 0 || ncol < 0 || !in || !out) return dmk_fail(ctx, DMK_ERR_INVALID, "gather2d: bad arguments");
    if (nrow == 0 || ncol == 0) return DMK_OK;
    FamScope fs(ctx, DMK_FAM_FIT);
    hipLaunchKernelGGL(gather2d_kernel, dim3(grid_for((long long)nrow * ncol)), dim3(256), 0, ctx->stream, nrow, ncol, row_idx,
                       col_idx, in, (long long)ld_in, out);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_ewise_mul(dmk_ctx *ctx, int mode, int64_t nrow, int64_t ncol, const double *A, const double *B, double *out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (mode < 0 || mode > 2 || (mode == 2 && (ncol & 1)) || nrow < 0 || ncol < 0 || !A || !B || !out)
        return dmk_fail(ctx, DMK_ERR_INVALID, "ewise_mul: bad arguments");
    if (nrow == 0 || ncol == 0) return DMK_OK;
    FamScope fs(ctx, DMK_FAM_FIT);
    hipLaunchKernelGGL(ewise_kernel, dim3(grid_for(nrow * ncol)), dim3(256), 0, ctx->stream, mode, (long long)nrow,
                       (long long)ncol, A, B, out);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_sub_sumsq(dmk_ctx *ctx, int64_t n, const double *a, const double *b, double *diff, double *sumsq_dev) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n < 0 || !a || !b || !sumsq_dev) return dmk_fail(ctx, DMK_ERR_INVALID, "sub_sumsq: bad arguments");
    FamScope fs(ctx, DMK_FAM_FIT);
    const long long per_block = std::max<long long>(((n + SS_BLOCKS - 1) / SS_BLOCKS + NT - 1) / NT * NT, NT);
    const int nblk = (int)std::max<long long>((n + per_block - 1) / per_block, 1);
    void *ws = nullptr;
    const int rc = dmk_scratch(ctx, SS_BLOCKS * sizeof(double), &ws);
    if (rc) return rc;
    double *part = static_cast<double *>(ws);
    hipLaunchKernelGGL(sub_sumsq_kernel, dim3(nblk), dim3(NT), 0, ctx->stream, (long long)n, per_block, a, b, diff, part);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, ctx->stream, nblk, part, sumsq_dev);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_fit_kmat(dmk_ctx *ctx, int n, int batch, const double *ew, const double *f, double beta, int nocc, double *K) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n <= 0 || batch <= 0 || !ew || !K || (beta > 0.0 && !f) || (beta <= 0.0 && (nocc < 0 || nocc > n)))
        return dmk_fail(ctx, DMK_ERR_INVALID, "fit_kmat: bad arguments");
    FamScope fs(ctx, DMK_FAM_FIT);
    hipLaunchKernelGGL(fit_kmat_kernel, dim3(grid_for((long long)batch * n * n)), dim3(256), 0, ctx->stream, n, batch, ew, f, beta, nocc, K);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_scatter2d_add_f64(dmk_ctx *ctx, int m, const int32_t *idx, const double *src, double alpha, double *dst, int64_t ld_dst,
                          int elem_stride) {
    if (!ctx) return DMK_ERR_INVALID;
    if (m < 0 || !idx || !src || !dst || ld_dst <= 0 || elem_stride <= 0)
        return dmk_fail(ctx, DMK_ERR_INVALID, "scatter2d_add: bad arguments");
    if (m == 0) return DMK_OK;
    FamScope fs(ctx, DMK_FAM_FIT);
    hipLaunchKernelGGL(scatter2d_kernel, dim3(grid_for((long long)m * m)), dim3(256), 0, ctx->stream, m, idx, src, alpha, dst,
                       (long long)ld_dst, elem_stride);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_axpy_f64(dmk_ctx *ctx, int64_t n, double alpha, const double *x, double *y) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n < 0 || !x || !y) return dmk_fail(ctx, DMK_ERR_INVALID, "axpy: bad arguments");
    if (n == 0) return DMK_OK;
    FamScope fs(ctx, DMK_FAM_FIT);
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (long long)n, alpha, x, y);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_vcor_dV_dparam(dmk_ctx *ctx, int nent, int nb, const double *G, int64_t ldg, const int32_t *nz_ptr,
                       const int32_t *nz_i, const int32_t *nz_j, const double *nz_val, const int64_t *out_off, double *dV) {
    if (!ctx) return DMK_ERR_INVALID;
    if (nent < 0 || nb <= 0 || !G || !nz_ptr || !nz_i || !nz_j || !nz_val || !out_off || !dV)
        return dmk_fail(ctx, DMK_ERR_INVALID, "vcor_dV_dparam: bad arguments");
    if (nent == 0) return DMK_OK;
    if (nent > 65535) return dmk_fail(ctx, DMK_ERR_INVALID, "vcor_dV_dparam: more than 65535 entries per call");
    FamScope fs(ctx, DMK_FAM_FIT);
    const long long npair = (long long)nb * (nb + 1) / 2;
    hipLaunchKernelGGL(dv_dparam_kernel, dim3(grid_for(npair, 256), nent), dim3(256), 0, ctx->stream, nent, nb, (long long)ldg, G,
                       nz_ptr, nz_i, nz_j, nz_val, dV, reinterpret_cast<const long long *>(out_off));
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

// ---- the whole T = 0 objective of FitVcorEmb as ONE call (routine/slater.py:1059-1124 errfunc) ------------------------------
// V_emb = v0 + t v1 -> H = embH1 + V_emb -> eigenpairs (warm Ogita-Aishima refinement from the previous basis, ENQUEUED without a
// host look at its verdict) -> T = 0 occupations of both spin channels (one launch) -> rho on the fitted block, residual, sum of
// squares -> a pinned host record polled by the caller: ~15 launches issued back to back from C and ONE wait, where the Python
// chain paid 7 - 20 us of interpreter time between 25 launches and two synchronising read-backs per evaluation.
// Returns DMK_OK with *status = 0 (objective valid: *f2 = sum of squares, w / occ / Vp / drho hold this evaluation's levels,
// occupations, eigenvector rows and residual for the gradient), 1 (the refinement did not verify every matrix within `npass`
// passes -- nothing usable was produced, the caller falls back to the synchronous solver) or 2 (non-finite levels).
int dmk_fit_objective(dmk_ctx *ctx, const dmk_fit_args *a, double *f2, int *status, int *settle_pass) {
    if (!ctx) return DMK_ERR_INVALID;
    if (!a || !f2 || !status || a->nb <= 0 || a->spin < 1 || a->spin > 2 || a->nidx <= 0 || !a->v0 || !a->H1 || !a->H || !a->Vp ||
        !a->w || !a->occ || !a->fit_idx || !a->W || !a->target || !a->drho || !a->work || !a->slot)
        return dmk_fail(ctx, DMK_ERR_INVALID, "fit_objective: bad arguments");
    const int nb = a->nb, spin = a->spin, nidx = a->nidx;
    FitSlot *slot = static_cast<FitSlot *>(a->slot);
    // work: [0, 2048) partial sums | [2048] sum of squares | [2056, 2056 + 8 spin) occupation info | then 2 spin ints of verdicts
    double *part = a->work, *ss_dev = a->work + 2048, *occ_info = a->work + 2056;
    int *verdict = reinterpret_cast<int *>(a->work + 2056 + 8 * 4);
    const int gx = (nidx + 15) / 16;
    if (gx * gx * spin > 2048) return dmk_fail(ctx, DMK_ERR_INVALID, "fit_objective: %d fitted indices exceed the partial-sum table", nidx);
    const unsigned long long seq = ++ctx->fit_seq;
    // with the ray's norm bounds the unpack kernel also leaves |H|, the state words and the counters in the refinement's workspace
    static const bool ray_norm_on = !(getenv("DMK_FIT_RAY_NORM") && atoi(getenv("DMK_FIT_RAY_NORM")) == 0);
    const bool norm_here = ray_norm_on && a->ray_norm != nullptr && a->v1 != nullptr && spin <= 256;
    RfWorkspace W;
    int rc = DMK_OK;
    if (norm_here) {
        rc = rf_workspace(ctx, nb, spin, &W);
        if (rc) return rc;
    }
    {
        FamScope fs(ctx, DMK_FAM_FIT);
        hipLaunchKernelGGL(fit_ray_unpack_kernel, dim3(grid_for((long long)nb * nb * spin)), dim3(256), 0, ctx->stream, nb, spin,
                           a->v0, a->v1, a->t, a->H1, a->H, norm_here ? a->ray_norm : nullptr, norm_here ? W.anorm : nullptr,
                           norm_here ? W.state : nullptr, norm_here ? W.arrive : nullptr);
        DMK_CHECK_LAUNCH(ctx);
    }
    rc = dmk_eigh_refine_enqueue(ctx, nb, spin, a->H, a->Vp, a->w, a->Vp, a->npass, verdict, norm_here ? 1 : 0);
    if (rc) return rc;
    rc = dmk_assign_occ_zero_t_batch(ctx, nb, spin, a->w, a->nelec, a->has_mu0 ? a->mu0 : nullptr, (a->has_mu0 ? 1 : 0) | 4,
                                     a->tol_deg, a->occ, occ_info);
    if (rc) return rc;
    {
        FamScope fs(ctx, DMK_FAM_FIT);
        hipLaunchKernelGGL(fit_rho_diff_kernel, dim3(gx, gx, spin), dim3(NT), 0, ctx->stream, nb, nidx, a->Vp, a->occ, a->fit_idx, a->W,
                           a->target, a->drho, part);
        hipLaunchKernelGGL(fit_final_kernel, dim3(1), dim3(256), 0, ctx->stream, gx * gx * spin, part, ss_dev, spin, verdict, occ_info,
                           slot, seq);
        DMK_CHECK_LAUNCH(ctx);
    }
    // wait for the record: poll the pinned sequence number (the kernel releases it at system scope after the payload); a
    // stream synchronisation is the fallback when polling is switched off or takes implausibly long
    static const bool poll = !(getenv("DMK_FIT_POLL") && atoi(getenv("DMK_FIT_POLL")) == 0);
    bool seen = false;
    if (poll) {
        volatile unsigned long long *ps = &slot->seq;
        for (long spins = 0; spins < 200000000L; ++spins) {
            if (__atomic_load_n(ps, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
            __builtin_ia32_pause();
        }
    }
    if (!seen) {
        DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (__atomic_load_n(&slot->seq, __ATOMIC_ACQUIRE) != seq)
            return dmk_fail(ctx, DMK_ERR_STATE, "fit_objective: the result record was not written");
    }
    *f2 = slot->f2;
    int st = 0, worst_pass = 0;
    for (int i = 0; i < spin; ++i) {
        if ((int)slot->verdict[i] != 1 && (int)slot->verdict[i] != 3) st = 1;     // 1 verified, 3 settled by prediction
        worst_pass = std::max(worst_pass, (int)slot->verdict[spin + i]);
    }
    if (st == 0)
        for (int i = 0; i < spin; ++i)
            if (slot->occ_status[i] != 0.0) st = 2;
    *status = st;
    if (settle_pass) *settle_pass = worst_pass;
    return DMK_OK;
}

}  // extern "C"
