// K1b -- eigenpairs of the real symmetric tridiagonal T = (d, e) with every per-eigenvector vector in LDS
//        (phase 2 of the batched eigensolver for 64 < n <= 256; reference: routine/mfd.py:77-83, one la.eigh per k-point).
//
// Round 2 / 3 ran this phase inside eigh_kernel with ONE LANE PER EIGENVALUE and a pivoted tridiagonal elimination whose
// seven n-vectors per eigenvector lived in a lane-major GLOBAL scratch (7 n^2 doubles per matrix: 968 MB for 432 x 200^2):
// every step of its first-order recurrences was a round trip through L2 / HBM -- 2.15 ms and 5.9 of the launch's 6.9 GB of
// HBM traffic (rocprofv3 PMC, profiles/r03_d_*) for an arithmetically tiny amount of work.
//
// Here one 256-thread workgroup per matrix keeps everything in LDS:
//   (a) block splitting at negligible couplings, (b) the k-th eigenvalue of its block by bisection on the Sturm count -- both
//       exactly as in eigh_kernel (one lane per eigenvalue, d / e / e^2 in LDS);
//   (c) the eigenvector from the TWISTED FACTORISATION of T - lam I (Parlett & Dhillon; LAPACK dlar1v without its RRR shifts):
//         forward   D+_1 = d_1 - lam,  D+_{i+1} = (d_{i+1} - lam) - e_i^2 / D+_i          (L+ D+ L+^T from the top)
//         backward  D-_n = d_n - lam,  D-_i     = (d_i - lam)     - e_i^2 / D-_{i+1}      (U- D- U-^T from the bottom)
//         gamma_i = D+_i + D-_i - (d_i - lam);  r = argmin |gamma_i|;  z_r = 1,
//         z_i = -(e_i / D+_i) z_{i+1} (i < r),   z_{i+1} = -(e_i / D-_{i+1}) z_i (i >= r),
//       which needs ONE n-vector per eigenvector: D+ is stored by the forward pass, the first backward pass only tracks
//       argmin |gamma| against it, a second backward pass (bottom -> r + 1) leaves e_{i-1} / D-_i in the slots below the twist,
//       and z overwrites the same slots.  A wave owns an 8-column slab of the n x 32 work array (row stride 33 doubles: the
//       transposed write-out of Z^T is conflict-light), so the four waves never meet at a barrier; n = 200: 64 KB of LDS, i.e.
//       TWO workgroups per CU -- all 432 matrices of C5 resident at once, and two waves per SIMD to overlap the dependent
//       chains (a 16-column slab, one workgroup per CU, measured 2.5 ms for 432 x 200: 0.65 ms of bisection + 0.35 ms of
//       vectors per matrix with nothing to hide the latencies behind, and two rounds of workgroups).
//       With lam accurate to eps |T| the residual is |gamma_r| |z_r| / |z| = O(n eps |T|); eigenvalues of one block closer than
//       1e-3 |T| form a cluster whose vectors are re-orthogonalised (modified Gram-Schmidt, twice), the same rule as before;
//   (d) the ACCEPTANCE TEST of eigh_kernel on every vector (|T z - lam z|_inf <= 64 n eps |T|, |z| = 1; NaN fails): a matrix with
//       a failed vector is only FLAGGED -- the caller then runs eigh_kernel's phase 2 (pivoted inverse iteration with its
//       dstein-style repair path) on the flagged matrices, so nothing is ever accepted on trust;
//   (e) rank sort, eigenvalues out.
// HBM traffic: d, e in; Z^T (n^2 doubles per matrix) out and once more through L2 for clusters / the acceptance test.
#include "common.h"
#include <cstdlib>
#include <vector>

namespace {

constexpr int TP_NT = 256;
constexpr int TP_NW = TP_NT / 64;
constexpr int TP_LB = 8;             // eigenvectors in flight per wave
constexpr int TP_LD = TP_NW * TP_LB + 1;   // row stride of the work array (doubles): odd, so the transposed write-out is conflict-light

struct TpArgs {
    int n, batch;
    const double *d, *e;     // batch x n (e[i] couples i and i + 1); left untouched
    double *Zt;              // batch x n x n: row j = eigenvector j of T (unsorted order: ascending inside each block)
    double *w;               // batch x n eigenvalues, ascending
    int *rank_out;           // batch x n: position of eigenvalue j in ascending order
    int *flags;              // batch: number of vectors that failed the acceptance test (or that fault injection sends to the legacy path)
    int inject;
    double *debug;           // optional batch x 4: first failed vector j, its residual / (rtol |T|), |z|^2 - 1, twist index (DMK_EIGH_DEBUG)
};

__device__ __forceinline__ double tp_wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double tp_rcp(double q) {          // 1 / q: hardware estimate + two Newton steps (full precision for normal q)
    double r = __builtin_amdgcn_rcp(q);
    r = r * (2.0 - q * r);
    r = r * (2.0 - q * r);
    return r;
}

__global__ __launch_bounds__(TP_NT) void tri_eigpairs_kernel(const TpArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int n = g.n, b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *dl = sm, *el = dl + n, *e2 = el + n, *lam = e2 + n, *bnorm = lam + n;
    int *bs = reinterpret_cast<int *>(bnorm + n), *be = bs + n, *bad = be + n, *rank = bad + n;
    double *Wk = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(rank + n) + 15) & ~static_cast<uintptr_t>(15));   // [n][TP_LD]
    const double *d = g.d + (size_t)b * n, *e = g.e + (size_t)b * n;
    double *Zt = g.Zt + (size_t)b * n * n;
    const double eps = 2.220446049250313e-16;
    long long tph[7];
    tph[0] = wall_clock64();

    for (int t = tid; t < n; t += TP_NT) {
        dl[t] = d[t];
        el[t] = (t + 1 < n) ? e[t] : 0.0;
    }
    __syncthreads();
    // ---- (a) unreduced blocks ----------------------------------------------------------------------------------------------
    if (tid == 0) {
        int s0 = 0;
        for (int i = 0; i < n; ++i) {
            const bool cut = (i == n - 1) || fabs(el[i]) <= eps * (fabs(dl[i]) + fabs(dl[i + 1]));
            if (!cut) continue;
            el[i] = 0.0;
            double nrm = 0.0;
            for (int q = s0; q <= i; ++q)
                nrm = fmax(nrm, fabs(dl[q]) + (q > s0 ? fabs(el[q - 1]) : 0.0) + (q < i ? fabs(el[q]) : 0.0));
            for (int q = s0; q <= i; ++q) { bs[q] = s0; be[q] = i + 1; bnorm[q] = nrm; }
            s0 = i + 1;
        }
    }
    __syncthreads();
    for (int t = tid; t < n; t += TP_NT) e2[t] = el[t] * el[t];
    __syncthreads();

    tph[1] = wall_clock64();
    // ---- (b) eigenvalues: bisection on the Sturm count, one lane per eigenvalue ----------------------------------------------
    for (int j = tid; j < n; j += TP_NT) {
        const int s0 = bs[j], t0 = be[j], m = t0 - s0, kk = j - s0;
        if (m == 1) { lam[j] = dl[s0]; continue; }
        const double tn = bnorm[j];
        double emax2 = 0.0, lo = dl[s0], hi = dl[s0];
        for (int i = s0; i < t0; ++i) {
            const double rad = (i > s0 ? fabs(el[i - 1]) : 0.0) + (i + 1 < t0 ? fabs(el[i]) : 0.0);
            lo = fmin(lo, dl[i] - rad);
            hi = fmax(hi, dl[i] + rad);
            if (i + 1 < t0) emax2 = fmax(emax2, e2[i]);
        }
        const double pivmin = 2.2250738585072014e-308 * fmax(1.0, emax2);
        lo -= 2.0 * eps * tn * m + 2.0 * pivmin;
        hi += 2.0 * eps * tn * m + 2.0 * pivmin;
        // plain bisection (measured: a quadrisection with three interleaved Sturm recurrences per lane is SLOWER, 0.75 vs 0.67 ms per
        // 200 x 200 matrix -- the loop is bound by f64 instruction issue (v_rcp_f64 + two Newton steps per pivot), not by latency)
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (!(mid > lo && mid < hi)) break;
            int cnt = 0;
            double q = dl[s0] - mid;
            if (fabs(q) < pivmin) q = -pivmin;
            cnt += q < 0.0 ? 1 : 0;
#pragma unroll 4
            for (int i = s0 + 1; i < t0; ++i) {
                q = (dl[i] - mid) - e2[i - 1] * tp_rcp(q);
                if (fabs(q) < pivmin) q = -pivmin;
                cnt += q < 0.0 ? 1 : 0;
            }
            if (cnt > kk) hi = mid; else lo = mid;
            if (hi - lo <= eps * (fabs(lo) + fabs(hi)) + 2.0 * pivmin) break;
        }
        lam[j] = 0.5 * (lo + hi);
    }
    __syncthreads();

    tph[2] = wall_clock64();
    // ---- (c) eigenvectors by twisted factorisation: wave w works in columns [8 w, 8 w + 8) of Wk ---------------------------
    {
        double *slab = Wk + wave * TP_LB;
        for (int j0 = wave * TP_LB; j0 < n; j0 += TP_NW * TP_LB) {
            const int l = lane & (TP_LB - 1);
            const int j = j0 + l;
            const bool act = lane < TP_LB && j < n;
            int s0 = 0, t0 = 0;
            double inv = 0.0;
            if (act) {
                s0 = bs[j]; t0 = be[j];
                double *x = slab + l;                                  // x[i * TP_LD]
                if (t0 - s0 == 1) {
                    x[(size_t)s0 * TP_LD] = 1.0;
                    inv = 1.0;
                } else {
                    const double lm = lam[j];
                    const double pert = fmax(eps * bnorm[j], 1e-300);
                    auto guard = [&](double v) { return fabs(v) < pert ? (v >= 0.0 ? pert : -pert) : v; };
                    // forward: D+_i into slot i.  (Storing the multipliers e_i / D+_i instead, so that the upward z recurrence is a bare
                    // multiply, measured slower: 0.63 vs 0.56 ms -- the extra multiply and fma in the two long passes cost more.)
                    double dp = guard(dl[s0] - lm);
                    x[(size_t)s0 * TP_LD] = dp;
                    for (int i = s0; i + 1 < t0; ++i) {
                        dp = guard((dl[i + 1] - lm) - e2[i] * tp_rcp(dp));
                        x[(size_t)(i + 1) * TP_LD] = dp;
                    }
                    // backward 1: D-_i on the fly, gamma_i against the stored D+_i, argmin
                    double dm = guard(dl[t0 - 1] - lm);
                    double gmin = fabs(x[(size_t)(t0 - 1) * TP_LD] + dm - (dl[t0 - 1] - lm));
                    int r = t0 - 1;
                    for (int i = t0 - 2; i >= s0; --i) {
                        dm = guard((dl[i] - lm) - e2[i] * tp_rcp(dm));
                        const double gm = fabs(x[(size_t)i * TP_LD] + dm - (dl[i] - lm));
                        if (gm < gmin) { gmin = gm; r = i; }
                    }
                    // backward 2: bottom -> r + 1, slot i <- e_{i-1} / D-_i  (the multiplier of z_{i-1} -> z_i)
                    dm = guard(dl[t0 - 1] - lm);
                    for (int i = t0 - 1; i > r; --i) {
                        const double rd = tp_rcp(dm);
                        x[(size_t)i * TP_LD] = el[i - 1] * rd;
                        dm = guard((dl[i - 1] - lm) - e2[i - 1] * rd);
                    }
                    // z: twist at r, up with D+, down with the stored multipliers; z overwrites the slots
                    double nr = 1.0, zc = 1.0;
                    for (int i = r - 1; i >= s0; --i) {
                        zc = -(el[i] * tp_rcp(x[(size_t)i * TP_LD])) * zc;
                        x[(size_t)i * TP_LD] = zc;
                        nr = fma(zc, zc, nr);
                    }
                    x[(size_t)r * TP_LD] = 1.0;
                    zc = 1.0;
                    for (int i = r + 1; i < t0; ++i) {
                        zc = -x[(size_t)i * TP_LD] * zc;
                        x[(size_t)i * TP_LD] = zc;
                        nr = fma(zc, zc, nr);
                    }
                    inv = 1.0 / sqrt(nr);
                }
            }
            // transposed write-out: eigenvector q of this batch -> row j0 + q of Z^T, lanes along the components (coalesced);
            // zero outside its block.  Wave-private slab: the LDS pipe keeps a wave's own accesses in order.
            for (int q = 0; q < TP_LB && j0 + q < n; ++q) {
                const int qs = __shfl(s0, q, 64), qt = __shfl(t0, q, 64);
                const double qi = __shfl(inv, q, 64);
                double *row = Zt + (size_t)(j0 + q) * n;
                for (int i = lane; i < n; i += 64)
                    row[i] = (i >= qs && i < qt) ? slab[(size_t)i * TP_LD + q] * qi : 0.0;
            }
        }
    }
    __threadfence_block();
    __syncthreads();

    tph[3] = wall_clock64();
    // ---- clusters: one wave each, members in ascending order (as in eigh_kernel) ----------------------------------------------
    {
        int cluster = -1;
        for (int j = 0; j < n; ++j) {
            const bool first = (j == bs[j]) || (lam[j] - lam[j - 1] > 1e-3 * bnorm[j]);
            if (!first) continue;
            ++cluster;
            if (cluster % TP_NW != wave) continue;
            const int s0 = bs[j], t0 = be[j];
            int last = j;
            while (last + 1 < t0 && lam[last + 1] - lam[last] <= 1e-3 * bnorm[j]) ++last;
            for (int q = j + 1; q <= last; ++q) {
                double *zq = Zt + (size_t)q * n;
                for (int pass = 0; pass < 2; ++pass)
                    for (int p = j; p < q; ++p) {
                        const double *zp = Zt + (size_t)p * n;
                        double dot = 0.0;
                        for (int i = s0 + lane; i < t0; i += 64) dot += zp[i] * zq[i];
                        dot = dmk_wave_sum(dot);
                        for (int i = s0 + lane; i < t0; i += 64) zq[i] -= dot * zp[i];
                    }
                double nr = 0.0;
                for (int i = s0 + lane; i < t0; i += 64) nr += zq[i] * zq[i];
                nr = dmk_wave_sum(nr);
                const double invn = nr > 0.0 ? 1.0 / sqrt(nr) : 0.0;
                for (int i = s0 + lane; i < t0; i += 64) zq[i] *= invn;
            }
        }
    }
    __threadfence_block();
    __syncthreads();

    tph[4] = wall_clock64();
    // ---- (d) acceptance test, one wave per vector -------------------------------------------------------------------------------
    const double rtol = 64.0 * n * eps;
    for (int j = wave; j < n; j += TP_NW) {
        const int s0 = bs[j], t0 = be[j];
        bool ok = true;
        double dbg_r = 0.0, dbg_z = 0.0;
        if (t0 - s0 > 1) {
            const double *z = Zt + (size_t)j * n;
            const double lj = lam[j];
            double r = 0.0, zn = 0.0;
            for (int i = s0 + lane; i < t0; i += 64) {
                const double zi = z[i];
                double t = (dl[i] - lj) * zi;
                if (i > s0) t += el[i - 1] * z[i - 1];
                if (i + 1 < t0) t += el[i] * z[i + 1];
                r = fmax(r, fabs(t));
                if (!(fabs(t) <= 1.7e308)) r = 1.7e308;
                zn += zi * zi;
            }
            r = tp_wave_max(r);
            zn = dmk_wave_sum(zn);
            ok = r <= rtol * fmax(bnorm[j], 1e-300) && fabs(zn - 1.0) <= 1e-8;
            dbg_r = r / (rtol * fmax(bnorm[j], 1e-300));
            dbg_z = zn - 1.0;
        }
        if (g.inject > 0 && (j % g.inject) == g.inject - 1) ok = false;
        if (!ok && lane == 0) {
            if (atomicAdd(&g.flags[b], 1) == 0 && g.debug) {
                g.debug[4 * b] = j; g.debug[4 * b + 1] = dbg_r; g.debug[4 * b + 2] = dbg_z; g.debug[4 * b + 3] = (double)(t0 - s0);
            }
        }
    }

    tph[5] = wall_clock64();
    // ---- (e) rank sort (ascending, stable) -----------------------------------------------------------------------------------------
    for (int j = tid; j < n; j += TP_NT) {
        const double dj = lam[j];
        int rk = 0;
        for (int q = 0; q < n; ++q) {
            const double dq = lam[q];
            rk += (dq < dj || (dq == dj && q < j)) ? 1 : 0;
        }
        g.w[(size_t)b * n + rk] = dj;
        g.rank_out[(size_t)b * n + j] = rk;
    }
    if (g.debug && b == 0 && tid == 0) {                     // phase clocks of matrix 0 (100 MHz wall clock)
        tph[6] = wall_clock64();
        for (int q = 0; q < 6; ++q) g.debug[4 * (size_t)g.batch + q] = (double)(tph[q + 1] - tph[q]);
    }
}

}  // namespace

size_t tri_eigpairs_lds(int n) { return (size_t)n * (5 * 8 + 4 * 4) + 16 + (size_t)n * TP_LD * 8; }

// 1 = launched (flags must then be honoured by the caller), 0 = shape not covered / DMK_EIGH_TRIPAIRS=0, < 0 on error
int launch_tri_eigpairs(dmk_ctx *ctx, int n, int batch, const double *d, const double *e, double *Zt, double *w, int *rank_out,
                        int *flags, int inject) {
    if (const char *en = getenv("DMK_EIGH_TRIPAIRS")) if (atoi(en) == 0) return 0;
    const size_t lds = tri_eigpairs_lds(n);
    if (n < 2 || lds > 160 * 1024 - 512) return 0;
    TpArgs g;
    g.n = n; g.batch = batch; g.d = d; g.e = e; g.Zt = Zt; g.w = w; g.rank_out = rank_out; g.flags = flags; g.inject = inject;
    g.debug = nullptr;
    const bool debug = getenv("DMK_EIGH_DEBUG") != nullptr;
    if (debug) {
        DMK_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&g.debug), ((size_t)batch * 4 + 8) * sizeof(double)));
        DMK_HIP(ctx, hipMemsetAsync(g.debug, 0, ((size_t)batch * 4 + 8) * sizeof(double), ctx->stream));
    }
    DMK_HIP(ctx, hipMemsetAsync(flags, 0, (size_t)batch * sizeof(int), ctx->stream));
    DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(tri_eigpairs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(tri_eigpairs_kernel, dim3(batch), dim3(TP_NT), lds, ctx->stream, g);
    DMK_CHECK_LAUNCH(ctx);
    if (debug) {                 // builder's lab switch: how many vectors went to the repair pass, and the first one of each matrix
        std::vector<int> hf(batch);
        std::vector<double> hd((size_t)batch * 4 + 8);
        DMK_HIP(ctx, hipMemcpy(hf.data(), flags, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost));
        DMK_HIP(ctx, hipMemcpy(hd.data(), g.debug, hd.size() * sizeof(double), hipMemcpyDeviceToHost));
        long long nbad = 0, nmat = 0;
        for (int b = 0; b < batch; ++b) { nbad += hf[b]; nmat += hf[b] != 0; }
        fprintf(stderr, "[tri_eigpairs n=%d batch=%d] %lld vectors of %lld matrices failed the acceptance test\n", n, batch, nbad, nmat);
        const double *tp = hd.data() + (size_t)batch * 4;
        fprintf(stderr, "   matrix 0 (us): load + split %.1f, bisection %.1f, twisted vectors %.1f, clusters %.1f, acceptance %.1f, sort %.1f\n",
                tp[0] * 1e-2, tp[1] * 1e-2, tp[2] * 1e-2, tp[3] * 1e-2, tp[4] * 1e-2, tp[5] * 1e-2);
        int shown = 0;
        for (int b = 0; b < batch && shown < 6; ++b)
            if (hf[b]) {
                fprintf(stderr, "   matrix %d: %d failed; first j = %d, residual / tol = %.3e, |z|^2 - 1 = %.3e, block size %d\n", b, hf[b],
                        (int)hd[4 * b], hd[4 * b + 1], hd[4 * b + 2], (int)hd[4 * b + 3]);
                ++shown;
            }
        (void)hipFree(g.debug);
    }
    return 1;
}
