// K1 -- batched complex-Hermitian (and real-symmetric) eigensolver, one workgroup per matrix.
//
// Replaces scipy.linalg.eigh (LAPACK zheevd / dsyevd) at routine/mfd.py:42-106
// (DiagRHF / DiagUHF[_symm]: one la.eigh per k-point and spin), routine/slater.py:278
// (eig bath) and lo/lowdin.py:87 (Loewdin metric).  Like LAPACK with lower=True only the
// lower triangle of the input is referenced.
//
// Algorithm (backward stable, no vendor library on the path):
//   1. Householder tridiagonalisation  A = Q T Q^H  (unblocked, the zhetd2 recurrence:
//      p = tau A22 v, w = p - (tau/2)(p^H v) v, A22 -= v w^H + w v^H); the trailing block is
//      kept as a full Hermitian matrix in HBM/L2 so that the matrix-vector product streams
//      contiguous rows across the 64 lanes of a wave; reflectors are saved row-major.
//   2. implicit-shift QL on the real tridiagonal (d, e): one lane runs the scalar recurrence
//      of a sweep and leaves its Givens pairs in LDS, then all lanes apply the sweep to Z^T
//      (row i of Z^T = eigenvector i of T, so a rotation touches two contiguous rows).
//   3. rank sort (ascending, stable) and back-transformation y = H_0 ... H_{n-2} z, four
//      eigenvectors per wave held in registers, reflector loads software-pipelined.
//
// Bound: latency / L2: algorithmic HBM traffic is
// 2*16*n^2 + 8*n bytes per matrix.
#include "common.h"

namespace {

constexpr int NT = 320;          // five waves: during the QL phase four apply rotation sweeps while the fifth produces them
constexpr int NW = NT / 64;
constexpr int NC = NT - 64;      // consumer threads of the QL phase

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {   // conj(a) * b
    return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double wave_sum(double v) { return dmk_wave_sum(v); }

struct EighArgs {
    int n, batch;
    const void *A;          // c128 or f64 (a_real)
    int a_real;
    const double *add;      // optional n x n f64 added to each matrix
    int add_group;          // matrices b and b' share add[(b / add_group)]
    double *w;              // batch x n
    void *Vt;               // c128 or f64 (v_real): batch x n x n, row = eigenvector
    int v_real;
    double2 *W, *Vh;        // workspaces batch x n x n
    double *Zt;             // batch x n x n
    double *d, *e;          // batch x n
    double2 *tau;           // batch x n
    int *status;            // device flag, set to 1 on non-convergence
};

// LDS carve (dynamic): vbuf[n] c128 | pbuf[n] c128 | cs[2][2n] f64 | red[3*NW + 2] f64 | 16 ints | dl[n] | el[n]
template <int R>
__global__ __launch_bounds__(NT) void eigh_kernel(const EighArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = g.n;
    double2 *vbuf = reinterpret_cast<double2 *>(smem);
    double2 *pbuf = vbuf + n;
    double *cs = reinterpret_cast<double *>(pbuf + n);
    double *red = cs + 4 * n;          // 3*NW + 2 doubles
    int *ired = reinterpret_cast<int *>(red + 3 * NW + 2);     // 16 ints

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t nn = (size_t)n * n;
    double2 *W = g.W + b * nn;
    double2 *Vh = g.Vh + b * nn;
    double *Zt = g.Zt + b * nn;
    double *d = g.d + (size_t)b * n;
    double *e = g.e + (size_t)b * n;
    double2 *tau = g.tau + (size_t)b * n;

    long long tphase[5];
    tphase[0] = wall_clock64();
    // ---- phase 0: W = Hermitian completion of the lower triangle (+ add), Zt = I ------------
    {
        const double *addm = g.add ? g.add + (size_t)(g.add_group > 0 ? b / g.add_group : 0) * nn : nullptr;
        for (size_t idx = tid; idx < nn; idx += NT) {
            const int i = (int)(idx / n), j = (int)(idx % n);
            const int lo_i = i >= j ? i : j, lo_j = i >= j ? j : i;
            double2 v;
            if (g.a_real) {
                v = make_double2(reinterpret_cast<const double *>(g.A)[b * nn + (size_t)lo_i * n + lo_j], 0.0);
            } else {
                v = reinterpret_cast<const double2 *>(g.A)[b * nn + (size_t)lo_i * n + lo_j];
                if (i < j) v.y = -v.y;
                if (i == j) v.y = 0.0;
            }
            if (addm) v.x += addm[(size_t)lo_i * n + lo_j];
            W[idx] = v;
            Zt[idx] = (i == j) ? 1.0 : 0.0;
        }
    }
    __syncthreads();

    tphase[1] = wall_clock64();
    // ---- phase 1: Householder tridiagonalisation ------------------------------------------------
    for (int k = 0; k + 1 < n; ++k) {
        const int m = n - k - 1;
        const double2 *rowk = W + (size_t)k * n + (k + 1);   // conj of the column below the diagonal
        // (a) norm of x[1:]
        double part = 0.0;
        for (int t = 1 + tid; t < m; t += NT) {
            const double2 x = rowk[t];
            part += x.x * x.x + x.y * x.y;
        }
        part = wave_sum(part);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        // (b) reflector parameters (every thread computes the same scalars)
        double xnorm2 = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) xnorm2 += red[q];
        double2 alpha = rowk[0];
        alpha.y = -alpha.y;                       // x[0] = conj(W[k][k+1])
        double2 tk = make_double2(0.0, 0.0), scale = make_double2(0.0, 0.0);
        double beta = alpha.x;
        if (!(xnorm2 == 0.0 && alpha.y == 0.0)) {
            const double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xnorm2);
            beta = alpha.x >= 0.0 ? -nrm : nrm;
            tk = make_double2((beta - alpha.x) / beta, -alpha.y / beta);
            const double dr = alpha.x - beta, di = alpha.y;
            const double den = dr * dr + di * di;
            scale = make_double2(dr / den, -di / den);   // 1 / (alpha - beta)
        }
        const bool active = !(tk.x == 0.0 && tk.y == 0.0);
        // (c) v -> LDS and reflector store
        for (int t = tid; t < m; t += NT) {
            double2 v;
            if (t == 0) v = make_double2(1.0, 0.0);
            else {
                double2 x = rowk[t];
                x.y = -x.y;
                v = cmul(x, scale);
            }
            vbuf[t] = v;
            Vh[(size_t)k * n + (k + 1) + t] = v;
        }
        if (tid == 0) {
            tau[k] = tk;
            d[k] = W[(size_t)k * n + k].x;
            e[k] = beta;
        }
        __syncthreads();
        if (active) {
            // (d) p = tau * A22 v   (one row per wave at a time; lanes along the row)
            for (int i = wave; i < m; i += NW) {
                const double2 *row = W + (size_t)(k + 1 + i) * n + (k + 1);
                double sr = 0.0, si = 0.0;
                for (int j = lane; j < m; j += 64) {
                    const double2 a = row[j], v = vbuf[j];
                    sr += a.x * v.x - a.y * v.y;
                    si += a.x * v.y + a.y * v.x;
                }
                sr = wave_sum(sr);
                si = wave_sum(si);
                if (lane == 0) pbuf[i] = cmul(tk, make_double2(sr, si));
            }
            __syncthreads();
            // (e) alpha2 = -1/2 tau (p^H v)
            double ar = 0.0, ai = 0.0;
            for (int t = tid; t < m; t += NT) {
                const double2 c = cmulc(pbuf[t], vbuf[t]);
                ar += c.x;
                ai += c.y;
            }
            ar = wave_sum(ar);
            ai = wave_sum(ai);
            if (lane == 0) { red[NW + 2 * wave] = ar; red[NW + 2 * wave + 1] = ai; }
            __syncthreads();
            double2 pv = make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < NW; ++q) { pv.x += red[NW + 2 * q]; pv.y += red[NW + 2 * q + 1]; }
            double2 a2 = cmul(tk, pv);
            a2.x *= -0.5;
            a2.y *= -0.5;
            __syncthreads();
            // (f) w = p + alpha2 v   (in place in pbuf)
            for (int t = tid; t < m; t += NT) {
                const double2 av = cmul(a2, vbuf[t]);
                pbuf[t] = make_double2(pbuf[t].x + av.x, pbuf[t].y + av.y);
            }
            __syncthreads();
            // (g) A22 -= v w^H + w v^H : one row per wave at a time, lanes along the row (coalesced, no index division);
            // two rows in flight per wave
            for (int i0 = wave; i0 < m; i0 += 2 * NW) {
                const int i1 = i0 + NW;
                const bool two = i1 < m;
                const double2 vi0 = vbuf[i0], wi0 = pbuf[i0];
                const double2 vi1 = two ? vbuf[i1] : make_double2(0.0, 0.0), wi1 = two ? pbuf[i1] : make_double2(0.0, 0.0);
                double2 *r0 = W + (size_t)(k + 1 + i0) * n + (k + 1);
                double2 *r1 = W + (size_t)(k + 1 + (two ? i1 : i0)) * n + (k + 1);
                for (int j = lane; j < m; j += 64) {
                    const double2 vj = vbuf[j], wj = pbuf[j];
                    double2 a0 = r0[j], a1 = r1[j];
                    a0.x -= vi0.x * wj.x + vi0.y * wj.y + wi0.x * vj.x + wi0.y * vj.y;
                    a0.y -= vi0.y * wj.x - vi0.x * wj.y + wi0.y * vj.x - wi0.x * vj.y;
                    if (j == i0) a0.y = 0.0;
                    r0[j] = a0;
                    if (two) {
                        a1.x -= vi1.x * wj.x + vi1.y * wj.y + wi1.x * vj.x + wi1.y * vj.y;
                        a1.y -= vi1.y * wj.x - vi1.x * wj.y + wi1.y * vj.x - wi1.x * vj.y;
                        if (j == i1) a1.y = 0.0;
                        r1[j] = a1;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        d[n - 1] = W[(size_t)(n - 1) * n + (n - 1)].x;
        e[n - 1] = 0.0;
        if (n >= 1) tau[n - 1] = make_double2(0.0, 0.0);
    }
    __syncthreads();

    tphase[2] = wall_clock64();
    // ---- phase 2: implicit QL, rotation sweeps applied to Zt ---------------------------------------
    // e[i] couples i and i+1.  The sequential recurrence that generates a sweep (one lane) and the application of a
    // sweep to Zt (all columns in parallel) only meet through the rotation list, so they run CONCURRENTLY: wave NW-1
    // produces sweeps into a double-buffered list in LDS, waves 0..NW-2 consume them, each on its own columns
    // (a sweep on a column only depends on the previous sweep on that column).  Hand-over is by sequence counters
    // in LDS.  The tridiagonal (d, e) lives in LDS; the split point is found by a 64-wide ballot; a rotation
    // costs one v_rsq_f64 + two Newton steps instead of a square root and two divisions; the sweep is applied with
    // the row loads of eight rotations in flight.
    {
        volatile int *ctl = ired;                 // [0] sweeps published, [1] producer finished, [2..2+NW-2] consumer progress
        int *meta = ired + 8;                     // [2][2] = (mm, cnt) per buffer
        double *dl = reinterpret_cast<double *>(ired + 16);   // [n]
        double *el = dl + n;                                  // [n]
        for (int t = tid; t < n; t += NT) {
            dl[t] = d[t];
            el[t] = e[t];
        }
        if (tid < 16) ired[tid] = 0;
        __syncthreads();
        const double eps = 2.220446049250313e-16;
        if (wave == NW - 1) {
            // ---------------- producer ----------------
            int k = 0;
            bool fail = false;
            for (int l = 0; l < n && !fail; ++l) {
                int iter = 0;
                while (true) {
                    int mm = n - 1;
                    for (int base = l; base < n - 1; base += 64) {
                        const int mq = base + lane;
                        bool small = false;
                        if (mq < n - 1) small = fabs(el[mq]) <= eps * (fabs(dl[mq]) + fabs(dl[mq + 1]));
                        const unsigned long long mask = __ballot(small);
                        if (mask != 0ull) {
                            mm = base + __ffsll((long long)mask) - 1;
                            break;
                        }
                    }
                    if (mm == l) break;
                    // buffer k & 1 is free once every consumer has finished sweep k - 2
                    if (k >= 2) {
                        while (true) {
                            int lowest = ctl[2];
#pragma unroll
                            for (int w = 1; w < NW - 1; ++w) lowest = min(lowest, ctl[2 + w]);
                            if (lowest >= k - 1) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    double *csb = cs + (size_t)(k & 1) * 2 * n;
                    if (lane == 0) {
                        int cnt = 0;     // rotation q acts on rows (mm-1-q, mm-q)
                        double gq = (dl[l + 1] - dl[l]) / (2.0 * el[l]);
                        double r = sqrt(gq * gq + 1.0);
                        gq = dl[mm] - dl[l] + el[l] / (gq + (gq >= 0.0 ? fabs(r) : -fabs(r)));
                        double s = 1.0, c = 1.0, p = 0.0;
                        int i = mm - 1;
                        bool under = false;
                        double d_hi = dl[mm];                          // d[i+1], untouched so far in this sweep
                        double e_i = el[i], d_i = dl[i];
                        for (; i >= l; --i) {
                            const double f = s * e_i;
                            const double bq = c * e_i;
                            const double x = f * f + gq * gq;
                            const int ip = i > l ? i - 1 : l;          // next iteration's (e, d), clamped: no branch
                            const double e_nx = el[ip], d_nx = dl[ip];
                            if (x == 0.0) {
                                el[i + 1] = 0.0;
                                dl[i + 1] = d_hi - p;
                                el[mm] = 0.0;
                                under = true;
                                break;
                            }
                            double y = __builtin_amdgcn_rsq(x);
                            y = y * (1.5 - 0.5 * x * y * y);
                            y = y * (1.5 - 0.5 * x * y * y);
                            r = x * y;
                            r = r + 0.5 * y * fma(-r, r, x);           // sqrt(x) to the last bit or so
                            el[i + 1] = r;
                            s = f * y;
                            c = gq * y;
                            gq = d_hi - p;
                            r = (d_i - gq) * s + 2.0 * c * bq;
                            p = s * r;
                            dl[i + 1] = gq + p;
                            gq = c * r - bq;
                            csb[2 * cnt] = c;
                            csb[2 * cnt + 1] = s;
                            ++cnt;
                            d_hi = d_i;
                            d_i = d_nx;
                            e_i = e_nx;
                        }
                        if (!under) {
                            dl[l] = d_hi - p;                          // d_hi == original d[l] here
                            el[l] = gq;
                            el[mm] = 0.0;
                        }
                        meta[2 * (k & 1)] = mm;
                        meta[2 * (k & 1) + 1] = cnt;
                    }
                    __threadfence_block();
                    ++k;
                    if (lane == 0) ctl[0] = k;
                    if (++iter > 80) {
                        if (lane == 0) *g.status = 1;
                        fail = true;
                        break;
                    }
                }
            }
            __threadfence_block();
            if (lane == 0) ctl[1] = 1;
        } else {
            // ---------------- consumers ----------------
            int k = 0;
            while (true) {
                const int fin = ctl[1];
                const int avail = ctl[0];
                if (avail <= k) {
                    if (fin) break;
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                __threadfence_block();
                const int mm = meta[2 * (k & 1)], cnt = meta[2 * (k & 1) + 1];
                const double *csb = cs + (size_t)(k & 1) * 2 * n;
                // rotation q on rows (i, i+1) of Zt with i = mm-1-q
                for (int rcol = tid; rcol < n; rcol += NC) {
                    double hi = Zt[(size_t)mm * n + rcol];
                    double lo[8], nx[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) nx[u] = (u < cnt) ? Zt[(size_t)(mm - 1 - u) * n + rcol] : 0.0;
                    for (int q0 = 0; q0 < cnt; q0 += 8) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) lo[u] = nx[u];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int q = q0 + 8 + u;
                            nx[u] = (q < cnt) ? Zt[(size_t)(mm - 1 - q) * n + rcol] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int q = q0 + u;
                            if (q < cnt) {
                                const double c = csb[2 * q], sn = csb[2 * q + 1];
                                Zt[(size_t)(mm - q) * n + rcol] = sn * lo[u] + c * hi;
                                hi = c * lo[u] - sn * hi;
                            }
                        }
                    }
                    Zt[(size_t)(mm - cnt) * n + rcol] = hi;
                }
                ++k;
                __threadfence_block();
                if (lane == 0) ctl[2 + wave] = k;
            }
        }
        __syncthreads();
        for (int t = tid; t < n; t += NT) d[t] = dl[t];
        if (b == 0 && tid == 0) {
            long long *tp = reinterpret_cast<long long *>(g.status + 2);
            tp[4] = 0; tp[5] = 0; tp[6] = 0; tp[7] = ctl[0];
        }
    }
    __syncthreads();

    tphase[3] = wall_clock64();
    // ---- phase 3: sort + back-transformation ------------------------------------------------------
    // rank[j] = position of eigenvalue j in ascending order (stable); kept in cs[] as ints
    int *rank = reinterpret_cast<int *>(cs);
    for (int j = tid; j < n; j += NT) {
        const double dj = d[j];
        int rk = 0;
        for (int q = 0; q < n; ++q) {
            const double dq = d[q];
            rk += (dq < dj || (dq == dj && q < j)) ? 1 : 0;
        }
        rank[j] = rk;
        g.w[(size_t)b * n + rk] = dj;
    }
    __syncthreads();

    constexpr int E = (R <= 4) ? 4 : 1;
    for (int m0 = wave * E; m0 < n; m0 += NW * E) {
        double2 y[E][R];
#pragma unroll
        for (int q = 0; q < E; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r;
                const int mq = m0 + q;
                y[q][r] = (mq < n && i < n) ? make_double2(Zt[(size_t)mq * n + i], 0.0) : make_double2(0.0, 0.0);
            }
        double2 vnext[R];
        auto loadv = [&](int k, double2 *v) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r;
                v[r] = (k >= 0 && i > k && i < n) ? Vh[(size_t)k * n + i] : make_double2(0.0, 0.0);
            }
        };
        loadv(n - 2, vnext);
        for (int k = n - 2; k >= 0; --k) {
            double2 v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = vnext[r];
            loadv(k - 1, vnext);
            const double2 tk = tau[k];
            if (tk.x == 0.0 && tk.y == 0.0) continue;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double2 c = cmulc(v[r], y[q][r]);
                    sr += c.x;
                    si += c.y;
                }
                sr = wave_sum(sr);
                si = wave_sum(si);
                const double2 f = cmul(tk, make_double2(sr, si));
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double2 u = cmul(f, v[r]);
                    y[q][r].x -= u.x;
                    y[q][r].y -= u.y;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const int mq = m0 + q;
            if (mq >= n) continue;
            const size_t orow = (size_t)b * nn + (size_t)rank[mq] * n;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int i = lane + 64 * r;
                if (i < n) {
                    if (g.v_real) reinterpret_cast<double *>(g.Vt)[orow + i] = y[q][r].x;
                    else reinterpret_cast<double2 *>(g.Vt)[orow + i] = y[q][r];
                }
            }
        }
    }
    if (b == 0 && tid == 0) {       // phase clocks of matrix 0 (100 MHz wall clock), read with DMK_EIGH_TIMING=1
        long long *tp = reinterpret_cast<long long *>(g.status + 2);
        tphase[4] = wall_clock64();
        for (int q = 0; q < 4; ++q) tp[q] = tphase[q + 1] - tphase[q];
    }
}

int launch_eigh(dmk_ctx *ctx, int n, int batch, const void *A, int a_real, const double *add, int add_group, double *w,
                void *Vt, int v_real) {
    if (n <= 0 || batch <= 0) return DMK_OK;
    if (n > 1024) return dmk_fail(ctx, DMK_ERR_INVALID, "eigh: n = %d exceeds the supported maximum of 1024", n);
    const size_t nn = (size_t)n * n;
    const size_t per = nn * (16 + 16 + 8) + (size_t)n * (8 + 8 + 16);
    const size_t total = per * batch + 256;
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, total, &ws);
    if (rc) return rc;
    char *p = reinterpret_cast<char *>(ws);
    EighArgs g;
    g.n = n; g.batch = batch; g.A = A; g.a_real = a_real; g.add = add; g.add_group = add_group;
    g.w = w; g.Vt = Vt; g.v_real = v_real;
    g.status = reinterpret_cast<int *>(p); p += 256;
    g.W = reinterpret_cast<double2 *>(p); p += nn * 16 * batch;
    g.Vh = reinterpret_cast<double2 *>(p); p += nn * 16 * batch;
    g.tau = reinterpret_cast<double2 *>(p); p += (size_t)n * 16 * batch;
    g.Zt = reinterpret_cast<double *>(p); p += nn * 8 * batch;
    g.d = reinterpret_cast<double *>(p); p += (size_t)n * 8 * batch;
    g.e = reinterpret_cast<double *>(p);
    DMK_HIP(ctx, hipMemsetAsync(g.status, 0, sizeof(int), ctx->stream));
    // reflector rows are only partially written; clear so that masked lanes read zeros
    DMK_HIP(ctx, hipMemsetAsync(g.Vh, 0, nn * 16 * batch, ctx->stream));
    const size_t lds = (size_t)n * (16 + 16 + 32 + 16) + (3 * NW + 2) * 8 + 64 + 64;
    {
        FamScope fs(ctx, DMK_FAM_EIGH);
        if (lds > 48 * 1024) {
            const void *fn = n <= 64 ? reinterpret_cast<const void *>(eigh_kernel<1>)
                             : n <= 256 ? reinterpret_cast<const void *>(eigh_kernel<4>)
                                        : reinterpret_cast<const void *>(eigh_kernel<16>);
            DMK_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        if (n <= 64) hipLaunchKernelGGL(eigh_kernel<1>, dim3(batch), dim3(NT), lds, ctx->stream, g);
        else if (n <= 256) hipLaunchKernelGGL(eigh_kernel<4>, dim3(batch), dim3(NT), lds, ctx->stream, g);
        else hipLaunchKernelGGL(eigh_kernel<16>, dim3(batch), dim3(NT), lds, ctx->stream, g);
        DMK_CHECK_LAUNCH(ctx);
    }
    int status = 0;
    DMK_HIP(ctx, hipMemcpyAsync(&status, g.status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (status != 0) return dmk_fail(ctx, DMK_ERR_NOCONV, "eigh: QL iteration did not converge");
    if (getenv("DMK_EIGH_TIMING")) {
        long long tp[8];
        DMK_HIP(ctx, hipMemcpy(tp, g.status + 2, sizeof(tp), hipMemcpyDeviceToHost));
        fprintf(stderr, "[eigh n=%d batch=%d] matrix 0: init %.3f ms, tridiag %.3f ms, QL %.3f ms (scalar %.3f, apply %.3f; %lld rotations in "
                "%lld sweeps), sort+back %.3f ms\n", n, batch, tp[0] * 1e-5, tp[1] * 1e-5, tp[2] * 1e-5, tp[4] * 1e-5, tp[5] * 1e-5,
                tp[6], tp[7], tp[3] * 1e-5);
    }
    return DMK_OK;
}

}  // namespace

int launch_eigh_public(dmk_ctx *ctx, int n, int batch, const void *A, int a_real, const double *add, int add_group,
                       double *w, void *Vt, int v_real) {
    return launch_eigh(ctx, n, batch, A, a_real, add, add_group, w, Vt, v_real);
}

extern "C" {

int dmk_eigh_batched(dmk_ctx *ctx, int n, int batch, const void *A, const double *add, int add_group, double *w,
                     void *Vt) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n < 0 || batch < 0 || !A || !w || !Vt) return dmk_fail(ctx, DMK_ERR_INVALID, "eigh_batched: bad arguments");
    return launch_eigh(ctx, n, batch, A, 0, add, add_group, w, Vt, 0);
}

int dmk_eigh_batched_real(dmk_ctx *ctx, int n, int batch, const double *A, double *w, double *Vt) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n < 0 || batch < 0 || !A || !w || !Vt) return dmk_fail(ctx, DMK_ERR_INVALID, "eigh_batched_real: bad arguments");
    return launch_eigh(ctx, n, batch, A, 1, nullptr, 0, w, Vt, 1);
}

}  // extern "C"
