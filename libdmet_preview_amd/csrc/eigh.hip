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
//   2. eigenpairs of the real tridiagonal (d, e), parallel over the eigenvalues: block splitting,
//      bisection on the Sturm count, inverse iteration with a pivoted tridiagonal elimination,
//      Gram-Schmidt inside clusters of close eigenvalues (row i of Z^T = eigenvector i of T).
//   3. rank sort (ascending, stable) and back-transformation y = H_0 ... H_{n-2} z, four
//      eigenvectors per wave held in registers, reflector loads software-pipelined.
//
// Bound: latency / L2: algorithmic HBM traffic is
// 2*16*n^2 + 8*n bytes per matrix.
#include "common.h"

namespace {

constexpr int NT = 320;          // five waves per matrix
constexpr int NW = NT / 64;

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {   // conj(a) * b
    return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ double wave_sum(double v) { return dmk_wave_sum(v); }
__device__ __forceinline__ double wave_max(double v) {          // cold path only (residual check)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

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
    double *ws2;            // batch x 7 x n x n: lane-major scratch of the inverse iteration
    int *status;            // device flag: 2 = an eigenvector of the tridiagonal failed its residual check after the retries
    int *rank_out;          // batch x n: when set, phase 3 stops after the sort (w written, rank stored here) and the
                            // back-transformation is done by backtransform_kernel (eigh_tridiag.hip)
    int skip_tridiag;       // d, e, tau, Vh were produced by the CU-resident kernel (eigh_tridiag.hip): phases 0 and 1 are skipped
    int inject;             // fault injection (tests): every `inject`-th eigenvector is treated as failed and goes through the repair path
    const int *only_flagged;   // batch ints or null: when set, a workgroup whose flag is 0 returns at once -- this launch is then the
                               // REPAIR PASS behind tri_eigpairs_kernel (eigh_tripairs.hip), which flags the matrices it could not vouch for
};

// LDS carve (dynamic): vbuf[n] c128 | pbuf[n] c128 | cs[2][2n] f64 | red[3*NW + 2] f64 | 16 ints | dl[n] | el[n] | colpart[NW][n] c128
template <int R, int HR>
__global__ __launch_bounds__(NT) void eigh_kernel(const EighArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = g.n;
    double2 *vbuf = reinterpret_cast<double2 *>(smem);
    double2 *pbuf = vbuf + n;
    double *cs = reinterpret_cast<double *>(pbuf + n);
    double *red = cs + 4 * n;          // 3*NW + 2 doubles
    int *ired = reinterpret_cast<int *>(red + 3 * NW + 2);     // 16 ints
    double2 *colpart = reinterpret_cast<double2 *>((reinterpret_cast<uintptr_t>(reinterpret_cast<double *>(ired + 16) + 2 * n) + 15) &
                                                   ~static_cast<uintptr_t>(15));          // [NW][n] c128 (R <= 4 only), 16-B aligned

    const int b = blockIdx.x;
    if (g.only_flagged && g.only_flagged[b] == 0) return;      // uniform over the workgroup
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
    tphase[1] = tphase[0];
    if (!g.skip_tridiag) {
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
            // (d) p = tau * A22 v.  Up to n = 256 only the UPPER triangle of the trailing block is read (and kept up to
            // date): row i contributes a_ij v_j to y_i and conj(a_ij) v_i to y_j, j > i -- half the memory traffic of the
            // full-matrix product, and the traffic is what 432 concurrent 200 x 200 problems are bound by (PMC: 38.8 GB per
            // launch through HBM, the 691 MB of workspaces exceed the MALL).  The column contributions are accumulated in
            // registers per wave (lane <-> column) and combined through LDS.
            if constexpr (R <= 4) {
                double2 colacc[R];
#pragma unroll
                for (int c = 0; c < R; ++c) colacc[c] = make_double2(0.0, 0.0);
                // HR rows per wave and pass: all their loads are issued before the first reduction, otherwise every row pays
                // its own L2 / HBM round trip (the loop was latency bound: 35 us per Householder step at m = 200).  HR = 4
                // costs 136 VGPRs -- one workgroup per CU -- and is used when the batch fits one round anyway (64 x 136^2:
                // 9.3 -> 4.5 ms); HR = 2 keeps two workgroups per CU for large batches (432 x 200^2: 13.8 vs 20.8 ms)
                for (int ib = wave; ib < m; ib += HR * NW) {
                    double2 a[HR][R];
#pragma unroll
                    for (int u = 0; u < HR; ++u) {
                        const int i = ib + u * NW;
                        const double2 *row = W + (size_t)(k + 1 + (i < m ? i : 0)) * n + (k + 1);
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const int j = lane + 64 * c;
                            a[u][c] = (i < m && j >= i && j < m) ? row[j] : make_double2(0.0, 0.0);
                        }
                    }
                    double sr[HR], si[HR];
#pragma unroll
                    for (int u = 0; u < HR; ++u) {
                        const int i = ib + u * NW;
                        const double2 vi = vbuf[i < m ? i : 0];
                        sr[u] = 0.0;
                        si[u] = 0.0;
#pragma unroll
                        for (int c = 0; c < R; ++c) {
                            const int j = lane + 64 * c;
                            if (i < m && j >= i && j < m) {
                                const double2 v = vbuf[j], av = a[u][c];
                                sr[u] += av.x * v.x - av.y * v.y;
                                si[u] += av.x * v.y + av.y * v.x;
                                if (j > i) {                   // conj(a_ij) v_i
                                    colacc[c].x += av.x * vi.x + av.y * vi.y;
                                    colacc[c].y += av.x * vi.y - av.y * vi.x;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < HR; ++u) {
                        const int i = ib + u * NW;
                        const double tr = wave_sum(sr[u]), ti = wave_sum(si[u]);
                        if (lane == 0 && i < m) pbuf[i] = make_double2(tr, ti);
                    }
                }
#pragma unroll
                for (int c = 0; c < R; ++c) {
                    const int j = lane + 64 * c;
                    if (j < m) colpart[(size_t)wave * n + j] = colacc[c];
                }
                __syncthreads();
                for (int t = tid; t < m; t += NT) {
                    double2 y = pbuf[t];
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        const double2 cpart = colpart[(size_t)w * n + t];
                        y.x += cpart.x;
                        y.y += cpart.y;
                    }
                    pbuf[t] = cmul(tk, y);
                }
            } else {
                // full-matrix product for the large-n variant (one row per wave at a time; lanes along the row)
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
            // two rows in flight per wave; up to n = 256 only the upper triangle (j >= i) is maintained
            for (int i0 = wave; i0 < m; i0 += 2 * NW) {
                const int i1 = i0 + NW;
                const bool two = i1 < m;
                const double2 vi0 = vbuf[i0], wi0 = pbuf[i0];
                const double2 vi1 = two ? vbuf[i1] : make_double2(0.0, 0.0), wi1 = two ? pbuf[i1] : make_double2(0.0, 0.0);
                double2 *r0 = W + (size_t)(k + 1 + i0) * n + (k + 1);
                double2 *r1 = W + (size_t)(k + 1 + (two ? i1 : i0)) * n + (k + 1);
                const int jstart = (R <= 4) ? (i0 & ~63) + lane : lane;      // first 64-column chunk that reaches the diagonal
                for (int j = jstart; j < m; j += 64) {
                    const double2 vj = vbuf[j], wj = pbuf[j];
                    if (R > 4 || j >= i0) {
                        double2 a0 = r0[j];
                        a0.x -= vi0.x * wj.x + vi0.y * wj.y + wi0.x * vj.x + wi0.y * vj.y;
                        a0.y -= vi0.y * wj.x - vi0.x * wj.y + wi0.y * vj.x - wi0.x * vj.y;
                        if (j == i0) a0.y = 0.0;
                        r0[j] = a0;
                    }
                    if (two && (R > 4 || j >= i1)) {
                        double2 a1 = r1[j];
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
    }   // !skip_tridiag

    tphase[2] = wall_clock64();
    // ---- phase 2: eigenpairs of the real tridiagonal T = (d, e) by bisection + inverse iteration ----------------------
    // e[i] couples i and i+1.  Round 1 ran implicit-shift QL here: ~43 000 DEPENDENT rotations per 200 x 200 matrix, each
    // applied to two rows of Z through L2 -- 9.2 ms of the 26 ms launch and the same latency chain for every matrix.  The
    // work that replaces it is parallel over the EIGENVALUES (one lane each):
    //   (a) T is cut into unreduced blocks at negligible couplings (1 x 1 blocks give their eigenpair exactly, and
    //       eigenvectors of different blocks are orthogonal by support -- degeneracies ACROSS blocks cost nothing);
    //   (b) the k-th eigenvalue of a block by bisection on the Sturm count (negative pivots of the LDL^T recurrence);
    //   (c) its eigenvector by inverse iteration: Gaussian elimination with partial pivoting of the shifted tridiagonal
    //       block (kept as three upper diagonals, the multipliers and the swap flags in a lane-major scratch), a hashed
    //       pseudo-random start, three solves with rescaling, tiny pivots replaced by eps |T|;
    //   (d) eigenvalues of one block closer than 1e-3 |T| form a cluster whose vectors are re-orthogonalised
    //       (modified Gram-Schmidt, twice) by one wave per cluster -- independently iterated vectors of a (near-)degenerate
    //       group span the right invariant subspace (different starts) but are not orthogonal to each other.
    {
        double *dl = reinterpret_cast<double *>(ired + 16);   // [n] diagonal
        double *el = dl + n;                                  // [n] couplings
        int *bs = reinterpret_cast<int *>(cs);                // [n] first index of the block of i
        int *be = bs + n;                                     // [n] one past its last index
        double *lam = cs + n;                                 // [n] eigenvalue of lane j (ascending inside a block)
        double *bnorm = cs + 2 * n;                           // [n] 1-norm of the block of i
        double *ws = g.ws2 + (size_t)b * 7 * nn;              // seven [row][lane] arrays
        const double eps = 2.220446049250313e-16;
        double *e2 = reinterpret_cast<double *>(vbuf);        // [n] squared couplings (the reflector buffer is idle now)
        for (int t = tid; t < n; t += NT) {
            dl[t] = d[t];
            el[t] = (t + 1 < n) ? e[t] : 0.0;
        }
        for (size_t idx = tid; idx < nn; idx += NT) Zt[idx] = 0.0;
        __syncthreads();
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
        for (int t = tid; t < n; t += NT) e2[t] = el[t] * el[t];
        __syncthreads();
        // ---- inverse iteration, one lane per eigenvector; the pieces are shared by the first pass and by the repair path ----
        // (c1) elimination with partial pivoting of T - lm I on the rows of j's block (three upper diagonals, multipliers,
        //      swap flags; tiny pivots replaced by eps |T|)
        auto inv_factor = [&](const int j, const double lm) {
            const int s0 = bs[j], t0 = be[j];
            double *U0 = ws + j, *U1 = U0 + nn, *U2 = U1 + nn, *Lm = U2 + nn, *Pv = Lm + nn;
            const double pert = fmax(eps * bnorm[j], 1e-300);
            double cd = dl[s0] - lm, cu = el[s0];
            for (int i = s0; i + 1 < t0; ++i) {
                const double sub = el[i], nd = dl[i + 1] - lm, nu = (i + 2 < t0) ? el[i + 1] : 0.0;
                const size_t o = (size_t)i * n;
                if (fabs(cd) >= fabs(sub)) {
                    if (fabs(cd) < pert) cd = cd >= 0.0 ? pert : -pert;
                    const double mlt = sub / cd;
                    U0[o] = cd; U1[o] = cu; U2[o] = 0.0; Lm[o] = mlt; Pv[o] = 0.0;
                    cd = nd - mlt * cu;
                    cu = nu;
                } else {
                    const double mlt = cd / sub;
                    U0[o] = sub; U1[o] = nd; U2[o] = nu; Lm[o] = mlt; Pv[o] = 1.0;
                    cd = cu - mlt * nd;
                    cu = -mlt * nu;
                }
            }
            if (fabs(cd) < pert) cd = cd >= 0.0 ? pert : -pert;
            U0[(size_t)(t0 - 1) * n] = cd;
        };
        // (c2) start vector: hashed uniform numbers in (-1, 1), different for every matrix, eigenvalue, row and attempt
        auto inv_seed = [&](const int j, const unsigned long long salt) {
            const int s0 = bs[j], t0 = be[j];
            double *x = ws + j + 5 * nn;
            unsigned long long h = ((unsigned long long)b * 0x9E3779B97F4A7C15ull) ^ ((unsigned long long)(j + 1) * 0xC2B2AE3D27D4EB4Full) ^
                                   (salt * 0xD6E8FEB86659FD93ull);
            for (int i = s0; i < t0; ++i) {
                h += 0x9E3779B97F4A7C15ull;
                unsigned long long z = h;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z ^= z >> 31;
                x[(size_t)i * n] = (double)(long long)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
            }
        };
        auto inv_xmax = [&](const int j) {
            const int s0 = bs[j], t0 = be[j];
            const double *x = ws + j + 5 * nn;
            double xm = 0.0;
            for (int i = s0; i < t0; ++i) xm = fmax(xm, fabs(x[(size_t)i * n]));
            return xm;
        };
        // (c3) one solve (T - lm I) x_new = x / xm.  It reads one array and writes another (forward x -> y, backward y -> x), so
        //      the loads of a pass do not alias its stores and the compiler may run several rows ahead of the recurrence; the
        //      scale of the next right-hand side (max |x|) is gathered during the backward pass and returned.
        auto inv_solve = [&](const int j, const double xm_in) {
            const int s0 = bs[j], t0 = be[j];
            const double *U0 = ws + j, *U1 = U0 + nn, *U2 = U1 + nn, *Lm = U2 + nn, *Pv = Lm + nn;
            double *x = ws + j + 5 * nn;
            double *__restrict__ yv = x + nn;                  // second vector, lane-major like the others
            const double sc = xm_in > 0.0 ? 1.0 / xm_in : 1.0;
            double cur = x[(size_t)s0 * n] * sc;
#pragma unroll 4
            for (int i = s0; i + 1 < t0; ++i) {                // forward: row swaps and multipliers
                const size_t o = (size_t)i * n;
                double nxt = x[o + n] * sc;
                const double pv = Pv[o], ml = Lm[o];
                if (pv != 0.0) { const double tsw = cur; cur = nxt; nxt = tsw; }
                yv[o] = cur;
                cur = nxt - ml * cur;
            }
            yv[(size_t)(t0 - 1) * n] = cur;
            double x1 = 0.0, x2 = 0.0, xm = 0.0;
#pragma unroll 4
            for (int i = t0 - 1; i >= s0; --i) {               // backward: three upper diagonals
                const size_t o = (size_t)i * n;
                const double u0 = U0[o], u1 = U1[o], u2 = U2[o];
                double r = yv[o];
                if (i + 1 < t0) r -= u1 * x1 + u2 * x2;
                r /= u0;
                x[o] = r;
                xm = fmax(xm, fabs(r));
                x2 = x1;
                x1 = r;
            }
            return xm;
        };
        // (c4) normalised copy into row j of Z^T
        auto inv_store = [&](const int j, const double xm) {
            const int s0 = bs[j], t0 = be[j];
            const double *x = ws + j + 5 * nn;
            double nr = 0.0;
            const double sc = xm > 0.0 ? 1.0 / xm : 1.0;
            for (int i = s0; i < t0; ++i) { const double v = x[(size_t)i * n] * sc; nr += v * v; }
            const double inv = sc / sqrt(nr);
            for (int i = s0; i < t0; ++i) Zt[(size_t)j * n + i] = x[(size_t)i * n] * inv;
        };
        for (int j = tid; j < n; j += NT) {
            const int s0 = bs[j], t0 = be[j], m = t0 - s0, kk = j - s0;
            if (m == 1) {
                lam[j] = dl[s0];
                Zt[(size_t)j * n + s0] = 1.0;
                continue;
            }
            const double tn = bnorm[j];
            // ---- (b) bisection: Gershgorin interval, count(x) = number of eigenvalues below x
            double emax2 = 0.0, lo = dl[s0], hi = dl[s0];
            for (int i = s0; i < t0; ++i) {
                const double rad = (i > s0 ? fabs(el[i - 1]) : 0.0) + (i + 1 < t0 ? fabs(el[i]) : 0.0);
                lo = fmin(lo, dl[i] - rad);
                hi = fmax(hi, dl[i] + rad);
                if (i + 1 < t0) emax2 = fmax(emax2, el[i] * el[i]);
            }
            const double pivmin = 2.2250738585072014e-308 * fmax(1.0, emax2);
            lo -= 2.0 * eps * tn * m + 2.0 * pivmin;
            hi += 2.0 * eps * tn * m + 2.0 * pivmin;
            for (int it = 0; it < 200; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (!(mid > lo && mid < hi)) break;
                int cnt = 0;
                double q = dl[s0] - mid;
                if (fabs(q) < pivmin) q = -pivmin;
                cnt += q < 0.0 ? 1 : 0;
#pragma unroll 4
                for (int i = s0 + 1; i < t0; ++i) {
                    // 1 / q from v_rcp_f64 and two Newton steps (the count only needs the SIGN of the pivots to be right
                    // up to perturbations of a few ulp of |T|, which is the accuracy bisection delivers anyway)
                    double r = __builtin_amdgcn_rcp(q);
                    r = r * (2.0 - q * r);
                    r = r * (2.0 - q * r);
                    q = (dl[i] - mid) - e2[i - 1] * r;
                    if (fabs(q) < pivmin) q = -pivmin;
                    cnt += q < 0.0 ? 1 : 0;
                }
                if (cnt > kk) hi = mid; else lo = mid;
                if (hi - lo <= eps * (fabs(lo) + fabs(hi)) + 2.0 * pivmin) break;
            }
            const double lm = 0.5 * (lo + hi);
            lam[j] = lm;
            inv_factor(j, lm);
            inv_seed(j, 0ull);
            double xm = inv_xmax(j);
            // two solves: the shift is an eigenvalue to rounding, so ONE step already leaves the eigenvector with relative error
            // ~ eps |T| / gap and the second is the safety margin dstein's growth test usually stops at; every vector goes through
            // the acceptance test of phase (e) below, which repairs the rare one that needed more (the third solve cost a quarter of
            // the 5.9 GB this phase moves through its lane-major scratch)
            for (int iter = 0; iter < 2; ++iter) xm = inv_solve(j, xm);
            inv_store(j, xm);
        }
        __syncthreads();
        // ---- (d) clusters: one wave each, members in ascending order
        {
            int cluster = -1;
            for (int j = 0; j < n; ++j) {
                const bool first = (j == bs[j]) || (lam[j] - lam[j - 1] > 1e-3 * bnorm[j]);
                if (!first) continue;
                ++cluster;
                if (cluster % NW != wave) continue;
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
                            dot = wave_sum(dot);
                            for (int i = s0 + lane; i < t0; i += 64) zq[i] -= dot * zp[i];
                        }
                    double nr = 0.0;
                    for (int i = s0 + lane; i < t0; i += 64) nr += zq[i] * zq[i];
                    nr = wave_sum(nr);
                    const double inv = nr > 0.0 ? 1.0 / sqrt(nr) : 0.0;
                    for (int i = s0 + lane; i < t0; i += 64) zq[i] *= inv;
                }
            }
        }
        __syncthreads();
        // ---- (e) acceptance test and repair.  Three fixed solves from a random start are enough for every spectrum met so
        // far, but nothing above PROVES an eigenvector: an unlucky start, near-duplicate shifts inside one unreduced block or
        // a vector annihilated by the cluster Gram-Schmidt would flow silently into rho, the bath and the vcor fit.  Every
        // vector is therefore checked -- |T z - lam z|_inf <= 64 n eps |T| and |z| = 1 (NaN fails both) -- one wave per vector.
        // A failed vector is rebuilt by wave 0 the way LAPACK's dstein does it: perturbed shift, fresh start, and the
        // re-orthogonalisation against the accepted members of its cluster INSIDE the iteration; if that fails as well the
        // launch reports DMK_ERR_NOCONV instead of returning a wrong basis.
        int *bad = reinterpret_cast<int *>(pbuf);             // [n] (the Householder buffer is idle in this phase)
        const double rtol = 64.0 * n * eps;
        auto residual_ok = [&](const int j) {                 // wave-cooperative; every lane returns the verdict
            const int s0 = bs[j], t0 = be[j];
            if (t0 - s0 == 1) return true;
            const double *z = Zt + (size_t)j * n;
            const double lj = lam[j];
            double r = 0.0, zn = 0.0;
            for (int i = s0 + lane; i < t0; i += 64) {
                const double zi = z[i];
                double t = (dl[i] - lj) * zi;
                if (i > s0) t += el[i - 1] * z[i - 1];
                if (i + 1 < t0) t += el[i] * z[i + 1];
                r = fmax(r, fabs(t));
                if (!(fabs(t) <= 1.7e308)) r = 1.7e308;        // NaN / Inf: fmax would drop a NaN
                zn += zi * zi;
            }
            r = wave_max(r);
            zn = wave_sum(zn);
            return r <= rtol * fmax(bnorm[j], 1e-300) && fabs(zn - 1.0) <= 1e-8;
        };
        for (int j = wave; j < n; j += NW) {
            bool ok = residual_ok(j);
            if (g.inject > 0 && (j % g.inject) == g.inject - 1) ok = false;
            if (lane == 0) bad[j] = ok ? 0 : 1;
        }
        __syncthreads();
        if (wave == 0) {
            for (int j = 0; j < n; ++j) {
                if (!bad[j]) continue;
                const int s0 = bs[j], t0 = be[j];
                const double tn = bnorm[j], lj = lam[j];
                double *zj = Zt + (size_t)j * n;
                double *x = ws + j + 5 * nn;
                bool fixed = false;
                for (int attempt = 1; attempt <= 3 && !fixed; ++attempt) {
                    double xm = 0.0;
                    if (lane == 0) {
                        inv_factor(j, lj + ((attempt & 1) ? 4.0 : -4.0) * attempt * eps * tn);
                        inv_seed(j, (unsigned long long)attempt);
                        xm = inv_xmax(j);
                    }
                    for (int iter = 0; iter < 5 && !fixed; ++iter) {
                        if (lane == 0) {
                            xm = inv_solve(j, xm);
                            inv_store(j, xm);
                        }
                        __threadfence_block();
                        // orthogonalise against the accepted vectors of j's cluster (twice), renormalise, feed back
                        for (int pass = 0; pass < 2; ++pass)
                            for (int p = s0; p < t0; ++p) {
                                if (p == j || bad[p] || fabs(lam[p] - lj) > 1e-3 * tn) continue;
                                const double *zp = Zt + (size_t)p * n;
                                double dot = 0.0;
                                for (int i = s0 + lane; i < t0; i += 64) dot += zp[i] * zj[i];
                                dot = wave_sum(dot);
                                for (int i = s0 + lane; i < t0; i += 64) zj[i] -= dot * zp[i];
                            }
                        double nr = 0.0;
                        for (int i = s0 + lane; i < t0; i += 64) nr += zj[i] * zj[i];
                        nr = wave_sum(nr);
                        const double inv = nr > 0.0 ? 1.0 / sqrt(nr) : 0.0;
                        for (int i = s0 + lane; i < t0; i += 64) {
                            const double v = zj[i] * inv;
                            zj[i] = v;
                            x[(size_t)i * n] = v;
                        }
                        __threadfence_block();
                        xm = 1.0;                                   // the fed-back vector has unit 2-norm: no rescaling needed
                        fixed = iter >= 1 && residual_ok(j);
                    }
                }
                if (lane == 0) {
                    if (fixed) bad[j] = 0;
                    else g.status[0] = 2;
                }
                __threadfence_block();
            }
        }
        __syncthreads();
        for (int t = tid; t < n; t += NT) d[t] = lam[t];
        if (b == 0 && tid == 0) {
            long long *tp = reinterpret_cast<long long *>(g.status + 2);
            tp[4] = 0; tp[5] = 0; tp[6] = 0; tp[7] = 0;
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
        if (g.rank_out) g.rank_out[(size_t)b * n + j] = rk;
    }
    __syncthreads();
    if (g.rank_out) {
        if (b == 0 && tid == 0) {
            long long *tp = reinterpret_cast<long long *>(g.status + 2);
            tphase[4] = wall_clock64();
            for (int q = 0; q < 4; ++q) tp[q] = tphase[q + 1] - tphase[q];
        }
        return;
    }

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

}  // namespace
int launch_tridiag_resident(dmk_ctx *ctx, int n, int batch, const void *A, const double *add, int add_group, void *Vh, void *tau,
                            double *d, double *e);
int launch_backtransform(dmk_ctx *ctx, int n, int batch, const double *Zt, const void *Vh, const void *tau, const int *rank, void *Vt,
                         void *Tws);
int launch_tri_eigpairs(dmk_ctx *ctx, int n, int batch, const double *d, const double *e, double *Zt, double *w, int *rank_out,
                        int *flags, int inject);
namespace {

int launch_eigh(dmk_ctx *ctx, int n, int batch, const void *A, int a_real, const double *add, int add_group, double *w,
                void *Vt, int v_real) {
    if (n <= 0 || batch <= 0) return DMK_OK;
    // one workgroup per matrix: the LDS carve (80 n bytes) and the per-lane column slices (64 R columns) bound n
    if (n > 2000) return dmk_fail(ctx, DMK_ERR_INVALID, "eigh: n = %d exceeds the supported maximum of 2000", n);
    const size_t nn = (size_t)n * n;
    const size_t per = nn * (16 + 16 + 8 + 56) + (size_t)n * (8 + 8 + 16 + 8);
    const size_t total = per * batch + 256 + (size_t)batch * sizeof(int) + 256;
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, total, &ws);
    if (rc) return rc;
    char *p = reinterpret_cast<char *>(ws);
    EighArgs g;
    g.n = n; g.batch = batch; g.A = A; g.a_real = a_real; g.add = add; g.add_group = add_group;
    g.w = w; g.Vt = Vt; g.v_real = v_real;
    {   // fault injection for the repair path of the tridiagonal eigenvectors (tests/test_gpu_parity.py); never set in production
        const char *inj = getenv("DMK_EIGH_INJECT");
        g.inject = inj ? atoi(inj) : 0;
    }
    g.status = reinterpret_cast<int *>(p); p += 256;
    g.W = reinterpret_cast<double2 *>(p); p += nn * 16 * batch;
    g.Vh = reinterpret_cast<double2 *>(p); p += nn * 16 * batch;
    g.tau = reinterpret_cast<double2 *>(p); p += (size_t)n * 16 * batch;
    g.Zt = reinterpret_cast<double *>(p); p += nn * 8 * batch;
    g.d = reinterpret_cast<double *>(p); p += (size_t)n * 8 * batch;
    g.e = reinterpret_cast<double *>(p); p += (size_t)n * 8 * batch;
    g.ws2 = reinterpret_cast<double *>(p); p += nn * 56 * batch;
    int *rank_ws = reinterpret_cast<int *>(p); p += (size_t)n * 8 * batch;
    int *flag_ws = reinterpret_cast<int *>((reinterpret_cast<uintptr_t>(p) + 255) & ~static_cast<uintptr_t>(255));
    DMK_HIP(ctx, hipMemsetAsync(g.status, 0, sizeof(int), ctx->stream));
    g.skip_tridiag = 0;
    g.rank_out = nullptr;
    g.only_flagged = nullptr;

    const size_t lds = (size_t)n * (16 + 16 + 32 + 16) + (3 * NW + 2) * 8 + 64 + 64 + (n <= 256 ? (size_t)NW * n * 16 + 16 : 0);
    {
        FamScope fs(ctx, DMK_FAM_EIGH);
        // complex matrices of the north-star size: tridiagonalisation with the matrix resident in LDS + registers (one 512-thread
        // workgroup per matrix, HBM read once); this kernel then only does the tridiagonal eigenpairs and the back-transformation
        static const bool resident_on = !(getenv("DMK_EIGH_RESIDENT") && atoi(getenv("DMK_EIGH_RESIDENT")) == 0);
        if (resident_on && !a_real && n > 64) {
            const int rt = launch_tridiag_resident(ctx, n, batch, A, add, add_group, g.Vh, g.tau, g.d, g.e);
            if (rt < 0) return rt;
            g.skip_tridiag = rt;
            if (rt && !v_real) g.rank_out = rank_ws;         // back-transformation by the lane-per-eigenvector kernel
            if (g.rank_out) {
                // eigenpairs of the tridiagonal forms with every vector in LDS (eigh_tripairs.hip); the launch of eigh_kernel
                // below then only repairs the matrices it flagged (normally none: 432 workgroups that return at once)
                const int rp = launch_tri_eigpairs(ctx, n, batch, g.d, g.e, g.Zt, w, rank_ws, flag_ws, g.inject);
                if (rp < 0) return rp;
                if (rp) g.only_flagged = flag_ws;
            }
        }
        // single-kernel path: its reflector rows are only partially written; clear so that masked lanes read zeros (the resident
        // kernels write complete rows)
        if (!g.skip_tridiag) DMK_HIP(ctx, hipMemsetAsync(g.Vh, 0, nn * 16 * batch, ctx->stream));
        // HR (rows in flight per wave in the Householder matrix-vector product): see the kernel
        const bool few = batch <= 256;
        const void *fn = n <= 64 ? (few ? reinterpret_cast<const void *>(eigh_kernel<1, 4>) : reinterpret_cast<const void *>(eigh_kernel<1, 2>))
                         : n <= 256 ? (few ? reinterpret_cast<const void *>(eigh_kernel<4, 4>) : reinterpret_cast<const void *>(eigh_kernel<4, 2>))
                         : n <= 1024 ? reinterpret_cast<const void *>(eigh_kernel<16, 2>)
                                     : reinterpret_cast<const void *>(eigh_kernel<32, 2>);
        if (lds > 48 * 1024) DMK_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (n <= 64) {
            if (few) hipLaunchKernelGGL((eigh_kernel<1, 4>), dim3(batch), dim3(NT), lds, ctx->stream, g);
            else hipLaunchKernelGGL((eigh_kernel<1, 2>), dim3(batch), dim3(NT), lds, ctx->stream, g);
        } else if (n <= 256) {
            if (few) hipLaunchKernelGGL((eigh_kernel<4, 4>), dim3(batch), dim3(NT), lds, ctx->stream, g);
            else hipLaunchKernelGGL((eigh_kernel<4, 2>), dim3(batch), dim3(NT), lds, ctx->stream, g);
        } else if (n <= 1024) {
            hipLaunchKernelGGL((eigh_kernel<16, 2>), dim3(batch), dim3(NT), lds, ctx->stream, g);
        } else {
            hipLaunchKernelGGL((eigh_kernel<32, 2>), dim3(batch), dim3(NT), lds, ctx->stream, g);
        }
        DMK_CHECK_LAUNCH(ctx);
        if (g.rank_out) {
            const int rb = launch_backtransform(ctx, n, batch, g.Zt, g.Vh, g.tau, g.rank_out, Vt, g.W);   // W is idle on this path: T factors
            if (rb < 0) return rb;
        }
    }
    int status = 0;
    DMK_HIP(ctx, hipMemcpyAsync(&status, g.status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (status != 0)
        return dmk_fail(ctx, DMK_ERR_NOCONV, "eigh: an eigenvector of the tridiagonal form failed the residual test |T z - lam z| <= 64 n eps |T| "
                                            "after three repair attempts (n = %d, batch = %d; NaN / Inf in the input?)", n, batch);
    if (getenv("DMK_EIGH_TIMING")) {
        long long tp[8];
        DMK_HIP(ctx, hipMemcpy(tp, g.status + 2, sizeof(tp), hipMemcpyDeviceToHost));
        fprintf(stderr, "[eigh n=%d batch=%d] matrix 0: init %.3f ms, tridiag %.3f ms, bisection + inverse iteration %.3f ms, sort+back %.3f ms\n",
                n, batch, tp[0] * 1e-5, tp[1] * 1e-5, tp[2] * 1e-5, tp[3] * 1e-5);
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
