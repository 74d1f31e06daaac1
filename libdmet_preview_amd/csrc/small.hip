// Small model lattices (BASELINE configs 1 - 2: Hubbard cells of a few sites on a k-mesh of a few dozen points): the mean-field
// step and the Schmidt bath as ONE launch each.
//
//   small_meanfield_kernel   F_k + vcor -> eigenpairs of every (spin, k) block -> occupations / mu -> rho_k -> k -> R fold
//                            reference: routine/mfd.py:33-108 (Diag*), :887-957 (assignocc), :352-360 (density + FFTtoT),
//                            system/fourier.py:168-177 (k2R)
//   small_bath_kernel        env x imp block of the stripe -> thin SVD -> bath count -> Loewdin -> embedding basis
//                            reference: routine/slater.py:117-220 (_get_emb_basis_svd), lo/lowdin.py:83-101
//
// The general path runs these stages as ~10 dependent launches tuned for C4 / C5 sizes (batched Householder + bisection eigensolver,
// TSQR + one-sided Jacobi SVD, ...): at 36 matrices of 4 x 4 every one of them is pure latency -- 0.21 ms of kernels and as much
// again in launch gaps and stage synchronisations, where LAPACK on the host needs 0.23 ms (C1) / 2.8 ms (C2) for the whole step.
// Here ONE workgroup does a stage: the matrices live in LDS (element-major, matrix index fastest: conflict-free), one thread per
// (spin, k) block runs a cyclic complex Jacobi eigensolver, the occupation code is the same device function the stand-alone
// kernels call (occ_body.h: bit-identical mu), the fold is a direct sum against exact per-axis twiddles, the bath is Householder QR
// of the tall block (one fused reduction per column) + one-sided Jacobi on its nb x nb triangle -- no Gram-matrix shortcut, the
// singular values keep the relative accuracy the sigma >= tol_bath count needs.
// Limits (checked by the launchers, which return "not handled" so that the caller takes the general path): n <= 8 orbitals per
// cell, nb <= 8 bath columns, spin * nk <= 128 blocks (eight lanes per block), LDS carve-outs below.
#include "common.h"
#include <cmath>
#include <algorithm>

#include "occ_body.h"

namespace {

constexpr int SM_NT = OCC_NT;                  // the occupation bodies expect a workgroup of OCC_NT threads
constexpr int SM_MAXN = 8;

struct SmallMF {
    int n, nmat, nk, spin, mstride;
    int mesh[3];
    const double2 *F;                          // [nmat][n][n] complex, row-major (lower triangle is used, like numpy's eigh)
    const double *add;                         // optional real shift, [nmat / add_group][n][n]
    int add_group;
    OccArgs occ;                               // ew / occ / out point at the outputs (ew: [nmat][n], out: info[0..5))
    int zero_t;
    double2 *Vt;                               // [nmat][n][n]: ROW m = eigenvector m
    double2 *rho_k;                            // [nmat][n][n]
    double *rho_R;                             // [spin][nk][n*n]
    double *info;                              // [0..5) occupation info, [5] max |Im| of the fold, [6] Jacobi not converged, [7] sweeps
};

// hardware estimates + two Newton steps (v_rcp_f64 / v_rsq_f64 are good to ~2^-26; two steps square that twice: <= 1 ulp).  The
// libm sqrt / division of the rotation parameters were two thirds of the eigensolver phase at n = 4 (seven of them in a
// dependent chain per rotation).
__device__ __forceinline__ double sm_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
__device__ __forceinline__ double sm_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

__device__ __forceinline__ double block_max_f64(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < SM_NT / 64; ++w) t = fmax(t, sh[w]);
    return t;
}

__global__ __launch_bounds__(SM_NT) void small_meanfield_kernel(const SmallMF g) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];            // (16: the double2 views below use 128-bit LDS accesses)  Hr | Hi | Vr | Vi, each [n*n][mstride]; twiddles [3][128] complex; ranks [nmat][8] int
    __shared__ double shd[SM_NT / 64];
    const int n = g.n, nn = n * n, nmat = g.nmat, ms = g.mstride, tid = threadIdx.x;
    double *Hr = dyn, *Hi = Hr + (size_t)nn * ms, *Vr = Hi + (size_t)nn * ms, *Vi = Vr + (size_t)nn * ms;
    double2 *tw = reinterpret_cast<double2 *>(Vi + (size_t)nn * ms);
    // exact per-axis twiddles e^{+2 pi i a / n_d}: quarter turns exact, the rest from sincospi of the reduced fraction
    for (int t = tid; t < 3 * 128; t += SM_NT) {
        const int d = t / 128, a = t % 128, nd = g.mesh[d];
        double s = 0.0, c = 1.0;
        if (a < nd) sincospi(2.0 * (double)a / (double)nd, &s, &c);
        if (a < nd && (4 * a) % nd == 0) {         // multiples of a quarter turn
            const int qt = (4 * a) / nd;
            c = qt == 0 ? 1.0 : (qt == 2 ? -1.0 : 0.0);
            s = qt == 1 ? 1.0 : (qt == 3 ? -1.0 : 0.0);
        }
        tw[t] = double2{c, s};
    }
    int notconv = 0, nsweep = 0;
    const long long clk0 = wall_clock64();       // phase clocks of thread 0 (100 MHz) -> info[8..12], for the builder's profile
    long long clk1 = clk0, clk2 = clk0, clk3 = clk0, clk4 = clk0;
    // EIGHT lanes per matrix (lane r <-> row r in the column phase of a rotation, column r in the row phase): the loads and stores
    // of a phase are independent across the lanes of a matrix and issue together; lanes of one matrix sit in one wave, so the LDS
    // pipe keeps a phase's stores ahead of the next phase's loads (one thread per matrix walked ~160 dependent LDS accesses per
    // rotation: 110 us for 36 matrices of 4 x 4).
    const int m = tid >> 3, r = tid & 7;
    const bool mat_ok = m < nmat, act = mat_ok && r < n;
#define HR(i, j) Hr[(size_t)((i) * n + (j)) * ms + m]
#define HI(i, j) Hi[(size_t)((i) * n + (j)) * ms + m]
#define VR(i, j) Vr[(size_t)((i) * n + (j)) * ms + m]
#define VI(i, j) Vi[(size_t)((i) * n + (j)) * ms + m]
    int *rk = reinterpret_cast<int *>(tw + 3 * 128);          // [nmat][8]: sorted position of eigenpair l
    if (mat_ok) {
        const double2 *F = g.F + (size_t)m * nn;
        const double *ad = g.add ? g.add + (size_t)(m / g.add_group) * nn : nullptr;
        if (act) {
            const int i = r;
            for (int j = 0; j <= i; ++j) {                    // lower triangle of row i and its mirror image
                const double2 f = F[i * n + j];
                const double re = f.x + (ad ? ad[i * n + j] : 0.0), im = (i == j) ? 0.0 : f.y;
                HR(i, j) = re; HI(i, j) = im;
                HR(j, i) = re; HI(j, i) = -im;
            }
            for (int j = 0; j < n; ++j) { VR(i, j) = (i == j) ? 1.0 : 0.0; VI(i, j) = 0.0; }
        }
        // max over the 8 lanes of a matrix
        auto gmax = [&](double v) {
            v = fmax(v, __shfl_xor(v, 1, 64));
            v = fmax(v, __shfl_xor(v, 2, 64));
            v = fmax(v, __shfl_xor(v, 4, 64));
            return v;
        };
        double scale = 0.0;
        if (act)
            for (int j = 0; j < n; ++j) scale = fmax(scale, fmax(fabs(HR(r, j)), fabs(HI(r, j))));
        scale = gmax(scale);
        int sweep = 0;
        for (; sweep < 40; ++sweep) {
            double off = 0.0;
            if (act)
                for (int q = r + 1; q < n; ++q) off = fmax(off, fmax(fabs(HR(r, q)), fabs(HI(r, q))));
            off = gmax(off);
            if (!(off > 5.6e-17 * scale)) break;           // 2^-54: half an ulp of the largest element
            for (int p = 0; p < n; ++p)
                for (int q = p + 1; q < n; ++q) {
                    // cyclic Jacobi on the Hermitian matrix: J = D R with D = diag(.., 1 (p), .., e^{-i phi} (q), ..), phi = arg h_pq,
                    // R the real rotation that annihilates the (then real) pq element; H <- J^H H J, V <- V J
                    const double ar = HR(p, q), ai = HI(p, q);                 // broadcast reads: every lane forms the same J
                    const double mod2 = ar * ar + ai * ai;
                    if (!(mod2 > 1.0e-290) || !(mod2 > 1.0e-34 * scale * scale)) continue;
                    const double imod = sm_rsqrt(mod2);                      // 1 / |h_pq|
                    const double er = ar * imod, ei = ai * imod;             // e^{i phi}
                    const double tau = 0.5 * (HR(q, q) - HR(p, p)) * imod;
                    const double rt = 1.0 + tau * tau;
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) * sm_rcp(fabs(tau) + rt * sm_rsqrt(rt));
                    const double c = sm_rsqrt(1.0 + t * t), sn = t * c;
                    // J_pp = c, J_pq = s, J_qp = -s e^{-i phi}, J_qq = c e^{-i phi}
                    const double jqp_r = -sn * er, jqp_i = sn * ei, jqq_r = c * er, jqq_i = -c * ei;
                    if (act) {                                               // columns: X[r, p], X[r, q] <- (X J)[r, .]  (X = H and V)
                        const int k = r;
                        {
                            const double xpr = HR(k, p), xpi = HI(k, p), xqr = HR(k, q), xqi = HI(k, q);
                            HR(k, p) = c * xpr + (xqr * jqp_r - xqi * jqp_i);
                            HI(k, p) = c * xpi + (xqr * jqp_i + xqi * jqp_r);
                            HR(k, q) = sn * xpr + (xqr * jqq_r - xqi * jqq_i);
                            HI(k, q) = sn * xpi + (xqr * jqq_i + xqi * jqq_r);
                        }
                        {
                            const double xpr = VR(k, p), xpi = VI(k, p), xqr = VR(k, q), xqi = VI(k, q);
                            VR(k, p) = c * xpr + (xqr * jqp_r - xqi * jqp_i);
                            VI(k, p) = c * xpi + (xqr * jqp_i + xqi * jqp_r);
                            VR(k, q) = sn * xpr + (xqr * jqq_r - xqi * jqq_i);
                            VI(k, q) = sn * xpi + (xqr * jqq_i + xqi * jqq_r);
                        }
                    }
                    if (act) {                                               // rows: H[p, r], H[q, r] <- (J^H H)[., r]
                        const int k = r;
                        const double xpr = HR(p, k), xpi = HI(p, k), xqr = HR(q, k), xqi = HI(q, k);
                        // conj(J_pp) = c, conj(J_qp) = (jqp_r, -jqp_i); conj(J_pq) = s, conj(J_qq) = (jqq_r, -jqq_i)
                        double npr = c * xpr + (xqr * jqp_r + xqi * jqp_i);
                        double npi = c * xpi + (xqi * jqp_r - xqr * jqp_i);
                        double nqr = sn * xpr + (xqr * jqq_r + xqi * jqq_i);
                        double nqi = sn * xpi + (xqi * jqq_r - xqr * jqq_i);
                        if (k == q) { npr = 0.0; npi = 0.0; nqi = 0.0; }    // the annihilated element and the real diagonal
                        if (k == p) { nqr = 0.0; nqi = 0.0; npi = 0.0; }
                        HR(p, k) = npr; HI(p, k) = npi;
                        HR(q, k) = nqr; HI(q, k) = nqi;
                    }
                }
        }
        if (sweep >= 40) notconv = 1;
        nsweep = sweep;
        // ascending order, stable: lane a ranks eigenvalue a (ties by index); levels and eigenvector rows go to their sorted place
        if (act) {
            const int a0 = r;
            const double la = HR(a0, a0);
            int rank = 0;
            for (int b = 0; b < n; ++b) {
                const double lb = HR(b, b);
                rank += (lb < la || (lb == la && b < a0)) ? 1 : 0;
            }
            rk[m * 8 + a0] = rank;
            const_cast<double *>(g.occ.ew)[(size_t)m * n + rank] = la;
            for (int i = 0; i < n; ++i) g.Vt[(size_t)m * nn + rank * n + i] = double2{VR(i, a0), VI(i, a0)};
        }
    }
    __threadfence();
    __syncthreads();
    clk1 = wall_clock64();
    // occupations of ALL levels (one particle-number sector), the very code of dmk_assign_occ
    if (g.zero_t) occ_zero_t_body(g.occ);
    else occ_fermi_body(g.occ);
    __threadfence();
    __syncthreads();
    clk2 = wall_clock64();
    // rho_k also stays in LDS for the fold (the H arrays are free now: 2 nn ms doubles >= nmat nn complex numbers)
    double2 *rl = reinterpret_cast<double2 *>(Hr);
    if (act) {                                 // rho_k = (V occ) V^H: lane r forms row r
        const double *oc = g.occ.occ + (size_t)m * n;
        const int i = r;
        for (int j = 0; j < n; ++j) {
            double re = 0.0, im = 0.0;
            for (int l = 0; l < n; ++l) {
                const double o = oc[rk[m * 8 + l]];
                const double air = VR(i, l), aii = VI(i, l), bjr = VR(j, l), bji = VI(j, l);
                re += o * (air * bjr + aii * bji);                            // a conj(b)
                im += o * (aii * bjr - air * bji);
            }
            g.rho_k[(size_t)m * nn + i * n + j] = double2{re, im};
            rl[(size_t)m * nn + i * n + j] = double2{re, im};
        }
    }
#undef HR
#undef HI
#undef VR
#undef VI
    __threadfence();
    __syncthreads();
    clk3 = wall_clock64();
    // k -> R: rho_R[s][R][ij] = Re (1 / nk) sum_k e^{+2 pi i k.R} rho_k[s][k][ij]   (np.fft.ifftn over the mesh axes)
    const int nk = g.nk, n1 = g.mesh[1], n2 = g.mesh[2];
    const double inv = 1.0 / (double)nk;
    double imax = 0.0;
    for (int o = tid; o < g.spin * nk * nn; o += SM_NT) {
        const int ij = o % nn, R = (o / nn) % nk, s = o / (nn * nk);
        const int r0 = R / (n1 * n2), r1 = (R / n2) % n1, r2 = R % n2;
        const int n0 = g.mesh[0];
        double re = 0.0, im = 0.0;
        // k = (k0 n1 + k1) n2 + k2 walked as ONE loop: the twiddle indices (k_d r_d) mod n_d advance by r_d with a conditional wrap
        // and restart at 0 when their axis counter wraps -- no integer division inside (six of them per term made this loop two
        // thirds of the kernel at C2), and no loop nest: as three nested loops with trip counts like 6 x 6 x 1 every term paid the
        // LDS round trips of its twiddles and the loop-control branches in sequence (0.77 us per term); the flat loop is unrolled
        // and the reads of four terms are in flight together
        const double2 *v = rl + (size_t)s * nk * nn + ij;
        int i0 = 0, i1 = 0, i2 = 0, c1 = 0, c2 = 0;
#pragma unroll 4
        for (int k = 0; k < nk; ++k) {
            const double2 w0 = tw[i0], w1 = tw[128 + i1], w2 = tw[256 + i2], x = v[(size_t)k * nn];
            const double ar = w0.x * w1.x - w0.y * w1.y, ai = w0.x * w1.y + w0.y * w1.x;
            const double pr = ar * w2.x - ai * w2.y, pi = ar * w2.y + ai * w2.x;
            re += pr * x.x - pi * x.y;
            im += pr * x.y + pi * x.x;
            ++c2;
            i2 += r2; if (i2 >= n2) i2 -= n2;
            if (c2 == n2) {
                c2 = 0; i2 = 0; ++c1;
                i1 += r1; if (i1 >= n1) i1 -= n1;
                if (c1 == n1) {
                    c1 = 0; i1 = 0;
                    i0 += r0; if (i0 >= n0) i0 -= n0;
                }
            }
        }
        g.rho_R[o] = re * inv;
        imax = fmax(imax, fabs(im * inv));
    }
    imax = block_max_f64(imax, shd);
    const double nc = block_max_f64((double)notconv, shd);
    const double ns = block_max_f64((double)nsweep, shd);
    clk4 = wall_clock64();
    if (tid == 0) {
        g.info[5] = imax; g.info[6] = nc; g.info[7] = ns;
        g.info[8] = 0.01 * (double)(clk1 - clk0); g.info[9] = 0.01 * (double)(clk2 - clk1);      // us: eigensolver, occupations,
        g.info[10] = 0.01 * (double)(clk3 - clk2); g.info[11] = 0.01 * (double)(clk4 - clk3);    // density, fold
    }
}

// ---- bath ---------------------------------------------------------------------------------------------------------------
struct SmallBath {
    int n0, n1, n2, nlo, spin, nenv, nb, nimp, nsites, orth, ncol_max;
    double tol;
    const double *rdm1;                        // [spin][nk][nlo][nlo] real stripe
    long long rdm1_stride;
    const int *env_idx, *bath_col, *virt_mask, *imp_idx;
    double *sigma;                             // [spin][nb]
    double *U;                                 // optional [spin][nenv][nb]
    double *basis;                             // [spin][nsites][ncol], ncol = nimp + min_s nbath_s, PACKED with that leading dimension
    int *iout;                                 // [0] ncol, [1 + s] nbath_s, [1 + spin] SVD sweeps not converged
    unsigned long long *dbg;                   // DMK_SMALL_TIMING: 100 MHz stamps of thread 0 at the phase boundaries (else null)
};

constexpr int SB_MAXB = 8;
constexpr int SB_NT = 256;                     // tall matrices of a few hundred rows: four waves keep the reductions short

// sum of `cnt` (<= SB_MAXB + 1) per-thread values over the workgroup, every thread receives all totals (fixed order)
__device__ void block_sum_vec(double *v, int cnt, double *sh /* [SB_MAXB + 1][SB_NT / 64] */) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    for (int c = 0; c < cnt; ++c) {
        const double s = dmk_wave_sum(v[c]);
        if (lane == 0) sh[c * (SB_NT / 64) + wave] = s;
    }
    __syncthreads();
    for (int c = 0; c < cnt; ++c) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < SB_NT / 64; ++w) t += sh[c * (SB_NT / 64) + w];
        v[c] = t;
    }
}

// One-sided (Hestenes) Jacobi on the columns of the nb x nb matrix M (LDS, row-major), run by ONE wave as EIGHT groups of eight
// lanes.  A sweep is a round-robin tournament: nb' - 1 steps (nb' = nb rounded up to even) of nb' / 2 mutually disjoint column
// pairs, group g takes pair g of the step.  Every lane of a group forms the three sums of its pair itself from LDS reads of the two
// columns (2 nb independent reads, three FMA chains of length nb) -- the lanes of a group run the same instructions on the same
// numbers, so they agree on the decision and on the rotation by construction -- and lane i < nb of the group then rotates row i.
// LDS operations of one wave complete in order, so the reads of a step see every write of the step before.
// (Round 5, first form: one pair at a time, lane i owned row i and the sums were 8-lane shuffle butterflies: ~1000 cycles per pair
// in four dependent cross-lane stages per sum and a chain of two reciprocals and two reciprocal square roots, 17-21 us for the SVD
// of a 4 x 4 triangle; it also needed products rounded on their own so that the lanes of a pair could not disagree.)
// Rotation from ONE reciprocal square root of the pair's (be - al, 2 ga) and one of (1 + cos 2theta) / 2 -- no division:
//     cos 2theta = |d| / h,  c = sqrt(u),  s = sign(d) g2 / (2 h c),   d = be - al, g2 = 2 ga, h = hypot(d, g2), u = (1 + cos 2theta) / 2
// (c and s share the factor rsqrt(u), so c^2 + s^2 = 1 to rounding whatever the estimate's error in h).
// On return the columns are mutually orthogonal: their norms are the singular values, the normalised columns the left singular
// vectors (for a symmetric positive semi-definite M: eigenvalues / eigenvectors).  Returns the number of sweeps, 60 = not
// converged.  Call with all 64 lanes of the wave.
__device__ int wave_onesided_jacobi(double *M, int nb) {
    const int lane = threadIdx.x & 63, i = lane & 7, grp = lane >> 3;
    const int np = (nb + 1) & ~1, nr = np - 1;                             // players, rounds
    const bool row_on = i < nb;
    int sweep = 0;
    for (; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int r = 0; r < nr; ++r) {
            // pair of this group in round r: (np - 1, r) for group 0, ((r + g) mod nr, (r - g) mod nr) otherwise
            int p = r + grp, q = r - grp;
            if (p >= nr) p -= nr;
            if (q < 0) q += nr;
            if (grp == 0) { p = r; q = nr; }
            if (p > q) { const int t = p; p = q; q = t; }
            const bool has = grp < (np >> 1) && q < nb;                    // (q = nb: the dummy player of an odd nb)
            double al = 0.0, be = 0.0, ga = 0.0;
            if (has) {
                for (int k = 0; k < nb; ++k) {
                    const double xr = M[k * nb + p], yr = M[k * nb + q];
                    al = fma(xr, xr, al); be = fma(yr, yr, be); ga = fma(xr, yr, ga);
                }
            }
            // rotate when |ga| > eps sqrt(al be) (squared: no square root)
            const bool rot = has && ga != 0.0 && ga * ga > 4.930380657631324e-32 * (al * be);
            if (rot) {
                const double d = be - al, g2 = 2.0 * ga;
                const double h2 = d * d + g2 * g2;
                if (h2 > 1.0e-290) {
                    double rh = __builtin_amdgcn_rsq(h2);
                    rh = rh * (1.5 - 0.5 * h2 * rh * rh);                  // one Newton step: the angle needs no more
                    const double u = 0.5 + 0.5 * fabs(d) * rh;             // in [1/2, 1]
                    const double ru = sm_rsqrt(u);
                    const double c = u * ru;
                    const double sn = (d >= 0.0 ? 0.5 : -0.5) * g2 * rh * ru;
                    if (row_on) {
                        const double x = M[i * nb + p], y = M[i * nb + q];
                        M[i * nb + p] = c * x - sn * y;
                        M[i * nb + q] = sn * x + c * y;
                    }
                    rotated = true;
                }
            }
        }
        if (!__any(rotated)) break;
    }
    return sweep;
}

__global__ __launch_bounds__(SB_NT) void small_bath_kernel(const SmallBath g) {
    extern __shared__ __attribute__((aligned(16))) double dyn[];            // A [spin][nenv][nb] (becomes Q-applied U) | R, Ur [spin][nb][nb] | tau [spin][nb] | X [nb][nb]
    __shared__ double shv[(SB_MAXB + 1) * (SB_NT / 64)];
    __shared__ int nbath_s[2];
    __shared__ int bad_s;
    // work arrays of the single-thread sections (nb x nb problems): LDS, not per-thread arrays -- those live in scratch memory,
    // where every access of the serial Jacobi loops is a round trip to L2
    __shared__ double Ms[SB_MAXB * SB_MAXB], Ws[SB_MAXB * SB_MAXB], sgs[SB_MAXB];
    __shared__ int ords[SB_MAXB];
    const int nenv = g.nenv, nb = g.nb, tid = threadIdx.x, spin = g.spin;
    double *Aall = dyn;
    double *Rall = Aall + (size_t)spin * nenv * nb;
    double *Uall = Rall + (size_t)spin * nb * nb;
    double *tauall = Uall + (size_t)spin * nb * nb;
    double *X = tauall + (size_t)spin * nb;
    if (tid == 0) bad_s = 0;
#define SB_STAMP(i) do { if (g.dbg && tid == 0) g.dbg[i] = wall_clock64(); } while (0)
    SB_STAMP(0);
    for (int s = 0; s < spin; ++s) {
        double *A = Aall + (size_t)s * nenv * nb, *Rm = Rall + (size_t)s * nb * nb, *Ur = Uall + (size_t)s * nb * nb;
        double *tau = tauall + (size_t)s * nb;
        const double *rd = g.rdm1 + (size_t)s * g.rdm1_stride;
        // gather: A[r][c] = big[env_idx[r]][bath_col[c]], big[(R1, p), (R2, q)] = rdm1[R1 - R2][p][q]   (bath.hip)
        for (int t = tid; t < nenv * nb; t += SB_NT) {
            const int r = t / nb, c = t % nb;
            const int e = g.env_idx[r], sc = g.bath_col[c];
            const int R1 = e / g.nlo, p = e % g.nlo, R2 = sc / g.nlo, q = sc % g.nlo;
            const int a0 = R1 / (g.n1 * g.n2), a1 = (R1 / g.n2) % g.n1, a2 = R1 % g.n2;
            const int b0 = R2 / (g.n1 * g.n2), b1 = (R2 / g.n2) % g.n1, b2 = R2 % g.n2;
            const int c0 = (a0 - b0 + g.n0) % g.n0, c1 = (a1 - b1 + g.n1) % g.n1, c2 = (a2 - b2 + g.n2) % g.n2;
            const int Rd = (c0 * g.n1 + c1) * g.n2 + c2;
            A[t] = rd[((size_t)Rd * g.nlo + p) * g.nlo + q];
        }
        __syncthreads();
        if (s == 0) SB_STAMP(1);
        // Householder QR, column by column: v = x + sign(x_j) |x| e_j (stored in place below the diagonal, v_j kept in `vj`),
        // tau = 2 / (v.v); one fused reduction gives |x|^2, a second one the dots v . a_k of all later columns
        for (int j = 0; j < nb && j < nenv; ++j) {
            double acc[SB_MAXB + 1];
            double nrm2 = 0.0;
            for (int r = j + tid; r < nenv; r += SB_NT) nrm2 = fma(A[r * nb + j], A[r * nb + j], nrm2);
            acc[0] = nrm2;
            block_sum_vec(acc, 1, shv);
            const double xn = sqrt(acc[0]);
            const double xj = A[j * nb + j];
            const double alpha = xj >= 0.0 ? -xn : xn;                     // R_jj
            const double vj = xj - alpha;                                  // v_j (no cancellation)
            const double vv = acc[0] - xj * xj + vj * vj;                  // v . v
            const double tj = vv > 0.0 ? 2.0 / vv : 0.0;
            // dots with the later columns
            for (int k = 0; k <= SB_MAXB; ++k) acc[k] = 0.0;
            for (int r = j + tid; r < nenv; r += SB_NT) {
                const double vr = (r == j) ? vj : A[r * nb + j];
                for (int k = j + 1; k < nb; ++k) acc[k - j - 1] = fma(vr, A[r * nb + k], acc[k - j - 1]);
            }
            block_sum_vec(acc, nb - j - 1, shv);
            for (int r = j + tid; r < nenv; r += SB_NT) {
                const double vr = (r == j) ? vj : A[r * nb + j];
                for (int k = j + 1; k < nb; ++k) A[r * nb + k] -= tj * acc[k - j - 1] * vr;
            }
            __syncthreads();
            if (tid == 0) {
                tau[j] = tj;
                // the reflector stays below the diagonal scaled so that its j-th component is 1; R_jj replaces the diagonal
                Rm[j * nb + j] = alpha;
                for (int k = j + 1; k < nb; ++k) Rm[j * nb + k] = A[j * nb + k];
                for (int k = 0; k < j; ++k) Rm[j * nb + k] = 0.0;
                A[j * nb + j] = vj;                                        // keep v_j itself: H = I - tau v v^T with the stored v
            }
            __syncthreads();
        }
        if (s == 0) SB_STAMP(2);
        // one-sided Jacobi SVD of the nb x nb triangle R = Ur diag(sigma) W^T: columns rotated until mutually orthogonal
        if (tid < 64) {
            const int k = nb < nenv ? nb : nenv;
            double *M = Ms;
            for (int e = tid; e < nb * nb; e += 64) M[e] = (e / nb) < k ? Rm[e] : 0.0;
            const int sweeps = wave_onesided_jacobi(M, nb);
            if (tid == 0 && sweeps >= 60) bad_s = 1;
            if (s == 0 && g.dbg && tid == 0) { g.dbg[7] = wall_clock64(); g.dbg[8] = (unsigned long long)sweeps; }
            // sigma = column norms, descending (stable: ties by index), Ur = normalised columns in that order
            if (tid < nb) {
                double a2 = 0.0;
                for (int i = 0; i < nb; ++i) a2 = fma(M[i * nb + tid], M[i * nb + tid], a2);
                sgs[tid] = sqrt(a2);
            }
            if (tid < nb) {                                                // same wave: sgs is complete (LDS program order)
                const double mine = sgs[tid];
                int pos = 0;
                for (int c = 0; c < nb; ++c) pos += (sgs[c] > mine || (sgs[c] == mine && c < tid)) ? 1 : 0;
                ords[pos] = tid;
            }
            if (tid < nb) {
                const int src = ords[tid];
                const double sv = sgs[src];
                g.sigma[s * nb + tid] = sv;
                const double inv = sv > 0.0 ? 1.0 / sv : 0.0;
                for (int i = 0; i < nb; ++i) Ur[i * nb + tid] = M[i * nb + src] * inv;
            }
            if (tid == 0) {
                int cnt = 0;
                for (int c = 0; c < nb; ++c) cnt += (sgs[c] >= g.tol) ? 1 : 0;
                nbath_s[s] = cnt;
            }
        }
        __syncthreads();
        if (s == 0) SB_STAMP(3);
        // U = Q [Ur; 0]: the reflectors applied in reverse order to the rows of [Ur; 0] -- into the storage of A, whose reflector
        // columns are consumed as they are applied (column j's reflector lives in A[j.., j])
        // First move the reflectors out of the way: V[r][j] = A[r][j] (r >= j) is needed until step j; U overwrites A row-wise, so
        // the reflector components are copied to registers per row batch instead -- simpler: keep a second array?  nenv * nb
        // doubles more of LDS is affordable for the sizes this kernel accepts: Y lives behind X.
        double *Y = X + nb * nb;                                           // [nenv][nb]
        for (int t = tid; t < nenv * nb; t += SB_NT) {
            const int r = t / nb, c = t % nb;
            Y[t] = r < nb ? Ur[r * nb + c] : 0.0;
        }
        __syncthreads();
        for (int j = (nb < nenv ? nb : nenv) - 1; j >= 0; --j) {
            double acc[SB_MAXB + 1];
            for (int k = 0; k < nb; ++k) acc[k] = 0.0;
            for (int r = j + tid; r < nenv; r += SB_NT) {
                const double vr = A[r * nb + j];
                for (int k = 0; k < nb; ++k) acc[k] = fma(vr, Y[r * nb + k], acc[k]);
            }
            block_sum_vec(acc, nb, shv);
            const double tj = tau[j];
            for (int r = j + tid; r < nenv; r += SB_NT) {
                const double vr = A[r * nb + j];
                for (int k = 0; k < nb; ++k) Y[r * nb + k] -= tj * acc[k] * vr;
            }
            __syncthreads();
        }
        // U of this spin -> A's storage (the reflectors are no longer needed), and to global memory when asked for
        for (int t = tid; t < nenv * nb; t += SB_NT) {
            A[t] = Y[t];
            if (g.U) g.U[(size_t)s * nenv * nb + t] = Y[t];
        }
        __syncthreads();
    }
    SB_STAMP(4);
    // ---- assemble: basis[s] = [ imp identity | env rows: B X ], B = U[:, :nbath_s] with virtual rows zeroed, X = (B^T B)^-1/2 ----
    int nbf = nb;
    for (int s = 0; s < spin; ++s) nbf = nbath_s[s] < nbf ? nbath_s[s] : nbf;
    const int ncol = g.nimp + nbf;
    for (size_t t = tid; t < (size_t)spin * g.nsites * ncol; t += SB_NT) g.basis[t] = 0.0;
    __threadfence();
    __syncthreads();
    SB_STAMP(5);
    for (int s = 0; s < spin; ++s) {
        double *A = Aall + (size_t)s * nenv * nb;
        double *bs = g.basis + (size_t)s * g.nsites * ncol;
        const int nbs = nbath_s[s];
        for (int i = tid; i < g.nimp && i < ncol; i += SB_NT) bs[(size_t)g.imp_idx[i] * ncol + i] = 1.0;
        if (nbs == 0) continue;
        if (g.orth) {
            for (int t = tid; t < nenv * nb; t += SB_NT)
                if (g.virt_mask[t / nb]) A[t] = 0.0;
            __syncthreads();
            // metric S = B^T B (nbs x nbs), one fused reduction per row of S
            for (int i = 0; i < nbs; ++i) {
                double acc[SB_MAXB + 1];
                for (int k = 0; k < nbs; ++k) acc[k] = 0.0;
                for (int r = tid; r < nenv; r += SB_NT) {
                    const double bi = A[r * nb + i];
                    for (int k = 0; k < nbs; ++k) acc[k] = fma(bi, A[r * nb + k], acc[k]);
                }
                block_sum_vec(acc, nbs, shv);
                if (tid == 0)
                    for (int k = 0; k < nbs; ++k) X[i * nb + k] = acc[k];
            }
            __syncthreads();
            if (tid < 64) {
                // X = S^-1/2 = sum_{e_m > 1e-14} v_m v_m^T / sqrt(e_m) (lo/lowdin.py:83-91).  S is symmetric positive semi-definite:
                // the one-sided Jacobi on its columns leaves S J = V Lambda -- column norms = eigenvalues, normalised columns =
                // eigenvectors (the general path takes the same route, bath.hip dmk_bath_assemble)
                double *Sm = Ms, *Vm = Ws;
                for (int e = tid; e < nbs * nbs; e += 64) Sm[e] = X[(e / nbs) * nb + (e % nbs)];
                const int sweeps = wave_onesided_jacobi(Sm, nbs);
                if (tid == 0 && sweeps >= 60) bad_s = 1;
                if (tid < nbs) {
                    double a2 = 0.0;
                    for (int i = 0; i < nbs; ++i) a2 = fma(Sm[i * nbs + tid], Sm[i * nbs + tid], a2);
                    const double e = sqrt(a2);
                    sgs[tid] = e;
                    const double inv = e > 0.0 ? 1.0 / e : 0.0;
                    for (int i = 0; i < nbs; ++i) Vm[i * nbs + tid] = Sm[i * nbs + tid] * inv;
                }
                if (tid < nbs) {                                           // row tid of X
                    for (int k = 0; k < nbs; ++k) {
                        double a = 0.0;
                        for (int m = 0; m < nbs; ++m) {
                            const double e = sgs[m];
                            if (e > 1.0e-14) a += Vm[tid * nbs + m] * Vm[k * nbs + m] / sqrt(e);
                        }
                        X[tid * nb + k] = a;
                    }
                }
            }
            __syncthreads();
        }
        for (int t = tid; t < nenv * nbs; t += SB_NT) {
            const int r = t / nbs, c = t % nbs;
            double v;
            if (g.orth) {
                v = 0.0;
                for (int j = 0; j < nbs; ++j) v += A[r * nb + j] * X[j * nb + c];
            } else {
                v = A[r * nb + c];
            }
            if (g.nimp + c < ncol) bs[(size_t)g.env_idx[r] * ncol + g.nimp + c] = v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        g.iout[0] = ncol;
        for (int s = 0; s < spin; ++s) g.iout[1 + s] = nbath_s[s];
        g.iout[1 + spin] = bad_s;
    }
    SB_STAMP(6);
#undef SB_STAMP
}

}  // namespace

extern "C" {

int dmk_small_meanfield(dmk_ctx *ctx, const int mesh[3], int n, int spin, const void *Fock_k, const double *add, int add_group,
                        double nelec, double beta, double mu0, int flags, double thr_deg, double fit_tol, double *ew, double *occ,
                        void *Vt, void *rho_k, double *rho_R, double *info_dev, int *handled) {
    if (!ctx) return DMK_ERR_INVALID;
    if (!handled || !mesh || !Fock_k || !ew || !occ || !Vt || !rho_k || !rho_R || !info_dev || n < 1 || spin < 1 || spin > 2)
        return dmk_fail(ctx, DMK_ERR_INVALID, "small_meanfield: bad arguments");
    *handled = 0;
    const long long nk = (long long)mesh[0] * mesh[1] * mesh[2];
    if (mesh[0] < 1 || mesh[1] < 1 || mesh[2] < 1) return dmk_fail(ctx, DMK_ERR_INVALID, "small_meanfield: bad mesh");
    const long long nmat = spin * nk;
    if (n > SM_MAXN || nmat > SM_NT / 8 || nk > 128 || mesh[0] > 128 || mesh[1] > 128 || mesh[2] > 128) return DMK_OK;
    if (add && add_group < 1) return dmk_fail(ctx, DMK_ERR_INVALID, "small_meanfield: add_group must be positive");
    const bool zero_t = !(beta < INFINITY);
    const long long nlev = nmat * n;
    if (zero_t && (nelec < 0.0 || nelec > (double)nlev || nelec != std::floor(nelec)))
        return dmk_fail(ctx, DMK_ERR_INVALID, "small_meanfield: T = 0 needs an integer 0 <= nelec <= %lld levels", nlev);
    if (!zero_t && !(beta > 0.0)) return dmk_fail(ctx, DMK_ERR_INVALID, "small_meanfield: beta must be positive");
    SmallMF g;
    g.n = n; g.nmat = (int)nmat; g.nk = (int)nk; g.spin = spin;
    g.mstride = (int)nmat | 1;                                             // odd stride: element-major arrays stay conflict-free
    g.mesh[0] = mesh[0]; g.mesh[1] = mesh[1]; g.mesh[2] = mesh[2];
    g.F = static_cast<const double2 *>(Fock_k); g.add = add; g.add_group = add ? add_group : 1;
    g.occ.ew = ew; g.occ.n = nlev; g.occ.nelec = nelec; g.occ.beta = beta; g.occ.mu0 = mu0; g.occ.thr = thr_deg;
    g.occ.tol = fit_tol > 0.0 ? fit_tol : 1e-12;
    g.occ.has_mu0 = (flags & 1) ? 1 : 0; g.occ.fix_mu = (flags & 2) ? 1 : 0; g.occ.sorted = 0;
    g.occ.occ = occ; g.occ.out = info_dev;
    g.zero_t = zero_t ? 1 : 0;
    g.Vt = static_cast<double2 *>(Vt); g.rho_k = static_cast<double2 *>(rho_k); g.rho_R = rho_R; g.info = info_dev;
    const size_t lds = ((size_t)4 * n * n * g.mstride + 2 * 3 * 128) * sizeof(double) + (size_t)nmat * 8 * sizeof(int);
    if (lds > 120 * 1024) return DMK_OK;
    if (lds > 48 * 1024)
        DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(small_meanfield_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FamScope fs(ctx, DMK_FAM_EIGH);
    hipLaunchKernelGGL(small_meanfield_kernel, dim3(1), dim3(SM_NT), lds, ctx->stream, g);
    DMK_CHECK_LAUNCH(ctx);
    *handled = 1;
    return DMK_OK;
}

int dmk_small_bath(dmk_ctx *ctx, const int mesh[3], int nlo, int spin, const double *rdm1, int64_t rdm1_stride, const int32_t *env_idx,
                   int nenv, const int32_t *bath_col, int nb, const int32_t *virt_mask, int orth, const int32_t *imp_idx, int nimp,
                   int nsites, double tol_bath, double *sigma, double *U, double *basis, int32_t *iout_dev, int *handled) {
    if (!ctx) return DMK_ERR_INVALID;
    if (!handled || !mesh || !rdm1 || !env_idx || !bath_col || !imp_idx || !sigma || !basis || !iout_dev || nlo < 1 || spin < 1 ||
        spin > 2 || nenv < 1 || nb < 1 || nimp < 0 || nsites < 1 || (orth && !virt_mask))
        return dmk_fail(ctx, DMK_ERR_INVALID, "small_bath: bad arguments");
    *handled = 0;
    if (nb > SB_MAXB) return DMK_OK;
    const size_t lds = ((size_t)spin * nenv * nb + (size_t)2 * spin * nb * nb + (size_t)spin * nb + (size_t)nb * nb +
                        (size_t)nenv * nb) * sizeof(double);
    if (lds > 120 * 1024) return DMK_OK;
    SmallBath g;
    g.n0 = mesh[0]; g.n1 = mesh[1]; g.n2 = mesh[2]; g.nlo = nlo; g.spin = spin; g.nenv = nenv; g.nb = nb; g.nimp = nimp;
    g.nsites = nsites; g.orth = orth ? 1 : 0; g.ncol_max = nimp + nb; g.tol = tol_bath;
    g.rdm1 = rdm1; g.rdm1_stride = rdm1_stride; g.env_idx = env_idx; g.bath_col = bath_col; g.virt_mask = virt_mask;
    g.imp_idx = imp_idx; g.sigma = sigma; g.U = U; g.basis = basis; g.iout = iout_dev;
    static const bool timing = getenv("DMK_SMALL_TIMING") != nullptr;       // the builder's phase profile; never set in production
    static unsigned long long *dbg = nullptr;
    if (timing && !dbg) DMK_HIP(ctx, hipMalloc(&dbg, 16 * sizeof(unsigned long long)));
    g.dbg = timing ? dbg : nullptr;
    if (lds > 48 * 1024)
        DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(small_bath_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
    FamScope fs(ctx, DMK_FAM_BATH);
    hipLaunchKernelGGL(small_bath_kernel, dim3(1), dim3(SB_NT), lds, ctx->stream, g);
    DMK_CHECK_LAUNCH(ctx);
    if (timing) {
        unsigned long long t[9];
        DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        DMK_HIP(ctx, hipMemcpy(t, dbg, sizeof(t), hipMemcpyDeviceToHost));
        fprintf(stderr, "[small_bath nenv=%d nb=%d spin=%d] us: gather %.2f, QR %.2f, SVD %.2f, apply Q (+ spin 1) %.2f, clear basis %.2f, "
                        "Loewdin + scatter %.2f; total %.2f; Jacobi of the SVD alone %.2f (%llu sweeps)\n", nenv, nb, spin, 0.01 * (t[1] - t[0]), 0.01 * (t[2] - t[1]), 0.01 * (t[3] - t[2]),
                0.01 * (t[4] - t[3]), 0.01 * (t[5] - t[4]), 0.01 * (t[6] - t[5]), 0.01 * (t[6] - t[0]), 0.01 * (t[7] - t[2]), t[8]);
    }
    *handled = 1;
    return DMK_OK;
}

}  // extern "C"
