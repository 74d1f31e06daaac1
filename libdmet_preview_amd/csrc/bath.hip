// K4 -- Schmidt-decomposition bath (SVD flavour).
//
// Replaces routine/slater.py:117-220 (`_get_emb_basis_svd`):
//   :167-175  gather of rdm1_env_imp from the stripe (incl. the lattice.expand branch,
//             system/lattice.py:304-337, done here as an index map  big[R1,R2] = A[R1-R2]
//             without materialising the (ncells*nlo)^2 matrix),
//   :180      scipy.linalg.svd (LAPACK dgesdd) of the tall-skinny (nenv x nb) block,
//   :200-202  virtual-row projection + Loewdin (lo/lowdin.py:83-101),
//   :212-213  scatter into C_lo_eo.
//
// SVD = Householder QR of the tall matrix + one-sided Jacobi SVD of the small R factor in LDS (high relative
// accuracy for the singular values that are compared with tol_bath = 1e-9; a Gram-matrix shortcut would square the
// condition number) + application of Q to [U_r; 0].
// The QR is a TSQR (communication-avoiding, still Householder and backward stable): every workgroup factors its own row
// slab entirely in LDS, the nb x nb R factors are stacked four at a time and factored again, level by level, until one
// R is left; Q is never formed, its reflectors stay where the slabs / stacks were, and Q [U_r; 0] is applied down the
// same tree.  ~12 launches per spin instead of the ~225 of the column-at-a-time version it replaces (kept below for
// nb > 64), and the tall matrix is read and written a constant number of times.
// Bound: HBM / launch latency: algorithmic bytes
// 8*nenv*(nb + nbath).
#include "common.h"
#include <algorithm>
#include <vector>

int launch_eigh_public(dmk_ctx *ctx, int n, int batch, const void *A, int a_real, const double *add, int add_group,
                       double *w, void *Vt, int v_real);

namespace {

constexpr int NT = 256;
constexpr int NWAVE = NT / 64;
constexpr int MAXB = 256;     // row-slab blocks for the tall-matrix kernels

__device__ __forceinline__ double wave_sum(double v) { return dmk_wave_sum(v); }

// A[r][c] = big[env_idx[r]][bath_col[c]],  big[(R1,p),(R2,q)] = rdm1[R1 - R2][p][q]
__global__ void gather_env_imp_kernel(int n0, int n1, int n2, int nlo, const double *__restrict__ rdm1,
                                      const int *__restrict__ env_idx, int nenv, const int *__restrict__ bath_col,
                                      int nb, double *__restrict__ A, long long rdm1_bstride = 0, long long A_bstride = 0) {
    rdm1 += (long long)blockIdx.y * rdm1_bstride;             // batch (spin channel) = blockIdx.y
    A += (long long)blockIdx.y * A_bstride;
    const long long total = (long long)nenv * nb;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(t / nb), c = (int)(t % nb);
        const int e = env_idx[r], s = bath_col[c];
        const int R1 = e / nlo, p = e % nlo, R2 = s / nlo, q = s % nlo;
        const int a0 = R1 / (n1 * n2), a1 = (R1 / n2) % n1, a2 = R1 % n2;
        const int b0 = R2 / (n1 * n2), b1 = (R2 / n2) % n1, b2 = R2 % n2;
        const int c0 = (a0 - b0 + n0) % n0, c1 = (a1 - b1 + n1) % n1, c2 = (a2 - b2 + n2) % n2;
        const int Rd = (c0 * n1 + c1) * n2 + c2;
        A[t] = rdm1[((long long)Rd * nlo + p) * nlo + q];
    }
}

// partial[blk][j] = sum over this block's rows r >= k of  V[r][kcol] * M[r][j]   (j in [j0, ncols))
// V and M are row-major with leading dims ldv / ldm.  Lanes run along j.
__global__ __launch_bounds__(NT) void col_dots_kernel(int nrows, int k, const double *__restrict__ V, int ldv, int kcol,
                                                      const double *__restrict__ M, int ldm, int j0, int ncols,
                                                      double *__restrict__ partial, int v_unit_diag,
                                                      double *__restrict__ rowk_out) {
    extern __shared__ double sh[];   // NWAVE x ncols
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // snapshot of row k of M for the next kernel (which rewrites that row while other
    // workgroups still need its old values)
    if (rowk_out != nullptr && blockIdx.x == 0)
        for (int j = j0 + threadIdx.x; j < ncols; j += NT) rowk_out[j] = M[(long long)k * ldm + j];
    const int rows_per_blk = (nrows - k + gridDim.x - 1) / gridDim.x;
    const int r0 = k + blockIdx.x * rows_per_blk;
    const int r1 = min(nrows, r0 + rows_per_blk);
    for (int jb = j0; jb < ncols; jb += 64) {
        const int j = jb + lane;
        double acc = 0.0;
        for (int r = r0 + wave; r < r1; r += NWAVE) {
            double v = V[(long long)r * ldv + kcol];
            if (v_unit_diag && r == k) v = 1.0;
            // QR mode (rowk_out given): the j == kcol entry is the norm of the part BELOW the
            // diagonal (computed directly, never as skk - alpha^2, which cancels)
            if (j < ncols && !(rowk_out != nullptr && r == k && j == kcol)) acc += v * M[(long long)r * ldm + j];
        }
        if (j < ncols) sh[wave * ncols + j] = acc;
    }
    __syncthreads();
    for (int j = j0 + threadIdx.x; j < ncols; j += NT) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) s += sh[w * ncols + j];
        partial[(long long)blockIdx.x * ncols + j] = s;
    }
}

// Householder QR step k on A (nrows x nb): uses partial[blk][j] = sum_{r>=k} A[r][k] A[r][j] for j > k
// and partial[blk][k] = sum_{r>k} A[r][k]^2.
__global__ __launch_bounds__(NT) void qr_apply_kernel(int nrows, int nb, int k, double *__restrict__ A,
                                                      const double *__restrict__ partial, int nblk_partial,
                                                      const double *__restrict__ rowk, double *__restrict__ tau_out) {
    extern __shared__ double sh[];   // g[nb] | fac[nb]
    __shared__ double s_tau, s_beta, s_inv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // deterministic reduction of the partials (fixed order), redundantly per block
    for (int j = k + threadIdx.x; j < nb; j += NT) {
        double s = 0.0;
        for (int q = 0; q < nblk_partial; ++q) s += partial[(long long)q * nb + j];
        sh[j] = s;
    }
    __syncthreads();
    const double alpha = rowk[k];
    if (threadIdx.x == 0) {
        const double xnorm2 = sh[k];
        double tau = 0.0, beta = alpha, inv = 0.0;
        if (xnorm2 > 0.0) {
            const double nrm = sqrt(alpha * alpha + xnorm2);
            beta = alpha >= 0.0 ? -nrm : nrm;
            tau = (beta - alpha) / beta;
            inv = 1.0 / (alpha - beta);
        }
        s_tau = tau; s_beta = beta; s_inv = inv;
    }
    __syncthreads();
    const double tau = s_tau, beta = s_beta, inv = s_inv;
    // fac_j = tau * v^T a_j,  v = [1; x * inv]:  v^T a_j = A[k][j] + (g_j - alpha A[k][j]) * inv
    for (int j = k + 1 + threadIdx.x; j < nb; j += NT) {
        const double akj = rowk[j];
        sh[nb + j] = tau * (akj + (sh[j] - alpha * akj) * inv);   // sh[j] includes the r = k term
    }
    __syncthreads();
    const int rows_per_blk = (nrows - k + gridDim.x - 1) / gridDim.x;
    const int r0 = k + blockIdx.x * rows_per_blk;
    const int r1 = min(nrows, r0 + rows_per_blk);
    for (int r = r0 + wave; r < r1; r += NWAVE) {
        const double ark = (r == k) ? alpha : A[(long long)r * nb + k];
        const double vr = (r == k) ? 1.0 : ark * inv;
        for (int j = k + 1 + lane; j < nb; j += 64)
            if (tau != 0.0) A[(long long)r * nb + j] -= sh[nb + j] * vr;
        if (lane == 0) A[(long long)r * nb + k] = (r == k) ? beta : vr;   // reflector stored in place
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) tau_out[k] = tau;
}

// U (nrows x nc) <- H_k U with H_k = I - tau v v^T, v from column k of QR-factored A (unit diagonal)
__global__ __launch_bounds__(NT) void q_apply_kernel(int nrows, int nb, int nc, int k, const double *__restrict__ A,
                                                     const double *__restrict__ tau_arr, double *__restrict__ U,
                                                     const double *__restrict__ partial, int nblk_partial) {
    extern __shared__ double sh[];   // t[nc]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double tau = tau_arr[k];
    if (tau == 0.0) return;
    for (int j = threadIdx.x; j < nc; j += NT) {
        double s = 0.0;
        for (int q = 0; q < nblk_partial; ++q) s += partial[(long long)q * nc + j];
        sh[j] = tau * s;
    }
    __syncthreads();
    const int rows_per_blk = (nrows - k + gridDim.x - 1) / gridDim.x;
    const int r0 = k + blockIdx.x * rows_per_blk;
    const int r1 = min(nrows, r0 + rows_per_blk);
    for (int r = r0 + wave; r < r1; r += NWAVE) {
        const double vr = (r == k) ? 1.0 : A[(long long)r * nb + k];
        for (int j = lane; j < nc; j += 64) U[(long long)r * nc + j] -= sh[j] * vr;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// TSQR building blocks.  Both kernels work on `m` consecutive rows [r0, r0 + m) of a row-major matrix with leading
// dimension nb: a leaf's slab of the gathered matrix A, or -- tree levels -- four stacked nb x nb R factors of the level
// below (block b of a stack occupies rows [b nb, (b + 1) nb), so a node's inputs are contiguous as well).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TS_MAXROWS = 256;      // rows per node: 8 waves x 32 register slots
constexpr int TS_FANIN = 4;

struct TsqrArgs {
    int nb;                 // columns
    int rows_total;         // rows of the whole matrix at this level
    int rows_per_node;      // rows of a node (the last one may be shorter)
    double *M;              // rows_total x nb: in = matrix, out = reflectors (unit diagonal implied) below / R on and above the diagonal
    double *tau;            // nodes x nb
    double *Rout;           // nodes x nb x nb, zero below the diagonal and beyond the node's rows (input of the next level)
    long long bstride;      // elements between the workspaces of consecutive batch members (blockIdx.y): M, tau and Rout all live in it
};

__device__ __forceinline__ double lane_bcast(double v, int src) {      // src wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

constexpr int TS_NT = 512, TS_NW = TS_NT / 64;       // eight waves per node: rows are dealt round-robin to the waves

// Householder QR of one node, the node's rows held in REGISTERS: lane <-> column (nb <= 64), wave w owns rows w, w + 8, ...
// The first version kept the node in LDS and spent 12 us per column waiting on LDS round trips (0.68 ms per level at 224 x 56;
// rocprof); here a column step is register arithmetic: column k is broadcast out of lane k with v_readlane and the scalar feeds
// v_fma_f64 directly.  Two things keep the inner loops free of per-row conditions (the second version still spent 9 us per
// column on exec-mask bookkeeping around every row):
//   * a finished row is written out and its registers are cleared, so a zero row contributes nothing to any sum or update;
//   * after every group of eight columns the slots are SHIFTED down by one (slot i <- slot i + 1), so the pivot row of column
//     k = 8 g + c is always slot 0 of wave c -- a compile-time register, a wave-uniform owner.
// One barrier per column: the per-wave partial sums and the pivot-row snapshot go through a double-buffered LDS slab (the
// buffer of step k is last read in step k, and step k + 2 cannot write it before every wave passed the barrier of step k + 1).
//   h_j = sum_{r >= k} a_rk a_rj (j >= k);  |x|^2 below the diagonal = h_k - a_kk^2 is NOT used -- g_k is summed without the
//   pivot row;  tau, beta from g_k and a_kk;  a_rj -= tau (a_kj + inv g_j) v_r,  v = [1; a_rk inv];  reflector kept in column k.
template <int RPT>
__global__ __launch_bounds__(TS_NT) void tsqr_factor_kernel(const TsqrArgs g) {
    __shared__ double part[2][TS_NW + 1][64];        // [buffer][wave | pivot-row snapshot][column]
    __shared__ double taus[64];
    const int nb = g.nb;
    const int node = blockIdx.x;
    const int r0 = node * g.rows_per_node;
    const int m = min(g.rows_per_node, g.rows_total - r0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // uniform: keeps the owner tests scalar
    const int j = lane;
    const long long boff = (long long)blockIdx.y * g.bstride;
    double *Mrows = g.M + boff + (size_t)r0 * nb;
    double *R = g.Rout + boff + (size_t)node * nb * nb;
    double a[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = wave + TS_NW * i;
        a[i] = (r < m && j < nb) ? Mrows[(size_t)r * nb + j] : 0.0;
    }
    if (threadIdx.x < 64) taus[threadIdx.x] = 0.0;
    const int kmax = min(m, nb);
    const int ngroups = (kmax + TS_NW - 1) / TS_NW;
    for (int grp = 0; grp < ngroups; ++grp) {
#pragma unroll
        for (int c = 0; c < TS_NW; ++c) {
            const int k = grp * TS_NW + c;
            if (k >= kmax) break;                                  // uniform
            const int buf = k & 1;
            // ---- partial sums over this wave's live rows; slot 0 holds row 8 grp + wave: finished if wave < c, pivot if wave == c
            double acc = 0.0;
            if (wave == c) part[buf][TS_NW][j] = a[0];
            else if (wave > c) acc = lane_bcast(a[0], k) * a[0];
#pragma unroll
            for (int i = 1; i < RPT; ++i) acc += lane_bcast(a[i], k) * a[i];
            part[buf][wave][j] = acc;
            __syncthreads();
            double gj = 0.0, xnorm2 = 0.0;
#pragma unroll
            for (int w = 0; w < TS_NW; ++w) {
                gj += part[buf][w][j];
                xnorm2 += part[buf][w][k];
            }
            const double alpha = part[buf][TS_NW][k];
            double tau = 0.0, beta = alpha, inv = 0.0;
            if (xnorm2 > 0.0) {
                const double nrm = sqrt(alpha * alpha + xnorm2);
                beta = alpha >= 0.0 ? -nrm : nrm;
                tau = (beta - alpha) / beta;
                inv = 1.0 / (alpha - beta);
            }
            const double fac = (j > k) ? tau * (part[buf][TS_NW][j] + inv * gj) : 0.0;     // tau v^T a_j on the columns to update
            const bool diag = (j == k);
            if (wave == c) {                                       // pivot row: v = 1
                a[0] = diag ? beta : a[0] - fac;
            } else if (wave > c) {
                const double vr = lane_bcast(a[0], k) * inv;
                a[0] = diag ? vr : a[0] - fac * vr;
            }
#pragma unroll
            for (int i = 1; i < RPT; ++i) {
                const double vr = lane_bcast(a[i], k) * inv;
                a[i] = diag ? vr : a[i] - fac * vr;
            }
            if (threadIdx.x == 0) taus[k] = tau;
        }
        // rows 8 grp .. 8 grp + 7 are final (R on and right of the diagonal, reflector entries left of it): write them out,
        // then shift the slots so that the next group's pivot rows sit in slot 0 again
        {
            const int r = grp * TS_NW + wave;
            if (r < m && j < nb) Mrows[(size_t)r * nb + j] = a[0];
            if (r < nb && j < nb) R[(size_t)r * nb + j] = (r <= j && r < m) ? a[0] : 0.0;
        }
#pragma unroll
        for (int i = 0; i + 1 < RPT; ++i) a[i] = a[i + 1];
        a[RPT - 1] = 0.0;
    }
    __syncthreads();
    // rows that never became pivots (m > nb): slot i now holds row 8 (ngroups + i) + wave
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = (ngroups + i) * TS_NW + wave;
        if (r < m && j < nb) Mrows[(size_t)r * nb + j] = a[i];
        if (r < nb && j < nb) R[(size_t)r * nb + j] = 0.0;         // m < nb: rows of the R block beyond the node's rows
    }
    for (int t = threadIdx.x; t < nb; t += TS_NT) g.tau[boff + (size_t)node * nb + t] = taus[t];
}

struct TsqrApplyArgs {
    int nb, nc;
    int rows_total, rows_per_node;
    const double *V;        // rows_total x nb reflectors of this level (tsqr_factor_kernel's in-place output)
    const double *tau;      // nodes x nb
    const double *X;        // nodes x nb x nc: the node's input block (top rows of [X; 0])
    double *Y;              // rows_total x nc: Q_node [X; 0]; a tree level writes the X blocks of the level below, a leaf rows of U
    long long bstride;      // batch stride of the workspace (V, tau, X and the tree levels' Y)
    long long y_bstride;    // batch stride of Y (the leaf level writes U: nenv x nc per batch member)
};

// Y = H_0 H_1 ... H_{kmax-1} [X; 0] for one node, Y in registers (lane <-> column of Y, nc <= 64; slot i of wave w <-> row
// w + 8 i).  The reflector entries a wave needs in step k are one per lane (lane i keeps v_{w + 8 i, k}, zero above the
// diagonal so that no row needs a condition): they come straight from global memory, fetched one step ahead, and are broadcast
// with readlane.  One barrier per step (double-buffered partial sums, see the factor kernel).
template <int RPT>
__global__ __launch_bounds__(TS_NT) void tsqr_apply_kernel(const TsqrApplyArgs g) {
    __shared__ double part[2][TS_NW][64];
    const int nb = g.nb, nc = g.nc;
    const int node = blockIdx.x;
    const int r0 = node * g.rows_per_node;
    const int m = min(g.rows_per_node, g.rows_total - r0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane;
    const long long boff = (long long)blockIdx.y * g.bstride;
    const double *X = g.X + boff + (size_t)node * nb * nc;
    double y[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = wave + TS_NW * i;
        y[i] = (r < m && r < nb && j < nc) ? X[(size_t)r * nc + j] : 0.0;
    }
    const int kmax = min(m, nb);
    const int myrow = wave + TS_NW * lane;                       // the row whose reflector entries this lane carries (lane < RPT)
    const bool carrier = lane < RPT && myrow < m;
    const double *vcol = g.V + boff + (size_t)(r0 + (carrier ? myrow : 0)) * nb;
    double vnext = (carrier && kmax > 0) ? vcol[kmax - 1] : 0.0;
    int buf = 0;                                                 // toggles per EXECUTED step (skipped steps have no barrier)
    for (int k = kmax - 1; k >= 0; --k) {
        double vmine = vnext;
        if (myrow == k) vmine = 1.0;
        if (myrow < k || !carrier) vmine = 0.0;
        if (k > 0) vnext = carrier ? vcol[k - 1] : 0.0;
        const double tau = g.tau[boff + (size_t)node * nb + k];
        if (tau == 0.0) continue;                               // uniform: H_k = I
        buf ^= 1;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc += lane_bcast(vmine, i) * y[i];
        part[buf][wave][j] = acc;
        __syncthreads();
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < TS_NW; ++w) t += part[buf][w][j];
        t *= tau;
#pragma unroll
        for (int i = 0; i < RPT; ++i) y[i] -= t * lane_bcast(vmine, i);
    }
    double *Yrows = g.Y + (long long)blockIdx.y * g.y_bstride + (size_t)r0 * nc;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = wave + TS_NW * i;
        if (r < m && j < nc) Yrows[(size_t)r * nc + j] = y[i];
    }
}

// One-sided Jacobi SVD of the upper-triangular R (top nb x nb of A).  One workgroup.
// Output: sigma (descending), Utop (nb x nb row-major, column j <-> sigma[j]).
// upper_only: the input is an R factor (entries below the diagonal are ignored).  Vt_out (optional):
// right singular vectors, row j <-> sigma[j]; needs a second LDS image.
__global__ __launch_bounds__(NT) void jacobi_svd_kernel(int nb, const double *__restrict__ A, int lda,
                                                        double *__restrict__ sigma, double *__restrict__ Utop,
                                                        int *__restrict__ status, int upper_only,
                                                        double *__restrict__ Vt_out) {
    extern __shared__ double sh[];
    const int N = nb + (nb & 1);              // padded to even
    const int ld = nb + 1;                    // column stride (odd: conflict-free column walks)
    double *G = sh;                           // [N][ld] column-major: G[j*ld + i] = R[i][j]
    double *nrm = G + (size_t)N * ld;         // [N]
    double *V = nrm + N;                      // [N][ld] (only when Vt_out != nullptr)
    __shared__ int s_rot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < N * ld; t += NT) {
        const int j = t / ld, i = t % ld;
        G[t] = (j < nb && i < nb && (!upper_only || i <= j)) ? A[(long long)i * lda + j] : 0.0;
        if (Vt_out) V[t] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    const double eps = 2.220446049250313e-16;
    int sweep = 0;
    for (; sweep < 60; ++sweep) {
        if (threadIdx.x == 0) s_rot = 0;
        __syncthreads();
        for (int round = 0; round < N - 1; ++round) {
            for (int pi = wave; pi < N / 2; pi += NWAVE) {
                int p, q;
                if (pi == 0) { p = N - 1; q = round; }
                else { p = (round + pi) % (N - 1); q = (round - pi + (N - 1)) % (N - 1); }
                if (p > q) { const int t = p; p = q; q = t; }
                if (q >= nb) continue;       // padding column
                double a = 0.0, b = 0.0, c = 0.0;
                for (int i = lane; i < nb; i += 64) {
                    const double gp = G[p * ld + i], gq = G[q * ld + i];
                    a += gp * gp; b += gq * gq; c += gp * gq;
                }
                a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
                if (fabs(c) > eps * sqrt(a * b) && c != 0.0) {
                    const double zeta = (b - a) / (2.0 * c);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                    for (int i = lane; i < nb; i += 64) {
                        const double gp = G[p * ld + i], gq = G[q * ld + i];
                        G[p * ld + i] = cs * gp - sn * gq;
                        G[q * ld + i] = sn * gp + cs * gq;
                        if (Vt_out) {
                            const double vp = V[p * ld + i], vq = V[q * ld + i];
                            V[p * ld + i] = cs * vp - sn * vq;
                            V[q * ld + i] = sn * vp + cs * vq;
                        }
                    }
                    if (lane == 0) atomicAdd(&s_rot, 1);
                }
            }
            __syncthreads();
        }
        const int rot = s_rot;
        __syncthreads();
        if (rot == 0) break;
    }
    if (threadIdx.x == 0 && sweep >= 60) *status = 2;
    // norms, ranks (descending, stable), normalised columns
    for (int j = wave; j < nb; j += NWAVE) {
        double a = 0.0;
        for (int i = lane; i < nb; i += 64) a += G[j * ld + i] * G[j * ld + i];
        a = wave_sum(a);
        if (lane == 0) nrm[j] = sqrt(a);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nb; j += NT) {
        const double sj = nrm[j];
        int rk = 0;
        for (int q = 0; q < nb; ++q) rk += (nrm[q] > sj || (nrm[q] == sj && q < j)) ? 1 : 0;
        sigma[rk] = sj;
        const double inv = sj > 0.0 ? 1.0 / sj : 0.0;
        for (int i = 0; i < nb; ++i) Utop[(long long)i * nb + rk] = G[j * ld + i] * inv;
        if (Vt_out)
            for (int i = 0; i < nb; ++i) Vt_out[(long long)rk * nb + i] = V[j * ld + i];
    }
}

// One-sided Jacobi SVD for nb <= 64 without right vectors (the bath's R factor): same rotations in the same round-robin
// order as jacobi_svd_kernel, organised for latency.  Eight waves; a column is ONE element per lane, so a pair (p, q) is
// two LDS reads per lane, three wave sums through the DPP crossbar (interleaved, no LDS) and the rotation in registers;
// every wave loads all the pairs it owns in a round before it reduces any of them.  One barrier per round.
// (rocprof on the first TSQR version: the general kernel took 4.4 ms at nb = 56 -- 42 % of the bath; this one ~0.2 ms.)
constexpr int JF_NT = 1024, JF_NW = JF_NT / 64, JF_MAXP = 2;     // 32 pairs per round at most = 2 per wave
// 1 / x and 1 / sqrt(x) from the hardware estimates + two Newton steps (full double accuracy for the normal range these
// rotation parameters live in; an IEEE division / square root costs three times as many instructions)
__device__ __forceinline__ double jf_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    return r * (2.0 - x * r);
}
__device__ __forceinline__ double jf_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    return y * (1.5 - 0.5 * x * y * y);
}
__global__ __launch_bounds__(JF_NT) void jacobi_svd_fast_kernel(int nb, const double *__restrict__ A, int lda,
                                                                double *__restrict__ sigma, double *__restrict__ Utop,
                                                                int *__restrict__ status, int upper_only,
                                                                long long a_bstride = 0, long long u_bstride = 0,
                                                                const int *__restrict__ skip = nullptr) {
    if (skip && skip[0]) return;              // the caller's shortcut already produced the result (lowdin_taylor_kernel)
    // batch member = blockIdx.x: its matrix, its nb singular values, its vectors, its two status ints
    A += (long long)blockIdx.x * a_bstride;
    Utop += (long long)blockIdx.x * u_bstride;
    sigma += (long long)blockIdx.x * nb;
    status += 2 * blockIdx.x;
    __shared__ double G[64 * 65];             // column-major, column stride 65: G[j * 65 + i] = R[i][j]
    __shared__ double nrm[64];
    __shared__ int s_rot;
    const int N = nb + (nb & 1), ld = 65;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int t = threadIdx.x; t < 64 * ld; t += JF_NT) {
        const int j = t / ld, i = t % ld;
        G[t] = (j < nb && i < nb && (!upper_only || i <= j)) ? A[(long long)i * lda + j] : 0.0;
    }
    if (threadIdx.x == 0) s_rot = 0;
    __syncthreads();
    const double eps2 = 2.220446049250313e-16 * 2.220446049250313e-16;
    const int npairs = N / 2;
    int sweep = 0;
    for (; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int round = 0; round < N - 1; ++round) {
            int pp[JF_MAXP], qq[JF_MAXP];
            double gp[JF_MAXP], gq[JF_MAXP];
#pragma unroll
            for (int u = 0; u < JF_MAXP; ++u) {
                const int pi = wave + JF_NW * u;
                int p = -1, q = -1;
                if (pi < npairs) {
                    if (pi == 0) { p = N - 1; q = round; }
                    else { p = (round + pi) % (N - 1); q = (round - pi + (N - 1)) % (N - 1); }
                    if (p > q) { const int t = p; p = q; q = t; }
                    if (q >= nb) p = -1;          // padding column
                }
                pp[u] = p; qq[u] = q;
                gp[u] = p >= 0 ? G[p * ld + lane] : 0.0;
                gq[u] = p >= 0 ? G[q * ld + lane] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < JF_MAXP; ++u) {
                if (pp[u] < 0) continue;          // uniform
                const double a = wave_sum(gp[u] * gp[u]), b = wave_sum(gq[u] * gq[u]), c = wave_sum(gp[u] * gq[u]);
                // rotate unless the pair is orthogonal to working precision: |c| <= eps sqrt(a b), tested without the root
                if (c * c > eps2 * (a * b) && c != 0.0) {
                    const double zeta = (b - a) * 0.5 * jf_rcp(c);
                    const double az = fabs(zeta);
                    double t;
                    if (az > 1e150) t = 0.5 * jf_rcp(az);                       // 1 + zeta^2 would overflow
                    else t = jf_rcp(az + (1.0 + zeta * zeta) * jf_rsqrt(1.0 + zeta * zeta));
                    if (zeta < 0.0) t = -t;
                    const double cs = jf_rsqrt(1.0 + t * t), sn = cs * t;
                    G[pp[u] * ld + lane] = cs * gp[u] - sn * gq[u];
                    G[qq[u] * ld + lane] = sn * gp[u] + cs * gq[u];
                    rotated = 1;
                }
            }
            __syncthreads();
        }
        if (rotated && lane == 0) atomicAdd(&s_rot, 1);
        __syncthreads();
        const int rot = s_rot;
        __syncthreads();
        if (threadIdx.x == 0) s_rot = 0;
        if (rot == 0) break;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sweep >= 60) status[0] = 2;
        status[1] = sweep + 1;
    }
    for (int j = wave; j < nb; j += JF_NW) {
        const double v = G[j * ld + lane];
        const double a = wave_sum(v * v);
        if (lane == 0) nrm[j] = sqrt(a);
    }
    __syncthreads();
    // ranks (descending, stable) and normalised columns: column j of G goes to column rank(j) of Utop
    for (int j = wave; j < nb; j += JF_NW) {
        const double sj = nrm[j];
        int rk = 0;
        for (int q = 0; q < nb; ++q) rk += (nrm[q] > sj || (nrm[q] == sj && q < j)) ? 1 : 0;
        if (lane == 0) sigma[rk] = sj;
        const double inv = sj > 0.0 ? 1.0 / sj : 0.0;
        if (lane < nb) Utop[(long long)lane * nb + rk] = G[j * ld + lane] * inv;
    }
}

// C (M x N) = A (M x K) op(B), row-major f64, small K and N (tall-skinny times small);
// op(B) = B (K x N) or, with transB, B^T with B stored N x K
__global__ void rowmat_small_kernel(long long M, int N, int K, const double *__restrict__ A, const double *__restrict__ B,
                                    int transB, double *__restrict__ C) {
    const long long total = M * N;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / N;
        const int c = (int)(t % N);
        double s = 0.0;
        if (transB)
            for (int k = 0; k < K; ++k) s += A[r * K + k] * B[(long long)c * K + k];
        else
            for (int k = 0; k < K; ++k) s += A[r * K + k] * B[(long long)k * N + c];
        C[t] = s;
    }
}

// particle weight of the Nambu bath columns (routine/bcs.py:92): w[j] = sum_{c, p < keep} U[c][p][j]^2,
// U viewed as (ncell, period, nb); one workgroup per column, fixed-order reduction
__global__ __launch_bounds__(NT) void bcs_weight_kernel(int ncell, int period, int keep, int nb,
                                                        const double *__restrict__ U, double *__restrict__ w) {
    __shared__ double red[NWAVE];
    const int j = blockIdx.x;
    double s = 0.0;
    const int rows = ncell * keep;
    for (int t = threadIdx.x; t < rows; t += NT) {
        const int c = t / keep, p = t % keep;
        const double v = U[((long long)c * period + p) * nb + j];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < NWAVE; ++i) tot += red[i];
        w[j] = tot;
    }
}

// routine/bcs.py:88-103: basis (2, ncells, 2n, n + nval);  identity on the impurity block of cell 0,
// alpha bath = columns order[:nval] of U, beta bath = columns order[nval:] with the particle / hole halves swapped
__global__ void bcs_assemble_kernel(int ncells, int n, int nval, const double *__restrict__ U,
                                    const int *__restrict__ order, double *__restrict__ basis) {
    const int ncol = n + nval, n2 = 2 * n, nb = 2 * nval;
    const long long total = 2LL * ncells * n2 * ncol;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(t % ncol);
        const int r = (int)((t / ncol) % n2);
        const int c = (int)((t / ((long long)ncol * n2)) % ncells);
        const int s = (int)(t / ((long long)ncol * n2 * ncells));
        double v = 0.0;
        if (c == 0) {
            v = (col < n && r == col) ? 1.0 : 0.0;
        } else if (col >= n) {
            const int j = col - n;
            const int src_r = (s == 0) ? r : (r + n) % n2;
            const int src_c = order[s == 0 ? j : nval + j];
            v = U[((long long)(c - 1) * n2 + src_r) * nb + src_c];
        }
        basis[t] = v;
    }
}

// out (batch, r_out, c_out) = zero-padded copy of in (batch, r_in, c_in)   (slater_helper.py:517-518)
__global__ void pad_block_kernel(int batch, long long r_in, long long c_in, const double *__restrict__ in,
                                 long long r_out, long long c_out, double *__restrict__ out) {
    const long long total = (long long)batch * r_out * c_out;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const long long c = t % c_out, r = (t / c_out) % r_out, b = t / (c_out * r_out);
        out[t] = (r < r_in && c < c_in) ? in[(b * r_in + r) * c_in + c] : 0.0;
    }
}

__global__ void mask_copy_kernel(int nenv, int nb, int nbath, const double *__restrict__ U,
                                 const int *__restrict__ virt_mask, int apply_mask, double *__restrict__ B) {
    const long long total = (long long)nenv * nbath;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(t / nbath), c = (int)(t % nbath);
        B[t] = (apply_mask && virt_mask[r]) ? 0.0 : U[(long long)r * nb + c];
    }
}

// partial Gram: partial[blk][i*nc + j] = sum_{rows of blk} B[r][i] B[r][j]
__global__ __launch_bounds__(NT) void gram_partial_kernel(int nrows, int nc, const double *__restrict__ B,
                                                          double *__restrict__ partial) {
    const int rows_per_blk = (nrows + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(nrows, r0 + rows_per_blk);
    for (int t = threadIdx.x; t < nc * nc; t += NT) {
        const int i = t / nc, j = t % nc;
        double s = 0.0;
        for (int r = r0; r < r1; ++r) s += B[(long long)r * nc + i] * B[(long long)r * nc + j];
        partial[(long long)blockIdx.x * nc * nc + t] = s;
    }
}
// the same partial Gram with the row slab staged through LDS in chunks of GRAM_CHUNK rows (nc <= 64): the version above reads
// every B entry nc times from L2 (0.33 ms at 43144 x 56); here a chunk is read once, coalesced, and the nc^2 products run
// out of LDS with lanes along j (conflict-free) and i broadcast
constexpr int GRAM_CHUNK = 96;
__global__ __launch_bounds__(NT) void gram_partial_lds_kernel(int nrows, int nc, const double *__restrict__ B,
                                                              double *__restrict__ partial) {
    __shared__ double Bs[GRAM_CHUNK * 65];
    const int rows_per_blk = (nrows + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(nrows, r0 + rows_per_blk);
    constexpr int EPT = 16;                          // entries per thread: 64 * 64 / 256
    double acc[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) acc[u] = 0.0;
    const int ld = nc + 1;
    for (int c0 = r0; c0 < r1; c0 += GRAM_CHUNK) {
        const int rows = min(GRAM_CHUNK, r1 - c0);
        __syncthreads();
        for (int t = threadIdx.x; t < rows * nc; t += NT) Bs[(t / nc) * ld + (t % nc)] = B[(long long)c0 * nc + t];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = threadIdx.x + NT * u;
            if (e < nc * nc) {
                const int i = e / nc, j = e % nc;
                double sacc = 0.0;
                for (int r = 0; r < rows; ++r) sacc += Bs[r * ld + i] * Bs[r * ld + j];
                acc[u] += sacc;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int e = threadIdx.x + NT * u;
        if (e < nc * nc) partial[(long long)blockIdx.x * nc * nc + e] = acc[u];
    }
}
__global__ void reduce_partials_kernel(int n, int nblk, const double *__restrict__ partial, double *__restrict__ out) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int q = 0; q < nblk; ++q) s += partial[(long long)q * n + t];
        out[t] = s;
    }
}
// X = sum_{m: e_m > tol} v_m v_m^T / sqrt(e_m)   from Vt rows (lo/lowdin.py:83-91)
__global__ void inv_sqrt_kernel(int n, const double *__restrict__ e, const double *__restrict__ Vt, double tol,
                                double *__restrict__ X) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n * n; t += gridDim.x * blockDim.x) {
        const int i = t / n, j = t % n;
        double s = 0.0;
        for (int m = 0; m < n; ++m)
            if (e[m] > tol) s += Vt[m * n + i] * Vt[m * n + j] / sqrt(e[m]);
        X[t] = s;
    }
}
// the same from eigenvectors stored as COLUMNS (U[i][m], the layout of jacobi_svd_fast_kernel)
__global__ void inv_sqrt_cols_kernel(int n, const double *__restrict__ e, const double *__restrict__ U, double tol,
                                     double *__restrict__ X, const int *__restrict__ skip = nullptr) {
    if (skip && skip[0]) return;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n * n; t += gridDim.x * blockDim.x) {
        const int i = t / n, j = t % n;
        double s = 0.0;
        for (int m = 0; m < n; ++m)
            if (e[m] > tol) s += U[i * n + m] * U[j * n + m] / sqrt(e[m]);
        X[t] = s;
    }
}
// Loewdin factor of a metric that is the identity up to rounding -- the usual case: the bath vectors are columns of the SVD's
// U.  With E = S - I, S^-1/2 = I - E/2 + 3/8 E^2 - 5/16 E^3 ...; an element of E^3 is bounded by n^2 max|E|^3, so the shortcut is
// taken when 5/16 n^2 max|E|^3 <= 2e-17 (n = 64: max|E| <= 2.5e-7; a U from the SVD has max|E| ~ 1e-15); the kernel then sets flag[0] = 1 and the Jacobi
// eigensolver of the metric (1 ms of latency for a 56 x 56 matrix) and inv_sqrt_cols_kernel return at once.  Anything else
// (rank-deficient or genuinely non-orthogonal vectors) leaves flag[0] = 0 and takes the eigendecomposition with its 1e-14 cut.
__global__ __launch_bounds__(256) void lowdin_taylor_kernel(int n, const double *__restrict__ S, double *__restrict__ X,
                                                            int *__restrict__ flag) {
    __shared__ double E[64 * 65];
    __shared__ double red[4];
    double m = 0.0;
    for (int t = threadIdx.x; t < n * n; t += 256) {
        const int i = t / n, j = t % n;
        const double e = S[t] - (i == j ? 1.0 : 0.0);
        E[i * 65 + j] = e;
        m = fmax(m, (fabs(e) <= 1.7e308) ? fabs(e) : 1.0);           // NaN / Inf: not near the identity
    }
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const double emax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (0.3125 * (double)n * (double)n * emax * emax * emax > 2.0e-17) {          // the dropped cubic term would be visible
        if (threadIdx.x == 0) flag[0] = 0;
        return;
    }
    for (int t = threadIdx.x; t < n * n; t += 256) {
        const int i = t / n, j = t % n;
        double e2 = 0.0;
        for (int k = 0; k < n; ++k) e2 = fma(E[i * 65 + k], E[k * 65 + j], e2);
        X[t] = (i == j ? 1.0 : 0.0) - 0.5 * E[i * 65 + j] + 0.375 * e2;
    }
    if (threadIdx.x == 0) flag[0] = 1;
}
// basis[env_idx[r]][nimp + c] = sum_j B[r][j] X[j][c]   (or B itself when X == nullptr)
__global__ void scatter_bath_kernel(int nenv, int nbath, const double *__restrict__ B, const double *__restrict__ X,
                                    const int *__restrict__ env_idx, int nimp, int ncol_basis,
                                    double *__restrict__ basis) {
    const long long total = (long long)nenv * nbath;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(t / nbath), c = (int)(t % nbath);
        double s;
        if (X) {
            s = 0.0;
            for (int j = 0; j < nbath; ++j) s += B[(long long)r * nbath + j] * X[j * nbath + c];
        } else {
            s = B[t];
        }
        if (nimp + c < ncol_basis) basis[(long long)env_idx[r] * ncol_basis + nimp + c] = s;
    }
}
__global__ void scatter_imp_kernel(int nimp, const int *__restrict__ imp_idx, int ncol_basis, double *__restrict__ basis) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nimp && i < ncol_basis) basis[(long long)imp_idx[i] * ncol_basis + i] = 1.0;
}

int nblocks_for(int nrows) {
    int nb = (nrows + 4 * NWAVE - 1) / (4 * NWAVE);
    return nb < 1 ? 1 : (nb > MAXB ? MAXB : nb);
}

}  // namespace



// TSQR path of dmk_bath_svd for `batch` matrices (the spin channels) that share the index maps: every launch carries the batch
// in a grid dimension, so the dependent chain of ~12 launches is paid once.  Returns 1 if handled, 0 if the caller must use the
// column-at-a-time path, < 0 on error.
static int bath_svd_tsqr(dmk_ctx *ctx, const int mesh[3], int nlo, int batch, const double *rdm1, long long rdm1_bstride,
                         const int32_t *env_idx, int nenv, const int32_t *bath_col, int nb, double *sigma, double *U) {
    static const bool tsqr_enabled = [] { const char *e = getenv("DMK_BATH_TSQR"); return !(e && atoi(e) == 0); }();
    if (!tsqr_enabled || 4 * nb > TS_MAXROWS) return 0;
    // ---- TSQR plan: leaves of m0 rows, then stacks of four R factors per node until one is left ----------------------
    const int m0 = nenv <= TS_MAXROWS ? nenv : std::min(TS_MAXROWS, std::max(TS_FANIN * nb, (nenv + 255) / 256));
    std::vector<int> nodes;                        // nodes[l] = workgroups of level l
    nodes.push_back((nenv + m0 - 1) / m0);
    while (nodes.back() > 1) nodes.push_back((nodes.back() + TS_FANIN - 1) / TS_FANIN);
    const int L = (int)nodes.size();
    size_t tot_nodes = 0;
    for (int n : nodes) tot_nodes += (size_t)n;
    // workspace per batch member (doubles): A | tau (all levels) | R stacks (output of every level) | X stacks (input of every
    // level's apply); then 2 status ints per member
    const size_t szA = (size_t)nenv * nb, szBlk = (size_t)nb * nb;
    const size_t per = szA + tot_nodes * nb + 2 * tot_nodes * szBlk;
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, (per * batch + 2 * (size_t)batch + 8) * sizeof(double), &ws);
    if (rc) return rc;
    double *A = reinterpret_cast<double *>(ws);
    double *tau = A + szA;
    double *Rst = tau + tot_nodes * nb;
    double *Xst = Rst + tot_nodes * szBlk;
    int *status = reinterpret_cast<int *>(A + per * batch);
    std::vector<size_t> off(L + 1, 0);             // node offset of level l in tau / Rst / Xst
    for (int l = 0; l < L; ++l) off[l + 1] = off[l] + (size_t)nodes[l];
    DMK_HIP(ctx, hipMemsetAsync(status, 0, 2 * (size_t)batch * sizeof(int), ctx->stream));
    {
        long long total = (long long)nenv * nb;
        int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(gather_env_imp_kernel, dim3(blocks, batch), dim3(256), 0, ctx->stream, mesh[0], mesh[1], mesh[2], nlo,
                           rdm1, env_idx, nenv, bath_col, nb, A, rdm1_bstride, (long long)per);
        DMK_CHECK_LAUNCH(ctx);
    }
    // up the tree: level l factors rows of (l == 0 ? A : the R stack written by level l - 1)
    for (int l = 0; l < L; ++l) {
        TsqrArgs a;
        a.nb = nb;
        a.rows_total = l == 0 ? nenv : nodes[l - 1] * nb;
        a.rows_per_node = l == 0 ? m0 : TS_FANIN * nb;
        a.M = l == 0 ? A : Rst + off[l - 1] * szBlk;
        a.tau = tau + off[l] * nb;
        a.Rout = Rst + off[l] * szBlk;
        a.bstride = (long long)per;
        const dim3 grid(nodes[l], batch);
        // register tile: 2 / 4 / 8 / 16 / 32 rows per thread (16 ... 256 rows per node)
        if (a.rows_per_node <= 64) hipLaunchKernelGGL(tsqr_factor_kernel<8>, grid, dim3(TS_NT), 0, ctx->stream, a);
        else if (a.rows_per_node <= 128) hipLaunchKernelGGL(tsqr_factor_kernel<16>, grid, dim3(TS_NT), 0, ctx->stream, a);
        else hipLaunchKernelGGL(tsqr_factor_kernel<32>, grid, dim3(TS_NT), 0, ctx->stream, a);
        DMK_CHECK_LAUNCH(ctx);
    }
    // SVD of the root R; its left singular vectors are the root's X block
    double *Rroot = Rst + off[L - 1] * szBlk, *Xroot = Xst + off[L - 1] * szBlk;
    hipLaunchKernelGGL(jacobi_svd_fast_kernel, dim3(batch), dim3(JF_NT), 0, ctx->stream, nb, Rroot, nb, sigma, Xroot, status, 1,
                       (long long)per, (long long)per);
    DMK_CHECK_LAUNCH(ctx);
    // down the tree: U = Q [U_r; 0]
    for (int l = L - 1; l >= 0; --l) {
        TsqrApplyArgs a;
        a.nb = nb; a.nc = nb;
        a.rows_total = l == 0 ? nenv : nodes[l - 1] * nb;
        a.rows_per_node = l == 0 ? m0 : TS_FANIN * nb;
        a.V = l == 0 ? A : Rst + off[l - 1] * szBlk;
        a.tau = tau + off[l] * nb;
        a.X = Xst + off[l] * szBlk;
        a.Y = l == 0 ? U : Xst + off[l - 1] * szBlk;
        a.bstride = (long long)per;
        a.y_bstride = l == 0 ? (long long)szA : (long long)per;
        const dim3 grid(nodes[l], batch);
        if (a.rows_per_node <= 64) hipLaunchKernelGGL(tsqr_apply_kernel<8>, grid, dim3(TS_NT), 0, ctx->stream, a);
        else if (a.rows_per_node <= 128) hipLaunchKernelGGL(tsqr_apply_kernel<16>, grid, dim3(TS_NT), 0, ctx->stream, a);
        else hipLaunchKernelGGL(tsqr_apply_kernel<32>, grid, dim3(TS_NT), 0, ctx->stream, a);
        DMK_CHECK_LAUNCH(ctx);
    }
    std::vector<int> st(2 * (size_t)batch, 0);
    DMK_HIP(ctx, hipMemcpyAsync(st.data(), status, st.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < batch; ++b)
        if (st[2 * b] != 0) return dmk_fail(ctx, DMK_ERR_NOCONV, "bath_svd: Jacobi SVD did not converge");
    if (getenv("DMK_BATH_TIMING"))
        fprintf(stderr, "[bath_svd %d x (%d x %d)] %d leaves of %d rows, %d levels, %d Jacobi sweeps\n", batch, nenv, nb, nodes[0], m0, L, st[1]);
    return 1;
}

extern "C" {

int dmk_bath_svd_batched(dmk_ctx *ctx, const int mesh[3], int nlo, int batch, const double *rdm1, int64_t rdm1_stride,
                         const int32_t *env_idx, int nenv, const int32_t *bath_col, int nb, double *sigma, double *U) {
    if (!ctx) return DMK_ERR_INVALID;
    if (!mesh || nlo <= 0 || batch <= 0 || !rdm1 || !env_idx || !bath_col || nenv <= 0 || nb <= 0 || !sigma || !U)
        return dmk_fail(ctx, DMK_ERR_INVALID, "bath_svd: bad arguments");
    if (nb > 120) return dmk_fail(ctx, DMK_ERR_INVALID, "bath_svd: nb = %d exceeds the supported maximum of 120", nb);
    if (nenv < nb) return dmk_fail(ctx, DMK_ERR_INVALID, "bath_svd: needs nenv >= nb (tall matrix)");
    {
        FamScope fs(ctx, DMK_FAM_BATH);
        int rc = bath_svd_tsqr(ctx, mesh, nlo, batch, rdm1, (long long)rdm1_stride, env_idx, nenv, bath_col, nb, sigma, U);
        if (rc != 0) return rc < 0 ? rc : DMK_OK;
    }
    for (int b = 0; b < batch; ++b) {
        int rc = dmk_bath_svd(ctx, mesh, nlo, rdm1 + (size_t)b * rdm1_stride, env_idx, nenv, bath_col, nb, sigma + (size_t)b * nb,
                              U + (size_t)b * nenv * nb);
        if (rc) return rc;
    }
    return DMK_OK;
}

int dmk_bath_svd(dmk_ctx *ctx, const int mesh[3], int nlo, const double *rdm1, const int32_t *env_idx, int nenv,
                 const int32_t *bath_col, int nb, double *sigma, double *U) {
    if (!ctx) return DMK_ERR_INVALID;
    if (!mesh || nlo <= 0 || !rdm1 || !env_idx || !bath_col || nenv <= 0 || nb <= 0 || !sigma || !U)
        return dmk_fail(ctx, DMK_ERR_INVALID, "bath_svd: bad arguments");
    if (nb > 120) return dmk_fail(ctx, DMK_ERR_INVALID, "bath_svd: nb = %d exceeds the supported maximum of 120", nb);
    if (nenv < nb) return dmk_fail(ctx, DMK_ERR_INVALID, "bath_svd: needs nenv >= nb (tall matrix)");
    FamScope fs(ctx, DMK_FAM_BATH);
    {
        int rc = bath_svd_tsqr(ctx, mesh, nlo, 1, rdm1, 0, env_idx, nenv, bath_col, nb, sigma, U);
        if (rc != 0) return rc < 0 ? rc : DMK_OK;
    }
    // column-at-a-time Householder QR (nb > 64)
    // workspace: A (nenv x nb) | partial (MAXB x nb) | tau (nb) | rowk (nb) | Utop (nb x nb) | status
    const size_t szA = (size_t)nenv * nb, szP = (size_t)MAXB * nb;
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, (szA + szP + 2 * (size_t)nb + (size_t)nb * nb + 8) * sizeof(double), &ws);
    if (rc) return rc;
    double *A = reinterpret_cast<double *>(ws);
    double *partial = A + szA;
    double *tau = partial + szP;
    double *rowk = tau + nb;
    double *Utop = rowk + nb;
    int *status = reinterpret_cast<int *>(Utop + (size_t)nb * nb);
    DMK_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int), ctx->stream));
    {
        long long total = (long long)nenv * nb;
        int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(gather_env_imp_kernel, dim3(blocks), dim3(256), 0, ctx->stream, mesh[0], mesh[1], mesh[2], nlo,
                           rdm1, env_idx, nenv, bath_col, nb, A);
        DMK_CHECK_LAUNCH(ctx);
    }
    // Householder QR, one column at a time
    for (int k = 0; k < nb; ++k) {
        const int nblk = nblocks_for(nenv - k);
        hipLaunchKernelGGL(col_dots_kernel, dim3(nblk), dim3(NT), NWAVE * nb * sizeof(double), ctx->stream, nenv, k, A,
                           nb, k, A, nb, k, nb, partial, 0, rowk);
        DMK_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(qr_apply_kernel, dim3(nblk), dim3(NT), 2 * nb * sizeof(double), ctx->stream, nenv, nb, k, A,
                           partial, nblk, rowk, tau);
        DMK_CHECK_LAUNCH(ctx);
    }
    // SVD of R
    {
        const int N = nb + (nb & 1);
        const size_t lds = ((size_t)N * (nb + 1) + N) * sizeof(double);
        if (lds > 48 * 1024)
            DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_svd_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(jacobi_svd_kernel, dim3(1), dim3(NT), lds, ctx->stream, nb, A, nb, sigma, Utop, status, 1,
                           (double *)nullptr);
        DMK_CHECK_LAUNCH(ctx);
    }
    // U = Q [Utop; 0]
    DMK_HIP(ctx, hipMemsetAsync(U, 0, szA * sizeof(double), ctx->stream));
    DMK_HIP(ctx, hipMemcpyAsync(U, Utop, (size_t)nb * nb * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    for (int k = nb - 1; k >= 0; --k) {
        const int nblk = nblocks_for(nenv - k);
        hipLaunchKernelGGL(col_dots_kernel, dim3(nblk), dim3(NT), NWAVE * nb * sizeof(double), ctx->stream, nenv, k, A,
                           nb, k, U, nb, 0, nb, partial, 1, (double *)nullptr);
        DMK_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(q_apply_kernel, dim3(nblk), dim3(NT), nb * sizeof(double), ctx->stream, nenv, nb, nb, k, A,
                           tau, U, partial, nblk);
        DMK_CHECK_LAUNCH(ctx);
    }
    int st = 0;
    DMK_HIP(ctx, hipMemcpyAsync(&st, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st != 0) return dmk_fail(ctx, DMK_ERR_NOCONV, "bath_svd: Jacobi SVD did not converge");
    return DMK_OK;
}

int dmk_bath_assemble(dmk_ctx *ctx, const double *U, int nenv, int nb, int nbath, const int32_t *virt_mask, int orth,
                      const int32_t *env_idx, const int32_t *imp_idx, int nimp, int nsites, int ncol_basis,
                      double *basis) {
    if (!ctx) return DMK_ERR_INVALID;
    if (!U || nenv <= 0 || nb <= 0 || nbath < 0 || nbath > nb || !env_idx || !imp_idx || nimp < 0 || nsites <= 0 ||
        ncol_basis <= 0 || !basis || (orth && !virt_mask))
        return dmk_fail(ctx, DMK_ERR_INVALID, "bath_assemble: bad arguments");
    DMK_HIP(ctx, hipMemsetAsync(basis, 0, (size_t)nsites * ncol_basis * sizeof(double), ctx->stream));
    if (nimp > 0) {
        FamScope fs(ctx, DMK_FAM_BATH);
        hipLaunchKernelGGL(scatter_imp_kernel, dim3((nimp + 255) / 256), dim3(256), 0, ctx->stream, nimp, imp_idx,
                           ncol_basis, basis);
        DMK_CHECK_LAUNCH(ctx);
    }
    if (nbath == 0) return DMK_OK;
    const int GB = 128;
    const size_t szB = (size_t)nenv * nbath, szS = (size_t)nbath * nbath;
    // the context's SECOND scratch: the eigensolver below uses the first one (round 2 paid a hipMalloc / hipFree pair and a
    // device synchronisation here on every call)
    void *ws2 = nullptr;
    int rc = dmk_scratch2(ctx, (szB + (size_t)GB * szS + 3 * szS + nbath + 2) * sizeof(double), &ws2);
    if (rc) return rc;
    double *wsd = reinterpret_cast<double *>(ws2);
    double *B = wsd, *partial = B + szB, *S = partial + (size_t)GB * szS, *Vt = S + szS, *X = Vt + szS, *ev = X + szS;
    auto cleanup = [&]() {};
    {
        FamScope fs(ctx, DMK_FAM_BATH);
        const int blocks = (int)std::min<long long>(((long long)szB + 255) / 256, 4096);
        hipLaunchKernelGGL(mask_copy_kernel, dim3(blocks), dim3(256), 0, ctx->stream, nenv, nb, nbath, U, virt_mask,
                           orth ? 1 : 0, B);
        if (orth) {
            if (nbath <= 64) hipLaunchKernelGGL(gram_partial_lds_kernel, dim3(GB), dim3(NT), 0, ctx->stream, nenv, nbath, B, partial);
            else hipLaunchKernelGGL(gram_partial_kernel, dim3(GB), dim3(NT), 0, ctx->stream, nenv, nbath, B, partial);
            hipLaunchKernelGGL(reduce_partials_kernel, dim3(std::min<int>(64, (int)((szS + 255) / 256))), dim3(256), 0, ctx->stream, (int)szS,
                               GB, partial, S);
        }
    }
    if (orth && nbath <= 64) {
        // eigenpairs of the symmetric positive semi-definite metric: its singular vectors ARE its eigenvectors and the one-sided
        // Jacobi kernel delivers them (with high relative accuracy for the small eigenvalues the 1e-14 cut looks at) in ~0.15 ms;
        // the general batched eigensolver needs 0.55 ms for one 56 x 56 matrix (latency chain of its three phases).
        // The kernel stores the vectors as COLUMNS of its output (here the `Vt` buffer): inv_sqrt_cols_kernel reads that layout.
        FamScope fs(ctx, DMK_FAM_BATH);
        int *st = reinterpret_cast<int *>(ev + nbath);
        DMK_HIP(ctx, hipMemsetAsync(st, 0, 4 * sizeof(int), ctx->stream));
        hipLaunchKernelGGL(lowdin_taylor_kernel, dim3(1), dim3(256), 0, ctx->stream, nbath, S, X, st + 2);
        hipLaunchKernelGGL(jacobi_svd_fast_kernel, dim3(1), dim3(JF_NT), 0, ctx->stream, nbath, S, nbath, ev, Vt, st, 0, 0LL, 0LL, st + 2);
        hipLaunchKernelGGL(inv_sqrt_cols_kernel, dim3(16), dim3(256), 0, ctx->stream, nbath, ev, Vt, 1e-14, X, st + 2);
    } else if (orth) {
        rc = launch_eigh_public(ctx, nbath, 1, S, 1, nullptr, 0, ev, Vt, 1);
        if (rc) { cleanup(); return rc; }
    }
    {
        FamScope fs(ctx, DMK_FAM_BATH);
        if (orth && nbath > 64)
            hipLaunchKernelGGL(inv_sqrt_kernel, dim3(16), dim3(256), 0, ctx->stream, nbath, ev, Vt, 1e-14, X);
        const int blocks = (int)std::min<long long>(((long long)szB + 255) / 256, 4096);
        hipLaunchKernelGGL(scatter_bath_kernel, dim3(blocks), dim3(256), 0, ctx->stream, nenv, nbath, B,
                           orth ? X : (const double *)nullptr, env_idx, nimp, ncol_basis, basis);
    }
    hipError_t le = hipGetLastError();
    cleanup();
    if (le != hipSuccess) return dmk_fail(ctx, DMK_ERR_HIP, "bath_assemble: launch failed: %s", hipGetErrorString(le));
    if (orth && nbath <= 64) {
        int st = 0;
        DMK_HIP(ctx, hipMemcpyAsync(&st, ev + nbath, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (st != 0) return dmk_fail(ctx, DMK_ERR_NOCONV, "bath_assemble: Jacobi eigensolver of the Loewdin metric did not converge");
    }
    return rc;
}


int dmk_svd_small(dmk_ctx *ctx, int n, const double *A, double *sigma, double *U, double *Vt) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n <= 0 || !A || !sigma || !U || !Vt) return dmk_fail(ctx, DMK_ERR_INVALID, "svd_small: bad arguments");
    if (n > 84) return dmk_fail(ctx, DMK_ERR_INVALID, "svd_small: n = %d exceeds the supported maximum of 84", n);
    FamScope fs(ctx, DMK_FAM_BATH);
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, 256, &ws);
    if (rc) return rc;
    int *status = reinterpret_cast<int *>(ws);
    DMK_HIP(ctx, hipMemsetAsync(status, 0, sizeof(int), ctx->stream));
    const int N = n + (n & 1);
    const size_t lds = (2 * (size_t)N * (n + 1) + N) * sizeof(double);
    if (lds > 48 * 1024)
        DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_svd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(jacobi_svd_kernel, dim3(1), dim3(NT), lds, ctx->stream, n, A, n, sigma, U, status, 0, Vt);
    DMK_CHECK_LAUNCH(ctx);
    int st = 0;
    DMK_HIP(ctx, hipMemcpyAsync(&st, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st != 0) return dmk_fail(ctx, DMK_ERR_NOCONV, "svd_small: Jacobi SVD did not converge");
    return DMK_OK;
}

int dmk_dgemm_nn_small(dmk_ctx *ctx, int64_t M, int N, int K, const double *A, const double *B, int transB,
                       double *C) {
    if (!ctx) return DMK_ERR_INVALID;
    if (M < 0 || N <= 0 || K <= 0 || !A || !B || !C) return dmk_fail(ctx, DMK_ERR_INVALID, "dgemm_nn_small: bad arguments");
    if (M == 0) return DMK_OK;
    FamScope fs(ctx, DMK_FAM_MISC);
    const int blocks = (int)std::min<long long>((M * N + 255) / 256, 8192);
    hipLaunchKernelGGL(rowmat_small_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (long long)M, N, K, A, B, transB, C);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_bcs_weight(dmk_ctx *ctx, int ncell, int period, int keep, int nb, const double *U, double *w) {
    if (!ctx) return DMK_ERR_INVALID;
    if (ncell < 0 || period <= 0 || keep < 0 || keep > period || nb <= 0 || !U || !w)
        return dmk_fail(ctx, DMK_ERR_INVALID, "bcs_weight: bad arguments");
    FamScope fs(ctx, DMK_FAM_BATH);
    hipLaunchKernelGGL(bcs_weight_kernel, dim3(nb), dim3(NT), 0, ctx->stream, ncell, period, keep, nb, U, w);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_bcs_assemble(dmk_ctx *ctx, int ncells, int n, int nval, const double *U, const int *order, double *basis) {
    if (!ctx) return DMK_ERR_INVALID;
    if (ncells <= 0 || n <= 0 || nval <= 0 || nval > n || !U || !order || !basis)
        return dmk_fail(ctx, DMK_ERR_INVALID, "bcs_assemble: bad arguments");
    FamScope fs(ctx, DMK_FAM_BATH);
    const long long total = 2LL * ncells * 2 * n * (n + nval);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(bcs_assemble_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ncells, n, nval, U, order, basis);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_pad_block_f64(dmk_ctx *ctx, int batch, int64_t r_in, int64_t c_in, const double *in, int64_t r_out,
                      int64_t c_out, double *out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (batch < 0 || r_in < 0 || c_in < 0 || r_out < r_in || c_out < c_in || !in || !out)
        return dmk_fail(ctx, DMK_ERR_INVALID, "pad_block: bad arguments");
    const long long total = (long long)batch * r_out * c_out;
    if (total == 0) return DMK_OK;
    FamScope fs(ctx, DMK_FAM_MISC);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(pad_block_kernel, dim3(blocks), dim3(256), 0, ctx->stream, batch, (long long)r_in, (long long)c_in,
                       in, (long long)r_out, (long long)c_out, out);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

}  // extern "C"
