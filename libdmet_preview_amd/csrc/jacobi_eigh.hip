// K13 -- low-latency symmetric eigensolver for a FEW small real matrices: parallel block Jacobi over several CUs.
//
// routine/slater.py:1075, 1098 call la.eigh(embHeff[s], ovlp_emb[s]) once per objective / gradient evaluation of the
// vcor fit, hundreds to thousands of times in sequence on a (spin, nemb, nemb) stack (2 x 256 x 256 at C5).  The
// batched Householder + QL kernel (eigh.hip) is built for throughput -- one workgroup per matrix, 27 ms of latency
// at n = 256 -- and its QL phase is a sequential recurrence.  Here latency is what counts:
//
//   * one-sided (Hestenes) Jacobi on M = A + c I (c >= |A|_F makes M positive definite, so the singular vectors of M
//     ARE the eigenvectors and nothing mixes +l with -l): rows g_j of G = V M are rotated in pairs until they are
//     mutually orthogonal; then g_j = (l_j + c) v_j, l_j = g_j . v_j - c.
//   * the n vectors are cut into blocks of 16; a workgroup owns a PAIR of blocks for one step (both blocks of G and V
//     in LDS, 128 KB at n = 256), rotates all cross pairs (16 inner steps of 16 disjoint pairs, one pair per wave at a
//     time, lanes along the vector, DPP reductions), and the block pairs follow a round-robin tournament: n/32
//     workgroups per matrix, n/16 - 1 device-wide hand-overs per sweep instead of n - 1.
//   * WARM START: successive fit matrices differ by a line-search step, so the previous eigenvectors V0 make
//     G = V0 M almost orthogonal already and 2-3 sweeps (instead of 8-10) reach the same tolerance.
//
// Hand-over between workgroups of one matrix is an agent-scope release / acquire on a monotonic counter
// (__threadfence: L2 write-back + invalidate, the XCDs do not share an L2).  All workgroups of the launch are
// co-resident (batch * n/32 <= 256 CUs is checked).  Bit-reproducible: fixed pair order, no atomics on data.
#include "common.h"

namespace {

constexpr int JT = 256;
constexpr int JWV = JT / 64;

struct JacArgs {
    int n;            // padded dimension (multiple of 2 b)
    int batch, b, nb, max_sweeps;
    double *G, *V;    // batch x n x n, row j = vector j
    unsigned *bar;    // one monotonic counter per matrix
    int *abort_flag;  // set when a hand-over timed out
    int *flags;       // batch x max_sweeps: "a rotation happened in this sweep"
    int *sweeps_done; // batch
    double tol;
};

// Returns false when the hand-over timed out (a workgroup of the matrix is not running: the launch was not co-resident).
__device__ __forceinline__ bool matrix_barrier(unsigned *bar, unsigned target, int *abort_flag) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: this workgroup's rows are visible device-wide
        atomicAdd(bar, 1u);
        long long spins = 0;
        int good = 1;
        while (atomicAdd(bar, 0u) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1LL << 26) || atomicOr(abort_flag, 0)) {
                atomicOr(abort_flag, 1);
                good = 0;
                break;
            }
        }
        __threadfence();                                   // acquire
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

// NP Hestenes rotations of disjoint row pairs (rp[k], rq[k]) by one wave, lanes along the vectors (LDS images Gs / Vs,
// leading dimension n).  The 3 NP dot products and their DPP reductions are issued together so that the reductions and
// the sqrt / div chains of the NP pairs overlap.  Returns true if any pair was rotated.
template <int NP>
__device__ __forceinline__ bool rotate_pairs(double *Gs, double *Vs, int n, const int (&rp)[NP], const int (&rq)[NP], double tol,
                                             int lane) {
    double a[NP], b[NP], c[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) a[k] = b[k] = c[k] = 0.0;
    for (int i = lane; i < n; i += 64) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const double x = Gs[(size_t)rp[k] * n + i], y = Gs[(size_t)rq[k] * n + i];
            a[k] = fma(x, x, a[k]);
            b[k] = fma(y, y, b[k]);
            c[k] = fma(x, y, c[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        a[k] = dmk_wave_sum(a[k]);
        b[k] = dmk_wave_sum(b[k]);
        c[k] = dmk_wave_sum(c[k]);
    }
    double cs[NP], sn[NP];
    bool rot[NP], any = false;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        // |c| > tol sqrt(a b)  <=>  c^2 > tol^2 a b ; reciprocals and reciprocal square roots from the hardware
        // approximations + two Newton steps (full double precision, a fraction of the IEEE division / sqrt sequences)
        rot[k] = (c[k] * c[k] > tol * tol * (a[k] * b[k])) && c[k] != 0.0;
        const double cc = rot[k] ? c[k] : 1.0;
        const double zeta = 0.5 * (b[k] - a[k]) * fast_rcp(cc);
        const double h2 = fma(zeta, zeta, 1.0);
        const double hyp = h2 * fast_rsqrt(h2);                 // sqrt(1 + zeta^2)
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) * fast_rcp(fabs(zeta) + hyp);
        cs[k] = fast_rsqrt(fma(t, t, 1.0));
        sn[k] = cs[k] * t;
        any |= rot[k];
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (!rot[k]) continue;                               // wave-uniform
        double *gp = Gs + (size_t)rp[k] * n, *gq = Gs + (size_t)rq[k] * n;
        double *vp = Vs + (size_t)rp[k] * n, *vq = Vs + (size_t)rq[k] * n;
        for (int i = lane; i < n; i += 64) {
            const double x = gp[i], y = gq[i];
            gp[i] = cs[k] * x - sn[k] * y;
            gq[i] = sn[k] * x + cs[k] * y;
            const double u = vp[i], w = vq[i];
            vp[i] = cs[k] * u - sn[k] * w;
            vq[i] = sn[k] * u + cs[k] * w;
        }
    }
    return any;
}

template <int B>
__global__ __launch_bounds__(JT) void jacobi_eigh_kernel(JacArgs g) {
    extern __shared__ double sh[];
    constexpr int b = B, NP = B / JWV;
    const int n = g.n, nb = g.nb, nwg = nb / 2;
    double *Gs = sh;                       // [2b][n]
    double *Vs = sh + (size_t)2 * b * n;   // [2b][n]
    const int mat = blockIdx.x / nwg, w = blockIdx.x % nwg;
    double *G = g.G + (size_t)mat * n * n, *V = g.V + (size_t)mat * n * n;
    unsigned *bar = g.bar + mat;
    int *flags = g.flags + (size_t)mat * g.max_sweeps;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned epoch = 0;

    auto load_blocks = [&](int P, int Q) {
        const size_t rows = (size_t)b * n;
        for (size_t t = tid; t < rows; t += JT) {
            Gs[t] = G[(size_t)P * rows + t];
            Gs[rows + t] = G[(size_t)Q * rows + t];
            Vs[t] = V[(size_t)P * rows + t];
            Vs[rows + t] = V[(size_t)Q * rows + t];
        }
        __syncthreads();
    };
    auto store_blocks = [&](int P, int Q) {
        __syncthreads();
        const size_t rows = (size_t)b * n;
        for (size_t t = tid; t < rows; t += JT) {
            G[(size_t)P * rows + t] = Gs[t];
            G[(size_t)Q * rows + t] = Gs[rows + t];
            V[(size_t)P * rows + t] = Vs[t];
            V[(size_t)Q * rows + t] = Vs[rows + t];
        }
    };

    int sweep = 0;
    for (; sweep < g.max_sweeps; ++sweep) {
        bool rotated = false;
        // ---- pairs inside the two blocks this workgroup starts with (round robin inside each block) ----
        load_blocks(2 * w, 2 * w + 1);
        for (int r = 0; r < b - 1; ++r) {
            int rp[NP], rq[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {                 // b/2 pairs per block, two blocks: b pairs, NP per wave
                const int pi = wave + JWV * k;
                const int blk = pi / (b / 2), q0 = pi % (b / 2);
                int p, q;
                if (q0 == 0) { p = b - 1; q = r; }
                else { p = (r + q0) % (b - 1); q = (r - q0 + (b - 1)) % (b - 1); }
                rp[k] = blk * b + p;
                rq[k] = blk * b + q;
            }
            rotated |= rotate_pairs<NP>(Gs, Vs, n, rp, rq, g.tol, lane);
            __syncthreads();
        }
        store_blocks(2 * w, 2 * w + 1);
        ++epoch;
        if (!matrix_barrier(bar, epoch * nwg, g.abort_flag)) return;
        // ---- cross pairs: round-robin tournament over the blocks ----
        for (int t = 0; t < nb - 1; ++t) {
            int P, Q;
            if (w == 0) { P = nb - 1; Q = t; }
            else { P = (t + w) % (nb - 1); Q = (t - w + (nb - 1)) % (nb - 1); }
            load_blocks(P, Q);
            for (int s = 0; s < b; ++s) {
                int rp[NP], rq[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int i = wave + JWV * k;
                    rp[k] = i;
                    rq[k] = b + (i + s) % b;
                }
                rotated |= rotate_pairs<NP>(Gs, Vs, n, rp, rq, g.tol, lane);
                __syncthreads();
            }
            store_blocks(P, Q);
            if (t == nb - 2 && __syncthreads_or(rotated ? 1 : 0) && tid == 0) atomicOr(&flags[sweep], 1);
            ++epoch;
            if (!matrix_barrier(bar, epoch * nwg, g.abort_flag)) return;
        }
        int any = 0;
        if (tid == 0) any = atomicOr(&flags[sweep], 0);
        any = __syncthreads_or(any);
        if (!any) { ++sweep; break; }
    }
    if (w == 0 && tid == 0) g.sweeps_done[mat] = sweep;
}

// Shift c >= spectral radius of the symmetric matrix Bm (n0 x n0): min(Gershgorin row sum, Frobenius norm), rigorous.
// from_lower: only the lower triangle of Bm is valid (cold start on A); otherwise Bm = V0 A V0^T is stored in full and is
// nearly diagonal, which makes the bound tight (a tight shift keeps the eigenvector error at eps |A| instead of eps |A|_F).
__global__ __launch_bounds__(JT) void jacobi_shift_kernel(int n0, int from_lower, const double *__restrict__ Bm,
                                                          double *__restrict__ cshift) {
    __shared__ double red[2 * JWV];
    const int mat = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *a = Bm + (size_t)mat * n0 * n0;
    double fro = 0.0, gmax = 0.0;
    for (int r = wave; r < n0; r += JWV) {
        double rs = 0.0;
        for (int c = lane; c < n0; c += 64) {
            const double v = from_lower ? a[(size_t)(r >= c ? r : c) * n0 + (r >= c ? c : r)] : a[(size_t)r * n0 + c];
            rs += fabs(v);
            fro = fma(v, v, fro);
        }
        rs = dmk_wave_sum(rs);
        gmax = fmax(gmax, rs);
    }
    fro = dmk_wave_sum(fro);
    if (lane == 0) { red[wave] = fro; red[JWV + wave] = gmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double f = sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const double gsh = fmax(fmax(red[JWV], red[JWV + 1]), fmax(red[JWV + 2], red[JWV + 3]));
        const double bound = fmin(f, gsh);
        cshift[mat] = bound > 0.0 ? 1.015625 * bound : 1.0;
    }
}

__global__ void jacobi_init_kernel(int n0, int n, int batch, const double *__restrict__ A, const double *__restrict__ V0A,
                                   const double *__restrict__ V0, const double *__restrict__ cshift, double *__restrict__ G,
                                   double *__restrict__ V) {
    const long long total = (long long)batch * n * n;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int mat = (int)(t / ((long long)n * n));
        const int r = (int)((t / n) % n), col = (int)(t % n);
        const double c = cshift[mat];
        double gv, vv;
        if (r < n0 && col < n0) {
            const size_t o = (size_t)mat * n0 * n0 + (size_t)r * n0 + col;
            if (V0) {
                vv = V0[o];
                gv = V0A[o] + c * vv;                      // row r of V0 (A + c I)
            } else {
                vv = (r == col) ? 1.0 : 0.0;
                gv = A[(size_t)mat * n0 * n0 + (size_t)(r >= col ? r : col) * n0 + (r >= col ? col : r)] + (r == col ? c : 0.0);
            }
        } else {
            vv = (r == col) ? 1.0 : 0.0;
            gv = (r == col) ? c : 0.0;
        }
        G[t] = gv;
        V[t] = vv;
    }
}

// Vu (n0 x n0, contiguous) = the converged vectors without the padding
__global__ void jacobi_gather_kernel(int n0, int n, int batch, const double *__restrict__ V, double *__restrict__ Vu) {
    const long long total = (long long)batch * n0 * n0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int mat = (int)(t / ((long long)n0 * n0));
        const int r = (int)((t / n0) % n0), c = (int)(t % n0);
        Vu[t] = V[(size_t)mat * n * n + (size_t)r * n + c];
    }
}

// eigenvalues as Rayleigh quotients l_j = (v_j A) . v_j (error eps |A|, second order in the vector error), ascending
// stable rank, sorted output (rows of Vt = eigenvectors)
__global__ __launch_bounds__(JT) void jacobi_finish_kernel(int n0, const double *__restrict__ Vu, const double *__restrict__ VA,
                                                           double *__restrict__ w, double *__restrict__ Vt) {
    extern __shared__ double ev[];           // [n0] + ranks
    const int mat = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *Vm = Vu + (size_t)mat * n0 * n0, *Tm = VA + (size_t)mat * n0 * n0;
    double *inrm = ev + n0;                  // 1 / |v_j|: thousands of rotations leave |v_j| = 1 + O(1e-13)
    for (int j = wave; j < n0; j += JWV) {
        double s = 0.0, q = 0.0;
        for (int i = lane; i < n0; i += 64) {
            const double v = Vm[(size_t)j * n0 + i];
            s = fma(Tm[(size_t)j * n0 + i], v, s);
            q = fma(v, v, q);
        }
        s = dmk_wave_sum(s);
        q = dmk_wave_sum(q);
        if (lane == 0) {
            ev[j] = s / q;
            inrm[j] = 1.0 / sqrt(q);
        }
    }
    __syncthreads();
    int *rank = reinterpret_cast<int *>(ev + 2 * n0);
    for (int j = threadIdx.x; j < n0; j += JT) {
        const double dj = ev[j];
        int rk = 0;
        for (int q = 0; q < n0; ++q) rk += (ev[q] < dj || (ev[q] == dj && q < j)) ? 1 : 0;
        w[(size_t)mat * n0 + rk] = dj;
        rank[j] = rk;
    }
    __syncthreads();
    for (int j = wave; j < n0; j += JWV) {
        const size_t orow = (size_t)mat * n0 * n0 + (size_t)rank[j] * n0;
        for (int i = lane; i < n0; i += 64) Vt[orow + i] = Vm[(size_t)j * n0 + i] * inrm[j];
    }
}

// full symmetric copy of a lower-triangle-valid matrix
__global__ void jacobi_symm_kernel(int n0, int batch, const double *__restrict__ A, double *__restrict__ Af) {
    const long long total = (long long)batch * n0 * n0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long mat = t / ((long long)n0 * n0);
        const int r = (int)((t / n0) % n0), c = (int)(t % n0);
        Af[t] = A[mat * n0 * n0 + (size_t)(r >= c ? r : c) * n0 + (r >= c ? c : r)];
    }
}

}  // namespace

int launch_dgemm_small_nn(dmk_ctx *ctx, int M, int N, int K, int batch, const double *A, const double *B, double *C);

extern "C" {

int dmk_eigh_jacobi_real(dmk_ctx *ctx, int n, int batch, const double *A, const double *V0, double *w, double *Vt,
                         int *sweeps_out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n <= 0 || batch <= 0 || !A || !w || !Vt) return dmk_fail(ctx, DMK_ERR_INVALID, "eigh_jacobi: bad arguments");
    int b = 16;
    int npad = ((n + 2 * b - 1) / (2 * b)) * (2 * b);
    if ((size_t)4 * b * npad * 8 > 144 * 1024) {
        b = 8;
        npad = ((n + 2 * b - 1) / (2 * b)) * (2 * b);
    }
    const size_t lds = (size_t)4 * b * npad * sizeof(double);
    if (lds > 144 * 1024) return dmk_fail(ctx, DMK_ERR_INVALID, "eigh_jacobi: n = %d exceeds the supported maximum of 576", n);
    const int nb = npad / b, nwg = nb / 2;
    if ((long long)batch * nwg > 256)
        return dmk_fail(ctx, DMK_ERR_INVALID, "eigh_jacobi: batch * n/32 = %lld workgroups would not be co-resident; use "
                        "dmk_eigh_batched_real for large batches", (long long)batch * nwg);
    FamScope fs(ctx, DMK_FAM_EIGH);
    const int max_sweeps = 40;
    const size_t nn = (size_t)npad * npad;
    const size_t b_mat = ((nn * 8 * batch) + 255) & ~(size_t)255;
    const size_t b_tmp = (((size_t)n * n * 8 * batch) + 255) & ~(size_t)255;
    void *ws = nullptr;
    int rc = dmk_scratch(ctx, 2 * b_mat + 3 * b_tmp + 8192 + (size_t)batch * (max_sweeps + 4) * 8, &ws);
    if (rc) return rc;
    char *p = static_cast<char *>(ws);
    double *G = reinterpret_cast<double *>(p); p += b_mat;
    double *V = reinterpret_cast<double *>(p); p += b_mat;
    double *T1 = reinterpret_cast<double *>(p); p += b_tmp;      // V0 A, later Vu
    double *T2 = reinterpret_cast<double *>(p); p += b_tmp;      // V0 A V0^T, later Vu A
    double *Af = reinterpret_cast<double *>(p); p += b_tmp;      // full symmetric A (cold start)
    double *cshift = reinterpret_cast<double *>(p); p += ((size_t)batch * 8 + 255) & ~(size_t)255;
    unsigned *bar = reinterpret_cast<unsigned *>(p); p += ((size_t)batch * 4 + 255) & ~(size_t)255;
    int *sweeps_done = reinterpret_cast<int *>(p); p += ((size_t)batch * 4 + 255) & ~(size_t)255;
    int *abort_flag = reinterpret_cast<int *>(p); p += 256;
    int *flags = reinterpret_cast<int *>(p);
    DMK_HIP(ctx, hipMemsetAsync(bar, 0, (size_t)(p - reinterpret_cast<char *>(bar)) + (size_t)batch * max_sweeps * 4, ctx->stream));
    const long long tot0 = (long long)batch * n * n;
    const unsigned g0 = (unsigned)std::min<long long>((tot0 + 255) / 256, 8192);
    const double *Afull = A;
    if (V0) {
        rc = launch_dgemm_small_nn(ctx, n, n, n, batch, V0, A, T1);        // rows of V0 A  (A symmetric, full storage)
        if (rc) return rc;
        rc = dmk_dgemm_batched(ctx, 0, 1, n, n, n, batch, 1.0, T1, n, (int64_t)n * n, V0, n, (int64_t)n * n, 0.0, T2, n,
                               (int64_t)n * n);                           // V0 A V0^T: nearly diagonal
        if (rc) return rc;
        hipLaunchKernelGGL(jacobi_shift_kernel, dim3(batch), dim3(JT), 0, ctx->stream, n, 0, T2, cshift);
    } else {
        hipLaunchKernelGGL(jacobi_symm_kernel, dim3(g0), dim3(256), 0, ctx->stream, n, batch, A, Af);
        Afull = Af;
        hipLaunchKernelGGL(jacobi_shift_kernel, dim3(batch), dim3(JT), 0, ctx->stream, n, 1, A, cshift);
    }
    const long long total = (long long)batch * nn;
    hipLaunchKernelGGL(jacobi_init_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, ctx->stream,
                       n, npad, batch, A, V0 ? T1 : (const double *)nullptr, V0, cshift, G, V);
    DMK_CHECK_LAUNCH(ctx);
    JacArgs g;
    g.n = npad; g.batch = batch; g.b = b; g.nb = nb; g.max_sweeps = max_sweeps;
    g.G = G; g.V = V; g.bar = bar; g.flags = flags; g.sweeps_done = sweeps_done; g.abort_flag = abort_flag;
    g.tol = sqrt((double)npad) * 2.220446049250313e-16;
    if (b == 16) {
        DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_eigh_kernel<16>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(jacobi_eigh_kernel<16>, dim3(batch * nwg), dim3(JT), lds, ctx->stream, g);
    } else {
        DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(jacobi_eigh_kernel<8>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(jacobi_eigh_kernel<8>, dim3(batch * nwg), dim3(JT), lds, ctx->stream, g);
    }
    DMK_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(jacobi_gather_kernel, dim3(g0), dim3(256), 0, ctx->stream, n, npad, batch, V, T1);
    rc = launch_dgemm_small_nn(ctx, n, n, n, batch, T1, Afull, T2);        // rows v_j A
    if (rc) return rc;
    hipLaunchKernelGGL(jacobi_finish_kernel, dim3(batch), dim3(JT), (size_t)n * 16 + (size_t)n * 4 + 16, ctx->stream, n, T1, T2, w, Vt);
    DMK_CHECK_LAUNCH(ctx);
    std::vector<int> sw(batch);
    int aborted = 0;
    DMK_HIP(ctx, hipMemcpyAsync(sw.data(), sweeps_done, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipMemcpyAsync(&aborted, abort_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (aborted) return dmk_fail(ctx, DMK_ERR_STATE, "eigh_jacobi: workgroup hand-over timed out (launch not co-resident)");
    int worst = 0;
    for (int i = 0; i < batch; ++i) worst = std::max(worst, sw[i]);
    if (sweeps_out) *sweeps_out = worst;
    if (worst >= max_sweeps) return dmk_fail(ctx, DMK_ERR_NOCONV, "eigh_jacobi: no convergence in %d sweeps", max_sweeps);
    return DMK_OK;
}

}  // extern "C"
