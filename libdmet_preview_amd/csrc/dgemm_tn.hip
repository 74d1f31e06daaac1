// K7 -- ERI contraction  C (M x N) += alpha * X^T Y  on the f64 matrix cores.
//
// Replaces the lib.dot(Lij.T, Lij, alpha, eri, 1) calls of
// basis_transform/eri_transform.py:455-476 (`_Lij_s4_to_eri`).  X and Y are the
// tril-packed (L|ab) planes, K x M and K x N row-major (K = naux or 2*naux with
// the Re and Im planes stacked), so both MFMA operands are "K-major with unit
// stride along the tile edge": global rows stream straight into an LDS image
// [k][m] with no transpose, and a fragment read is 16 consecutive doubles.
//
// Tiling (CDNA4, wave64): 256 threads = 2 x 2 waves, workgroup tile 128 x 128,
// wave tile 64 x 64 = 4 x 4 v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs),
// BK = 16, double-buffered LDS (2 x 2 x 16 x 144 x 8 B = 72 KiB -> 2 workgroups
// per CU = 2 waves per SIMD, which the f64 matrix pipe needs to stay busy).
// Rows of the LDS image are padded to 144 doubles so that the two k-rows a
// ds_read_b64 lane-group touches fall on disjoint bank halves.
//
// Roofline: f64 MFMA: 2*K*M*N flop per call against
// 8*(K*(M+N) + 2*M*N) bytes.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_LD = BM + 16;          // 144 doubles: row stride = 128 B mod 256 B
constexpr int NTHREADS = 256;

template <bool VEC2>
__global__ __launch_bounds__(NTHREADS, 2) void dgemm_tn_acc_kernel(
    int M, int N, int K, double alpha, const double *__restrict__ X, int64_t ldx,
    const double *__restrict__ Y, int64_t ldy, double *__restrict__ C, int64_t ldc,
    int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * BK * LDS_LD];
    double *As = lds;                       // [2][BK][LDS_LD]
    double *Bs = lds + 2 * BK * LDS_LD;     // [2][BK][LDS_LD]

    // ---- tile selection: XCD-contiguous, grouped (8 tile-rows per group) -------------
    const unsigned nblocks = (unsigned)tiles_m * (unsigned)tiles_n;
    const unsigned lid = xcd_remap(blockIdx.x, nblocks);
    constexpr unsigned GROUP = 8;
    const unsigned per_group = GROUP * (unsigned)tiles_n;
    const unsigned g = lid / per_group;
    const unsigned first_m = g * GROUP;
    const unsigned gsize = min((unsigned)tiles_m - first_m, GROUP);
    const unsigned in_g = lid - g * per_group;
    const int tm = (int)(first_m + in_g % gsize);
    const int tn = (int)(in_g / gsize);
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- global -> register staging --------------------------------------------------
    // slab = BK x 128 doubles per operand = 1024 double2; 4 double2 per thread per operand
    constexpr int PER = (BK * BM / 2) / NTHREADS;   // 4
    double2 ra[PER], rb[PER];
    int lk[PER], lc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int e = tid + i * NTHREADS;
        lk[i] = e / (BM / 2);
        lc[i] = (e % (BM / 2)) * 2;
    }

    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int k = k0 + lk[i];
            const int mm = m0 + lc[i], nn = n0 + lc[i];
            double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
            if (k < K) {
                const double *px = X + (int64_t)k * ldx + mm;
                const double *py = Y + (int64_t)k * ldy + nn;
                if (VEC2) {
                    if (mm + 1 < M) va = *reinterpret_cast<const double2 *>(px);
                    else if (mm < M) va.x = px[0];
                    if (nn + 1 < N) vb = *reinterpret_cast<const double2 *>(py);
                    else if (nn < N) vb.x = py[0];
                } else {
                    if (mm < M) va.x = px[0];
                    if (mm + 1 < M) va.y = px[1];
                    if (nn < N) vb.x = py[0];
                    if (nn + 1 < N) vb.y = py[1];
                }
            }
            ra[i] = va;
            rb[i] = vb;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            *reinterpret_cast<double2 *>(&As[(buf * BK + lk[i]) * LDS_LD + lc[i]]) = ra[i];
            *reinterpret_cast<double2 *>(&Bs[(buf * BK + lk[i]) * LDS_LD + lc[i]]) = rb[i];
        }
    };

    d4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

    const int nkt = (K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();

    const int frag_k = lane >> 4, frag_x = lane & 15;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) gload((kt + 1) * BK);
        const double *Ab = As + buf * BK * LDS_LD + wm * 64 + frag_x;
        const double *Bb = Bs + buf * BK * LDS_LD + wn * 64 + frag_x;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = Ab[(kk * 4 + frag_k) * LDS_LD + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bb[(kk * 4 + frag_k) * LDS_LD + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C += alpha * acc ; D layout: row = (lane>>4) + 4 r, col = lane & 15 --
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + i * 16 + frag_k + 4 * r;
            if (row >= M) continue;
            double *crow = C + (int64_t)row * ldc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + frag_x;
                if (col < N) crow[col] += alpha * acc[i][j][r];
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// "big" variant: 512-thread workgroup, 256 x 128 tile, LDS-DMA ring.
//
// tools/mfma_f64_probe.hip: a register-resident v_mfma_f64_16x16x4_f64 stream sustains 77.5 TFLOP/s
// (64.0 cycles per MFMA per SIMD).  rocprof PMC on the 128 x 128 kernel above: MFMA pipe 75 % busy,
// 20 % of wave-cycles parked on memory waits, 31 GB fetched per launch against 9.5 GB algorithmic --
// the scarce resource is the L2 -> LDS feed and its latency, not the matrix pipe.  So the eight waves
// of a 512-thread workgroup (2 per SIMD, wave tile 64 x 64 = 16 accumulators in 128 VGPRs, no AGPR
// traffic) share one 256 x 128 tile: half the L2 bytes per flop of two independent 128 x 128
// workgroups.  Operands arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, the copy
// engine writes LDS directly) into a 3-stage ring (3 x 52 KiB) issued two K-tiles ahead and retired
// with a counted s_waitcnt vmcnt(6) plus ONE raw s_barrier per K-tile (BK = 16 -> 64 MFMAs = 4096
// pipe cycles per wave between barriers).  LDS rows keep a 128-B-mod-256-B stride (A: 272, B: 144
// doubles) so every ds_read_b64 lane group is conflict-free.  Out-of-range lanes re-read clamped valid
// columns (their data only reaches masked outputs), so every wave issues exactly six loads per tile
// and the vmcnt arithmetic holds.  Requires K % 16 == 0, even M, N, ldx, ldy and 16-B aligned bases;
// anything else takes the kernel above.
constexpr int GBM = 256, GBN = 128, GBK = 16, GD = 3, GNT = 512;
constexpr int GA_LD = GBM + 16, GB_LD = GBN + 16;
constexpr int GA_STAGE = GBK * GA_LD, GB_STAGE = GBK * GB_LD;      // doubles
constexpr int G_STAGE = GA_STAGE + GB_STAGE;

typedef __attribute__((address_space(3))) void lds_void_t;

__global__ __launch_bounds__(GNT, 2) void dgemm_tn_acc_big_kernel(
    int M, int N, int K, double alpha, const double *__restrict__ X, int64_t ldx,
    const double *__restrict__ Y, int64_t ldy, double *__restrict__ C, int64_t ldc,
    int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) double lds[GD * G_STAGE];

    const unsigned nblocks = (unsigned)tiles_m * (unsigned)tiles_n;
    const unsigned lid = xcd_remap(blockIdx.x, nblocks);
    constexpr unsigned GROUP = 4;
    const unsigned per_group = GROUP * (unsigned)tiles_n;
    const unsigned g = lid / per_group;
    const unsigned first_m = g * GROUP;
    const unsigned gsize = min((unsigned)tiles_m - first_m, GROUP);
    const unsigned in_g = lid - g * per_group;
    const int tm = (int)(first_m + in_g % gsize);
    const int tn = (int)(in_g / gsize);
    const int m0 = tm * GBM, n0 = tn * GBN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frag_k = lane >> 4, frag_x = lane & 15;

    int ca0 = m0 + 2 * lane, ca1 = m0 + 128 + 2 * lane, cb = n0 + 2 * lane;
    if (ca0 + 1 >= M) ca0 = M - 2;
    if (ca1 + 1 >= M) ca1 = M - 2;
    if (cb + 1 >= N) cb = N - 2;
    const double *pA0 = X + ca0, *pA1 = X + ca1, *pB = Y + cb;

    auto issue = [&](int t) {
        double *st = lds + (t % GD) * G_STAGE;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int k = wave * 2 + r;
            const int64_t kg = (int64_t)(t * GBK + k);
            glds16(pA0 + kg * ldx, lds_addr_of(st + k * GA_LD));
            glds16(pA1 + kg * ldx, lds_addr_of(st + k * GA_LD + 128));
            glds16(pB + kg * ldy, lds_addr_of(st + GA_STAGE + k * GB_LD));
        }
    };

    d4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

    const int T = K / GBK;
    issue(0);
    if (T > 1) issue(1);
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < T) issue(t + 2);
        const double *Ab = lds + (t % GD) * G_STAGE + wm * 64 + frag_x;
        const double *Bb = lds + (t % GD) * G_STAGE + GA_STAGE + wn * 64 + frag_x;
#pragma unroll
        for (int kk = 0; kk < GBK / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = Ab[(kk * 4 + frag_k) * GA_LD + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bb[(kk * 4 + frag_k) * GB_LD + j * 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + i * 16 + frag_k + 4 * r;
            if (row >= M) continue;
            double *crow = C + (int64_t)row * ldc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + frag_x;
                // fire-and-forget f64 atomic instead of load-add-store: every C element has exactly ONE writer
                // per launch (tiles are disjoint, launches are stream-ordered), so the sum is still deterministic,
                // but the wave no longer sits out one HBM round trip per element row (lab: 60.1 -> 63.6 TF)
                if (col < N) unsafeAtomicAdd(&crow[col], alpha * acc[i][j][r]);
            }
        }
    }
}

}  // namespace

int launch_dgemm_tn_acc(dmk_ctx *ctx, int M, int N, int K, double alpha, const double *X,
                        int64_t ldx, const double *Y, int64_t ldy, double *C, int64_t ldc) {
    if (M <= 0 || N <= 0 || K <= 0) return DMK_OK;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int64_t nblocks = (int64_t)tiles_m * tiles_n;
    if (nblocks > 0x7fffffffLL) return dmk_fail(ctx, DMK_ERR_INVALID, "dgemm_tn: grid too large");
    const bool vec2 = ((ldx & 1) == 0) && ((ldy & 1) == 0) &&
                      ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
    static const bool big_enabled = [] { const char *e = getenv("DMK_DGEMM_BIG"); return !(e && atoi(e) == 0); }();
    if (big_enabled && vec2 && (K % GBK) == 0 && (M % 2) == 0 && (N % 2) == 0 && M >= 2 && N >= 2 &&
        (int64_t)M * N >= 4 * GBM * GBN) {
        const int btm = (M + GBM - 1) / GBM, btn = (N + GBN - 1) / GBN;
        FamScope fs(ctx, DMK_FAM_DGEMM);
        hipLaunchKernelGGL(dgemm_tn_acc_big_kernel, dim3((unsigned)(btm * btn)), dim3(GNT), 0, ctx->stream, M, N, K,
                           alpha, X, ldx, Y, ldy, C, ldc, btm, btn);
        DMK_CHECK_LAUNCH(ctx);
        return DMK_OK;
    }
    FamScope fs(ctx, DMK_FAM_DGEMM);
    if (vec2)
        hipLaunchKernelGGL(dgemm_tn_acc_kernel<true>, dim3((unsigned)nblocks), dim3(NTHREADS), 0,
                           ctx->stream, M, N, K, alpha, X, ldx, Y, ldy, C, ldc, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL(dgemm_tn_acc_kernel<false>, dim3((unsigned)nblocks), dim3(NTHREADS), 0,
                           ctx->stream, M, N, K, alpha, X, ldx, Y, ldy, C, ldc, tiles_m, tiles_n);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}
