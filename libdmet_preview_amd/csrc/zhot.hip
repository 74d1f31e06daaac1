// K6 hot kernels -- the density-fitted half transform at production tile sizes.
//
//   step 1  Ut[L][q][a] = sum_p Lpq[L][p][q] conj(C_i[p][a])                 (r_e2, first index)
//   step 2  S_L[a][b]   = sum_q Ut[L][q][a] C_j[q][b] (+ sum_q C_j[q][a] Ut[L][q][b])   (second index
//           + lib.hermi_sum of the time-reversal partner), a >= b, tril-packed and ACCUMULATED
//           into the Re / Im planes of Lij_s4
//   reference: basis_transform/eri_transform.py:368-378, 403-434
//
// Why a second implementation next to the generic zgemm.hip: rocprof PMC on the generic
// 256-thread kernels showed the f64 matrix pipe only 56-62 % busy with 30-45 % of wave cycles
// parked on memory waits, and cutting the MFMA count by 25 % (3M) changed nothing -- the half
// transform was bound by the L2 -> LDS feed and its latency, not by the pipe
// (tools/mfma_f64_probe.hip: 77.5 TFLOP/s is reachable from registers).  Hence:
//   * 512-thread workgroups (8 waves, 2 per SIMD) share one large tile, so the L2 bytes per MFMA
//     drop 2-3x;
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4) into a 4-stage ring, issued three
//     K-tiles ahead and retired by a counted s_waitcnt vmcnt(N) + ONE raw s_barrier per K-tile;
//     no staging registers, no scratch, so the counted waits are never drained by the compiler;
//   * step 2 never computes a block above the diagonal AND keeps every wave busy: per L one
//     workgroup takes the 128 x 128 off-diagonal square (8 blocks per wave), a second one takes
//     the two 128 x 128 diagonal triangles with block rows paired (w, 7 - w) so that each wave
//     owns exactly 9 of the 72 blocks; both segments of the symmetrised product read the SAME
//     two LDS panels (U and C over all 256 columns), loaded once per K-tile.
// Complex arithmetic is 4M (four real MFMAs per complex tile step, neg:[1,0,0] for Ai*Bi): the
// 3M form needs 1.5x the accumulators and does not fit 2 waves/SIMD without spilling.
//
// Constraints (else the caller falls back to zgemm.hip): nao % 8 == 0, nemb == 256 for step 2
// (nemb <= 256 and % 16 == 0 for step 1 tiles are masked), 16-B aligned operands.
#include "common.h"
#include <cstdlib>
#include <type_traits>

// Ablation switches for tools/zhot_lab.hip only (product builds leave ZHOT_ABL at 0):
//   1 = no LDS-DMA after the prologue, 2 = no epilogue, 4 = no barrier (results wrong, timing only)
#ifndef ZHOT_ABL
#define ZHOT_ABL 0
#endif

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int HNT = 512;

#define ZMFMA4(ACC_RE, ACC_IM, A, B)                                                              \
    do {                                                                                          \
        ACC_RE = __builtin_amdgcn_mfma_f64_16x16x4f64((A).x, (B).x, ACC_RE, 0, 0, 0);             \
        ACC_IM = __builtin_amdgcn_mfma_f64_16x16x4f64((A).x, (B).y, ACC_IM, 0, 0, 0);             \
        ACC_RE = __builtin_amdgcn_mfma_f64_16x16x4f64((A).y, (B).y, ACC_RE, 0, 0, 1);             \
        ACC_IM = __builtin_amdgcn_mfma_f64_16x16x4f64((A).y, (B).x, ACC_IM, 0, 0, 0);             \
    } while (0)

// =============================================================================================
// step 1: flattened M-blocks (batch L folded into M), tile 128 x 128, BK = 8, 4-stage ring
// =============================================================================================
constexpr int H1_BM = 128, H1_BN = 128, H1_BK = 8, H1_D = 4;
constexpr int H1_STAGE = H1_BK * (H1_BM + H1_BN);     // double2 elements per stage (32 KiB)

struct H1Args {
    const double2 *Lpq;    // [nL][nao][nao]   element (p, q) at p*nao + q
    const double2 *Ci;     // [nao][nemb]
    double2 *Ut;           // [nL][nao][nemb]
    int nL, nao, nemb, nblk;   // nblk = ceil(nao / 16)
    int tiles_m, tiles_n;
    unsigned nblocks;
};

__global__ __launch_bounds__(HNT, 2) void half1_kernel(const H1Args g) {
    __shared__ __attribute__((aligned(16))) double2 lds[H1_D * H1_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;            // 2 (M) x 4 (N) waves, wave tile 64 x 32
    const int frag_k = lane >> 4, frag_x = lane & 15;

    const unsigned lid = xcd_remap(blockIdx.x, g.nblocks);
    const int tile_m = (int)(lid / (unsigned)g.tiles_n), tile_n = (int)(lid % (unsigned)g.tiles_n);
    const int n0 = tile_n * H1_BN;
    const long long nao = g.nao, nemb = g.nemb;

    // ---- per-lane LDS-DMA sources (this wave loads K-row `wave` of every tile: 2 A + 2 B instrs)
    const double2 *srcA[2], *srcB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = 64 * h + lane;
        const int gb = tile_m * (H1_BM / 16) + (m >> 4);
        int L = gb / g.nblk;
        int q = (gb - L * g.nblk) * 16 + (m & 15);
        if (L >= g.nL) L = g.nL - 1;
        if (q >= g.nao) q = g.nao - 1;
        srcA[h] = g.Lpq + (long long)L * nao * nao + q + (long long)wave * nao;
        int col = n0 + 64 * h + lane;
        if (col >= g.nemb) col = g.nemb - 1;
        srcB[h] = g.Ci + col + (long long)wave * nemb;
    }
    auto issue = [&](int t) {
        double2 *st = lds + (t % H1_D) * H1_STAGE;
        const long long k0 = (long long)t * H1_BK;
        glds16(srcA[0] + k0 * nao, lds_addr_of(st + wave * H1_BM));
        glds16(srcA[1] + k0 * nao, lds_addr_of(st + wave * H1_BM + 64));
        glds16(srcB[0] + k0 * nemb, lds_addr_of(st + H1_BK * H1_BM + wave * H1_BN));
        glds16(srcB[1] + k0 * nemb, lds_addr_of(st + H1_BK * H1_BM + wave * H1_BN + 64));
    };

    d4_t acc_re[4][2], acc_im[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc_re[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
            acc_im[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
        }

    const int T = g.nao / H1_BK;
    issue(0);
    if (T > 1) issue(1);
    if (T > 2) issue(2);
    for (int t = 0; t < T; ++t) {
        const int later = T - 1 - t;                    // tiles issued after tile t (at most 2 here)
        if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ZHOT_ABL & 4)) __builtin_amdgcn_s_barrier();
        if (!(ZHOT_ABL & 1) && t + 3 < T) issue(t + 3);
        const double2 *Ab = lds + (t % H1_D) * H1_STAGE + wm * 64 + frag_x;
        const double2 *Bb = lds + (t % H1_D) * H1_STAGE + H1_BK * H1_BM + wn * 32 + frag_x;
#pragma unroll
        for (int kk = 0; kk < H1_BK / 4; ++kk) {
            double2 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = Ab[(kk * 4 + frag_k) * H1_BM + i * 16];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b[j] = Bb[(kk * 4 + frag_k) * H1_BN + j * 16];
                b[j].y = -b[j].y;                       // conj(C_i)
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) ZMFMA4(acc_re[i][j], acc_im[i][j], a[i], b[j]);
        }
    }

    // ---- epilogue: Ut[L][q][a] ---------------------------------------------------------------
    if (ZHOT_ABL & 2) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s += acc_re[i][j][0] + acc_im[i][j][1] + acc_re[i][j][2] + acc_im[i][j][3];
        if (s == 12345.678) g.Ut[tid].x = s;
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gb = tile_m * (H1_BM / 16) + wm * 4 + i;
        const int L = gb / g.nblk;
        if (L >= g.nL) continue;
        const int qb = (gb - L * g.nblk) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qb + frag_k + 4 * r;
            if (q >= g.nao) continue;
            double2 *row = g.Ut + ((long long)L * nao + q) * nemb;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 32 + j * 16 + frag_x;
                if (col < g.nemb) row[col] = make_double2(acc_re[i][j][r], acc_im[i][j][r]);
            }
        }
    }
}

// =============================================================================================
// step 2 (nemb == 256): per L two workgroups, both streaming the full-width U and C panels
// =============================================================================================
constexpr int H2_N = 256, H2_BK = 4, H2_D = 4;
constexpr int H2_STAGE = H2_BK * 2 * H2_N;            // double2 per stage: U[4][256] | C[4][256] = 32 KiB

constexpr int H2_MAXSLOT = 16;
struct H2Args {
    const double2 *Ut;     // [nslot][nL][nao][256]: step-1 outputs of `nslot` consecutive AO blocks
    const double2 *Cj[H2_MAXSLOT];   // [nao][256] of each block
    int sym[H2_MAXSLOT];   // add the time-reversal partner term of that block?
    long long slot_stride; // elements between the Ut of consecutive slots
    double *planes;        // [(ri * naux + L) * npair + pair]
    long long naux, npair;
    int nL, nao, nslot;
    unsigned nblocks;
};

__device__ __forceinline__ void pack_acc(const H2Args &g, int L, int row, int col, double vr, double vi) {
    if (ZHOT_ABL & 2) {
        if (vr == 12345.678) g.planes[0] = vi;
        return;
    }
    if (row >= col) {
        const long long idx = (long long)row * (row + 1) / 2 + col;
        double *pr = g.planes + (long long)L * g.npair + idx;
        double *pi = g.planes + (g.naux + (long long)L) * g.npair + idx;
        // single writer per address per launch -> deterministic; no load latency in the epilogue
        unsafeAtomicAdd(pr, vr);
        unsafeAtomicAdd(pi, vi);
    }
}

__global__ __launch_bounds__(HNT, 2) void half2_kernel(const H2Args g) {
    __shared__ __attribute__((aligned(16))) double2 lds[H2_D * H2_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frag_k = lane >> 4, frag_x = lane & 15;
    const unsigned lid = xcd_remap(blockIdx.x, g.nblocks);
    const int L = (int)(lid >> 1), type = (int)(lid & 1);
    const long long nemb = H2_N;
    const int Tb = g.nao / H2_BK;            // K-tiles per AO block
    const int T = Tb * g.nslot;              // the ring runs straight through all queued blocks: the
                                             // accumulators (and the epilogue) are shared by nslot blocks

    // LDS-DMA sources: a stage is 16 pieces of 64 complex: pieces 0..3 = U row 0, ..., 12..15 = U row 3? no:
    // layout U[4][256] then C[4][256]: piece p < 16 -> U row p / 4, cols 64 (p % 4); p >= 16 -> C likewise.
    // 32 pieces per stage, 8 waves -> 4 per wave.
    const double2 *Ubase = g.Ut + (long long)L * g.nao * nemb;
    long long soff[4];       // per-lane element offset of this wave's four pieces inside a block's U or C panel
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int piece = wave + 8 * h;                  // 0..31: pieces 0-15 = U rows 0-3, 16-31 = C rows 0-3
        const int rowk = (piece & 15) >> 2, c0 = (piece & 3) * 64;
        soff[h] = (long long)rowk * nemb + c0 + lane;
    }
    // pieces wave and wave + 8 are always U, wave + 16 and wave + 24 always C
    auto issue_tile = [&](int tt) {
        const int slot = tt / Tb, t = tt - slot * Tb;
        double2 *st = lds + (tt % H2_D) * H2_STAGE;
        const long long k0 = (long long)t * H2_BK * nemb;
        const double2 *ub = Ubase + (long long)slot * g.slot_stride + k0;
        const double2 *cb = g.Cj[slot] + k0;
        glds16(ub + soff[0], lds_addr_of(st + (wave) * 64));
        glds16(ub + soff[1], lds_addr_of(st + (wave + 8) * 64));
        glds16(cb + soff[2], lds_addr_of(st + (wave + 16) * 64));
        glds16(cb + soff[3], lds_addr_of(st + (wave + 24) * 64));
    };

    if (type == 1) {
        // ---- two diagonal 128 x 128 triangles: waves 0-3 -> [0,128), waves 4-7 -> [128,256) -----------
        const int d0 = (wave >> 2) * 128;
        // tri_body issues its own loads: give it the 4-piece source table through a 2-entry view per call
        // (pieces wave and wave + 8 belong to U rows, wave + 16 / + 24 to C rows)
        auto run = [&](auto tag) {
            constexpr int R1 = decltype(tag)::value;
            constexpr int R2 = 7 - R1;
            d4_t re1[R1 + 1], im1[R1 + 1], re2[R2 + 1], im2[R2 + 1];
#pragma unroll
            for (int c = 0; c <= R1; ++c) { re1[c] = d4_t{0.0, 0.0, 0.0, 0.0}; im1[c] = d4_t{0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
            for (int c = 0; c <= R2; ++c) { re2[c] = d4_t{0.0, 0.0, 0.0, 0.0}; im2[c] = d4_t{0.0, 0.0, 0.0, 0.0}; }
            auto issue = issue_tile;
            issue(0);
            if (T > 1) issue(1);
            if (T > 2) issue(2);
            for (int t = 0; t < T; ++t) {
                const int later = T - 1 - t;
                if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!(ZHOT_ABL & 4)) __builtin_amdgcn_s_barrier();
                if (!(ZHOT_ABL & 1) && t + 3 < T) issue(t + 3);
                const double2 *U = lds + (t % H2_D) * H2_STAGE + frag_k * H2_N + d0 + frag_x;
                const double2 *C = U + H2_BK * H2_N;
                {
                    const double2 a1 = U[R1 * 16], a2 = U[R2 * 16];
                    double2 b[R2 + 1];
#pragma unroll
                    for (int c = 0; c <= R2; ++c) b[c] = C[c * 16];
#pragma unroll
                    for (int c = 0; c <= R1; ++c) ZMFMA4(re1[c], im1[c], a1, b[c]);
#pragma unroll
                    for (int c = 0; c <= R2; ++c) ZMFMA4(re2[c], im2[c], a2, b[c]);
                }
                if (g.sym[t / Tb]) {
                    const double2 a1 = C[R1 * 16], a2 = C[R2 * 16];
                    double2 b[R2 + 1];
#pragma unroll
                    for (int c = 0; c <= R2; ++c) b[c] = U[c * 16];
#pragma unroll
                    for (int c = 0; c <= R1; ++c) ZMFMA4(re1[c], im1[c], a1, b[c]);
#pragma unroll
                    for (int c = 0; c <= R2; ++c) ZMFMA4(re2[c], im2[c], a2, b[c]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row1 = d0 + R1 * 16 + frag_k + 4 * r, row2 = d0 + R2 * 16 + frag_k + 4 * r;
#pragma unroll
                for (int c = 0; c <= R1; ++c) pack_acc(g, L, row1, d0 + c * 16 + frag_x, re1[c][r], im1[c][r]);
#pragma unroll
                for (int c = 0; c <= R2; ++c) pack_acc(g, L, row2, d0 + c * 16 + frag_x, re2[c][r], im2[c][r]);
            }
        };
        switch (wave & 3) {
            case 0: run(std::integral_constant<int, 0>{}); break;
            case 1: run(std::integral_constant<int, 1>{}); break;
            case 2: run(std::integral_constant<int, 2>{}); break;
            default: run(std::integral_constant<int, 3>{}); break;
        }
        return;
    }

    // ---- off-diagonal square: rows [128,256) x cols [0,128); waves 4 (M) x 2 (N), wave tile 32 x 64 ----
    const int wm = wave >> 1, wn = wave & 1;
    d4_t acc_re[2][4], acc_im[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc_re[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
            acc_im[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
        }
    auto issue = issue_tile;
    issue(0);
    if (T > 1) issue(1);
    if (T > 2) issue(2);
    for (int t = 0; t < T; ++t) {
        const int later = T - 1 - t;
        if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ZHOT_ABL & 4)) __builtin_amdgcn_s_barrier();
        if (!(ZHOT_ABL & 1) && t + 3 < T) issue(t + 3);
        const double2 *U = lds + (t % H2_D) * H2_STAGE + frag_k * H2_N + frag_x;
        const double2 *C = U + H2_BK * H2_N;
        {
            double2 a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = U[128 + wm * 32 + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = C[wn * 64 + j * 16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ZMFMA4(acc_re[i][j], acc_im[i][j], a[i], b[j]);
        }
        if (g.sym[t / Tb]) {
            double2 a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = C[128 + wm * 32 + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = U[wn * 64 + j * 16];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ZMFMA4(acc_re[i][j], acc_im[i][j], a[i], b[j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 128 + wm * 32 + i * 16 + frag_k + 4 * r;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                pack_acc(g, L, row, wn * 64 + j * 16 + frag_x, acc_re[i][j][r], acc_im[i][j][r]);
        }
}

bool hot_enabled() {
    static const bool on = [] { const char *e = getenv("DMK_ERI_HOT"); return !(e && atoi(e) == 0); }();
    return on;
}

}  // namespace

// Returns 1 if the hot path handled the launch, 0 if the caller must use the generic kernel, < 0 on error.
int launch_half1_hot(dmk_ctx *ctx, const void *Lpq, const void *Ci, void *Ut, int nL, int nao, int nemb) {
    if (!hot_enabled() || (nao % H1_BK) != 0 || nao < 3 * H1_BK || nemb < 64 || (long long)nL * nao < 4 * H1_BM) return 0;
    if ((reinterpret_cast<uintptr_t>(Lpq) | reinterpret_cast<uintptr_t>(Ci) | reinterpret_cast<uintptr_t>(Ut)) & 15) return 0;
    H1Args a;
    a.Lpq = reinterpret_cast<const double2 *>(Lpq);
    a.Ci = reinterpret_cast<const double2 *>(Ci);
    a.Ut = reinterpret_cast<double2 *>(Ut);
    a.nL = nL; a.nao = nao; a.nemb = nemb;
    a.nblk = (nao + 15) / 16;
    const long long total_blk = (long long)nL * a.nblk;
    a.tiles_m = (int)((total_blk + H1_BM / 16 - 1) / (H1_BM / 16));
    a.tiles_n = (nemb + H1_BN - 1) / H1_BN;
    a.nblocks = (unsigned)(a.tiles_m * a.tiles_n);
    FamScope fs(ctx, DMK_FAM_ZGEMM_HALF1);
    hipLaunchKernelGGL(half1_kernel, dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a);
    DMK_CHECK_LAUNCH(ctx);
    return 1;
}

int launch_half2_hot(dmk_ctx *ctx, const void *Ut, long long slot_stride, int nslot, const void *const *Cj,
                     const int *sym, double *planes, long long naux, long long npair, int nL, int nao, int nemb) {
    if (!hot_enabled() || nemb != H2_N || (nao % H2_BK) != 0 || nao < 3 * H2_BK || nslot < 1 || nslot > H2_MAXSLOT)
        return 0;
    if (reinterpret_cast<uintptr_t>(Ut) & 15) return 0;
    H2Args a;
    a.Ut = reinterpret_cast<const double2 *>(Ut);
    for (int i = 0; i < H2_MAXSLOT; ++i) {
        a.Cj[i] = reinterpret_cast<const double2 *>(Cj[i < nslot ? i : 0]);
        a.sym[i] = i < nslot ? sym[i] : 0;
        if (reinterpret_cast<uintptr_t>(a.Cj[i]) & 15) return 0;
    }
    a.slot_stride = slot_stride;
    a.planes = planes; a.naux = naux; a.npair = npair;
    a.nL = nL; a.nao = nao; a.nslot = nslot;
    a.nblocks = (unsigned)(2 * nL);
    FamScope fs(ctx, DMK_FAM_ZGEMM_HALF2);
    hipLaunchKernelGGL(half2_kernel, dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a);
    DMK_CHECK_LAUNCH(ctx);
    return 1;
}

int half2_hot_usable(int nao, int nemb) { return hot_enabled() && nemb == H2_N && (nao % H2_BK) == 0 && nao >= 3 * H2_BK; }
int half2_hot_maxslot() { return H2_MAXSLOT; }
