// K6 hot kernels -- the density-fitted half transform at production tile sizes.
//
//   step 1  Ut[L][q][a] = sum_p Lpq[L][p][q] conj(C_i[p][a])                 (r_e2, first index)
//   step 2  S_L[a][b]   = sum_q Ut[L][q][a] C_j[q][b] (+ sum_q C_j[q][a] Ut[L][q][b])   (second index
//           + lib.hermi_sum of the time-reversal partner), a >= b, tril-packed and ACCUMULATED
//           into the Re / Im planes of Lij_s4
//   reference: basis_transform/eri_transform.py:368-378, 403-434
//
// Why a second implementation next to the generic zgemm.hip (numbers: MI355X, rocprof PMC and the
// ablation labs under tools/):
//   * the generic register-staged kernels kept the f64 matrix pipe only 56-62 % busy with 30-45 % of
//     the wave cycles parked in s_waitcnt / s_barrier, and cutting the MFMA count by 25 % (3M) changed
//     nothing: the half transform was latency-bound on its operand feed, not pipe-bound
//     (tools/mfma_f64_probe.hip: 77.5 TFLOP/s is reachable from registers);
//   * operands therefore arrive by LDS-DMA (global_load_lds_dwordx4) into a 3-4 stage ring, issued
//     two to three K-tiles ahead and retired by a counted s_waitcnt vmcnt(N) + ONE raw s_barrier per
//     K-tile; no staging registers, no scratch (round 4: the LDS-DMA pieces are addressed as scalar tile base + one loop-invariant
//     32-bit offset VGPR per piece, which removed half2_kernel's last 18 spilled VGPRs), so the counted waits are never drained
//     by the compiler;
//   * 256-thread workgroups, TWO per CU: a 512-thread workgroup sharing a larger tile halves the L2
//     bytes per flop but runs its two waves per SIMD in barrier lock-step and measured 10 % slower
//     than two independent workgroups that desynchronise by themselves (tools/gemm_lab*.hip);
//   * step 2 never computes a 16 x 16 block above the diagonal AND keeps every wave equally loaded:
//     per L four workgroups -- the two 64 x 128 halves of the off-diagonal square (8 blocks per wave)
//     and the two 128 x 128 diagonal triangles with block rows paired (w, 7 - w) so that each wave owns
//     exactly 9 of a triangle's 36 blocks; in a triangle both segments of the symmetrised product read
//     the SAME two LDS panels;
//   * step 2 runs straight through up to 16 queued AO blocks per launch: accumulators and the
//     tril-pack epilogue are shared, and the epilogue uses fire-and-forget f64 atomics (exactly one
//     writer per plane element per launch, so the sum stays deterministic).
// Complex arithmetic is 3M (Karatsuba: three real MFMAs per complex tile step, see cmfma below); with LDS-DMA
// there are no staging registers, so the 1.5x accumulator set still fits two waves per SIMD without spilling.
//
// Constraints (else the caller falls back to zgemm.hip): nao % 8 == 0, nemb == 256 for step 2,
// 16-B aligned operands.
#include "common.h"
#include <cstdlib>
#include <algorithm>
#include <type_traits>

#include "zhot_common.h"

namespace {

// =============================================================================================
// step 1: flattened M-blocks (batch L folded into M), tile 128 x 64, BK = 8, 3-stage ring (72 KiB)
// =============================================================================================
constexpr int H1_BM = 128, H1_BN = 64, H1_BK = 8, H1_D = 3;

// out[L][m][n] = sum_k A[L][k][m] * op(B[k][n]),  op = conj (step 1: B = C_i) or identity (step 2 of general nemb: B = C_j).
// A is K-major: element (k, m) of batch L at L * (nao * mrows) + k * mrows + m  (step 1: Lpq, mrows = nao; step 2: Ut,
// mrows = nemb).  The batch index is folded into M in 16-row blocks.
struct H1Args {
    const double2 *Lpq;    // A
    const double2 *Ci;     // B [nao][nemb]
    double2 *Ut;           // out [nL][mrows][nemb]
    int nL, nao, nemb, nblk;   // nao = K; nblk = ceil(mrows / 16)
    int mrows;
    // K loop bound: nao rounded up to the K tile.  B must hold kdim rows, ZERO beyond nao (the pipeline keeps a padded copy of
    // C_ao_emb); the A rows of the padding are the clamped row nao - 1 -- finite numbers against zeros, never read past a block
    int kdim;
    int tiles_m, tiles_n;
    unsigned nblocks;
    // both spin channels in one launch: the same A tile (AO block) against C_i of spin 0 / spin 1 into their own Ut;
    // the workgroups of one M tile are adjacent (n tile fastest, then spin), so they share the A tile in L2
    int nspin;
    long long b_spin_stride, out_spin_stride;
    // several queued AO blocks in one launch (one ramp-up / drain instead of one per block): block `slot` is the A
    // operand Lpq + slot * a_slot_stride, its B operand Ci + bk[slot] * b_k_stride, its output Ut + slot * out_slot_stride
    int nslot;
    unsigned per_slot;      // workgroups per block = tiles_m * tiles_n * nspin
    long long a_slot_stride, out_slot_stride, b_k_stride;
    int bk[16];
};

#define H1_PICK_BK(G, SLOT)                                                                        \
    ((SLOT) == 0 ? (G).bk[0] : (SLOT) == 1 ? (G).bk[1] : (SLOT) == 2 ? (G).bk[2] : (SLOT) == 3 ? (G).bk[3]      \
     : (SLOT) == 4 ? (G).bk[4] : (SLOT) == 5 ? (G).bk[5] : (SLOT) == 6 ? (G).bk[6] : (SLOT) == 7 ? (G).bk[7]    \
     : (SLOT) == 8 ? (G).bk[8] : (SLOT) == 9 ? (G).bk[9] : (SLOT) == 10 ? (G).bk[10] : (SLOT) == 11 ? (G).bk[11] \
     : (SLOT) == 12 ? (G).bk[12] : (SLOT) == 13 ? (G).bk[13] : (SLOT) == 14 ? (G).bk[14] : (G).bk[15])

// BM = 128 (wave tile 64 x 32, 8 accumulator tiles, two workgroups per CU) or BM = 64 (wave tile 32 x 32, 4 tiles, three
// workgroups per CU).  NARROW: the output tile is 48 instead of 64 columns wide -- the four waves are stacked along M
// (wave tile BM / 4 x 48: 6 accumulator tiles at BM = 128) -- for embedding dimensions that three 48-column tiles cover
// with less padding than 64-column ones (C4: nemb 136 -> 144 instead of 192 computed columns).  The B panel keeps its
// 64-column LDS rows (lanes 48-63 of a piece land in the padding), so the LDS-DMA issue pattern is the same for both.
// The M index is the FLAT row (L, q) -> L * mrows + q with no padding between batches: a 16-row block may straddle two L
// (every lane carries its own source address, and Ut[L][q][:] is one contiguous array of nL * mrows rows).
// LAB (tools/zhot_lab.hip only; the product instantiates LAB = 0, where every `if constexpr` below folds away): ablation bits that
// remove one ingredient of the K loop at a time so that its share of the time can be MEASURED on the real kernel -- 1: no Ut
// stores, 2: no LDS-DMA after the prologue (the ring keeps stale tiles), 4: no s_barrier.  Results of a LAB != 0 instantiation are
// meaningless by construction.
// KPAD: the K loop runs over g.kdim > g.nao (an AO dimension off the K tile); only then are the A rows clamped -- the instantiation
// for dimensions on the tile is the kernel of rounds 2 - 5, instruction for instruction (the clamp measured 2.4 % on it).
template <bool CONJB, int BM, int OCC, bool NARROW, int LAB = 0, bool KPAD = false>
__global__ __launch_bounds__(HNT, OCC) void half1_kernel(const H1Args g) {
    constexpr int MI = NARROW ? BM / 64 : BM / 32;       // 16-row blocks per wave
    constexpr int NJ = NARROW ? 3 : 2;                   // 16-column blocks per wave
    constexpr int BN = NARROW ? 48 : H1_BN;              // columns of the output tile
    constexpr int AH = BM / 64;                          // 1 KiB pieces per K row of the A panel
    constexpr int STAGE = H1_BK * (BM + H1_BN);
    __shared__ __attribute__((aligned(16))) double2 lds[H1_D * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps the LDS-DMA addressing scalar
    // 2 (M) x 2 (N) waves with wave tile (BM / 2) x 32, or 4 (M) x 1 waves with wave tile (BM / 4) x 48
    const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
    const int frag_k = lane >> 4, frag_x = lane & 15;

    const unsigned lid_all = xcd_remap(blockIdx.x, g.nblocks);
    const int slot = (int)(lid_all / g.per_slot);
    const unsigned lid = lid_all - (unsigned)slot * g.per_slot;
    const unsigned per_m = (unsigned)(g.tiles_n * g.nspin);
    const int tile_m = (int)(lid / per_m);
    const unsigned rest = lid - (unsigned)tile_m * per_m;
    const int sp = (int)(rest / (unsigned)g.tiles_n), tile_n = (int)(rest - (unsigned)sp * (unsigned)g.tiles_n);
    const int n0 = tile_n * BN;
    const long long nao = g.nao, nemb = g.nemb, mrows = g.mrows;
    const long long rows_total = (long long)g.nL * mrows;
    const double2 *const Asl = g.Lpq + (long long)slot * g.a_slot_stride;
    const double2 *const Bsp = g.Ci + (long long)sp * g.b_spin_stride + (long long)H1_PICK_BK(g, slot) * g.b_k_stride;
    double2 *const Osp = g.Ut + (long long)sp * g.out_spin_stride + (long long)slot * g.out_slot_stride;

    // ---- LDS-DMA sources: wave w streams K rows 2w, 2w+1 (A: 2 x 1 KiB per row, B: 1 KiB).  Per lane only a loop-invariant byte
    //      offset from the block's base (one VGPR per piece; a block is <= 512 MB); the K-row part of the address is scalar ----
    unsigned voffA[AH], voffB;
#pragma unroll
    for (int h = 0; h < AH; ++h) {
        long long r = (long long)tile_m * BM + 64 * h + lane;
        if (r >= rows_total) r = rows_total - 1;         // clamped lanes only ever feed masked outputs
        const long long L = r / mrows, q = r - L * mrows;
        voffA[h] = (unsigned)((L * nao * mrows + q) * 16);
    }
    {
        int col = n0 + lane;
        if (col >= g.nemb) col = g.nemb - 1;
        voffB = (unsigned)(col * 16);
    }
    auto issue = [&](int t) {
        double2 *st = lds + (t % H1_D) * STAGE;
        const int k0 = wave * 2;
        const long long kg = (long long)t * H1_BK + k0;
        const double2 *a0 = Asl + kg * mrows, *a1 = a0 + mrows, *b0 = Bsp + kg * nemb, *b1 = b0 + nemb;      // wave-uniform
        if constexpr (KPAD) {                                // rows of the padding: the block's last row (see H1Args::kdim)
            const int last = (int)nao - 1, k32 = t * H1_BK + k0;
            a0 = Asl + (long long)(k32 < last ? k32 : last) * mrows;
            a1 = Asl + (long long)(k32 + 1 < last ? k32 + 1 : last) * mrows;
        }
        if constexpr (AH == 2) {
            glds16s_x6(voffA[0], voffA[1], voffB, voffA[0], voffA[1], voffB, a0, a0, b0, a1, a1, b1, lds_addr_of(st + k0 * BM),
                       lds_addr_of(st + k0 * BM + 64), lds_addr_of(st + H1_BK * BM + k0 * H1_BN), lds_addr_of(st + (k0 + 1) * BM),
                       lds_addr_of(st + (k0 + 1) * BM + 64), lds_addr_of(st + H1_BK * BM + (k0 + 1) * H1_BN));
        } else {
            glds16s_x4(voffA[0], voffB, voffA[0], voffB, a0, b0, a1, b1, lds_addr_of(st + k0 * BM),
                       lds_addr_of(st + H1_BK * BM + k0 * H1_BN), lds_addr_of(st + (k0 + 1) * BM),
                       lds_addr_of(st + H1_BK * BM + (k0 + 1) * H1_BN));
        }
    };

    cacc acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) cacc_zero(acc[i][j]);

    const int T = (KPAD ? g.kdim : g.nao) / H1_BK;
    issue(0);
    if (T > 1) issue(1);
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) {                                                     // tile t landed; tile t+1 may be in flight
            if (AH == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (!(LAB & 4)) __builtin_amdgcn_s_barrier();
        if constexpr (LAB & 2) { if (t + 2 < T && g.nslot < 0) issue(t + 2); }
        else { if (t + 2 < T) issue(t + 2); }
        const double2 *Ab = lds + (t % H1_D) * STAGE + wm * (MI * 16) + frag_x;
        const double2 *Bb = lds + (t % H1_D) * STAGE + H1_BK * BM + wn * 32 + frag_x;
#pragma unroll
        for (int kk = 0; kk < H1_BK / 4; ++kk) {
            cfrag a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = cfrag_of(lds_frag(&Ab[(kk * 4 + frag_k) * BM + i * 16]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                double2 v = lds_frag(&Bb[(kk * 4 + frag_k) * H1_BN + j * 16]);
                if (CONJB) v.y = -v.y;                  // conj(C_i)
                b[j] = cfrag_of(v);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) cmfma(acc[i][j], a[i], b[j]);
        }
    }

    // ---- epilogue: Ut[L][q][a] = row (L * mrows + q) of one contiguous (nL * mrows) x nemb array ----------------------
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long rr = (long long)tile_m * BM + (wm * MI + i) * 16 + frag_k + 4 * r;
            if (rr >= rows_total) continue;
            if constexpr (LAB & 1) { if (g.nslot >= 0) continue; }
            double2 *row = Osp + rr * nemb;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = n0 + wn * 32 + j * 16 + frag_x;
                if (col < g.nemb) row[col] = make_double2(cacc_re(acc[i][j], r), cacc_im(acc[i][j], r));
            }
        }
    }
}

// =============================================================================================
// step 2 (nemb == 256): per L four workgroups
//   type 0 / 1 : rows [128,192) / [192,256) x cols [0,128) of the off-diagonal square (8 blocks / wave)
//   type 2 / 3 : diagonal triangle [0,128)^2 / [128,256)^2, block rows (w, 7-w) per wave (9 blocks / wave)
// =============================================================================================
constexpr int H2_N = 256, H2_BK = 4;
constexpr int H2_MAXSLOT = 16;
constexpr int H2S_STAGE = H2_BK * 384, H2S_D = 3;     // square: Ua[4][64] | Cb[4][128] | Ca[4][64] | Ub[4][128] (24 KiB)
constexpr int H2T_STAGE = H2_BK * 256, H2T_D = 4;     // triangle: U[4][128] | C[4][128]                        (16 KiB)
constexpr int H2_LDS = (H2S_STAGE * H2S_D > H2T_STAGE * H2T_D) ? H2S_STAGE * H2S_D : H2T_STAGE * H2T_D;

struct H2Args {
    const double2 *Ut;     // [nslot][nL][nao][256]: step-1 outputs of `nslot` consecutive AO blocks
    const double2 *Cj[H2_MAXSLOT];   // [nao][256] of each block
    unsigned symmask;      // bit s: add the time-reversal partner term of block s?
    long long slot_stride; // elements between the Ut of consecutive slots
    double *planes;        // [(ri * naux + L) * npair + pair]
    long long naux, npair;
    int nL, nao, nslot;
    int kdim;              // K loop bound: nao rounded up to the K tile; Cj holds kdim rows, zero beyond nao (Ut rows of the padding
                           // are whatever follows in the pipeline's own, initialised buffer: finite numbers against zeros)
    unsigned nblocks;
    // both spin channels in ONE launch (4 nL nspin workgroups): one ramp-up / drain per group of queued blocks instead
    // of one per spin (measured at C5: 2 x 8.19 ms -> 15.75 ms).  Cutting the last rounds of workgroups into shorter
    // ones (slot ranges accumulated through partial buffers) was tried on top and measured slower: the launch is
    // throughput-bound in steady state, the extra epilogues cost more than the shorter drain saves.
    int nspin;
    long long ut_spin_stride, cj_spin_stride, planes_spin_stride;   // elements between the spin channels
    // every queued block carries the time-reversal partner term: the 16 DIAGONAL 16 x 16 blocks then skip segment 2 and are
    // completed as P + P^T in the epilogue (S_rr = U_r^T C_r + C_r^T U_r, and the second product is the transpose of the
    // first): 256 instead of 272 block products per L, and triangle waves issue 16 per K step like the square ones
    int fold_diag;
};

// Kernel-argument arrays must only be indexed with compile-time constants, and the argument struct must
// never be passed by reference: either makes hipcc copy the whole struct to scratch (private memory), whose
// loads need s_waitcnt vmcnt(0) -- draining the LDS-DMA ring -- and whose per-dispatch scratch set-up cost
// ~17 ms per launch when this kernel first did it.
#define H2_PICK_CJ(G, SLOT)                                                                        \
    ((SLOT) == 0 ? (G).Cj[0] : (SLOT) == 1 ? (G).Cj[1] : (SLOT) == 2 ? (G).Cj[2] : (SLOT) == 3 ? (G).Cj[3]      \
     : (SLOT) == 4 ? (G).Cj[4] : (SLOT) == 5 ? (G).Cj[5] : (SLOT) == 6 ? (G).Cj[6] : (SLOT) == 7 ? (G).Cj[7]    \
     : (SLOT) == 8 ? (G).Cj[8] : (SLOT) == 9 ? (G).Cj[9] : (SLOT) == 10 ? (G).Cj[10] : (SLOT) == 11 ? (G).Cj[11] \
     : (SLOT) == 12 ? (G).Cj[12] : (SLOT) == 13 ? (G).Cj[13] : (SLOT) == 14 ? (G).Cj[14] : (G).Cj[15])

// LAB: ablation bits of tools/zhot_lab.hip, as in half1_kernel (1: no plane atomics, 2: no LDS-DMA after the prologue, 4: no
// s_barrier); the product instantiates LAB = 0.
template <int LAB = 0, bool RE = false>
__global__ __launch_bounds__(HNT, 2) void half2_kernel(const H2Args g) {
    __shared__ __attribute__((aligned(16))) double2 lds[H2_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar LDS-DMA addressing
    const int frag_k = lane >> 4, frag_x = lane & 15;
    const unsigned lid = xcd_remap(blockIdx.x, g.nblocks);
    const int Lall = (int)(lid >> 2), type = (int)(lid & 3);
    const int sp = Lall >= g.nL ? 1 : 0;     // nspin <= 2
    const int L = Lall - sp * g.nL;
    const long long nemb = H2_N;
    const int Tb = g.kdim / H2_BK;           // K-tiles per AO block
    const int T = Tb * g.nslot;              // the ring runs straight through all queued blocks
    const double2 *Ubase = g.Ut + (long long)sp * g.ut_spin_stride + (long long)L * g.nao * nemb;
    double *const g_planes = g.planes + (long long)sp * g.planes_spin_stride;
    const long long cj_off = (long long)sp * g.cj_spin_stride;
    const long long g_naux = g.naux, g_npair = g.npair, g_slot_stride = g.slot_stride;
    const unsigned g_symmask = g.symmask;
    const bool fold = g.fold_diag != 0;

    if (type >= 2) {
        // ---------------- diagonal triangle [d0, d0+128)^2 ----------------------------------------
        const int d0 = (type - 2) * 128;
        // stage = 16 pieces of 64 complex: piece p < 8 -> U row p/2, half p%2 ; p >= 8 -> C likewise; 4 pieces per wave
        unsigned voff[4];                      // byte offset of this lane's 16 B inside a K-tile of the operand
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int piece = wave + 4 * h;
            voff[h] = (unsigned)((((piece & 7) >> 1) * (int)nemb + d0 + (piece & 1) * 64 + lane) * 16);
        }
        // running issue state (wave-uniform, SGPRs): no division and no kernel-argument load per K-tile
        int is_t = 0, is_slot = 0, is_stage = 0;
        const double2 *is_ub = Ubase, *is_cb = H2_PICK_CJ(g, 0) + cj_off;
        // (spreading the four pieces of a tile over the MFMA stream of a K step, instead of a burst after the barrier,
        // measured 2.5 % slower: the inline-asm DMA statements pin the compiler's MFMA / ds_read schedule)
        auto issue_advance = [&]() {
            is_stage = is_stage + 1 == H2T_D ? 0 : is_stage + 1;
            if (++is_t == Tb) {
                is_t = 0;
                ++is_slot;
                is_ub = Ubase + (long long)is_slot * g_slot_stride;
                is_cb = H2_PICK_CJ(g, is_slot) + cj_off;
            } else {
                is_ub += H2_BK * nemb;
                is_cb += H2_BK * nemb;
            }
        };
        auto issue = [&]() {
            double2 *st = lds + is_stage * H2T_STAGE;
            // scalar tile bases + loop-invariant per-lane byte offsets: no vector ALU work per piece (common.h glds16s_x4)
            glds16s_x4(voff[0], voff[1], voff[2], voff[3], is_ub, is_ub, is_cb, is_cb, lds_addr_of(st + wave * 64),
                       lds_addr_of(st + (wave + 4) * 64), lds_addr_of(st + (wave + 8) * 64), lds_addr_of(st + (wave + 12) * 64));
            issue_advance();
        };
        auto run = [&](auto tag) {
            constexpr int R1 = decltype(tag)::value;
            constexpr int R2 = 7 - R1;
            cacc acc1[R1 + 1], acc2[R2 + 1];
#pragma unroll
            for (int c = 0; c <= R1; ++c) cacc_zero(acc1[c]);
#pragma unroll
            for (int c = 0; c <= R2; ++c) cacc_zero(acc2[c]);
            issue();
            if (T > 1) issue();
            if (T > 2) issue();
            int c_t = 0, c_stage = 0;
            unsigned c_sym = g_symmask & 1u, c_mask = g_symmask;
            for (int t = 0; t < T; ++t) {
                const int later = T - 1 - t;
                if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (!(LAB & 4)) __builtin_amdgcn_s_barrier();
                if constexpr (LAB & 2) { if (t + 3 < T && g.nslot < 0) issue(); }
                else { if (t + 3 < T) issue(); }
                const double2 *U = lds + c_stage * H2T_STAGE + frag_k * 128 + frag_x;
                c_stage = c_stage + 1 == H2T_D ? 0 : c_stage + 1;
                const double2 *C = U + H2_BK * 128;
                {   // segment 1: S[a][b] += U[q][a] C[q][b]   (one B fragment live at a time)
                    const cfrag a1 = cfrag_of_t<RE>(lds_frag(&U[R1 * 16])), a2 = cfrag_of_t<RE>(lds_frag(&U[R2 * 16]));
#pragma unroll
                    for (int c = 0; c <= R2; ++c) {
                        const cfrag b = cfrag_of_t<RE>(lds_frag(&C[c * 16]));
                        if (c <= R1) cmfma_t<RE>(acc1[c <= R1 ? c : 0], a1, b);
                        cmfma_t<RE>(acc2[c], a2, b);
                    }
                }
                if (c_sym) {   // segment 2: S[a][b] += C[q][a] U[q][b]   (same two panels)
                    const cfrag a1 = cfrag_of_t<RE>(lds_frag(&C[R1 * 16])), a2 = cfrag_of_t<RE>(lds_frag(&C[R2 * 16]));
#pragma unroll
                    for (int c = 0; c <= R2; ++c) {
                        const cfrag b = cfrag_of_t<RE>(lds_frag(&U[c * 16]));
                        if (c < R1 || (c == R1 && !fold)) cmfma_t<RE>(acc1[c <= R1 ? c : 0], a1, b);
                        if (c < R2 || !fold) cmfma_t<RE>(acc2[c], a2, b);
                    }
                }
                if (++c_t == Tb) {
                    c_t = 0;
                    c_mask >>= 1;
                    c_sym = c_mask & 1u;
                }
            }
            if (fold) {
                // diagonal blocks hold P = U_r^T C_r only: add P^T through a wave-private LDS tile (the ring is idle now)
                __syncthreads();
                double *tr = reinterpret_cast<double *>(lds) + wave * (2 * 16 * 17);
                auto fold_block = [&](cacc &acc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tr[(frag_k + 4 * r) * 17 + frag_x] = cacc_re(acc, r);
                        if constexpr (!RE) tr[272 + (frag_k + 4 * r) * 17 + frag_x] = cacc_im(acc, r);
                    }
                    double tre[4], tim[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                      // the LDS pipe keeps a wave's own accesses in order
                        tre[r] = tr[frag_x * 17 + frag_k + 4 * r];
                        tim[r] = RE ? 0.0 : tr[272 + frag_x * 17 + frag_k + 4 * r];
                    }
                    // fold into the T1 / T2 / T3 representation: Re += tre, Im += tim  (T1 += tre, T3 += tre + tim)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc.p[r] += tre[r];
                        if constexpr (!RE) acc.t[r] += tre[r] + tim[r];
                    }
                };
                fold_block(acc1[R1]);
                fold_block(acc2[R2]);
            }
            if constexpr (LAB & 1) { if (g.nslot >= 0) return; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row1 = d0 + R1 * 16 + frag_k + 4 * r, row2 = d0 + R2 * 16 + frag_k + 4 * r;
#pragma unroll
                for (int c = 0; c <= R1; ++c)
                    pack_acc_t<RE>(g_planes, g_naux, g_npair, L, row1, d0 + c * 16 + frag_x, acc1[c], r);
#pragma unroll
                for (int c = 0; c <= R2; ++c)
                    pack_acc_t<RE>(g_planes, g_naux, g_npair, L, row2, d0 + c * 16 + frag_x, acc2[c], r);
            }
        };
        switch (wave) {
            case 0: run(std::integral_constant<int, 0>{}); break;
            case 1: run(std::integral_constant<int, 1>{}); break;
            case 2: run(std::integral_constant<int, 2>{}); break;
            default: run(std::integral_constant<int, 3>{}); break;
        }
        return;
    }

    // ---------------- off-diagonal half square: rows [r0, r0+64) x cols [0,128) -------------------------
    const int r0 = 128 + 64 * type;
    const int wm = wave >> 1, wn = wave & 1;            // wave tile 32 x 64
    // stage = 24 pieces of 64 complex: 0-3 Ua rows, 4-11 Cb (row*2+half), 12-15 Ca rows, 16-23 Ub (row*2+half)
    unsigned voff[6];
#pragma unroll
    for (int h = 0; h < 6; ++h) {
        const int piece = wave + 4 * h;
        int row, col;
        if (piece < 4) { row = piece; col = r0; }
        else if (piece < 12) { row = (piece - 4) >> 1; col = ((piece - 4) & 1) * 64; }
        else if (piece < 16) { row = piece - 12; col = r0; }
        else { row = (piece - 16) >> 1; col = ((piece - 16) & 1) * 64; }
        voff[h] = (unsigned)((row * (int)nemb + col + lane) * 16);
    }
    int is_t = 0, is_slot = 0, is_stage = 0;
    const double2 *is_ub = Ubase, *is_cb = H2_PICK_CJ(g, 0) + cj_off;
    auto issue = [&]() {
        double2 *st = lds + is_stage * H2S_STAGE;
        // pieces wave + 4 h: h = 0 Ua, 1-2 Cb, 3 Ca, 4-5 Ub -- which operand a piece belongs to does not depend on the wave, so the
        // bases are the two scalar tile pointers and the per-lane part is a loop-invariant byte offset (common.h glds16s_x6)
        glds16s_x6(voff[0], voff[1], voff[2], voff[3], voff[4], voff[5], is_ub, is_cb, is_cb, is_cb, is_ub, is_ub, lds_addr_of(st + wave * 64),
                   lds_addr_of(st + (wave + 4) * 64), lds_addr_of(st + (wave + 8) * 64), lds_addr_of(st + (wave + 12) * 64),
                   lds_addr_of(st + (wave + 16) * 64), lds_addr_of(st + (wave + 20) * 64));
        is_stage = is_stage + 1 == H2S_D ? 0 : is_stage + 1;
        if (++is_t == Tb) {
            is_t = 0;
            ++is_slot;
            is_ub = Ubase + (long long)is_slot * g_slot_stride;
            is_cb = H2_PICK_CJ(g, is_slot) + cj_off;
        } else {
            is_ub += H2_BK * nemb;
            is_cb += H2_BK * nemb;
        }
    };
    cacc acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cacc_zero(acc[i][j]);
    issue();
    if (T > 1) issue();
    int c_t = 0, c_stage = 0;
    unsigned c_sym = g_symmask & 1u, c_mask = g_symmask;
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (!(LAB & 4)) __builtin_amdgcn_s_barrier();
        if constexpr (LAB & 2) { if (t + 2 < T && g.nslot < 0) issue(); }
        else { if (t + 2 < T) issue(); }
        const double2 *Ua = lds + c_stage * H2S_STAGE + frag_k * 64 + wm * 32 + frag_x;
        const double2 *Cb = lds + c_stage * H2S_STAGE + 256 + frag_k * 128 + wn * 64 + frag_x;
        c_stage = c_stage + 1 == H2S_D ? 0 : c_stage + 1;
        const double2 *Ca = Ua + 768;
        const double2 *Ub = Cb + 768;
        {
            cfrag a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = cfrag_of_t<RE>(lds_frag(&Ua[i * 16]));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = cfrag_of_t<RE>(lds_frag(&Cb[j * 16]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) cmfma_t<RE>(acc[i][j], a[i], b[j]);
        }
        if (c_sym) {
            cfrag a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = cfrag_of_t<RE>(lds_frag(&Ca[i * 16]));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = cfrag_of_t<RE>(lds_frag(&Ub[j * 16]));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) cmfma_t<RE>(acc[i][j], a[i], b[j]);
        }
        if (++c_t == Tb) {
            c_t = 0;
            c_mask >>= 1;
            c_sym = c_mask & 1u;
        }
    }
    if constexpr (LAB & 1) { if (g.nslot >= 0) return; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + wm * 32 + i * 16 + frag_k + 4 * r;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                pack_acc_t<RE>(g_planes, g_naux, g_npair, L, row, wn * 64 + j * 16 + frag_x, acc[i][j], r);
        }
}

bool hot_enabled() {
    static const bool on = [] { const char *e = getenv("DMK_ERI_HOT"); return !(e && atoi(e) == 0); }();
    return on;
}

}  // namespace

// shapes the flattened kernel covers when the flattened row count is nL x nao (step 1)
int half1_hot_usable(int nL, int nao, int nemb) {
    // the per-lane part of an LDS-DMA source address is a 32-bit byte offset from the block's base (glds16s): an AO block must stay
    // below 4 GiB (C5: 512 MB); larger ones take the generic kernels
    // (nao need not be a multiple of the K tile: the K loop runs over hot_kdim(nao) with a zero-padded B operand)
    return hot_enabled() && nao >= 2 * H1_BK && nemb >= 32 && (long long)nL * nao >= 4 * H1_BM &&
           (long long)nL * nao * nao * 16 < (1LL << 32);
}

// Auxiliary rows one step-1 launch may cover: the per-lane part of an LDS-DMA source address is a 32-bit byte offset from the
// block's base, so a launch spans < 4 GiB of its AO block; blocks beyond that (naux nao^2 >= 2^28) are transformed in several
// launches over ranges of L (capi.hip).  DMK_ERI_HOT_LCHUNK caps it (tests: the cut on small shapes).
int half1_hot_max_rows(int nao) {
    long long rows = ((1LL << 32) - 1) / ((long long)nao * nao * 16);
    if (const char *e = getenv("DMK_ERI_HOT_LCHUNK")) { const int v = atoi(e); if (v > 0 && v < rows) rows = v; }
    return (int)std::min<long long>(rows, 0x7fffffff);
}

// K loop bound of the hot kernels for an AO dimension: the next multiple of the step-1 K tile (8; the step-2 tiles are 4)
int hot_kdim(int nao) { return (nao + H1_BK - 1) / H1_BK * H1_BK; }

// Returns 1 if the hot path handled the launch, 0 if the caller must use the generic kernel, < 0 on error.
// kdim: K loop bound (0: K itself, which must then be a multiple of the K tile); B holds kdim rows, zero beyond K.
static int launch_flat_hot(dmk_ctx *ctx, const void *A, const void *B, void *out, int nL, int K, int mrows, int N, bool conjB,
                           int fam, int nspin = 1, long long b_spin_stride = 0, long long out_spin_stride = 0, int nslot = 1,
                           long long a_slot_stride = 0, long long out_slot_stride = 0, long long b_k_stride = 0,
                           const int *bk = nullptr, int kdim = 0) {
    if (kdim == 0) kdim = K;
    if (kdim < K || (kdim % H1_BK) != 0) return 0;
    if (!half1_hot_usable(nL, K, N) || (long long)nL * mrows < 4 * H1_BM) return 0;
    if ((long long)nL * K * mrows * 16 >= (1LL << 32)) return 0;           // see half1_hot_usable
    static const int bm = [] { const char *e = getenv("DMK_ERI_H1_BM"); return (e && atoi(e) == 64) ? 64 : 128; }();
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(out)) & 15) return 0;
    H1Args a;
    a.Lpq = reinterpret_cast<const double2 *>(A);
    a.Ci = reinterpret_cast<const double2 *>(B);
    a.Ut = reinterpret_cast<double2 *>(out);
    a.nL = nL; a.nao = K; a.nemb = N; a.mrows = mrows; a.kdim = kdim;
    a.nblk = (mrows + 15) / 16;
    a.tiles_m = (int)(((long long)nL * mrows + bm - 1) / bm);        // flat rows: no padding between the nL batches
    // output tile width: 64 columns (2 x 2 waves) or 48 (4 x 1 waves), whichever pads N less; DMK_ERI_H1_BN = 64 | 48 overrides
    int bn = (((N + 47) / 48) * 48 < ((N + 63) / 64) * 64) ? 48 : 64;
    if (const char *e = getenv("DMK_ERI_H1_BN")) { const int v = atoi(e); if (v == 48 || v == 64) bn = v; }
    if (bm != 128) bn = 64;
    a.tiles_n = (N + bn - 1) / bn;
    a.nspin = nspin; a.b_spin_stride = b_spin_stride; a.out_spin_stride = out_spin_stride;
    if (nslot < 1 || nslot > 16) return 0;
    a.nslot = nslot; a.a_slot_stride = a_slot_stride; a.out_slot_stride = out_slot_stride; a.b_k_stride = b_k_stride;
    for (int i = 0; i < 16; ++i) a.bk[i] = (bk && i < nslot) ? bk[i] : 0;
    a.per_slot = (unsigned)(a.tiles_m * a.tiles_n * nspin);
    if ((unsigned long long)a.per_slot * (unsigned)nslot > 0x7fffffffull) return 0;
    a.nblocks = a.per_slot * (unsigned)nslot;
    FamScope fs(ctx, fam);
    fs.mfma_flops(6.0 * (double)a.nblocks * bm * bn * (double)kdim);
    const bool kp = kdim != K;
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a); };
    if (bm == 128 && bn == 48) {
        if (conjB) kp ? go(half1_kernel<true, 128, 2, true, 0, true>) : go(half1_kernel<true, 128, 2, true>);
        else kp ? go(half1_kernel<false, 128, 2, true, 0, true>) : go(half1_kernel<false, 128, 2, true>);
    } else if (bm == 128) {
        if (conjB) kp ? go(half1_kernel<true, 128, 2, false, 0, true>) : go(half1_kernel<true, 128, 2, false>);
        else kp ? go(half1_kernel<false, 128, 2, false, 0, true>) : go(half1_kernel<false, 128, 2, false>);
    } else {
        if (conjB) kp ? go(half1_kernel<true, 64, 3, false, 0, true>) : go(half1_kernel<true, 64, 3, false>);
        else kp ? go(half1_kernel<false, 64, 3, false, 0, true>) : go(half1_kernel<false, 64, 3, false>);
    }
    DMK_CHECK_LAUNCH(ctx);
    return 1;
}

int launch_half1_hot(dmk_ctx *ctx, const void *Lpq, const void *Ci, void *Ut, int nL, int nao, int nemb, int nspin,
                     long long ci_spin_stride, long long ut_spin_stride, int kdim) {
    if (nspin < 1 || nspin > 2) return 0;
    return launch_flat_hot(ctx, Lpq, Ci, Ut, nL, nao, nao, nemb, true, DMK_FAM_ZGEMM_HALF1, nspin, ci_spin_stride, ut_spin_stride,
                           1, 0, 0, 0, nullptr, kdim);
}

// Step 1 of `nslot` queued AO blocks in one launch: block s at Lpq + s * a_slot_stride, transformed with
// C[spin][ki[s]] (C: [spin][nk][nao][nemb], spin stride ci_spin_stride) into Ut + s * ut_slot_stride (+ spin stride).
int launch_half1_hot_multi(dmk_ctx *ctx, const void *Lpq, long long a_slot_stride, int nslot, const int *ki, const void *C,
                           void *Ut, long long ut_slot_stride, int nL, int nao, int nemb, int nspin, long long ci_spin_stride,
                           long long ut_spin_stride, int kdim) {
    if (nspin < 1 || nspin > 2) return 0;
    if (kdim == 0) kdim = nao;
    return launch_flat_hot(ctx, Lpq, C, Ut, nL, nao, nao, nemb, true, DMK_FAM_ZGEMM_HALF1, nspin, ci_spin_stride, ut_spin_stride,
                           nslot, a_slot_stride, ut_slot_stride, (long long)kdim * nemb, ki, kdim);
}

int launch_half2_hot(dmk_ctx *ctx, const void *Ut, long long slot_stride, int nslot, const void *const *Cj,
                     const int *sym, double *planes, long long naux, long long npair, int nL, int nao, int nemb, int nspin,
                     long long ut_spin_stride, long long cj_spin_stride, long long planes_spin_stride, int kdim, int re_only) {
    if (kdim == 0) kdim = nao;
    if (!hot_enabled() || nemb != H2_N || kdim < nao || (kdim % H2_BK) != 0 || nao < 3 * H2_BK || nslot < 1 || nslot > H2_MAXSLOT ||
        nspin < 1 || nspin > 2)
        return 0;
    if (reinterpret_cast<uintptr_t>(Ut) & 15) return 0;
    H2Args a;
    a.Ut = reinterpret_cast<const double2 *>(Ut);
    a.symmask = 0;
    for (int i = 0; i < H2_MAXSLOT; ++i) {
        a.Cj[i] = reinterpret_cast<const double2 *>(Cj[i < nslot ? i : 0]);
        if (i < nslot && sym[i]) a.symmask |= 1u << i;
        if (reinterpret_cast<uintptr_t>(a.Cj[i]) & 15) return 0;
    }
    a.slot_stride = slot_stride;
    a.planes = planes; a.naux = naux; a.npair = npair;
    a.nL = nL; a.nao = nao; a.nslot = nslot; a.kdim = kdim;
    a.nspin = nspin;
    a.ut_spin_stride = ut_spin_stride; a.cj_spin_stride = cj_spin_stride; a.planes_spin_stride = planes_spin_stride;
    a.nblocks = (unsigned)(4 * nL * nspin);
    a.fold_diag = (a.symmask == (nslot >= 32 ? 0xffffffffu : ((1u << nslot) - 1u))) ? 1 : 0;
    FamScope fs(ctx, DMK_FAM_ZGEMM_HALF2);
    {   // 136 of the 256 16 x 16 blocks per L and spin; a block with the time-reversal partner term runs a second segment
        // (without the 16 diagonal blocks when the whole group is symmetrised: they are folded in the epilogue)
        double blocks = 0.0;
        for (int i = 0; i < nslot; ++i) blocks += 136.0 + (sym[i] ? (a.fold_diag ? 120.0 : 136.0) : 0.0);
        fs.mfma_flops((re_only ? 4.0 : 6.0) * blocks * 256.0 * (double)kdim * (double)nL * (double)nspin);
    }
    if (re_only) hipLaunchKernelGGL((half2_kernel<0, true>), dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a);
    else hipLaunchKernelGGL((half2_kernel<0, false>), dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a);
    DMK_CHECK_LAUNCH(ctx);
    return 1;
}

int half2_hot_usable(int nao, int nemb) { return hot_enabled() && nemb == H2_N && nao >= 3 * H2_BK; }
int half2_hot_maxslot() { return H2_MAXSLOT; }
