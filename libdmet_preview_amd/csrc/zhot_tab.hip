// K6 hot step 2 for a GENERAL embedding dimension (any nemb >= 32; nemb == 256 keeps its specialised kernel in zhot.hip).
//
//   S_L[a][b] = sum_q Ut[L][q][a] C_j[q][b] (+ sum_q C_j[q][a] Ut[L][q][b]),  a >= b, tril-packed and ACCUMULATED
//   into the Re / Im planes of Lij_s4              reference: basis_transform/eri_transform.py:368-378, 403-434
//
// The reference's r_e2 is shape agnostic; round 1 only had the LDS-DMA ring kernels for nemb = 256 and sent every
// other embedding size (C4: nemb = 136) through the generic register-staged zgemm, one launch per AO block and spin
// (0.1 - 0.3 ms launches: ramp-up / drain bound, 23 TF algorithmic).  This kernel keeps the structure that made the
// 256 kernel fast -- operands by LDS-DMA into a 3-4 stage ring retired by counted s_waitcnt, ONE barrier per K step,
// 3M complex product, two 256-thread workgroups per CU, the ring running straight through up to 16 queued AO blocks,
// fire-and-forget atomics with one writer per plane element -- and replaces its hard-wired block ownership by a
// host-built TABLE (cached in the context per nemb):
//   * the lower triangle of the nb x nb grid of 16 x 16 blocks (nb = ceil(nemb / 16)) is cut along segments of 7
//     blocks into diagonal TRIANGLES (<= 28 blocks, operand panels U[.][128], C[.][128]: 16 KiB per stage, 4 stages)
//     and off-diagonal RECTANGLES of <= 4 x 7 blocks (panels Ua[.][64], Cb[.][128], Ca[.][64], Ub[.][128]: 24 KiB per
//     stage, 3 stages) -- one workgroup per item and auxiliary index L;
//   * inside an item every wave owns an explicit list of <= 8 blocks (an even row-major split of the item's blocks),
//     read from the table into SGPRs; a block costs two (with the
//     time-reversal partner four) fragment reads per K step instead of sharing them along a row -- 42 B/clk/CU of LDS
//     bandwidth against 128 available -- in exchange for ONE code path for every shape.
//   * embedding spaces of up to 12 blocks (nemb <= 192) use WIDE items instead: panels U[.][192], C[.][192] over the
//     whole matrix (24 KiB per stage, 3 stages) and the triangle's blocks dealt out evenly -- no light workgroups.
//   Block efficiency (useful blocks / 4 waves x longest list): nemb 136 -> 94 %, 200 -> 95 %, 272 -> 93 %.
// Panels wider than the matrix re-read clamped valid columns; they only ever feed masked outputs.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "zhot_common.h"

namespace {

constexpr int T_BK = 4;
constexpr int T_MAXSLOT = 16;
constexpr int T_ITEM = 48;                             // ints per table item
constexpr int TD_STAGE = T_BK * 256;                   // diagonal item: U[4][128] | C[4][128]                          (16 KiB)
constexpr int TR_STAGE = T_BK * 384;                   // rectangle:     Ua[4][64] | Cb[4][128] | Ca[4][64] | Ub[4][128] (24 KiB)
                                                       // wide item:     U[4][192] | C[4][192] (whole matrix, nemb <= 192) (24 KiB)
constexpr int T_WIDE_MAXNB = 12;

// Two occupancy points of the same kernel:
//   Cfg2: <= 8 accumulator tiles per wave (192 VGPRs + fragments), TWO workgroups per CU, rings of 4 / 3 stages (72 KiB);
//   Cfg3: <= 5 accumulator tiles per wave (120 VGPRs + fragments; 6 spill at the 170-register limit), THREE workgroups per CU, rings of 3 / 2 stages (48 KiB) -- the
//         third wave per SIMD covers the barrier / LDS round trips of the other two (the same lever that took the
//         contraction kernel from 68 to 71 TF).
struct Cfg2 { static constexpr int MAXBLK = 8, SEG = 7, OCC = 2, TD_D = 4, TR_D = 3; };
struct Cfg3 { static constexpr int MAXBLK = 5, SEG = 5, OCC = 3, TD_D = 3, TR_D = 2; };
template <class CFG> constexpr int lds_elems() {
    return (TR_STAGE * CFG::TR_D > TD_STAGE * CFG::TD_D) ? TR_STAGE * CFG::TR_D : TD_STAGE * CFG::TD_D;
}

struct H2TArgs {
    const double2 *Ut;               // [nslot][nL][nao][nemb]
    const double2 *Cj[T_MAXSLOT];    // [nao][nemb] of each queued block
    unsigned symmask;
    long long slot_stride;
    double *planes;
    long long naux, npair;
    int nL, nao, nslot, nemb;
    int kdim;                        // K loop bound: nao rounded up to the K tile (zhot.hip H2Args::kdim)
    unsigned nblocks;
    int nspin;
    long long ut_spin_stride, cj_spin_stride, planes_spin_stride;
    int fold_diag;                   // every queued block is symmetrised: diagonal blocks run one segment and are folded (zhot.hip)
    const int *table;                // nitems x T_ITEM: kind, R0, C0, nblk[4], pad, entries[4][8] = (row block << 8) | col block (local)
    int nitems;
    // SUB-GROUPS: the queue of nslot blocks is cut into nsub runs of sub_slots consecutive blocks; every (L, item) has one
    // workgroup PER RUN, and run p >= 1 accumulates into its own copy of the planes (planes_sub + (p - 1) sub_stride) that the
    // pipeline adds to the kL's planes in a fixed order when the kL ends -- still exactly one writer per plane element and
    // launch.  nsub x more, nsub x shorter workgroups: a launch of 1.6 rounds of resident workgroups (C4: 1248 on 768
    // slots, the last round 62 % full) becomes one of 6.5.
    int nsub, sub_slots;
    double *planes_sub;
    long long sub_stride;
    unsigned per_sub;                // workgroups per run = nitems * nL * nspin
};

// kernel-argument arrays are only ever indexed by constants (see zhot.hip)
#define T_PICK_CJ(G, SLOT)                                                                         \
    ((SLOT) == 0 ? (G).Cj[0] : (SLOT) == 1 ? (G).Cj[1] : (SLOT) == 2 ? (G).Cj[2] : (SLOT) == 3 ? (G).Cj[3]      \
     : (SLOT) == 4 ? (G).Cj[4] : (SLOT) == 5 ? (G).Cj[5] : (SLOT) == 6 ? (G).Cj[6] : (SLOT) == 7 ? (G).Cj[7]    \
     : (SLOT) == 8 ? (G).Cj[8] : (SLOT) == 9 ? (G).Cj[9] : (SLOT) == 10 ? (G).Cj[10] : (SLOT) == 11 ? (G).Cj[11] \
     : (SLOT) == 12 ? (G).Cj[12] : (SLOT) == 13 ? (G).Cj[13] : (SLOT) == 14 ? (G).Cj[14] : (G).Cj[15])

// LAB: ablation bits of tools/zhot_lab.hip as in zhot.hip (1: no plane atomics, 2: no LDS-DMA after the prologue, 4: no
// s_barrier); the product instantiates LAB = 0.
template <class CFG, int LAB = 0, bool RE = false>
__global__ __launch_bounds__(HNT, CFG::OCC) void half2_tab_kernel(const H2TArgs g) {
    constexpr int T_MAXBLK = CFG::MAXBLK;
    __shared__ __attribute__((aligned(16))) double2 lds[lds_elems<CFG>()];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frag_k = lane >> 4, frag_x = lane & 15;
    const unsigned lid_all = xcd_remap(blockIdx.x, g.nblocks);
    const int sub = (int)(lid_all / g.per_sub);
    const unsigned lid = lid_all - (unsigned)sub * g.per_sub;
    const int Lall = (int)(lid / (unsigned)g.nitems), item = (int)(lid - (unsigned)Lall * (unsigned)g.nitems);
    const int sp = Lall >= g.nL ? 1 : 0;
    const int L = Lall - sp * g.nL;
    const long long nemb = g.nemb;
    const int Tb = g.kdim / T_BK;
    const int slot0 = sub * g.sub_slots;                                           // this run: queue slots [slot0, slot0 + nmine)
    const int nmine = g.nslot - slot0 < g.sub_slots ? g.nslot - slot0 : g.sub_slots;
    const int T = Tb * nmine;
    const long long g_naux = g.naux, g_npair = g.npair, g_slot_stride = g.slot_stride;
    const double2 *Ubase = g.Ut + (long long)sp * g.ut_spin_stride + (long long)L * g.nao * nemb + (long long)slot0 * g_slot_stride;
    double *const g_planes = (sub == 0 ? g.planes + (long long)sp * g.planes_spin_stride
                                       : g.planes_sub + (long long)(sub - 1) * g.sub_stride + (long long)sp * 2LL * g_naux * g_npair);
    const long long cj_off = (long long)sp * g.cj_spin_stride;
    const unsigned g_symmask = g.symmask >> slot0;

    // ---- this workgroup's item and this wave's block list (wave-uniform: SGPRs) ----------------------------------
    const int *it = g.table + (long long)item * T_ITEM;
    const int kind = __builtin_amdgcn_readfirstlane(it[0]);
    const int R0 = __builtin_amdgcn_readfirstlane(it[1]), C0 = __builtin_amdgcn_readfirstlane(it[2]);
    const int nblk = __builtin_amdgcn_readfirstlane(it[3 + wave]);
    // diagonal blocks of this wave (the host puts them LAST in its list): with `fold_diag` they skip segment 2
    const int ndiag = g.fold_diag ? (__builtin_amdgcn_readfirstlane(it[7]) >> (8 * wave)) & 255 : 0;
    const int nb2 = nblk - ndiag;
    const bool wg_has_diag = g.fold_diag && __builtin_amdgcn_readfirstlane(it[7]) != 0;      // uniform over the workgroup
    int ro[T_MAXBLK], co[T_MAXBLK];              // local element offsets of each block inside the row / column panels
#pragma unroll
    for (int i = 0; i < T_MAXBLK; ++i) {
        const int e = __builtin_amdgcn_readfirstlane(it[8 + wave * T_MAXBLK + i]);
        ro[i] = (e >> 8) * 16;
        co[i] = (e & 255) * 16;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the table reads must not sit in the counted DMA ring

    const int r0 = R0 * 16, c0 = C0 * 16;
    auto clampcol = [&](int c) { return c < g.nemb ? c : g.nemb - 1; };

    // The whole item, specialised on its kind and on the LENGTH of this wave's block list: with a static list length the
    // K step is straight-line code (the compiler software-pipelines the fragment reads under the MFMAs) and the
    // accumulators are exactly 24 NB registers.  (A single loop with `if (i < nblk)` around every block made hipcc spill
    // 630 VGPRs once the reads were pipelined by hand, and left every block with an exposed ds_read round trip when
    // they were not: 37 TF on the pipe at C4 shapes.)
    auto run = [&](auto kindtag, auto nbtag) {
        constexpr int KIND = decltype(kindtag)::value, NB = decltype(nbtag)::value;
        constexpr int STAGE = KIND == 0 ? TD_STAGE : TR_STAGE, D = KIND == 0 ? CFG::TD_D : CFG::TR_D;   // wide == rectangle
        constexpr int NP = KIND == 0 ? 4 : 6;                            // LDS-DMA pieces per wave and stage
        // panel offsets inside a stage and panel widths (complex elements)
        constexpr int PUA = 0, PCB = KIND == 0 ? 512 : KIND == 1 ? 256 : 768, PCA = KIND == 0 ? 512 : 768,
                      PUB = KIND == 1 ? 1024 : 0;
        constexpr int PWR = KIND == 0 ? 128 : KIND == 1 ? 64 : 192, PWC = KIND == 2 ? 192 : 128;

        unsigned voff[NP];                                               // this lane's byte offset inside one K step of the operand
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            const int piece = wave + 4 * h;
            int row, col;
            if (KIND == 0) {
                // 16 pieces of 64 complex: piece p < 8 -> U row p/2, half p%2 ; p >= 8 -> C likewise
                row = (piece & 7) >> 1; col = r0 + (piece & 1) * 64 + lane;
            } else if (KIND == 2) {
                // 24 pieces: piece p < 12 -> U row p/3, third p%3 ; p >= 12 -> C likewise (the whole matrix width)
                const int q = piece >= 12 ? piece - 12 : piece;
                row = q / 3; col = (q % 3) * 64 + lane;
            } else {
                // 24 pieces: 0-3 Ua rows, 4-11 Cb (row*2+half), 12-15 Ca rows, 16-23 Ub (row*2+half)
                if (piece < 4) { row = piece; col = r0 + lane; }
                else if (piece < 12) { row = (piece - 4) >> 1; col = c0 + ((piece - 4) & 1) * 64 + lane; }
                else if (piece < 16) { row = piece - 12; col = r0 + lane; }
                else { row = (piece - 16) >> 1; col = c0 + ((piece - 16) & 1) * 64 + lane; }
            }
            voff[h] = (unsigned)((row * (int)nemb + clampcol(col)) * 16);
        }
        int is_t = 0, is_slot = slot0, is_stage = 0;
        const double2 *is_ub = Ubase, *is_cb = T_PICK_CJ(g, slot0) + cj_off;
        auto issue = [&]() {
            double2 *st = lds + is_stage * STAGE;
            // which operand piece wave + 4 h belongs to depends on h only (diagonal: U U C C; rectangle: Ua Cb Cb Ca Ub Ub; wide:
            // U U U C C C): scalar tile bases + loop-invariant per-lane byte offsets, no vector ALU work per piece (common.h)
            if constexpr (KIND == 0)
                glds16s_x4(voff[0], voff[1], voff[2], voff[3], is_ub, is_ub, is_cb, is_cb, lds_addr_of(st + wave * 64),
                           lds_addr_of(st + (wave + 4) * 64), lds_addr_of(st + (wave + 8) * 64), lds_addr_of(st + (wave + 12) * 64));
            else if constexpr (KIND == 1)
                glds16s_x6(voff[0], voff[1], voff[2], voff[3], voff[NP - 2], voff[NP - 1], is_ub, is_cb, is_cb, is_cb, is_ub, is_ub,
                           lds_addr_of(st + wave * 64), lds_addr_of(st + (wave + 4) * 64), lds_addr_of(st + (wave + 8) * 64),
                           lds_addr_of(st + (wave + 12) * 64), lds_addr_of(st + (wave + 16) * 64), lds_addr_of(st + (wave + 20) * 64));
            else
                glds16s_x6(voff[0], voff[1], voff[2], voff[3], voff[NP - 2], voff[NP - 1], is_ub, is_ub, is_ub, is_cb, is_cb, is_cb,
                           lds_addr_of(st + wave * 64), lds_addr_of(st + (wave + 4) * 64), lds_addr_of(st + (wave + 8) * 64),
                           lds_addr_of(st + (wave + 12) * 64), lds_addr_of(st + (wave + 16) * 64), lds_addr_of(st + (wave + 20) * 64));
            is_stage = is_stage + 1 == D ? 0 : is_stage + 1;
            if (++is_t == Tb) {
                is_t = 0;
                ++is_slot;
                is_ub = Ubase + (long long)(is_slot - slot0) * g_slot_stride;
                is_cb = T_PICK_CJ(g, is_slot) + cj_off;
            } else {
                is_ub += T_BK * nemb;
                is_cb += T_BK * nemb;
            }
        };

        cacc acc[NB > 0 ? NB : 1];
#pragma unroll
        for (int i = 0; i < NB; ++i) cacc_zero(acc[i]);

        issue();
        if (D > 2 && T > 1) issue();
        if (D > 3 && T > 2) issue();
        int c_t = 0, c_stage = 0;
        unsigned c_sym = g_symmask & 1u, c_mask = g_symmask;
        for (int t = 0; t < T; ++t) {
            // tile t must have landed; up to D - 2 younger tiles (NP loads each) may still be in flight
            const int later = T - 1 - t;
            const int young = later < D - 2 ? later : D - 2;
            if (young == 2) {
                if (NP == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else if (young == 1) {
                if (NP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr (!(LAB & 4)) __builtin_amdgcn_s_barrier();
            if constexpr (LAB & 2) { if (t + D - 1 < T && g.nslot < 0) issue(); }
            else { if (t + D - 1 < T) issue(); }
            const double2 *st = lds + c_stage * STAGE;
            c_stage = c_stage + 1 == D ? 0 : c_stage + 1;
            if (NB > 0) {
                const double2 *rowU = st + PUA + frag_k * PWR + frag_x, *colC = st + PCB + frag_k * PWC + frag_x;
#pragma unroll
                for (int i = 0; i < NB; ++i) {                          // S[a][b] += U[q][a] C[q][b]
                    const cfrag a = cfrag_of_t<RE>(lds_frag(rowU + ro[i])), b = cfrag_of_t<RE>(lds_frag(colC + co[i]));
                    cmfma_t<RE>(acc[i], a, b);
                }
                if (c_sym) {
                    const double2 *rowC = st + PCA + frag_k * PWR + frag_x, *colU = st + PUB + frag_k * PWC + frag_x;
#pragma unroll
                    for (int i = 0; i < NB; ++i) {                      // S[a][b] += C[q][a] U[q][b]
                        // at most the last two blocks of a list are diagonal ones: only they carry a (uniform) branch
                        if (i + 2 < NB || i < nb2) {
                            const cfrag a = cfrag_of_t<RE>(lds_frag(rowC + ro[i])), b = cfrag_of_t<RE>(lds_frag(colU + co[i]));
                            cmfma_t<RE>(acc[i], a, b);
                        }
                    }
                }
            }
            if (++c_t == Tb) {
                c_t = 0;
                c_mask >>= 1;
                c_sym = c_mask & 1u;
            }
        }
        if (wg_has_diag) {
            // folded diagonal blocks hold P = U_r^T C_r only: add P^T through a wave-private LDS tile (the ring is idle now)
            __syncthreads();
            double *tr = reinterpret_cast<double *>(lds) + wave * (2 * 16 * 17);
#pragma unroll
            for (int i = (NB >= 2 ? NB - 2 : 0); i < NB; ++i) {
                if (i >= nb2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tr[(frag_k + 4 * r) * 17 + frag_x] = cacc_re(acc[i], r);
                        if constexpr (!RE) tr[272 + (frag_k + 4 * r) * 17 + frag_x] = cacc_im(acc[i], r);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                  // the LDS pipe keeps a wave's own accesses in order
                        const double tre = tr[frag_x * 17 + frag_k + 4 * r], tim = RE ? 0.0 : tr[272 + frag_x * 17 + frag_k + 4 * r];
                        acc[i].p[r] += tre;                        // Re += tre, Im += tim in the (T1, T2, T3) representation
                        if constexpr (!RE) acc[i].t[r] += tre + tim;
                    }
                }
            }
        }
        if constexpr (LAB & 1) { if (g.nslot >= 0) return; }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pack_acc_t<RE>(g_planes, g_naux, g_npair, L, r0 + ro[i] + frag_k + 4 * r, c0 + co[i] + frag_x, acc[i], r, g.nemb);
    };
    auto by_len = [&](auto kindtag) {
        auto go = [&](auto nbtag) {
            if constexpr (decltype(nbtag)::value <= T_MAXBLK) run(kindtag, nbtag);
        };
        switch (nblk) {
            case 0: go(std::integral_constant<int, 0>{}); break;
            case 1: go(std::integral_constant<int, 1>{}); break;
            case 2: go(std::integral_constant<int, 2>{}); break;
            case 3: go(std::integral_constant<int, 3>{}); break;
            case 4: go(std::integral_constant<int, 4>{}); break;
            case 5: go(std::integral_constant<int, 5>{}); break;
            case 6: go(std::integral_constant<int, 6>{}); break;
            case 7: go(std::integral_constant<int, 7>{}); break;
            default: go(std::integral_constant<int, 8>{}); break;
        }
    };
    if (kind == 0) by_len(std::integral_constant<int, 0>{});
    else if (kind == 1) by_len(std::integral_constant<int, 1>{});
    else by_len(std::integral_constant<int, 2>{});
}

bool tab_enabled() {
    static const bool on = [] { const char *e = getenv("DMK_ERI_HOT"); return !(e && atoi(e) == 0); }();
    return on;
}

// Host side of the decomposition described at the top: items of the lower block triangle and the per-wave block lists.
void build_table(int nemb, int T_MAXBLK, int T_SEG, std::vector<int> &tab, double &useful_blocks, double &slots, double &folded) {
    const int nb = (nemb + 15) / 16;
    tab.clear();
    useful_blocks = slots = folded = 0.0;
    auto push_item = [&](int kind, int R0, int C0, std::vector<std::pair<int, int>> lists[4]) {
        const size_t base = tab.size();
        tab.resize(base + T_ITEM, 0);
        tab[base] = kind; tab[base + 1] = R0; tab[base + 2] = C0;
        size_t longest = 0;
        for (int w = 0; w < 4; ++w) {
            // diagonal blocks (same global block row and column) go last and are counted: they may skip segment 2
            std::vector<std::pair<int, int>> ordered, diag;
            for (auto &rc : lists[w]) ((kind != 1 && R0 + rc.first == C0 + rc.second) ? diag : ordered).push_back(rc);
            if (diag.size() > 2) {                            // the kernel folds at most the last two blocks of a list
                ordered.insert(ordered.end(), diag.begin(), diag.end() - 2);
                diag.erase(diag.begin(), diag.end() - 2);
            }
            tab[base + 7] |= (int)diag.size() << (8 * w);
            folded += (double)diag.size();
            ordered.insert(ordered.end(), diag.begin(), diag.end());
            lists[w] = ordered;
            tab[base + 3 + w] = (int)lists[w].size();
            longest = std::max(longest, lists[w].size());
            useful_blocks += (double)lists[w].size();
            for (size_t i = 0; i < lists[w].size(); ++i)
                tab[base + 8 + w * T_MAXBLK + i] = (lists[w][i].first << 8) | lists[w][i].second;
        }
        slots += 4.0 * (double)longest;
    };
    auto even_split = [](const std::vector<std::pair<int, int>> &all, std::vector<std::pair<int, int>> lists[4]) {
        const size_t base = all.size() / 4, extra = all.size() % 4;
        size_t pos = 0;
        for (int w = 0; w < 4; ++w) {
            const size_t n = base + ((size_t)w < extra ? 1 : 0);
            lists[w].assign(all.begin() + pos, all.begin() + pos + n);
            pos += n;
        }
    };
    if (nb <= T_WIDE_MAXNB) {
        // small embedding spaces: every workgroup sees the whole matrix width (panels U[.][192], C[.][192]), so any block can go
        // to any wave -- no light items (C4: 45 blocks -> 3 workgroups x 4 waves instead of a 28-block triangle plus a 14- and
        // a 3-block item).  Balanced in SEGMENTS, not in blocks: in an all-symmetrised group (the common case) an off-diagonal block runs two
        // segments per K step and a folded diagonal block one, so an even split of the row-major block list left waves with
        // 5 to 8 segments in one workgroup (C4: 81 useful of 92 occupied segment slots).  Diagonal blocks are dealt out first,
        // one per wave, then every off-diagonal block goes to the wave with the lightest load (ties: fewer blocks): C4 ->
        // nine waves of 3 + 1 blocks (7 segments) and three of 3 (6 segments), 81 of 84 slots; by block count (groups that
        // are not all symmetrised) the same lists are 4, 4, 4, 3 per workgroup.
        const int nblk_all = nb * (nb + 1) / 2;
        const int nwg = (nblk_all + 4 * T_MAXBLK - 1) / (4 * T_MAXBLK);
        const int nw = 4 * nwg;
        std::vector<std::vector<std::pair<int, int>>> wl(nw);
        std::vector<int> load(nw, 0);
        for (int d = 0; d < nb; ++d) {                                // diagonal blocks round-robin
            wl[d % nw].push_back({d, d});
            load[d % nw] += 1;
        }
        for (int r = 0; r < nb; ++r)
            for (int c = 0; c < r; ++c) {
                int best = -1;
                for (int w = 0; w < nw; ++w) {
                    if ((int)wl[w].size() >= T_MAXBLK) continue;
                    if (best < 0 || load[w] < load[best] || (load[w] == load[best] && wl[w].size() < wl[best].size())) best = w;
                }
                wl[best].push_back({r, c});
                load[best] += 2;
            }
        // workgroups of four waves with similar loads (a workgroup lasts as long as its heaviest wave)
        std::vector<int> order(nw);
        for (int w = 0; w < nw; ++w) order[w] = w;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return load[a] > load[b]; });
        for (int wg = 0; wg < nwg; ++wg) {
            std::vector<std::pair<int, int>> lists[4];
            for (int w = 0; w < 4; ++w) lists[w] = wl[order[4 * wg + w]];
            push_item(2, 0, 0, lists);
        }
        return;
    }
    for (int s0 = 0; s0 < nb; s0 += T_SEG) {
        const int w = std::min(T_SEG, nb - s0);
        {   // diagonal triangle of width w: w (w + 1) / 2 <= 28 blocks, row-major, cut evenly over the four waves
            std::vector<std::pair<int, int>> all, lists[4];
            for (int r = 0; r < w; ++r)
                for (int c = 0; c <= r; ++c) all.push_back({r, c});
            even_split(all, lists);
            push_item(0, s0, s0, lists);
        }
        for (int t0 = 0; t0 < s0; t0 += T_SEG) {          // rectangles below the diagonal: rows of this segment x cols of segment t0
            const int cw = T_SEG;                           // earlier segments are always full
            for (int rh = 0; rh < w; rh += 4) {
                const int rw = std::min(4, w - rh);
                std::vector<std::pair<int, int>> all, lists[4];
                for (int r = 0; r < rw; ++r)
                    for (int c = 0; c < cw; ++c) all.push_back({r, c});
                even_split(all, lists);
                push_item(1, s0 + rh, t0, lists);
            }
        }
    }
}

}  // namespace

int half2_tab_usable(int nao, int nemb) {
    return tab_enabled() && nemb >= 32 && nemb <= 4096 && nao >= 3 * T_BK;
}
int half2_tab_maxslot() { return T_MAXSLOT; }

// Returns 1 if handled, 0 if the caller must use the generic kernel, < 0 on error.  Arguments as launch_half2_hot (zhot.hip).
// Sub-groups of a step-2 launch (see H2TArgs): how many runs the queue of `nslot` blocks should be cut into so that the launch
// has at least ~6 rounds of resident workgroups; 1 when it already has, or when the queue is too short to cut.
int half2_tab_subgroups(dmk_ctx *ctx, int nL, int nao, int nemb, int nspin, int nslot, int max_sub) {
    (void)ctx; (void)nao; (void)nL; (void)nemb; (void)nspin;
    // Measured at C4 (round 4, profiles/r04_*): 1 / 2 / 4 runs per launch = 1.438 / 1.425 / 1.443 ms per 16-block launch, and each
    // extra run costs a pass over the planes when the kL ends -- the partial last round of workgroups is NOT what holds this
    // kernel back.  One run per launch therefore stays the default; DMK_ERI_TAB_SUB = 2..4 cuts the queue for experiments on
    // other shapes.
    int p = 1;
    if (const char *e = getenv("DMK_ERI_TAB_SUB")) p = atoi(e);
    return std::max(1, std::min(std::min(p, max_sub), std::max(1, nslot / 2)));
}

int launch_half2_tab(dmk_ctx *ctx, const void *Ut, long long slot_stride, int nslot, const void *const *Cj, const int *sym,
                     double *planes, long long naux, long long npair, int nL, int nao, int nemb, int nspin,
                     long long ut_spin_stride, long long cj_spin_stride, long long planes_spin_stride, int nsub,
                     double *planes_sub, long long sub_stride, int kdim, int re_only) {
    if (kdim == 0) kdim = nao;
    if (kdim < nao || (kdim % T_BK) != 0) return 0;
    if (!half2_tab_usable(nao, nemb) || nslot < 1 || nslot > T_MAXSLOT || nspin < 1 || nspin > 2) return 0;
    if (reinterpret_cast<uintptr_t>(Ut) & 15) return 0;
    // occupancy point (see Cfg2 / Cfg3).  Measured (MI355X, executed TF of this kernel): the evenly dealt WIDE items of small
    // embedding spaces gain from the third wave per SIMD (C4, nemb 136: 45.6 -> 54.8), the segment items of larger ones
    // lose more from their shorter block lists than they gain (nemb 256 routed here: 56.9 vs 50.4; its specialised
    // kernel: 66.5).  DMK_ERI_TAB_OCC = 2 | 3 overrides.
    // fault injection (tests): decline the launch so that the caller's generic fallback runs
    if (const char *e = getenv("DMK_ERI_TAB_DECLINE")) if (atoi(e) != 0) return 0;
    const char *occ_e = getenv("DMK_ERI_TAB_OCC");          // read per launch (a handful per second): tests toggle it
    const int occ_env = occ_e ? atoi(occ_e) : 0;
    const int occ = (occ_env == 2 || occ_env == 3) ? occ_env : ((nemb + 15) / 16 <= T_WIDE_MAXNB ? 3 : 2);
    const dmk_ctx::StepTable *tb = nullptr;
    for (auto &t : ctx->step2_tables)
        if (t.nemb == nemb && t.cfg == occ) tb = &t;
    if (!tb) {
        std::vector<int> h;
        double useful, slots, folded;
        if (occ == 2) build_table(nemb, Cfg2::MAXBLK, Cfg2::SEG, h, useful, slots, folded);
        else build_table(nemb, Cfg3::MAXBLK, Cfg3::SEG, h, useful, slots, folded);
        int *dev = nullptr;
        if (dmk_dev_alloc(ctx, reinterpret_cast<void **>(&dev), h.size() * sizeof(int)) != hipSuccess)
            return dmk_fail(ctx, DMK_ERR_NOMEM, "half2_tab: table allocation failed");
        if (hipMemcpy(dev, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(dev);
            return dmk_fail(ctx, DMK_ERR_HIP, "half2_tab: table upload failed");
        }
        ctx->step2_tables.push_back({nemb, occ, (int)(h.size() / T_ITEM), useful, folded, dev});
        tb = &ctx->step2_tables.back();
    }
    H2TArgs a;
    a.Ut = reinterpret_cast<const double2 *>(Ut);
    a.symmask = 0;
    double segs = 0.0;
    for (int i = 0; i < T_MAXSLOT; ++i) {
        a.Cj[i] = reinterpret_cast<const double2 *>(Cj[i < nslot ? i : 0]);
        if (i < nslot && sym[i]) a.symmask |= 1u << i;
        if (i < nslot) segs += sym[i] ? 2.0 : 1.0;
        if (reinterpret_cast<uintptr_t>(a.Cj[i]) & 15) return 0;
    }
    a.slot_stride = slot_stride;
    a.planes = planes; a.naux = naux; a.npair = npair;
    a.nL = nL; a.nao = nao; a.nslot = nslot; a.nemb = nemb; a.kdim = kdim;
    a.nspin = nspin;
    a.ut_spin_stride = ut_spin_stride; a.cj_spin_stride = cj_spin_stride; a.planes_spin_stride = planes_spin_stride;
    a.table = tb->dev; a.nitems = tb->nitems;
    a.fold_diag = (a.symmask == (nslot >= 32 ? 0xffffffffu : ((1u << nslot) - 1u))) ? 1 : 0;
    if (nsub < 1 || !planes_sub) nsub = 1;
    a.sub_slots = (nslot + nsub - 1) / nsub;
    a.nsub = (nslot + a.sub_slots - 1) / a.sub_slots;                 // runs that actually hold blocks
    a.planes_sub = planes_sub; a.sub_stride = sub_stride;
    const unsigned long long per_sub = (unsigned long long)tb->nitems * (unsigned)nL * (unsigned)nspin;
    const unsigned long long nblocks = per_sub * (unsigned)a.nsub;
    if (nblocks > 0x7fffffffull) return 0;
    a.per_sub = (unsigned)per_sub;
    a.nblocks = (unsigned)nblocks;
    FamScope fs(ctx, DMK_FAM_ZGEMM_HALF2);
    // a symmetrised block runs a second segment -- without the folded diagonal blocks when the whole group is symmetrised
    const double seg2 = (segs - (double)nslot) * (tb->useful_blocks - (a.fold_diag ? tb->folded_blocks : 0.0));
    fs.mfma_flops((re_only ? 4.0 : 6.0) * ((double)nslot * tb->useful_blocks + seg2) * 256.0 * (double)kdim * (double)nL * (double)nspin);
    if (occ == 2) {
        if (re_only) hipLaunchKernelGGL((half2_tab_kernel<Cfg2, 0, true>), dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a);
        else hipLaunchKernelGGL((half2_tab_kernel<Cfg2, 0, false>), dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a);
    } else {
        if (re_only) hipLaunchKernelGGL((half2_tab_kernel<Cfg3, 0, true>), dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a);
        else hipLaunchKernelGGL((half2_tab_kernel<Cfg3, 0, false>), dim3(a.nblocks), dim3(HNT), 0, ctx->stream, a);
    }
    DMK_CHECK_LAUNCH(ctx);
    return 1;
}
