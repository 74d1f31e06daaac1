// LAB ONLY (tools/zhot_lab.hip includes this after libdmet_preview_amd/csrc/zhot.hip): the PERSISTENT form of step 1 that round 4
// tried and did not ship.  Measured on MI355X (profiles/r04_e_zhot_lab.txt): the K loop alone gains (everything but MFMA + fragment
// reads removed: 94 % of the pipe peak against 92 % at C5 shapes, 89-91 % against 85 % at C4), but vmcnt retires in order, stores
// included, so the epilogue's stores sit in front of every LDS-DMA piece issued after them and the ring stalls on their completion
// once per output tile: 75-79 % against 87 % with one workgroup per tile at C5, 71-73 % against 72-77 % at C4.  Issuing a third K tile
// ahead of the stores recovers 4 points, spreading the workgroups' phases none.  A workgroup per tile leaves its stores behind when
// it exits and never waits for them; that form stays in the product.
#pragma once
namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// step 1, PERSISTENT form: the same tiles, K loop and ring as half1_kernel, but a launch holds OCC workgroups per CU and every
// workgroup walks a contiguous run of tiles with the LDS-DMA ring running STRAIGHT THROUGH the tile boundaries: the first two K
// tiles of the next output tile are issued under the last two K steps of the current one, so a tile pays neither the DMA ramp-up
// (one HBM round trip per 13 K steps at C4 where K = 104) nor an idle ring under its epilogue.
//   * vmcnt returns in order on gfx9, stores included: the epilogue's stores are issued AFTER the next tile's first two K tiles
//     and are therefore only waited for at the third K step of the next tile, a microsecond later; the first two K steps wait with
//     vmcnt(pieces + NST) where NST is the EXACT number of store instructions of the epilogue.  The stores are written in inline
//     asm for that reason (always issued: masked lanes are pointed at a sink; never counted, coalesced or skipped by the compiler).
//   * consecutive tiles of a run share the A tile (n tile fastest, then spin), which the run's first tile pulled into L2.
__device__ double2 h1_sink[64];
typedef double h1_d2_t __attribute__((ext_vector_type(2)));

template <bool CONJB, int BM, int OCC, bool NARROW, int LAB = 0>
__global__ __launch_bounds__(HNT, OCC) void half1p_kernel(const H1Args g) {
    constexpr int MI = NARROW ? BM / 64 : BM / 32;
    constexpr int NJ = NARROW ? 3 : 2;
    constexpr int BN = NARROW ? 48 : H1_BN;
    constexpr int AH = BM / 64;
    constexpr int STAGE = H1_BK * (BM + H1_BN);
    constexpr int NST = (LAB & 1) ? 0 : MI * 4 * NJ;                 // store instructions per wave and epilogue
    static_assert(2 * (2 * AH + 2) + NST <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) double2 lds[H1_D * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
    const int frag_k = lane >> 4, frag_x = lane & 15;

    const unsigned first = (unsigned)((unsigned long long)blockIdx.x * g.nblocks / gridDim.x);
    const unsigned last = (unsigned)((unsigned long long)(blockIdx.x + 1) * g.nblocks / gridDim.x);
    if (first >= last) return;
    const unsigned nao = (unsigned)g.nao, mrows = (unsigned)g.mrows;
    const unsigned rows_total = (unsigned)g.nL * mrows;              // < 2^31 (launch_flat_hot)
    const long long nemb = g.nemb;
    const int T = g.nao / H1_BK;

    // ---- the tile being FETCHED (runs up to two K steps ahead of the tile being computed) ----------------------------------
    int f_slot, f_tm, f_sp, f_tn;
    {
        f_slot = (int)(first / g.per_slot);
        const unsigned lid = first - (unsigned)f_slot * g.per_slot;
        const unsigned per_m = (unsigned)(g.tiles_n * g.nspin);
        f_tm = (int)(lid / per_m);
        const unsigned rest = lid - (unsigned)f_tm * per_m;
        f_sp = (int)(rest / (unsigned)g.tiles_n);
        f_tn = (int)(rest - (unsigned)f_sp * (unsigned)g.tiles_n);
    }
    const double2 *fA, *fB;                                          // wave-uniform bases of the fetched tile's block
    unsigned voffA[AH], voffB;
    auto set_rows = [&]() {
#pragma unroll
        for (int h = 0; h < AH; ++h) {
            unsigned r = (unsigned)f_tm * BM + 64 * h + lane;
            if (r >= rows_total) r = rows_total - 1;                 // clamped lanes only ever feed masked outputs
            const unsigned L = r / mrows, q = r - L * mrows;
            voffA[h] = (L * nao * mrows + q) * 16u;                  // < 2^32 (half1_hot_usable)
        }
    };
    auto set_fetch = [&]() {
        fA = g.Lpq + (long long)f_slot * g.a_slot_stride;
        fB = g.Ci + (long long)f_sp * g.b_spin_stride + (long long)H1_PICK_BK(g, f_slot) * g.b_k_stride;
        int col = f_tn * BN + lane;
        if (col >= g.nemb) col = g.nemb - 1;
        voffB = (unsigned)(col * 16);
    };
    set_rows();
    set_fetch();
    int f_t = 0, is = 0;                                             // K tile to issue next, ring stage it goes to
    auto issue = [&]() {
        double2 *st = lds + is * STAGE;
        const int k0 = wave * 2;
        const long long kg = (long long)f_t * H1_BK + k0;
        const double2 *a0 = fA + kg * mrows, *a1 = a0 + mrows, *b0 = fB + kg * nemb, *b1 = b0 + nemb;
        if constexpr (AH == 2) {
            glds16s_x6(voffA[0], voffA[1], voffB, voffA[0], voffA[1], voffB, a0, a0, b0, a1, a1, b1, lds_addr_of(st + k0 * BM),
                       lds_addr_of(st + k0 * BM + 64), lds_addr_of(st + H1_BK * BM + k0 * H1_BN), lds_addr_of(st + (k0 + 1) * BM),
                       lds_addr_of(st + (k0 + 1) * BM + 64), lds_addr_of(st + H1_BK * BM + (k0 + 1) * H1_BN));
        } else {
            glds16s_x4(voffA[0], voffB, voffA[0], voffB, a0, b0, a1, b1, lds_addr_of(st + k0 * BM),
                       lds_addr_of(st + H1_BK * BM + k0 * H1_BN), lds_addr_of(st + (k0 + 1) * BM),
                       lds_addr_of(st + H1_BK * BM + (k0 + 1) * H1_BN));
        }
        ++f_t;
        is = is == H1_D - 1 ? 0 : is + 1;
    };

    cacc acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) cacc_zero(acc[i][j]);

    issue();
    issue();                                                         // T >= 2
    // Every workgroup of the launch starts at the same time and does identical work: left alone they ALL stay in phase, run their
    // epilogues at the same time (no MFMA on the whole chip for that long) and send their stores to HBM as one burst -- 512 x 128 KiB
    // that then takes ten times longer to be acknowledged than the ring has K tiles in flight.  Each workgroup therefore starts at
    // its own sixteenth of a tile period (first K tiles already in flight): epilogues and store traffic are spread evenly.
    if constexpr (!(LAB & 8)) {
        const unsigned phase = (blockIdx.x * 0x9E3779B1u) >> 28;                       // 0 .. 15
        const int period = MI * NJ * T * 2 * 3 * 64 * OCC;                             // cycles of one output tile with OCC workgroups per CU
        const int naps = (int)(((long long)period * phase / 16) / (127 * 64));
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    bool fetching = true;
    int cs = 0, post_epi = 0;
    bool skip = false, extra = false;
    for (unsigned tile = first; tile < last; ++tile) {
        // the tile being COMPUTED: the fetch state has not moved past it yet (it moves at this tile's K step T - 2)
        const int c_slot = f_slot, c_tm = f_tm, c_sp = f_sp, c_tn = f_tn;
        for (int t = 0; t < T; ++t) {
            // ---- K tile t of this output tile has landed; at most one younger K tile (and, for two steps, the previous epilogue's
            //      stores, which are younger than both) may still be in flight ----
            if (tile + 1 == last && t == T - 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (post_epi > 0) {
                if (extra) {                                         // two younger K tiles + the stores
                    if (AH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 + NST) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NST) : "memory");
                    extra = false;
                } else {
                    if (AH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + NST) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NST) : "memory");
                }
                --post_epi;
            } else {
                if (AH == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            if constexpr (!(LAB & 4)) __builtin_amdgcn_s_barrier();
            if (skip) {
                skip = false;                                        // this step's K tile went out ahead of the epilogue
            } else if (fetching) {
                if (f_t == T) {                                      // the ring moves on to the next output tile of this run
                    if (tile + 1 < last) {
                        f_t = 0;
                        const int tm_old = f_tm;
                        if (++f_tn == g.tiles_n) {
                            f_tn = 0;
                            if (++f_sp == g.nspin) {
                                f_sp = 0;
                                if (++f_tm == g.tiles_m) { f_tm = 0; ++f_slot; }
                            }
                        }
                        if (f_tm != tm_old) set_rows();
                        set_fetch();
                    } else {
                        fetching = false;
                    }
                }
                if constexpr (LAB & 2) { if (fetching && g.nslot < 0) issue(); }
                else { if (fetching) issue(); }
            }
            const double2 *Ab = lds + cs * STAGE + wm * (MI * 16) + frag_x;
            const double2 *Bb = lds + cs * STAGE + H1_BK * BM + wn * 32 + frag_x;
            cs = cs == H1_D - 1 ? 0 : cs + 1;
#pragma unroll
            for (int kk = 0; kk < H1_BK / 4; ++kk) {
                cfrag a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = cfrag_of(lds_frag(&Ab[(kk * 4 + frag_k) * BM + i * 16]));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    double2 v = lds_frag(&Bb[(kk * 4 + frag_k) * H1_BN + j * 16]);
                    if (CONJB) v.y = -v.y;
                    b[j] = cfrag_of(v);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) cmfma(acc[i][j], a[i], b[j]);
            }
        }

        // ---- a THIRD K tile of the next output tile goes out ahead of the stores (into the stage the last K step just released:
        //      one extra barrier per output tile), so that the first wait that needs the stores to have completed is the fourth K
        //      step of the next tile instead of the third ----
        post_epi = 2;
        if constexpr (!(LAB & 16)) {
            if (fetching) {
                if constexpr (!(LAB & 4)) __builtin_amdgcn_s_barrier();
                if constexpr (LAB & 2) { if (g.nslot < 0) issue(); }
                else issue();
                skip = true;
                if constexpr (!(LAB & 2)) { extra = true; post_epi = 3; }
            }
        }
        // ---- epilogue of the computed tile: exactly NST store instructions per wave, then fresh accumulators -----------------
        double2 *const Osp = g.Ut + (long long)c_sp * g.out_spin_stride + (long long)c_slot * g.out_slot_stride;
        const int n0 = c_tn * BN;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned rr = (unsigned)c_tm * BM + (wm * MI + i) * 16 + frag_k + 4 * r;
                double2 *row = Osp + (long long)rr * nemb;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col = n0 + wn * 32 + j * 16 + frag_x;
                    double2 *p = (rr < rows_total && col < g.nemb) ? row + col : &h1_sink[lane];
                    h1_d2_t v;
                    v.x = cacc_re(acc[i][j], r);
                    v.y = cacc_im(acc[i][j], r);
                    if constexpr (LAB & 1) { if (g.nslot < 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
                    else if constexpr (LAB & 32) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
                    else if constexpr (LAB & 64) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) cacc_zero(acc[i][j]);
    }
}


// LAB ONLY: half1_kernel (one workgroup per tile, the product form) with NONTEMPORAL Ut stores -- the only change.  A workgroup's slot is
// released when its stores are acknowledged; the question is whether streaming stores shorten that.
template <bool CONJB, int BM, int OCC, bool NARROW, int LAB = 0>
__global__ __launch_bounds__(HNT, OCC) void half1nt_kernel(const H1Args g) {
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

    const int T = g.nao / H1_BK;
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
                if (col < g.nemb) {
                    h1_d2_t v;
                    v.x = cacc_re(acc[i][j], r);
                    v.y = cacc_im(acc[i][j], r);
                    __builtin_nontemporal_store(v, reinterpret_cast<h1_d2_t *>(row + col));
                }
            }
        }
    }
}


}  // namespace
