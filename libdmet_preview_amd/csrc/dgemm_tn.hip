// K7 -- ERI contraction  C (M x N) += alpha * X^T Y  on the f64 matrix cores.
//
// Replaces the lib.dot(Lij.T, Lij, alpha, eri, 1) calls of
// basis_transform/eri_transform.py:455-476 (`_Lij_s4_to_eri`).  X and Y are the
// tril-packed (L|ab) planes, K x M and K x N row-major (K = naux or 2*naux with
// the Re and Im planes stacked), so both MFMA operands are "K-major with unit
// stride along the tile edge": global rows stream straight into an LDS image
// [k][m] with no transpose, and a fragment read is 16 consecutive doubles.
//
// Two kernels share the tiling; the LDS-DMA one at the bottom is the production path, the
// register-staged one handles arbitrary K / odd sizes.
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
#include <algorithm>
#include <cstdlib>
#include <vector>

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
// LDS-DMA variant (the production path for the ERI contraction).
//
// Measurements that shaped it (tools/mfma_f64_probe.hip, tools/gemm_lab*.hip, MI355X):
//   * a register-resident v_mfma_f64_16x16x4_f64 stream sustains 77.5 TFLOP/s (64.0 cycles / MFMA / SIMD);
//   * the register-staged kernel above reaches 58 TF: PMC shows the pipe 75 % busy and 20 % of the
//     wave cycles parked in s_waitcnt / s_barrier;
//   * a 512-thread workgroup sharing a 256 x 128 tile (half the L2 bytes per flop) is NOT faster
//     (60 TF): its two waves per SIMD run in barrier lock-step, so nobody covers the LDS latency
//     after each barrier.  Two INDEPENDENT 256-thread workgroups per CU desynchronise by themselves;
//   * read-modify-write of C in the epilogue costs 4-7 %: every C element has exactly one writer per
//     launch (tiles are disjoint, launches are stream ordered), so a fire-and-forget f64 atomic gives
//     the same deterministic sum without the HBM round trip;
//   * round 1 ran a 2-stage ring (BK = 16) with s_waitcnt vmcnt(0) in front of every K-tile -- exactly one
//     tile in flight, the full L2 / HBM latency exposed once per tile -- and walked the lower tile triangle
//     row by row: 64.8 TF executed, 62 GB of HBM traffic per launch against 18 GB algorithmic (PMC).
// Hence: 128 x 128 tile, 4 waves (wave tile 64 x 64, 16 accumulators in 128 VGPRs, no AGPR traffic),
// BK = 8, operands by LDS-DMA (global_load_lds_dwordx4: no staging registers) into a THREE-stage ring with two tiles in
// flight, retired by a counted s_waitcnt vmcnt(4) + ONE raw s_barrier per K-tile -- and THREE workgroups per CU: PMC on the
// two-workgroup version showed the matrix pipe 88 % busy with every wave parked 9 % of its time at the barrier / LDS
// round trip after it; a third wave per SIMD covers those gaps.  It fits because the kernel needs 162 VGPRs (<= 170) and
// because the LDS rows are padded to 136 instead of 144 doubles (3 x 17 KiB x 3 workgroups = 153 KiB <= 160): that
// stride costs a 2-way bank conflict on the fragment reads, which use < 15 % of the LDS bandwidth here.
// Out-of-range lanes re-read clamped valid columns (only masked outputs see them), so every wave issues
// exactly 4 loads per tile.
// Tile order: a host-built table (cached in the context per shape) walks 8 x 8 SUPER-BLOCKS of tiles; the 64
// workgroups resident on one XCD (32 CUs x 2; xcd_remap hands each XCD a contiguous range of logical ids) are
// then one super-block: they stream 8 + 8 operand panels through that XCD's L2 instead of 1 + 64.
// Requires K % 8 == 0, even M, N, ldx, ldy, 16-B aligned bases; otherwise the kernel above runs.
constexpr int GBK = 8, GD = 3;
constexpr int G_LD = BM + 8;                   // 136 doubles: 17 KiB per stage, THREE workgroups per CU (see above)
constexpr int G_STAGE = 2 * GBK * G_LD;        // doubles per stage: A[8][136] | B[8][136]

// SYMM (X == Y, M == N): C is symmetric, so only tiles with tm >= tn are computed; an off-diagonal tile is
// also added, transposed, to C[tn-tile][tm-tile] (still one writer per element).  Saves ~1/2 of the aa and bb
// contractions, i.e. 1/3 of the UHF contraction work.
// SADDR: LDS-DMA pieces addressed as scalar row pointer + loop-invariant per-lane byte offset (common.h glds16s_x4) instead of
// per-lane 64-bit pointers advanced with the vector ALU.
// Mp / Np: columns of X / Y that may be LOADED (even; M <= Mp <= ldx): an odd pair count is padded by one zero column in the
// plane layout, the 16-byte loads then stay aligned and inside the row, and the stores are masked with the real M and N.
template <bool SYMM, bool SADDR>
__global__ __launch_bounds__(NTHREADS, 3) void dgemm_tn_acc_dma_kernel(
    int M, int N, int K, double alpha, const double *__restrict__ X, int64_t ldx,
    const double *__restrict__ Y, int64_t ldy, double *__restrict__ C, int64_t ldc,
    const unsigned *__restrict__ tile_table, unsigned nblocks, int seg_tiles, int64_t jumpA, int64_t jumpB, int Mp, int Np) {
    __shared__ __attribute__((aligned(16))) double lds[GD * G_STAGE];

    const unsigned lid = xcd_remap(blockIdx.x, nblocks);
    const unsigned packed = tile_table[lid];
    const int tm = (int)(packed >> 16), tn = (int)(packed & 0xffffu);
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar LDS-DMA addressing
    const int wm = wave >> 1, wn = wave & 1;
    const int frag_k = lane >> 4, frag_x = lane & 15;

    int ca = m0 + 2 * lane, cb = n0 + 2 * lane;
    if (ca + 1 >= Mp) ca = Mp - 2;
    if (cb + 1 >= Np) cb = Np - 2;
    // wave w streams K rows 2w, 2w+1 of both operands; SCALAR running row pointers advance one K-tile per issue, the per-lane
    // part of an address is the loop-invariant column byte offset (common.h glds16s_x4: no vector ALU work per piece)
    const unsigned voffA = (unsigned)ca * 8u, voffB = (unsigned)cb * 8u;
    const double *pA = X + (SADDR ? 0 : ca) + (int64_t)(2 * wave) * ldx, *pB = Y + (SADDR ? 0 : cb) + (int64_t)(2 * wave) * ldy;
    const int64_t stepA = (int64_t)GBK * ldx, stepB = (int64_t)GBK * ldy;
    int is_stage = 0;
    // K may be a stack of row SEGMENTS (the planes of several momentum transfers kL, or only their Re halves): after every
    // `seg_tiles` K-tiles the running pointers hop over the gap to the next segment (wave-uniform counter: scalar ops)
    int seg_left = seg_tiles;
    auto issue = [&]() {
        double *st = lds + is_stage * G_STAGE + (2 * wave) * G_LD;
        if constexpr (SADDR)
            glds16s_x4(voffA, voffA, voffB, voffB, pA, pA + ldx, pB, pB + ldy, lds_addr_of(st), lds_addr_of(st + G_LD),
                       lds_addr_of(st + GBK * G_LD), lds_addr_of(st + GBK * G_LD + G_LD));
        else
            glds16_x4(pA, pA + ldx, pB, pB + ldy, lds_addr_of(st), lds_addr_of(st + G_LD), lds_addr_of(st + GBK * G_LD),
                      lds_addr_of(st + GBK * G_LD + G_LD));
        pA += stepA;
        pB += stepB;
        if (--seg_left == 0) {
            pA += jumpA;
            pB += jumpB;
            seg_left = seg_tiles;
        }
        is_stage = is_stage + 1 == GD ? 0 : is_stage + 1;
    };

    d4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};

    const int T = K / GBK;
    issue();
    if (T > 1) issue();
    int c_stage = 0;
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // tile t landed; t+1 may be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... for every wave; everyone is done with tile t-1
        if (t + 2 < T) issue();                               // overwrite the stage tile t-1 lived in
        const double *Ab = lds + c_stage * G_STAGE + wm * 64 + frag_x;
        const double *Bb = lds + c_stage * G_STAGE + GBK * G_LD + wn * 64 + frag_x;
        c_stage = c_stage + 1 == GD ? 0 : c_stage + 1;
#pragma unroll
        for (int kk = 0; kk < GBK / 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = Ab[(kk * 4 + frag_k) * G_LD + i * 16];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bb[(kk * 4 + frag_k) * G_LD + j * 16];
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
                if (col < N) unsafeAtomicAdd(&crow[col], alpha * acc[i][j][r]);
            }
        }
    }
    if (SYMM && tm != tn) {
        // mirrored tile C[n-tile][m-tile] += acc^T.  Straight from the accumulator layout a store instruction would touch
        // 16 rows of C with 4 scattered doubles each (measured: the symmetric launch ran 63 TF against 68 TF for the
        // rectangular one, 55 TF at K = 800); instead each wave transposes its 64 x 64 block through the now idle LDS ring
        // (one 16-row block row at a time, row stride 65 doubles: conflict-free both ways) and stores 4 rows x 16
        // contiguous doubles per instruction, the same pattern as the direct store.
        __syncthreads();                                        // every wave is done reading the last K-tile
        double *tr = lds + wave * (16 * 65);
        const int rr = lane & 15, cc = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                           // one 16-row block row of the wave tile at a time
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) tr[(frag_k + 4 * r) * 65 + j * 16 + frag_x] = alpha * acc[i][j][r];
            // wave-private region: the LDS pipe returns a wave's own writes in order, no barrier needed
            const int row = m0 + wm * 64 + i * 16 + rr;         // contiguous along the mirrored row
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int c = 4 * it + cc;
                const int col = n0 + wn * 64 + c;
                if (row < M && col < N) unsafeAtomicAdd(&C[(int64_t)col * ldc + row], tr[rr * 65 + c]);
            }
        }
    }
}

}  // namespace

// Tile visiting order of the LDS-DMA kernel: 8 x 8 super-blocks of tiles, super-block rows walked in a serpentine
// (the 8 B panels at a row end are reused by the next row), tiles inside a super-block column by column; SYMM keeps
// tm >= tn only.  Packed (tm << 16 | tn); built once per shape and parked in the context.
// [lo, hi): restriction to a band of tiles -- tile COLUMNS tn for the symmetric launch (its direct writes then cover the
// lower part of that column band and its mirrored writes the rows of the band right of the diagonal: after the bands
// 0 .. b have run, the ROWS of band b are complete), tile ROWS tm for the rectangular one.  (-1, -1) = everything.
static const unsigned *dgemm_tile_table(dmk_ctx *ctx, int tiles_m, int tiles_n, bool symm, unsigned *count_out, int lo = -1,
                                        int hi = -1) {
    for (auto &t : ctx->tile_tables)
        if (t.tiles_m == tiles_m && t.tiles_n == tiles_n && t.symm == (int)symm && t.lo == lo && t.hi == hi) {
            *count_out = t.count;
            return t.dev;
        }
    int SB = 8;
    if (const char *e = getenv("DMK_DGEMM_SUPER")) SB = atoi(e) > 0 ? atoi(e) : 8;     // ablation: 1 = plain row-major order
    std::vector<unsigned> h;
    const int sm = (tiles_m + SB - 1) / SB, sn = (tiles_n + SB - 1) / SB;
    for (int Tm = 0; Tm < sm; ++Tm) {
        const int ncol = symm ? Tm + 1 : sn;
        for (int c = 0; c < ncol; ++c) {
            const int Tn = (Tm & 1) ? ncol - 1 - c : c;
            for (int jn = 0; jn < SB; ++jn)
                for (int im = 0; im < SB; ++im) {
                    const int tm = Tm * SB + im, tn = Tn * SB + jn;
                    if (tm >= tiles_m || tn >= tiles_n || (symm && tn > tm)) continue;
                    if (lo >= 0 && ((symm ? tn : tm) < lo || (symm ? tn : tm) >= hi)) continue;
                    h.push_back(((unsigned)tm << 16) | (unsigned)tn);
                }
        }
    }
    unsigned *dev = nullptr;
    if (dmk_dev_alloc(ctx, reinterpret_cast<void **>(&dev), std::max<size_t>(h.size(), 1) * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (!h.empty() && hipMemcpy(dev, h.data(), h.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(dev);
        return nullptr;
    }
    ctx->tile_tables.push_back({tiles_m, tiles_n, (int)symm, lo, hi, (unsigned)h.size(), dev});
    *count_out = (unsigned)h.size();
    return dev;
}

int launch_dgemm_tn_acc(dmk_ctx *ctx, int M, int N, int K, double alpha, const double *X,
                        int64_t ldx, const double *Y, int64_t ldy, double *C, int64_t ldc) {
    return launch_dgemm_tn_acc_seg(ctx, M, N, K, alpha, X, ldx, Y, ldy, C, ldc, 0, 0, 0, -1, -1, 0, 0);
}

// The same product with (i) K given as K / seg_rows row segments that start seg_stride_x / seg_stride_y ELEMENTS apart
// (seg_rows = 0: one contiguous segment) and (ii) the output restricted to the tile band [band_lo, band_hi) (128-row /
// 128-column tiles; -1: everything) -- see dgemm_tile_table.  The ERI pipeline stacks the planes of many kL along K and
// finishes the contraction band by band so that finished rows can be reduced over ranks while later bands are computed.
int launch_dgemm_tn_acc_seg(dmk_ctx *ctx, int M, int N, int K, double alpha, const double *X, int64_t ldx, const double *Y,
                            int64_t ldy, double *C, int64_t ldc, int seg_rows, int64_t seg_stride_x, int64_t seg_stride_y,
                            int band_lo, int band_hi, int Mp, int Np) {
    if (M <= 0 || N <= 0 || K <= 0) return DMK_OK;
    if (Mp < M) Mp = M;                 // loadable columns of X / Y (>= M / N, zero beyond): see the LDS-DMA kernel
    if (Np < N) Np = N;
    if (seg_rows <= 0 || seg_rows >= K) { seg_rows = K; seg_stride_x = seg_stride_y = 0; }
    if (K % seg_rows) return dmk_fail(ctx, DMK_ERR_INVALID, "dgemm_tn: K = %d is not a multiple of the segment length %d", K, seg_rows);
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int64_t nblocks = (int64_t)tiles_m * tiles_n;
    if (nblocks > 0x7fffffffLL) return dmk_fail(ctx, DMK_ERR_INVALID, "dgemm_tn: grid too large");
    const bool vec2 = ((ldx & 1) == 0) && ((ldy & 1) == 0) &&
                      ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) && ((seg_stride_x & 1) == 0) && ((seg_stride_y & 1) == 0);
    static const bool dma_enabled = [] { const char *e = getenv("DMK_DGEMM_DMA"); return !(e && atoi(e) == 0); }();
    if (dma_enabled && vec2 && (seg_rows % GBK) == 0 && K >= GBK && (Mp % 2) == 0 && (Np % 2) == 0 && Mp <= ldx && Np <= ldy &&
        M >= 2 && N >= 2 && tiles_m < 65536 && tiles_n < 65536) {
        static const bool symm_enabled = [] { const char *e = getenv("DMK_DGEMM_SYMM"); return !(e && atoi(e) == 0); }();
        const bool symm = symm_enabled && X == Y && ldx == ldy && seg_stride_x == seg_stride_y && M == N && tiles_m >= 2;
        unsigned count = 0;
        const unsigned *table = dgemm_tile_table(ctx, tiles_m, tiles_n, symm, &count, band_lo, band_hi);
        if (!table) return dmk_fail(ctx, DMK_ERR_NOMEM, "dgemm_tn: tile table allocation failed");
        if (count == 0) return DMK_OK;
        const int seg_tiles = seg_rows / GBK;
        const int64_t jumpA = seg_rows == K ? 0 : seg_stride_x - (int64_t)seg_rows * ldx;
        const int64_t jumpB = seg_rows == K ? 0 : seg_stride_y - (int64_t)seg_rows * ldy;
        FamScope fs(ctx, DMK_FAM_DGEMM);
        fs.mfma_flops(2.0 * (double)count * BM * BN * (double)K);
        // Measured (tools/contract_bench.py, rocprofv3 PMC; round 4): the scalar-base form is 1-3 % faster at the C5 size (N = 32896,
        // K = 1600: symmetric 24.82 -> 24.51 ms, rectangular 48.57 -> 47.16 ms) but DOUBLES the launch's L2-miss traffic (FETCH_SIZE
        // 22.1 -> 48.7 GB symmetric, 42.7 -> 82.0 GB rectangular; tools/fetch_probe.hip shows the counter reads both forms alike, so
        // the re-reads are real: the workgroups of a super-block drift apart and lose their shared panels in L2), and it is slower
        // on small pair spaces (N = 9316: symmetric K = 832 1.42 -> 1.51 ms).  The contraction therefore keeps per-lane pointers;
        // DMK_DGEMM_SADDR=1 selects the scalar-base form (read per launch: labs and tests toggle it).
        const char *se = getenv("DMK_DGEMM_SADDR");
        const bool saddr = se && atoi(se) != 0;
#define DGEMM_LAUNCH(SY, SA)                                                                                                  \
        hipLaunchKernelGGL((dgemm_tn_acc_dma_kernel<SY, SA>), dim3(count), dim3(NTHREADS), 0, ctx->stream, M, N, K, alpha, X, ldx, Y,   \
                           ldy, C, ldc, table, count, seg_tiles, jumpA, jumpB, Mp, Np)
        if (symm) { if (saddr) DGEMM_LAUNCH(true, true); else DGEMM_LAUNCH(true, false); }
        else { if (saddr) DGEMM_LAUNCH(false, true); else DGEMM_LAUNCH(false, false); }
#undef DGEMM_LAUNCH
        DMK_CHECK_LAUNCH(ctx);
        return DMK_OK;
    }
    // register-staged kernel (odd sizes): one launch per segment; a band is a row range of C (no mirroring here, so every
    // band writes its own rows over the full width)
    int m_lo = 0, m_hi = M;
    if (band_lo >= 0) {
        m_lo = std::min(M, band_lo * BM);
        m_hi = std::min(M, band_hi * BM);
        if (m_hi <= m_lo) return DMK_OK;
    }
    const int Mb = m_hi - m_lo;
    const int tiles_mb = (Mb + BM - 1) / BM;
    const int64_t nb2 = (int64_t)tiles_mb * tiles_n;
    const bool vec2b = vec2 && (m_lo % 2) == 0;
    for (int k0 = 0, sidx = 0; k0 < K; k0 += seg_rows, ++sidx) {
        const double *Xs = X + (seg_rows == K ? 0 : (int64_t)sidx * seg_stride_x) + m_lo;
        const double *Ys = Y + (seg_rows == K ? 0 : (int64_t)sidx * seg_stride_y);
        double *Cs = C + (int64_t)m_lo * ldc;
        FamScope fs(ctx, DMK_FAM_DGEMM);
        fs.mfma_flops(2.0 * (double)nb2 * BM * BN * (double)(((seg_rows + BK - 1) / BK) * BK));
        if (vec2b)
            hipLaunchKernelGGL(dgemm_tn_acc_kernel<true>, dim3((unsigned)nb2), dim3(NTHREADS), 0,
                               ctx->stream, Mb, N, seg_rows, alpha, Xs, ldx, Ys, ldy, Cs, ldc, tiles_mb, tiles_n);
        else
            hipLaunchKernelGGL(dgemm_tn_acc_kernel<false>, dim3((unsigned)nb2), dim3(NTHREADS), 0,
                               ctx->stream, Mb, N, seg_rows, alpha, Xs, ldx, Ys, ldy, Cs, ldc, tiles_mb, tiles_n);
        DMK_CHECK_LAUNCH(ctx);
    }
    return DMK_OK;
}
