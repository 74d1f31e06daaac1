// Complex (c128) MFMA GEMM family used by every dense complex product on the path:
//
//   K6  DF half transform  (basis_transform/eri_transform.py:403-434 -> pyscf _ao2mo.r_e2,
//       + lib.hermi_sum :372, lib.pack_tril :375, Lij_s4 accumulation :376-378)
//   K3  k <-> R folds as DFT-by-GEMM   (system/fourier.py:160-177)
//   K2  rho_k = (ev occ) ev^H          (routine/mfd.py:355-357)
//   K5  C_ao_emb = C_ao_lo basis_k     (basis_transform/make_basis.py:923-962, utils/misc.py:49-59)
//   a10 C^H h C style triple products  (basis_transform/make_basis.py:524-644)
//
//   C[m][n] = alpha * sum_seg sum_k opA(A_seg)[k][m] * opB(B_seg)[k][n]
//
// A complex product is evaluated as four real v_mfma_f64_16x16x4_f64 ("4M"; the
// neg:[1,0,0] modifier supplies the minus sign of Ai*Bi), conjugation is applied
// when a slab is staged into LDS.  Operands are staged as [k][m] images of
// interleaved (re, im) pairs so that a fragment read is one ds_read_b128 per
// lane and rows of 16 complex = 256 B keep every lane group conflict-free.
// Both operands may be "K-major" (unit stride along the tile edge: coalesced 16 B
// per lane, the layout every hot call uses) or "M-major" (generic fallback for
// the small batched products).
//
// Two accumulation segments let the time-reversal partner term of the half
// transform,  S[a][b] = sum_q U[q][a] C[q][b] + sum_q C[q][a] U[q][b],  land in
// the same accumulators; the PACK epilogue then adds the a >= b triangle into the
// Re / Im planes of Lij_s4 in exactly the layout the contraction kernel reads.
//
// Workgroup = 256 threads = 2 x 2 waves; wave tile = TM x TN blocks of 16 x 16.
#include "common.h"

namespace {

constexpr int BK = 8;
constexpr int NTHREADS = 256;

struct ZArgs {
    int M, N, K, batch, nseg;
    const void *A[2];
    const void *B[2];
    long long lda[2], ldb[2], strideA[2], strideB[2];
    int a_kmajor[2], b_kmajor[2], conjA[2], conjB[2], b_real[2];
    const double *kscaleB[2];
    double alpha;
    int flatten_m, lower_only;
    void *C;
    long long ldc, strideC;
    double *imag_max;
    double *planes;
    long long naux, npair;
    int tiles_m, tiles_n, per_batch, nblk;   // nblk = ceil(M/16) (flatten mode)
    unsigned nblocks;
};

// M3 = true: Karatsuba "3M" complex product -- T1 = Ar Br, T2 = Ai Bi, T3 = (Ar+Ai)(Br+Bi),
// Re = T1 - T2, Im = T3 - T1 - T2: three real MFMAs per complex tile step instead of four
// (the two operand sums are one v_add_f64 per fragment).  Normwise backward stable; the
// parity tests hold it to the same 1e-8 / 1e-10 budgets as the 4M form.
template <int TM, int TN, int EPI, bool M3>
__global__ __launch_bounds__(NTHREADS, 2) void zgemm_kernel(const ZArgs g) {
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16;
    constexpr int PA = BK * BM / NTHREADS, PB = BK * BN / NTHREADS;
    __shared__ __attribute__((aligned(16))) double2 lds[2 * BK * (BM + BN)];
    double2 *As = lds;                  // [2][BK][BM]
    double2 *Bs = lds + 2 * BK * BM;    // [2][BK][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frag_k = lane >> 4, frag_x = lane & 15;

    // ---- tile selection ------------------------------------------------------------
    const unsigned lid = xcd_remap(blockIdx.x, g.nblocks);
    int tile_m, tile_n, tile_batch = 0;
    if (g.flatten_m) {
        tile_m = (int)(lid / (unsigned)g.tiles_n);
        tile_n = (int)(lid % (unsigned)g.tiles_n);
    } else {
        tile_batch = (int)(lid / (unsigned)g.per_batch);
        int t = (int)(lid % (unsigned)g.per_batch);
        if (g.lower_only) {
            tile_m = 0;
            tile_n = 0;
            for (int tm = 0; tm < g.tiles_m; ++tm) {
                int cnt = ((tm + 1) * BM - 1) / BN + 1;
                cnt = cnt < g.tiles_n ? cnt : g.tiles_n;
                if (t < cnt) { tile_m = tm; tile_n = t; break; }
                t -= cnt;
            }
        } else {
            tile_m = t / g.tiles_n;
            tile_n = t % g.tiles_n;
        }
    }
    const int n0 = tile_n * BN;

    // map a tile-local row to (batch, global row)
    auto decode_row = [&](int m_local, int &b, int &row) {
        if (g.flatten_m) {
            const int gb = tile_m * (BM / 16) + (m_local >> 4);
            b = gb / g.nblk;
            row = (gb - b * g.nblk) * 16 + (m_local & 15);
        } else {
            b = tile_batch;
            row = tile_m * BM + m_local;
        }
    };

    // acc_re/acc_im hold (Re, Im) in 4M mode and (T1, T2) in 3M mode; acc_t3 only exists in 3M mode
    d4_t acc_re[TM][TN], acc_im[TM][TN], acc_t3[M3 ? TM : 1][M3 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc_re[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
            acc_im[i][j] = d4_t{0.0, 0.0, 0.0, 0.0};
            if (M3) acc_t3[M3 ? i : 0][M3 ? j : 0] = d4_t{0.0, 0.0, 0.0, 0.0};
        }
    // 16x16 blocks that are never needed cost no MFMA issue slots (wave-uniform mask): blocks entirely beyond the
    // matrix edge (a 136-wide problem on 64-wide tiles leaves an 8-row strip in the last tile row: 3 of its 4
    // block rows are empty) and, in lower-triangle mode, blocks entirely above the diagonal
    unsigned skip_mask = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col_min = n0 + wn * TN * 16 + j * 16;
            bool skip = col_min >= g.N;
            if (g.flatten_m) {
                const long long gb = (long long)tile_m * (BM / 16) + wm * TM + i;
                skip |= gb >= (long long)g.batch * g.nblk;
            } else {
                const int row_min = tile_m * BM + wm * TM * 16 + i * 16;
                skip |= row_min >= g.M;
                if (g.lower_only) skip |= (row_min + 15 < col_min);
            }
            if (skip) skip_mask |= 1u << (i * TN + j);
        }

    for (int s = 0; s < g.nseg; ++s) {
        const bool akm = g.a_kmajor[s] != 0, bkm = g.b_kmajor[s] != 0;
        const bool cja = g.conjA[s] != 0, cjb = g.conjB[s] != 0, breal = g.b_real[s] != 0;
        const long long lda = g.lda[s], ldb = g.ldb[s];
        const double *ksc = g.kscaleB[s];

        // per-thread element descriptors (fixed over the K loop)
        const double2 *pa[PA];
        int ka[PA], la[PA];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int e = tid + i * NTHREADS;
            int k, m;
            if (akm) { k = e / BM; m = e % BM; } else { m = e / BK; k = e % BK; }
            int b, row;
            decode_row(m, b, row);
            ka[i] = k;
            la[i] = k * BM + m;
            if (row < g.M && b < g.batch) {
                const double2 *base = reinterpret_cast<const double2 *>(g.A[s]) + (long long)b * g.strideA[s];
                pa[i] = akm ? base + (long long)k * lda + row : base + (long long)row * lda + k;
            } else {
                pa[i] = nullptr;
            }
        }
        const char *pb[PB];
        int kb[PB], lb[PB];
        const int bb = g.flatten_m ? 0 : tile_batch;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int e = tid + i * NTHREADS;
            int k, n;
            if (bkm) { k = e / BN; n = e % BN; } else { n = e / BK; k = e % BK; }
            kb[i] = k;
            lb[i] = k * BN + n;
            const int col = n0 + n;
            if (col < g.N) {
                const long long off = (long long)bb * g.strideB[s] + (bkm ? (long long)k * ldb + col : (long long)col * ldb + k);
                pb[i] = reinterpret_cast<const char *>(g.B[s]) + off * (breal ? 8 : 16);
            } else {
                pb[i] = nullptr;
            }
        }
        const long long stepA = akm ? lda * BK : BK;              // elements per K tile
        const long long stepB = (bkm ? ldb * BK : BK) * (breal ? 8 : 16);   // bytes per K tile

        double2 ra[PA], rb[PB];
        auto gload = [&](int kt) {
            const int k0 = kt * BK;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                double2 v = make_double2(0.0, 0.0);
                if (pa[i] != nullptr && k0 + ka[i] < g.K) v = pa[i][(long long)kt * stepA];
                if (cja) v.y = -v.y;
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                double2 v = make_double2(0.0, 0.0);
                const int k = k0 + kb[i];
                if (pb[i] != nullptr && k < g.K) {
                    const char *p = pb[i] + (long long)kt * stepB;
                    if (breal) v.x = *reinterpret_cast<const double *>(p);
                    else v = *reinterpret_cast<const double2 *>(p);
                    if (cjb) v.y = -v.y;
                    if (ksc != nullptr) {
                        const double sc = ksc[(long long)bb * g.K + k];
                        v.x *= sc;
                        v.y *= sc;
                    }
                }
                rb[i] = v;
            }
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int i = 0; i < PA; ++i) As[buf * BK * BM + la[i]] = ra[i];
#pragma unroll
            for (int i = 0; i < PB; ++i) Bs[buf * BK * BN + lb[i]] = rb[i];
        };

        const int nkt = (g.K + BK - 1) / BK;
        __syncthreads();            // previous segment's readers are done with the buffers
        gload(0);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nkt) gload(kt + 1);
            const double2 *Ab = As + buf * BK * BM + wm * TM * 16 + frag_x;
            const double2 *Bb = Bs + buf * BK * BN + wn * TN * 16 + frag_x;
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double2 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = Ab[(kk * 4 + frag_k) * BM + i * 16];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bb[(kk * 4 + frag_k) * BN + j * 16];
                if (M3) {
                    double as[TM], bs[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) as[i] = a[i].x + a[i].y;
#pragma unroll
                    for (int j = 0; j < TN; ++j) bs[j] = b[j].x + b[j].y;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (skip_mask & (1u << (i * TN + j))) continue;
                            acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc_re[i][j], 0, 0, 0);
                            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc_im[i][j], 0, 0, 0);
                            acc_t3[M3 ? i : 0][M3 ? j : 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                as[i], bs[j], acc_t3[M3 ? i : 0][M3 ? j : 0], 0, 0, 0);
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (skip_mask & (1u << (i * TN + j))) continue;
                            acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc_re[i][j], 0, 0, 0);
                            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].y, acc_im[i][j], 0, 0, 0);
                            acc_re[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc_re[i][j], 0, 0, 1);
                            acc_im[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].x, acc_im[i][j], 0, 0, 0);
                        }
                }
            }
            if (kt + 1 < nkt) lstore(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue ------------------------------------------------------------------
    double local_imag_max = 0.0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m_local = wm * TM * 16 + i * 16 + frag_k + 4 * r;
            int b, row;
            decode_row(m_local, b, row);
            if (row >= g.M || b >= g.batch) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * TN * 16 + j * 16 + frag_x;
                if (col >= g.N) continue;
                double vr, vi;
                if (M3) {
                    const double t1 = acc_re[i][j][r], t2 = acc_im[i][j][r], t3 = acc_t3[M3 ? i : 0][M3 ? j : 0][r];
                    vr = g.alpha * (t1 - t2);
                    vi = g.alpha * ((t3 - t1) - t2);
                } else {
                    vr = g.alpha * acc_re[i][j][r];
                    vi = g.alpha * acc_im[i][j][r];
                }
                if (EPI == ZEPI_STORE) {
                    double2 *C = reinterpret_cast<double2 *>(g.C) + (long long)b * g.strideC + (long long)row * g.ldc + col;
                    *C = make_double2(vr, vi);
                } else if (EPI == ZEPI_STORE_REAL) {
                    double *C = reinterpret_cast<double *>(g.C) + (long long)b * g.strideC + (long long)row * g.ldc + col;
                    *C = vr;
                    local_imag_max = fmax(local_imag_max, fabs(vi));
                } else {
                    if (row >= col) {
                        const long long idx = (long long)row * (row + 1) / 2 + col;
                        double *pr = g.planes + (long long)b * g.npair + idx;
                        double *pi = g.planes + (g.naux + (long long)b) * g.npair + idx;
                        *pr += vr;
                        *pi += vi;
                    }
                }
            }
        }
    }
    if (EPI == ZEPI_STORE_REAL && g.imag_max != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            local_imag_max = fmax(local_imag_max, __shfl_xor(local_imag_max, off, 64));
        if (lane == 0 && local_imag_max > 0.0)
            atomicMax(reinterpret_cast<unsigned long long *>(g.imag_max),
                      (unsigned long long)__double_as_longlong(local_imag_max));
    }
}

template <int TM, int TN, bool M3>
int launch_cfg(dmk_ctx *ctx, ZArgs &a, const ZGemm &g, int fam) {
    constexpr int BM = 2 * TM * 16, BN = 2 * TN * 16;
    a.tiles_n = (g.N + BN - 1) / BN;
    a.nblk = (g.M + 15) / 16;
    long long nblocks;
    if (g.flatten_m) {
        const long long total_blk = (long long)g.batch * a.nblk;
        a.tiles_m = (int)((total_blk + BM / 16 - 1) / (BM / 16));
        a.per_batch = 0;
        nblocks = (long long)a.tiles_m * a.tiles_n;
    } else {
        a.tiles_m = (g.M + BM - 1) / BM;
        if (g.lower_only) {
            int per = 0;
            for (int tm = 0; tm < a.tiles_m; ++tm) {
                int cnt = ((tm + 1) * BM - 1) / BN + 1;
                per += cnt < a.tiles_n ? cnt : a.tiles_n;
            }
            a.per_batch = per;
        } else {
            a.per_batch = a.tiles_m * a.tiles_n;
        }
        nblocks = (long long)a.per_batch * g.batch;
    }
    if (nblocks <= 0) return DMK_OK;
    if (nblocks > 0x7fffffffLL) return dmk_fail(ctx, DMK_ERR_INVALID, "zgemm: grid too large");
    a.nblocks = (unsigned)nblocks;
    FamScope fs(ctx, fam);
    // upper bound: 16 x 16 blocks beyond the matrix edge are skipped inside a tile
    fs.mfma_flops((M3 ? 6.0 : 8.0) * (double)nblocks * BM * BN * (double)(((g.K + 3) / 4) * 4) * (double)g.nseg);
    switch (g.epi) {
        case ZEPI_STORE:
            hipLaunchKernelGGL((zgemm_kernel<TM, TN, ZEPI_STORE, M3>), dim3(a.nblocks), dim3(NTHREADS), 0, ctx->stream, a);
            break;
        case ZEPI_STORE_REAL:
            hipLaunchKernelGGL((zgemm_kernel<TM, TN, ZEPI_STORE_REAL, M3>), dim3(a.nblocks), dim3(NTHREADS), 0, ctx->stream, a);
            break;
        default:
            hipLaunchKernelGGL((zgemm_kernel<TM, TN, ZEPI_PACK_ACC, M3>), dim3(a.nblocks), dim3(NTHREADS), 0, ctx->stream, a);
            break;
    }
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

}  // namespace

int launch_zgemm(dmk_ctx *ctx, const ZGemm &g, int fam) {
    if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return DMK_OK;
    if (g.nseg < 1 || g.nseg > 2) return dmk_fail(ctx, DMK_ERR_INVALID, "zgemm: nseg must be 1 or 2");
    ZArgs a;
    memset(&a, 0, sizeof(a));
    a.M = g.M; a.N = g.N; a.K = g.K; a.batch = g.batch; a.nseg = g.nseg;
    for (int s = 0; s < g.nseg; ++s) {
        const ZSeg &z = g.seg[s];
        if (z.A == nullptr || z.B == nullptr) return dmk_fail(ctx, DMK_ERR_INVALID, "zgemm: null operand");
        if (g.flatten_m && z.strideB != 0)
            return dmk_fail(ctx, DMK_ERR_INVALID, "zgemm: flatten_m needs a batch-invariant B");
        if ((reinterpret_cast<uintptr_t>(z.A) & 15) || (!z.b_real && (reinterpret_cast<uintptr_t>(z.B) & 15)))
            return dmk_fail(ctx, DMK_ERR_INVALID, "zgemm: operands must be 16-byte aligned");
        a.A[s] = z.A; a.B[s] = z.B;
        a.lda[s] = z.lda; a.ldb[s] = z.ldb; a.strideA[s] = z.strideA; a.strideB[s] = z.strideB;
        a.a_kmajor[s] = z.a_kmajor; a.b_kmajor[s] = z.b_kmajor;
        a.conjA[s] = z.conjA; a.conjB[s] = z.conjB; a.b_real[s] = z.b_real;
        a.kscaleB[s] = z.kscaleB;
    }
    a.alpha = g.alpha;
    a.flatten_m = g.flatten_m; a.lower_only = g.lower_only;
    a.C = g.C; a.ldc = g.ldc; a.strideC = g.strideC;
    a.imag_max = g.imag_max;
    a.planes = g.planes; a.naux = g.naux; a.npair = g.npair;
    if (g.epi == ZEPI_PACK_ACC) {
        if (g.planes == nullptr) return dmk_fail(ctx, DMK_ERR_INVALID, "zgemm: PACK epilogue without planes");
    } else if (g.C == nullptr) {
        return dmk_fail(ctx, DMK_ERR_INVALID, "zgemm: null C");
    }
    if (g.use_3m) return launch_cfg<2, 2, true>(ctx, a, g, fam);   // (4,2) would spill: 3 accumulator sets
    if (g.big_tile) return launch_cfg<4, 2, false>(ctx, a, g, fam);
    return launch_cfg<2, 2, false>(ctx, a, g, fam);
}
