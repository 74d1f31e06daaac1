// K1a -- Householder tridiagonalisation of a batch of complex Hermitian matrices with the matrix RESIDENT ON THE CU.
//
// Phase 1 of eigh.hip keeps the trailing block in HBM / L2 and streams it twice per Householder step (product, rank-2 update):
// 199 steps x two passes x a handful of dependent round trips = 29 us per step, 5.8 ms of the 10.6 ms a 200 x 200 matrix takes,
// and the PMC counters show 21x the algorithmic bytes.  A 200 x 200 Hermitian matrix is 20 100 complex numbers of lower triangle
// = 321 KB: more than the LDS (160 KB), less than LDS + register file.  So, for n <= 200 (the north-star nao), one 512-thread
// workgroup per matrix holds
//     rows   0 .. 111  packed in LDS (6 328 c128 = 101 KB; these rows retire first),
//     rows 112 .. 199  in registers: row i belongs to wave i % 8, slot (i - 112) / 8, lane l of chunk c holds column 64 c + l
//                      (2 + 2 + 8 x 3 + 4 = 32 chunks = 128 VGPRs per lane, every index a compile-time constant),
// reads the matrix from HBM exactly once and writes only the reflectors, d, e and tau.  Per step (same zhetd2 recurrence and
// reflector convention as eigh.hip, so phases 2 and 3 of eigh_kernel run unchanged on the result):
//     x = column k below the diagonal is already in LDS (captured by the lanes that own it during the previous update);
//     every wave forms |x|, tau, v for itself (same lanes, same order: bit-identical) -- no barrier;
//     p = tau A22 v from the LOWER triangle only: row sums a_ij v_j by one DPP reduction per owned row, column sums
//     conj(a_ij) v_i lane-wise (lane <-> column) and across the eight waves through LDS;
//     w = p - (tau / 2)(p^H v) v;  A22 -= v w^H + w v^H in place (registers / LDS), capturing column k + 1 and d[k + 1].
// Three barriers per step, no global memory inside the loop except the reflector row.
#include "common.h"

namespace {

constexpr int TD_NT = 512, TD_NW = 8;
constexpr int TD_RB = 112;                   // first register-resident row (multiple of 8)
constexpr int TD_NS = 11;                    // register slots per wave: rows TD_RB + 8 s + wave
constexpr int TD_NMAX = TD_RB + 8 * TD_NS;   // 200
__host__ __device__ constexpr int td_nc(int s) { return ((TD_RB + 8 * s + 7) >> 6) + 1; }     // 64-column chunks of slot s
__host__ __device__ constexpr int td_off(int s) {
    int o = 0;
    for (int t = 0; t < s; ++t) o += td_nc(t);
    return o;
}
constexpr int TD_NCH = td_off(TD_NS);        // 32 chunks per lane
constexpr int TD_TRI = TD_RB * (TD_RB + 1) / 2;
constexpr size_t TD_LDS = (size_t)(TD_TRI + 2 * TD_NMAX + TD_NMAX + 4 * TD_NMAX + TD_NW * TD_NMAX) * sizeof(double2);

__device__ __forceinline__ double2 td_cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 td_cmulc(double2 a, double2 b) {   // conj(a) * b
    return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
#if defined(__HIP_DEVICE_COMPILE__)
// wave sum that leaves the total in lane 63 (no broadcast: the owner lane stores it)
__device__ __forceinline__ double td_sum63(double v) {
    v += dmk_dpp_mov<0xb1, 0xf>(v);
    v += dmk_dpp_mov<0x4e, 0xf>(v);
    v += dmk_dpp_mov<0x124, 0xf>(v);
    v += dmk_dpp_mov<0x128, 0xf>(v);
    v += dmk_dpp_mov<0x142, 0xa>(v);
    v += dmk_dpp_mov<0x143, 0xc>(v);
    return v;
}
template <int CTRL> __device__ __forceinline__ double td_dpp(double v) { return dmk_dpp_mov<CTRL, 0xf>(v); }
#else
__device__ inline double td_sum63(double v) { return v; }      // host pass: never called
template <int CTRL> __device__ inline double td_dpp(double v) { return v; }
#endif

struct TdArgs {
    int n, batch;
    const double2 *A;       // batch x n x n, lower triangle referenced
    const double *add;      // optional real n x n added to every matrix of a group
    int add_group;
    double2 *Vh, *tau;      // reflectors (row k: v at columns k + 1 ..), batch x n
    double *d, *e;          // batch x n
};

__global__ __launch_bounds__(TD_NT, 1) void tridiag_resident_kernel(const TdArgs g) {
    extern __shared__ __attribute__((aligned(16))) char td_smem[];
    const int n = g.n, b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double2 *Lm = reinterpret_cast<double2 *>(td_smem);    // packed lower triangle of rows < TD_RB
    double2 *xbuf = Lm + TD_TRI;                           // [2][TD_NMAX] pivot column, double buffered
    double2 *pbuf = xbuf + 2 * TD_NMAX;                    // [TD_NMAX] p = tau A v
    double *rowp4 = reinterpret_cast<double *>(pbuf + TD_NMAX);   // [2 (re, im)][4 (16-lane rows)][TD_NMAX] row parts of A v
    double2 *colp = reinterpret_cast<double2 *>(rowp4 + 8 * TD_NMAX);   // [8][TD_NMAX] column parts, one per wave
    const size_t nn = (size_t)n * n;
    const double2 *A = g.A + b * nn;
    const double *addm = g.add ? g.add + (size_t)(g.add_group > 0 ? b / g.add_group : 0) * nn : nullptr;
    double2 *Vh = g.Vh + b * nn;
    double2 *tau = g.tau + (size_t)b * n;
    double *d = g.d + (size_t)b * n, *e = g.e + (size_t)b * n;

    auto load_a = [&](int i, int j) -> double2 {           // element (i, j) of the lower triangle, i < n, j <= i
        double2 v = A[(size_t)i * n + j];
        if (i == j) v.y = 0.0;
        if (addm) v.x += addm[(size_t)i * n + j];
        return v;
    };
    const int nl = n < TD_RB ? n : TD_RB;
    for (int i = w; i < nl; i += TD_NW)
        for (int j = lane; j <= i; j += 64) Lm[i * (i + 1) / 2 + j] = load_a(i, j);
    double2 ar[TD_NCH];
#pragma unroll
    for (int s = 0; s < TD_NS; ++s) {
#pragma unroll
        for (int c = 0; c < td_nc(s); ++c) {
            const int i = TD_RB + 8 * s + w, j = 64 * c + lane;
            ar[td_off(s) + c] = (i < n && j <= i) ? load_a(i, j) : make_double2(0.0, 0.0);
        }
    }
    for (int i = 1 + tid; i < n; i += TD_NT) xbuf[i] = load_a(i, 0);
    if (tid == 0) d[0] = load_a(0, 0).x;
    __syncthreads();

    for (int k = 0; k + 1 < n; ++k) {
        const double2 *x = xbuf + (k & 1) * TD_NMAX;
        double2 *xn = xbuf + ((k + 1) & 1) * TD_NMAX;
        // ---- reflector parameters, redundantly per wave ----
        double part = 0.0;
        for (int j = k + 2 + lane; j < n; j += 64) {
            const double2 t = x[j];
            part += t.x * t.x + t.y * t.y;
        }
        const double xnorm2 = dmk_wave_sum(part);
        const double2 alpha = x[k + 1];
        double2 tk = make_double2(0.0, 0.0), scale = make_double2(0.0, 0.0);
        double beta = alpha.x;
        if (!(xnorm2 == 0.0 && alpha.y == 0.0)) {
            const double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xnorm2);
            beta = alpha.x >= 0.0 ? -nrm : nrm;
            tk = make_double2((beta - alpha.x) / beta, -alpha.y / beta);
            const double dr = alpha.x - beta, di = alpha.y;
            const double den = dr * dr + di * di;
            scale = make_double2(dr / den, -di / den);     // 1 / (alpha - beta)
        }
        const bool active = !(tk.x == 0.0 && tk.y == 0.0);
        double2 vcol[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = 64 * c + lane;
            double2 v = make_double2(0.0, 0.0);
            if (j == k + 1) v = make_double2(1.0, 0.0);
            else if (j > k + 1 && j < n) v = td_cmul(x[j], scale);
            vcol[c] = v;
        }
        if (w == (k & 7)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 64 * c + lane;
                if (j < n) Vh[(size_t)k * n + j] = vcol[c];          // zeros up to the diagonal: the row is complete
            }
            if (lane == 0) {
                tau[k] = tk;
                e[k] = beta;
            }
        }
        auto v_of = [&](int i) -> double2 { return i == k + 1 ? make_double2(1.0, 0.0) : td_cmul(x[i], scale); };
        int i0 = k + 1;
        i0 += (w - i0) & 7;                                 // first row >= k + 1 owned by this wave
        double2 a2 = make_double2(0.0, 0.0);
        double2 wcol[4];
        if (active) {
            // ---- p = tau A22 v ----
            double2 cc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) cc[c] = make_double2(0.0, 0.0);
            // row sums of TWO rows are folded together: after the two quad stages lane l holds, for l % 4 = 0 .. 3, the partial
            // (row 0 re, row 0 im, row 1 re, row 1 im) of its quad; two more stages sum the quads of a 16-lane row and lanes
            // 12 .. 15 of each row store four partials per value (the consumer adds them): 27 instead of 72 instructions
            auto fold2 = [&](double2 r0, double2 r1, int ia, int ib) {
                const bool odd = lane & 1, hi = lane & 2;
                const double k0 = odd ? r0.y : r0.x, s0 = odd ? r0.x : r0.y;
                const double k1 = odd ? r1.y : r1.x, s1 = odd ? r1.x : r1.y;
                const double v0 = k0 + td_dpp<0xb1>(s0);          // quad_perm [1,0,3,2]
                const double v1 = k1 + td_dpp<0xb1>(s1);
                const double kk = hi ? v1 : v0, ss = hi ? v0 : v1;
                double u = kk + td_dpp<0x4e>(ss);                 // quad_perm [2,3,0,1]
                u += td_dpp<0x124>(u);                            // row_ror 4
                u += td_dpp<0x128>(u);                            // row_ror 8: every lane holds the total of its l % 4 class
                if ((lane & 12) == 12) {
                    const int row = hi ? ib : ia;
                    if (row >= 0) rowp4[((lane & 1) * 4 + (lane >> 4)) * TD_NMAX + row] = u;
                }
            };
            auto row_lds = [&](int i, double2 &rs) {                        // one LDS row (at most two chunks: i < 112)
                rs = make_double2(0.0, 0.0);
                if (i >= nl) return;
                const double2 vi = v_of(i);
                const double2 *row = Lm + i * (i + 1) / 2;
                if (lane <= i) {
                    const double2 a = row[lane];
                    rs.x = fma(a.x, vcol[0].x, rs.x); rs.x = fma(-a.y, vcol[0].y, rs.x);
                    rs.y = fma(a.x, vcol[0].y, rs.y); rs.y = fma(a.y, vcol[0].x, rs.y);
                    if (lane < i) {                                         // conj(a) * v_i
                        cc[0].x = fma(a.x, vi.x, cc[0].x); cc[0].x = fma(a.y, vi.y, cc[0].x);
                        cc[0].y = fma(a.x, vi.y, cc[0].y); cc[0].y = fma(-a.y, vi.x, cc[0].y);
                    }
                }
                if (i >= 64 && 64 + lane <= i) {
                    const double2 a = row[64 + lane];
                    rs.x = fma(a.x, vcol[1].x, rs.x); rs.x = fma(-a.y, vcol[1].y, rs.x);
                    rs.y = fma(a.x, vcol[1].y, rs.y); rs.y = fma(a.y, vcol[1].x, rs.y);
                    if (64 + lane < i) {
                        cc[1].x = fma(a.x, vi.x, cc[1].x); cc[1].x = fma(a.y, vi.y, cc[1].x);
                        cc[1].y = fma(a.x, vi.y, cc[1].y); cc[1].y = fma(-a.y, vi.x, cc[1].y);
                    }
                }
            };
            for (int i = i0; i < nl; i += 2 * TD_NW) {      // LDS rows, two per trip
                double2 r0, r1;
                row_lds(i, r0);
                row_lds(i + TD_NW, r1);
                fold2(r0, r1, i, i + TD_NW < nl ? i + TD_NW : -1);
            }
            double2 rreg[TD_NS + 1];
#pragma unroll
            for (int s = 0; s < TD_NS; ++s) {               // register rows
                const int i = TD_RB + 8 * s + w;
                double2 rs = make_double2(0.0, 0.0);
                if (i > k && i < n) {
                    const double2 vi = v_of(i);
#pragma unroll
                    for (int c = 0; c < td_nc(s); ++c) {
                        const double2 a = ar[td_off(s) + c];           // entries right of the diagonal are kept at zero
                        rs.x = fma(a.x, vcol[c].x, rs.x); rs.x = fma(-a.y, vcol[c].y, rs.x);
                        rs.y = fma(a.x, vcol[c].y, rs.y); rs.y = fma(a.y, vcol[c].x, rs.y);
                        double2 am = a;
                        if (c == td_nc(s) - 1 && 64 * c + lane >= i) am = make_double2(0.0, 0.0);   // strictly below the diagonal
                        cc[c].x = fma(am.x, vi.x, cc[c].x); cc[c].x = fma(am.y, vi.y, cc[c].x);
                        cc[c].y = fma(am.x, vi.y, cc[c].y); cc[c].y = fma(-am.y, vi.x, cc[c].y);
                    }
                }
                rreg[s] = rs;
            }
            rreg[TD_NS] = make_double2(0.0, 0.0);
#pragma unroll
            for (int s = 0; s < TD_NS; s += 2) {
                const int ia = TD_RB + 8 * s + w, ib = ia + 8;
                if (ib > k && ia < n)                       // at least one of the two rows is live (ia < ib)
                    fold2(rreg[s], rreg[s + 1], (ia > k && ia < n) ? ia : -1, (s + 1 < TD_NS && ib < n) ? ib : -1);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 64 * c + lane;
                if (j < n) colp[w * TD_NMAX + j] = cc[c];
            }
            __syncthreads();
            for (int t = k + 1 + tid; t < n; t += TD_NT) {
                double2 y = make_double2((rowp4[t] + rowp4[TD_NMAX + t]) + (rowp4[2 * TD_NMAX + t] + rowp4[3 * TD_NMAX + t]),
                                         (rowp4[4 * TD_NMAX + t] + rowp4[5 * TD_NMAX + t]) +
                                             (rowp4[6 * TD_NMAX + t] + rowp4[7 * TD_NMAX + t]));
#pragma unroll
                for (int q = 0; q < TD_NW; ++q) {
                    const double2 cpart = colp[q * TD_NMAX + t];
                    y.x += cpart.x;
                    y.y += cpart.y;
                }
                pbuf[t] = td_cmul(tk, y);
            }
            __syncthreads();
            // ---- alpha2 = -1/2 tau (p^H v), w = p + alpha2 v ----
            double pr = 0.0, pi = 0.0;
            double2 pj[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 64 * c + lane;
                pj[c] = (j > k && j < n) ? pbuf[j] : make_double2(0.0, 0.0);
                const double2 t = td_cmulc(pj[c], vcol[c]);
                pr += t.x;
                pi += t.y;
            }
            pr = dmk_wave_sum(pr);
            pi = dmk_wave_sum(pi);
            a2 = td_cmul(tk, make_double2(pr, pi));
            a2.x *= -0.5;
            a2.y *= -0.5;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double2 av = td_cmul(a2, vcol[c]);
                wcol[c] = make_double2(pj[c].x + av.x, pj[c].y + av.y);     // zero where v and p are zero (j <= k, j >= n)
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) wcol[c] = make_double2(0.0, 0.0);
        }
        // ---- A22 -= v w^H + w v^H (lower triangle) and capture of column k + 1 / d[k + 1] ----
        const int jn = k + 1, cn = jn >> 6, ln = jn & 63;
        auto w_of = [&](int i, double2 vi) -> double2 {
            const double2 p = pbuf[i], av = td_cmul(a2, vi);
            return make_double2(p.x + av.x, p.y + av.y);
        };
        auto upd = [&](double2 a, double2 vi, double2 wi, double2 vj, double2 wj) -> double2 {
            a.x = fma(-vi.x, wj.x, a.x); a.x = fma(-vi.y, wj.y, a.x); a.x = fma(-wi.x, vj.x, a.x); a.x = fma(-wi.y, vj.y, a.x);
            a.y = fma(-vi.y, wj.x, a.y); a.y = fma(vi.x, wj.y, a.y); a.y = fma(-wi.y, vj.x, a.y); a.y = fma(wi.x, vj.y, a.y);
            return a;
        };
        for (int i = i0; i < nl; i += TD_NW) {
            double2 *row = Lm + i * (i + 1) / 2;
            double2 vi = make_double2(0.0, 0.0), wi = vi;
            if (active) {
                vi = v_of(i);
                wi = w_of(i, vi);
            }
            if (lane <= i) {
                double2 a = row[lane];
                if (active) {
                    a = upd(a, vi, wi, vcol[0], wcol[0]);
                    if (lane == i) a.y = 0.0;
                    row[lane] = a;
                }
                if (cn == 0 && lane == ln) {
                    if (i > jn) xn[i] = a;
                    else d[jn] = a.x;
                }
            }
            if (i >= 64 && 64 + lane <= i) {
                double2 a = row[64 + lane];
                if (active) {
                    a = upd(a, vi, wi, vcol[1], wcol[1]);
                    if (64 + lane == i) a.y = 0.0;
                    row[64 + lane] = a;
                }
                if (cn == 1 && lane == ln) {
                    if (i > jn) xn[i] = a;
                    else d[jn] = a.x;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < TD_NS; ++s) {
            const int i = TD_RB + 8 * s + w;
            if (i > k && i < n) {
                double2 vi = make_double2(0.0, 0.0), wi = vi;
                if (active) {
                    vi = v_of(i);
                    wi = w_of(i, vi);
                }
#pragma unroll
                for (int c = 0; c < td_nc(s); ++c) {
                    double2 a = ar[td_off(s) + c];
                    if (active) {
                        double2 an = upd(a, vi, wi, vcol[c], wcol[c]);
                        if (c == td_nc(s) - 1) {                               // the chunk that holds the diagonal
                            const int j = 64 * c + lane;
                            if (j == i) an.y = 0.0;
                            if (j > i) an = make_double2(0.0, 0.0);
                        }
                        a = an;
                        ar[td_off(s) + c] = a;
                    }
                    if (c == cn && lane == ln) {
                        if (i > jn) xn[i] = a;
                        else d[jn] = a.x;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        e[n - 1] = 0.0;
        tau[n - 1] = make_double2(0.0, 0.0);
    }
}


// ---- K1a', tile layout: the same tridiagonalisation with the triangle as 16 x 16 tiles in MFMA ACCUMULATOR layout ---------------
// The row-layout kernel above is bound by its own instruction stream (8 f64 FMAs per complex element and step on the VALU plus one
// 64-lane reduction per row) while the matrix cores idle.  Here the lower triangle is 91 complex tiles (13 tile rows, n <= 208);
// element (R, C) of a tile sits in lane C + 16 (R % 4), register R / 4 -- the D layout of v_mfma_f64_16x16x4_f64 -- so that
//   * the rank-2 update A22 -= v w^H + w v^H is ONE MFMA per real tile: K = 4 = {v_r, v_i, w_r, w_i} (operands from three small LDS
//     tables written once per step), no VALU work at all;
//   * the row sums of the product are accumulated over the tiles of a row segment and reduced across 16 lanes once per FOUR rows
//     (a transposed DPP reduction of eight values, 54 instructions per segment), the column sums across the four lane groups;
//   * 72 tiles live in registers -- nine slots of 16 VGPRs per wave, runs of consecutive tiles of one tile row dealt to the waves
//     so that every wave keeps work until the end (t3_tab, tools-generated) -- and the 19 tiles that retire first (tile rows
//     0 .. 4 and the first tile of rows 5 .. 8) in LDS (ten register slots spill: 256 VGPRs + 196 B of scratch).
// Both partial sums of a wave go into its own slice of `ypart` with LDS atomics (one writer per address and instruction, program
// order within the wave: deterministic); four barriers per step.
constexpr int T3_NT = 512, T3_NW = 8, T3_N = 208, T3_RS = 9, T3_LS = 3, T3_SL = T3_RS + T3_LS;

struct T3Tab {
    signed char I[T3_NW][T3_SL], J[T3_NW][T3_SL];
    unsigned char last[T3_NW][T3_SL];          // the tile closes its row segment (flush the row sums)
};
__constant__ T3Tab t3_tab = {
    {{ 5,  5, 10, 10, 10, 11, 11, 11, 11,  7, -1, -1},
     { 6,  6,  9,  9,  9,  9,  9,  9,  9,  4,  3,  5},
     { 5,  5,  5,  8,  8,  8,  8, 11, 11,  4,  1,  2},
     { 7,  7,  7, 10, 10, 10, 10, 12, 12,  4,  6, -1},
     { 6,  6,  9,  9,  9, 12, 12, 12, 12,  4,  2,  1},
     { 8,  8,  8,  8, 11, 11, 11, 12, 12,  3,  4,  2},
     { 7,  7,  7,  7, 12, 12, 12, 12, 12,  8, -1, -1},
     { 6,  6, 10, 10, 10, 10, 11, 11, 11,  3,  3,  0}},
    {{ 1,  2,  8,  9, 10,  3,  4,  5,  6,  0, -1, -1},
     { 1,  2,  3,  4,  5,  6,  7,  8,  9,  1,  0,  0},
     { 3,  4,  5,  1,  2,  3,  4, 10, 11,  2,  0,  0},
     { 1,  2,  3,  4,  5,  6,  7,  9, 10,  0,  0, -1},
     { 3,  4,  0,  1,  2,  5,  6,  7,  8,  4,  2,  1},
     { 5,  6,  7,  8,  0,  1,  2,  3,  4,  3,  3,  1},
     { 4,  5,  6,  7,  0,  1,  2, 11, 12,  0, -1, -1},
     { 5,  6,  0,  1,  2,  3,  7,  8,  9,  2,  1,  0}},
    {{ 0,  1,  0,  0,  1,  0,  0,  0,  1,  1,  0,  0},
     { 0,  1,  0,  0,  0,  1,  0,  0,  1,  1,  1,  1},
     { 0,  0,  1,  0,  0,  0,  1,  0,  1,  1,  1,  1},
     { 0,  0,  1,  0,  0,  0,  1,  0,  1,  1,  1,  0},
     { 0,  1,  0,  0,  1,  0,  0,  0,  1,  1,  1,  1},
     { 0,  0,  0,  1,  0,  0,  1,  0,  1,  1,  1,  1},
     { 0,  0,  0,  1,  0,  0,  1,  0,  1,  1,  0,  0},
     { 0,  1,  0,  0,  0,  1,  0,  0,  1,  1,  1,  1}}};

constexpr size_t T3_LDS = (size_t)T3_NW * T3_LS * 256 * 16        // LDS tiles [wave][slot][r][lane] c128
                          + (size_t)(2 * T3_N + T3_N + T3_N + T3_NW * T3_N) * 16   // xbuf[2], pbuf, vbuf, ypart[8]
                          + (size_t)3 * T3_N * 4 * 8;             // AR, AI, BB operand tables

__global__ __launch_bounds__(T3_NT, 1) void tridiag_tiles_kernel(const TdArgs g) {
    extern __shared__ __attribute__((aligned(16))) char td_smem[];
    const int n = g.n, b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = lane & 15, q = lane >> 4;
    double2 *ltile = reinterpret_cast<double2 *>(td_smem);                 // [8][T3_LS][4][64]
    double2 *xbuf = ltile + T3_NW * T3_LS * 256;                            // [2][T3_N]
    double2 *pbuf = xbuf + 2 * T3_N;
    double2 *vbuf = pbuf + T3_N;                                            // [T3_N] reflector of the step
    double2 *ypart = vbuf + T3_N;                                           // [8][T3_N]
    double *AR = reinterpret_cast<double *>(ypart + T3_NW * T3_N);          // [T3_N][4]
    double *AI = AR + 4 * T3_N, *BB = AI + 4 * T3_N;
    const size_t nn = (size_t)n * n;
    const double2 *A = g.A + b * nn;
    const double *addm = g.add ? g.add + (size_t)(g.add_group > 0 ? b / g.add_group : 0) * nn : nullptr;
    double2 *Vh = g.Vh + b * nn;
    double2 *tau = g.tau + (size_t)b * n;
    double *d = g.d + (size_t)b * n, *e = g.e + (size_t)b * n;

    auto load_a = [&](int i, int j) -> double2 {           // Hermitian element (i, j) from the lower triangle (+ add); zero padding
        if (i >= n || j >= n) return make_double2(0.0, 0.0);
        const int hi = i >= j ? i : j, lo = i >= j ? j : i;
        double2 v = A[(size_t)hi * n + lo];
        if (i < j) v.y = -v.y;
        if (i == j) v.y = 0.0;
        if (addm) v.x += addm[(size_t)hi * n + lo];
        return v;
    };
    // this wave's tile list, read once (scalar registers; every later use is a compile-time slot index)
    int tcode[T3_SL];                                      // I | J << 8 | last << 16, or -1
#pragma unroll
    for (int s = 0; s < T3_SL; ++s) {
        const int I = t3_tab.I[w][s], J = t3_tab.J[w][s], L = t3_tab.last[w][s];
        tcode[s] = __builtin_amdgcn_readfirstlane(I < 0 ? -1 : (I | (J << 8) | (L << 16)));
    }
#define T3_I(s) (tcode[s] < 0 ? -1 : (tcode[s] & 0xff))
#define T3_J(s) (tcode[s] < 0 ? -1 : ((tcode[s] >> 8) & 0xff))
#define T3_L(s) (tcode[s] >= 0 && ((tcode[s] >> 16) & 1))
    d4_t tre[T3_RS], tim[T3_RS];
#pragma unroll
    for (int s = 0; s < T3_SL; ++s) {
        const int I = T3_I(s), J = T3_J(s);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double2 v = I >= 0 ? load_a(16 * I + q + 4 * r, 16 * J + C) : make_double2(0.0, 0.0);
            if (s < T3_RS) { tre[s < T3_RS ? s : 0][r] = v.x; tim[s < T3_RS ? s : 0][r] = v.y; }
            else ltile[((w * T3_LS + (s - T3_RS)) * 4 + r) * 64 + lane] = v;
        }
    }
    for (int i = tid; i < 2 * T3_N; i += T3_NT) xbuf[i] = (i >= 1 && i < n) ? load_a(i, 0) : make_double2(0.0, 0.0);
    for (int i = tid; i < T3_NW * T3_N; i += T3_NT) ypart[i] = make_double2(0.0, 0.0);
    if (tid == 0) d[0] = load_a(0, 0).x;
    __syncthreads();

    for (int k = 0; k + 1 < n; ++k) {
        const double2 *x = xbuf + (k & 1) * T3_N;
        double2 *xn = xbuf + ((k + 1) & 1) * T3_N;
        double part = 0.0;
        for (int j = k + 2 + lane; j < n; j += 64) {
            const double2 t = x[j];
            part += t.x * t.x + t.y * t.y;
        }
        const double xnorm2 = dmk_wave_sum(part);
        const double2 alpha = x[k + 1];
        double2 tk = make_double2(0.0, 0.0), scale = make_double2(0.0, 0.0);
        double beta = alpha.x;
        if (!(xnorm2 == 0.0 && alpha.y == 0.0)) {
            const double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xnorm2);
            beta = alpha.x >= 0.0 ? -nrm : nrm;
            tk = make_double2((beta - alpha.x) / beta, -alpha.y / beta);
            const double dr = alpha.x - beta, di = alpha.y;
            const double den = dr * dr + di * di;
            scale = make_double2(dr / den, -di / den);
        }
        const bool active = !(tk.x == 0.0 && tk.y == 0.0);
        // v_i: 1 at k + 1, x_i / (alpha - beta) below, 0 elsewhere -- once per step into LDS (the tiles read it ~25 times per wave)
        if (tid < T3_N) {
            double2 v = make_double2(0.0, 0.0);
            if (tid == k + 1) v = make_double2(1.0, 0.0);
            else if (tid > k + 1 && tid < n) v = td_cmul(x[tid], scale);
            vbuf[tid] = v;
        }
        __syncthreads();
        auto v_of = [&](int i) -> double2 { return vbuf[i]; };
        if (w == (k & 7)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 64 * c + lane;
                if (j < n) Vh[(size_t)k * n + j] = v_of(j);          // zeros up to the diagonal: the row is complete
            }
            if (lane == 0) {
                tau[k] = tk;
                e[k] = beta;
            }
        }
        double2 *yw = ypart + w * T3_N;
        if (active) {
            // ---- p = tau A22 v ----
            double2 racc[4], u[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) racc[r] = make_double2(0.0, 0.0);
            bool seg_open = false;
#pragma unroll
            for (int s = 0; s < T3_SL; ++s) {
                int code = tcode[s];
                asm volatile("" : "+s"(code));             // opaque: offsets derived from the tile coordinates are recomputed, not
                const int I = code < 0 ? -1 : (code & 0xff), J = code < 0 ? -1 : ((code >> 8) & 0xff);   // hoisted and spilled
                const bool last_of_seg = code >= 0 && ((code >> 16) & 1);
                const bool live = I >= 0 && 16 * J + 15 > k && 16 * J < n && 16 * I < n;
                if (live) {
                    double2 T[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (s < T3_RS) T[r] = make_double2(tre[s < T3_RS ? s : 0][r], tim[s < T3_RS ? s : 0][r]);
                        else T[r] = ltile[((w * T3_LS + (s - T3_RS)) * 4 + r) * 64 + lane];
                    }
                    if (!seg_open) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) u[r] = v_of(16 * I + q + 4 * r);
                        seg_open = true;
                    }
                    const double2 vc = v_of(16 * J + C);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        racc[r].x = fma(T[r].x, vc.x, racc[r].x); racc[r].x = fma(-T[r].y, vc.y, racc[r].x);
                        racc[r].y = fma(T[r].x, vc.y, racc[r].y); racc[r].y = fma(T[r].y, vc.x, racc[r].y);
                    }
                    if (I > J) {                           // Hermitian counterpart: y_col += conj(a) v_row
                        double cr = 0.0, ci = 0.0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            cr = fma(T[r].x, u[r].x, cr); cr = fma(T[r].y, u[r].y, cr);
                            ci = fma(T[r].x, u[r].y, ci); ci = fma(-T[r].y, u[r].x, ci);
                        }
                        const bool g1 = q & 1;             // lane groups 0 / 2 collect re, 1 / 3 im
                        double val = (g1 ? ci : cr) + __shfl_xor(g1 ? cr : ci, 16, 64);
                        val += __shfl_xor(val, 32, 64);
                        if (q < 2) atomicAdd(reinterpret_cast<double *>(yw + 16 * J + C) + q, val);
                    }
                }
                if (last_of_seg && seg_open) {
                    // eight values (four rows, re / im) over the 16 lanes of each lane group
                    const bool b0 = lane & 1, b1 = lane & 2;
                    double a_[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        a_[r] = (b0 ? racc[r].y : racc[r].x) + td_dpp<0xb1>(b0 ? racc[r].x : racc[r].y);      // quad_perm [1,0,3,2]
                    double p0 = (b1 ? a_[1] : a_[0]) + td_dpp<0x4e>(b1 ? a_[0] : a_[1]);                         // quad_perm [2,3,0,1]
                    double p1 = (b1 ? a_[3] : a_[2]) + td_dpp<0x4e>(b1 ? a_[2] : a_[3]);
                    p0 += td_dpp<0x124>(p0); p0 += td_dpp<0x128>(p0);                                            // row_ror 4, 8
                    p1 += td_dpp<0x124>(p1); p1 += td_dpp<0x128>(p1);
                    // lane (b1, b0) of quad 0 holds row b1, of quad 1 row 2 + b1; component b0
                    if (C < 8) {
                        const int row = 16 * I + q + 4 * (((C >> 2) << 1) + (b1 ? 1 : 0));   // D layout: register r holds row q + 4 r
                        atomicAdd(reinterpret_cast<double *>(yw + row) + (b0 ? 1 : 0), (C >> 2) ? p1 : p0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) racc[r] = make_double2(0.0, 0.0);
                    seg_open = false;
                }
            }
            __syncthreads();
            if (tid < T3_N) {
                double2 y = make_double2(0.0, 0.0);
#pragma unroll
                for (int qq = 0; qq < T3_NW; ++qq) {
                    const double2 o = ypart[qq * T3_N + tid];
                    y.x += o.x;
                    y.y += o.y;
                    ypart[qq * T3_N + tid] = make_double2(0.0, 0.0);
                }
                pbuf[tid] = (tid > k && tid < n) ? td_cmul(tk, y) : make_double2(0.0, 0.0);
            }
            __syncthreads();
            double pr = 0.0, pi = 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 64 * c + lane;
                if (j > k && j < n) {
                    const double2 t = td_cmulc(pbuf[j], v_of(j));
                    pr += t.x;
                    pi += t.y;
                }
            }
            pr = dmk_wave_sum(pr);
            pi = dmk_wave_sum(pi);
            double2 a2 = td_cmul(tk, make_double2(pr, pi));
            a2.x *= -0.5;
            a2.y *= -0.5;
            if (tid < T3_N) {
                const double2 v = v_of(tid), av = td_cmul(a2, v);
                const double2 wv = make_double2(pbuf[tid].x + av.x, pbuf[tid].y + av.y);
                double *ar = AR + 4 * tid, *ai = AI + 4 * tid, *bb = BB + 4 * tid;
                ar[0] = -v.x; ar[1] = -v.y; ar[2] = -wv.x; ar[3] = -wv.y;      // Re(A) -= v_r w_r + v_i w_i + w_r v_r + w_i v_i
                ai[0] = -v.y; ai[1] = v.x; ai[2] = -wv.y; ai[3] = wv.x;        // Im(A) -= v_i w_r - v_r w_i + w_i v_r - w_r v_i
                bb[0] = wv.x; bb[1] = wv.y; bb[2] = v.x; bb[3] = v.y;
            }
            __syncthreads();
        }
        // ---- A22 -= v w^H + w v^H on the matrix cores; capture of column k + 1 and d[k + 1] ----
        const int jn = k + 1, Jn = jn >> 4, Cn = jn & 15;
#pragma unroll
        for (int s = 0; s < T3_SL; ++s) {
            int code = tcode[s];
            asm volatile("" : "+s"(code));
            const int I = code < 0 ? -1 : (code & 0xff), J = code < 0 ? -1 : ((code >> 8) & 0xff);
            if (I >= 0 && 16 * J + 15 > k && 16 * J < n && 16 * I < n) {
                d4_t re, im;
                if (s < T3_RS) { re = tre[s < T3_RS ? s : 0]; im = tim[s < T3_RS ? s : 0]; }
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double2 t = ltile[((w * T3_LS + (s - T3_RS)) * 4 + r) * 64 + lane];
                        re[r] = t.x;
                        im[r] = t.y;
                    }
                }
                if (active) {
                    const double are = AR[4 * (16 * I + C) + q], aim = AI[4 * (16 * I + C) + q], bop = BB[4 * (16 * J + C) + q];
                    re = __builtin_amdgcn_mfma_f64_16x16x4f64(are, bop, re, 0, 0, 0);
                    im = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, bop, im, 0, 0, 0);
                    if (I == J) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (q + 4 * r == C) im[r] = 0.0;
                    }
                    if (s < T3_RS) { tre[s < T3_RS ? s : 0] = re; tim[s < T3_RS ? s : 0] = im; }
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) ltile[((w * T3_LS + (s - T3_RS)) * 4 + r) * 64 + lane] = make_double2(re[r], im[r]);
                    }
                }
                if (J == Jn && C == Cn) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * I + q + 4 * r;
                        if (row > jn && row < n) xn[row] = make_double2(re[r], im[r]);
                        else if (row == jn) d[jn] = re[r];
                    }
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        e[n - 1] = 0.0;
        tau[n - 1] = make_double2(0.0, 0.0);
    }
}

#undef T3_I
#undef T3_J
#undef T3_L

// ---- K1c: back-transformation y_m = H_0 ... H_{n-2} z_m, 16 eigenvectors per wave, components IN the lanes -------------------
// Phase 3 of eigh.hip keeps four eigenvectors per wave with the lanes along the components: every reflector costs two wave
// reductions per eigenvector (the dot product v^H y), 160 of its ~290 instructions.  Here lane l of a wave owns eigenvector
// m0 + (l & 15) and the components i = 4 t + (l >> 4), t = 0 .. 49, in registers: the dot product is a serial sum inside the lane
// plus ONE four-way exchange (lanes l, l ^ 16, l ^ 32, l ^ 48) per reflector; the reflector is read from an LDS copy shared by
// the seven waves of the workgroup (four distinct addresses per instruction, broadcast within each group of 16); retired
// components are skipped by wave-uniform branches in blocks of 20.
// Measured alternatives (432 x 200, MI355X): one private LDS copy per wave (every wave streams Vh itself) 2.32 ms; this kernel
// 2.29 ms; one eigenvector per lane with the reflector through the scalar cache as SGPR operands (no LDS reads at all) 4.35 ms --
// the s_load round trips of a block sit in front of its FMAs and two waves per SIMD do not cover them.  The phase is bound by
// the LDS return path (two 16-byte reads per component and reflector); the remedy is a blocked (WY) form on the matrix cores.
constexpr int BT_T = TD_NMAX / 4;            // components per lane
constexpr int BT_NW = 7;                     // 13 groups of 16 eigenvectors at n = 200: two workgroups of 7 / 6 groups per matrix
constexpr int BT_NT = 64 * BT_NW;

struct BtArgs {
    int n, batch, wgs_per_mat;
    const double *Zt;        // batch x n x n: row m = eigenvector m of the tridiagonal matrix
    const double2 *Vh, *tau;
    const int *rank;         // batch x n: output row of eigenvector m
    double2 *Vt;             // batch x n x n
};

__global__ __launch_bounds__(BT_NT, 1) void backtransform_kernel(const BtArgs g) {
    // one copy of the current reflector per workgroup: thread i < n carries v_i of the next reflector in a register while the
    // current one is applied; one barrier per reflector
    __shared__ __attribute__((aligned(16))) double2 vb[2][4 * BT_T];
    __shared__ double2 tb_s[2];
    const int n = g.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / g.wgs_per_mat;
    const int grp = (blockIdx.x % g.wgs_per_mat) * BT_NW + wave;
    const bool live = 16 * grp < n;              // a wave without eigenvectors still takes part in the barriers
    const size_t nn = (size_t)n * n;
    const int m = 16 * grp + (lane & 15), q = lane >> 4;
    const double *Zt = g.Zt + b * nn;
    const double2 *Vh = g.Vh + b * nn;
    const double2 *tau = g.tau + (size_t)b * n;
    double2 y[BT_T];
#pragma unroll
    for (int t = 0; t < BT_T; ++t) {
        const int i = 4 * t + q;
        y[t] = (live && m < n && i < n) ? make_double2(Zt[(size_t)m * n + i], 0.0) : make_double2(0.0, 0.0);
    }
    // row k of the reflectors: v_i for i > k (v_{k+1} = 1 is stored), zero elsewhere; thread 4 BT_T carries tau
    auto fetch = [&](int k) -> double2 {
        if (k < 0) return make_double2(0.0, 0.0);
        if (tid < 4 * BT_T) return (tid > k && tid < n) ? Vh[(size_t)k * n + tid] : make_double2(0.0, 0.0);
        if (tid == 4 * BT_T) return tau[k];
        return make_double2(0.0, 0.0);
    };
    double2 pre = fetch(n - 2);
    constexpr int TB = 5;                        // components are retired in blocks of 4 TB = 20 (one uniform branch per block)
    for (int k = n - 2; k >= 0; --k) {
        const int buf = k & 1;
        if (tid < 4 * BT_T) vb[buf][tid] = pre;
        else if (tid == 4 * BT_T) tb_s[buf] = pre;
        pre = fetch(k - 1);
        __syncthreads();                         // reflector k is in vb[buf]; the other buffer was last read before the previous barrier
        const double2 tk = tb_s[buf];
        if ((tk.x == 0.0 && tk.y == 0.0) || !live) continue;
        const double2 *v = vb[buf];
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int tb = 0; tb < BT_T / TB; ++tb) {
            if (4 * TB * (tb + 1) - 1 > k) {     // wave-uniform: some component of this block is still live (v is zero below k + 1)
#pragma unroll
                for (int t = TB * tb; t < TB * (tb + 1); ++t) {
                    const double2 vi = v[4 * t + q];
                    sr = fma(vi.x, y[t].x, sr);                   // conj(v) * y
                    sr = fma(vi.y, y[t].y, sr);
                    si = fma(vi.x, y[t].y, si);
                    si = fma(-vi.y, y[t].x, si);
                }
            }
        }
        sr += __shfl_xor(sr, 16, 64);
        si += __shfl_xor(si, 16, 64);
        sr += __shfl_xor(sr, 32, 64);
        si += __shfl_xor(si, 32, 64);
        const double2 f = td_cmul(tk, make_double2(sr, si));
#pragma unroll
        for (int tb = 0; tb < BT_T / TB; ++tb) {
            if (4 * TB * (tb + 1) - 1 > k) {
#pragma unroll
                for (int t = TB * tb; t < TB * (tb + 1); ++t) {
                    const double2 vi = v[4 * t + q];
                    y[t].x = fma(-f.x, vi.x, y[t].x);
                    y[t].x = fma(f.y, vi.y, y[t].x);
                    y[t].y = fma(-f.x, vi.y, y[t].y);
                    y[t].y = fma(-f.y, vi.x, y[t].y);
                }
            }
        }
    }
    if (live && m < n) {
        double2 *out = g.Vt + b * nn + (size_t)g.rank[(size_t)b * n + m] * n;
#pragma unroll
        for (int t = 0; t < BT_T; ++t) {
            const int i = 4 * t + q;
            if (i < n) out[i] = y[t];
        }
    }
}

// ---- K1d: blocked (compact WY) back-transformation on the matrix cores -------------------------------------------------------
// Sixteen reflectors at a time: H_k0 ... H_k0+15 = I - V T V^H (T upper triangular, larft), so a block is applied as
//     X = V^H Y,   X <- T X,   Y <- Y - V X
// and with the register layout of K1c -- lane (e = l & 15, q = l >> 4) of a wave owns eigenvector e and, in register r of tile c,
// component 16 c + q + 4 r -- every operand of the three products is already where v_mfma_f64_16x16x4_f64 wants it:
//   * Y tile c IS an accumulator tile (rows = components, columns = eigenvectors), and register r' of tile c' is the B operand
//     of the K chunk (components 4 t .. 4 t + 3, t = 4 c' + r') of X = V^H Y;
//   * X comes out as an accumulator tile (rows = reflectors) whose register j is, in the same lane, the B operand of K chunk j
//     (reflectors 4 j .. 4 j + 3) of T X and of V X: no cross-lane traffic at all;
//   * the A operands (V^H, T, V) are read from LDS: the reflector block reflector-major with rows padded to 209 elements
//     (conflict-free both along the reflectors and along the components), T padded to 17.
// One LDS read per component and SIXTEEN reflectors instead of two per reflector; the T factors come from a small kernel.
constexpr int WY_NB = 16, WY_LDV = 257, WY_NT = 64 * BT_NW, WY_NTILE = T3_N / 16;   // rows padded: conflict-free both ways, room for whole 64-lane DMA pieces

struct WyArgs {
    int n, batch, wgs_per_mat, nblk;
    const double *Zt;
    const double2 *Vh, *tau;
    double2 *T;              // batch x nblk x 16 x 16
    const int *rank;
    double2 *Vt;
};

// T of every block of 16 reflectors (forward, columnwise: T_jj = tau_j, T(0:j, j) = -tau_j T(0:j, 0:j) (V(:, 0:j)^H v_j))
__global__ __launch_bounds__(256) void wy_tfactor_kernel(const WyArgs g) {
    __shared__ double2 gram[WY_NB][WY_NB + 1], Ts[WY_NB][WY_NB + 1];
    __shared__ __attribute__((aligned(16))) double2 Vb[WY_NB][T3_N + 1];      // the block's reflector rows (one coalesced pass)
    const int n = g.n, b = blockIdx.x / g.nblk, blk = blockIdx.x % g.nblk, k0 = WY_NB * blk;
    const double2 *Vh = g.Vh + (size_t)b * n * n;
    const double2 *tau = g.tau + (size_t)b * n;
    {
        double2 pre[WY_NB * T3_N / 256];                   // all thirteen loads of a thread in flight at once
#pragma unroll
        for (int u = 0; u < WY_NB * T3_N / 256; ++u) {
            const int t = threadIdx.x + 256 * u, r = t / T3_N, i = t % T3_N, k = k0 + r;
            pre[u] = (k + 1 < n && i > k && i < n) ? Vh[(size_t)k * n + i] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < WY_NB * T3_N / 256; ++u) {
            const int t = threadIdx.x + 256 * u;
            Vb[t / T3_N][t % T3_N] = pre[u];
        }
    }
    __syncthreads();
    const int a = threadIdx.x >> 4, c = threadIdx.x & 15;
    double2 s = make_double2(0.0, 0.0);
    const int kc = k0 + c;
    if (a < c && kc + 1 < n) {                     // g_ac = v_a^H v_c; v_c is zero up to kc, one at kc + 1
        for (int i = kc + 1; i < n; ++i) {
            const double2 va = Vb[a][i], vc = Vb[c][i];
            s.x = fma(va.x, vc.x, s.x); s.x = fma(va.y, vc.y, s.x);
            s.y = fma(va.x, vc.y, s.y); s.y = fma(-va.y, vc.x, s.y);
        }
    }
    gram[a][c] = s;
    __syncthreads();
    // row i of T depends only on its own earlier entries (T_ij = -tau_j sum_{l = i}^{j-1} T_il g_lj): one thread per row keeps
    // the row in registers and walks the columns without a barrier
    if (threadIdx.x < WY_NB) {
        const int i = threadIdx.x;
        double2 row[WY_NB];
#pragma unroll
        for (int j = 0; j < WY_NB; ++j) {
            const int kj = k0 + j;
            const double2 tj = (kj + 1 < n) ? tau[kj] : make_double2(0.0, 0.0);
            double2 v = make_double2(0.0, 0.0);
            if (j == i) v = tj;
            else if (j > i) {
                double2 acc = make_double2(0.0, 0.0);
#pragma unroll
                for (int l = 0; l < j; ++l) {
                    if (l >= i) {
                        const double2 t = td_cmul(row[l], gram[l][j]);
                        acc.x += t.x;
                        acc.y += t.y;
                    }
                }
                const double2 r = td_cmul(tj, acc);
                v = make_double2(-r.x, -r.y);
            }
            row[j] = v;
            Ts[i][j] = v;
        }
    }
    __syncthreads();
    g.T[((size_t)b * g.nblk + blk) * 256 + threadIdx.x] = Ts[a][c];
}

__global__ __launch_bounds__(WY_NT, 1) void backtransform_wy_kernel(const WyArgs g) {
    extern __shared__ __attribute__((aligned(16))) char td_smem[];
    double2 *Vs2 = reinterpret_cast<double2 *>(td_smem);                // [2][16][WY_LDV] reflector blocks, reflector-major, double buffered
    double2 *Ts = Vs2 + 2 * WY_NB * WY_LDV;                             // [16][17]
    const int n = g.n;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = lane & 15, q = lane >> 4;
    const int b = blockIdx.x / g.wgs_per_mat;
    const int grp = (blockIdx.x % g.wgs_per_mat) * BT_NW + wave;
    const bool live = 16 * grp < n;
    const size_t nn = (size_t)n * n;
    const int m = 16 * grp + e;
    const double *Zt = g.Zt + b * nn;
    const double2 *Vh = g.Vh + b * nn;
    d4_t yr[WY_NTILE], yi[WY_NTILE];
#pragma unroll
    for (int c = 0; c < WY_NTILE; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * c + q + 4 * r;
            yr[c][r] = (live && m < n && i < n) ? Zt[(size_t)m * n + i] : 0.0;
            yi[c][r] = 0.0;
        }
    }
    // Reflector block `blk` into buffer blk & 1 by LDS-DMA: 16 rows x 4 pieces of 64 entries, dealt to the waves.  Row k of Vh holds
    // v_i for i > k and zeros up to the diagonal (the tridiagonalisation kernels write complete rows); lanes outside [0, n) and
    // rows of reflectors that do not exist read Vh[0][0], which is such a zero.
    auto stage = [&](int blk) {
        double2 *dst = Vs2 + (blk & 1) * WY_NB * WY_LDV;
        for (int p = wave; p < WY_NB * 4; p += BT_NW) {
            const int r = p >> 2, i = 64 * (p & 3) + lane, k = WY_NB * blk + r;
            const bool ok = k + 1 < n && i < n;
            const double2 *src = ok ? Vh + (size_t)k * n + i : Vh;            // Vh[0][0]: zero (row 0 is written in full)
            glds16(src, lds_addr_of(dst + r * WY_LDV + 64 * (p & 3)));
        }
    };
    stage(g.nblk - 1);
    for (int blk = g.nblk - 1; blk >= 0; --blk) {
        const int k0 = WY_NB * blk;
        const double2 *Vs = Vs2 + (blk & 1) * WY_NB * WY_LDV;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of block blk have landed
        __syncthreads();                                   // ... everyone's; and block blk + 1 (other buffer, Ts) is no longer read
        if (blk > 0) stage(blk - 1);                       // in flight underneath this block's products
        if (tid < 256) Ts[(tid >> 4) * 17 + (tid & 15)] = g.T[((size_t)b * g.nblk + blk) * 256 + tid];
        __syncthreads();
        if (!live) continue;
        const int c0 = (k0 + 1) >> 4;                      // first tile with a live component
        // ---- X = V^H Y ----
        d4_t xr = d4_t{0.0, 0.0, 0.0, 0.0}, xi = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < WY_NTILE; ++c) {
            if (c >= c0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {              // K chunk: components 16 c + 4 r' ... no: components (16 c + q' + 4 r), q' = 0 .. 3
                    const double2 v = Vs[e * WY_LDV + 16 * c + q + 4 * r];      // A: lane (reflector e, k = q) -> conj(V)[comp][e]
                    xr = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, yr[c][r], xr, 0, 0, 0);
                    xr = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, yi[c][r], xr, 0, 0, 0);
                    xi = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, yi[c][r], xi, 0, 0, 0);
                    xi = __builtin_amdgcn_mfma_f64_16x16x4f64(-v.y, yr[c][r], xi, 0, 0, 0);
                }
            }
        }
        // ---- X <- T X ----
        d4_t zr = d4_t{0.0, 0.0, 0.0, 0.0}, zi = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double2 tt = Ts[e * 17 + q + 4 * j];     // A: lane (row e, k = q) -> T[e][reflector of chunk j]
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(tt.x, xr[j], zr, 0, 0, 0);
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(-tt.y, xi[j], zr, 0, 0, 0);
            zi = __builtin_amdgcn_mfma_f64_16x16x4f64(tt.x, xi[j], zi, 0, 0, 0);
            zi = __builtin_amdgcn_mfma_f64_16x16x4f64(tt.y, xr[j], zi, 0, 0, 0);
        }
        // ---- Y <- Y - V X ----
#pragma unroll
        for (int c = 0; c < WY_NTILE; ++c) {
            if (c >= c0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double2 v = Vs[(q + 4 * j) * WY_LDV + 16 * c + e];    // A: lane (component 16 c + e, k = q) -> V[comp][reflector]
                    yr[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-v.x, zr[j], yr[c], 0, 0, 0);
                    yr[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, zi[j], yr[c], 0, 0, 0);
                    yi[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-v.x, zi[j], yi[c], 0, 0, 0);
                    yi[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-v.y, zr[j], yi[c], 0, 0, 0);
                }
            }
        }
    }
    if (live && m < n) {
        double2 *out = g.Vt + b * nn + (size_t)g.rank[(size_t)b * n + m] * n;
#pragma unroll
        for (int c = 0; c < WY_NTILE; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * c + q + 4 * r;
                if (i < n) out[i] = make_double2(yr[c][r], yi[c][r]);
            }
    }
}

}  // namespace

// Tridiagonalise `batch` complex Hermitian n x n matrices (n <= 200) into (d, e, tau, Vh) in the layout phases 2 / 3 of
// eigh_kernel read.  Returns 1 when launched, 0 when the shape is outside the resident kernel, < 0 on error.
int launch_tridiag_resident(dmk_ctx *ctx, int n, int batch, const void *A, const double *add, int add_group, void *Vh, void *tau,
                            double *d, double *e) {
    if (n < 2 || n > TD_NMAX || batch <= 0) return 0;
    TdArgs g;
    g.n = n; g.batch = batch; g.A = reinterpret_cast<const double2 *>(A); g.add = add; g.add_group = add_group;
    g.Vh = reinterpret_cast<double2 *>(Vh); g.tau = reinterpret_cast<double2 *>(tau); g.d = d; g.e = e;
    // per launch: the attribute belongs to the (function, device) pair and a process may hold contexts on several devices
    DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_resident_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)TD_LDS));
    DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_tiles_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS));
    // read per call (tests run both layouts in one process): the tile layout is the default, DMK_EIGH_TILES=0 selects the row layout
    const bool tiles_on = !(getenv("DMK_EIGH_TILES") && atoi(getenv("DMK_EIGH_TILES")) == 0);
    if (tiles_on) hipLaunchKernelGGL(tridiag_tiles_kernel, dim3(batch), dim3(T3_NT), T3_LDS, ctx->stream, g);
    else hipLaunchKernelGGL(tridiag_resident_kernel, dim3(batch), dim3(TD_NT), TD_LDS, ctx->stream, g);
    DMK_CHECK_LAUNCH(ctx);
    return 1;
}

// Back-transformation of the tridiagonal eigenvectors Zt (rows) with the reflectors (Vh, tau) of launch_tridiag_resident /
// eigh_kernel phase 1 into the rows rank[m] of Vt (c128).  n <= 200.  Tws: batch x ceil((n - 1) / 16) x 256 c128 of workspace for
// the T factors of the blocked form (nullptr: one reflector at a time).  Returns 1 when launched, 0 when out of range.
int launch_backtransform(dmk_ctx *ctx, int n, int batch, const double *Zt, const void *Vh, const void *tau, const int *rank, void *Vt,
                         void *Tws) {
    if (n < 2 || n > TD_NMAX || batch <= 0) return 0;
    // read per call: the blocked form on the matrix cores is the default, DMK_EIGH_WY=0 applies the reflectors one at a time
    const bool wy = Tws && !(getenv("DMK_EIGH_WY") && atoi(getenv("DMK_EIGH_WY")) == 0);
    if (wy) {
        WyArgs g;
        g.n = n; g.batch = batch;
        g.wgs_per_mat = ((n + 15) / 16 + BT_NW - 1) / BT_NW;
        g.nblk = (n - 1 + WY_NB - 1) / WY_NB;
        g.Zt = Zt; g.Vh = reinterpret_cast<const double2 *>(Vh); g.tau = reinterpret_cast<const double2 *>(tau);
        g.T = reinterpret_cast<double2 *>(Tws); g.rank = rank; g.Vt = reinterpret_cast<double2 *>(Vt);
        const size_t lds = (size_t)(2 * WY_NB * WY_LDV + WY_NB * 17) * sizeof(double2);
        DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(backtransform_wy_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   // per (function, device)
        hipLaunchKernelGGL(wy_tfactor_kernel, dim3((unsigned)(batch * g.nblk)), dim3(256), 0, ctx->stream, g);
        hipLaunchKernelGGL(backtransform_wy_kernel, dim3((unsigned)(batch * g.wgs_per_mat)), dim3(WY_NT), lds, ctx->stream, g);
        DMK_CHECK_LAUNCH(ctx);
        return 1;
    }
    BtArgs g;
    g.n = n; g.batch = batch;
    g.wgs_per_mat = ((n + 15) / 16 + BT_NW - 1) / BT_NW;
    g.Zt = Zt; g.Vh = reinterpret_cast<const double2 *>(Vh); g.tau = reinterpret_cast<const double2 *>(tau); g.rank = rank;
    g.Vt = reinterpret_cast<double2 *>(Vt);
    hipLaunchKernelGGL(backtransform_kernel, dim3((unsigned)(batch * g.wgs_per_mat)), dim3(BT_NT), 0, ctx->stream, g);
    DMK_CHECK_LAUNCH(ctx);
    return 1;
}
