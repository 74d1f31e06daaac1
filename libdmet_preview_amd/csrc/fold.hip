// K3 -- k <-> R folds of a FULL Gamma-centred mesh as one fused mixed-radix pass.
//
//   R2k:  out[k] = sum_R e^{-i k.R} in[R]              reference: system/fourier.py:160-166 (FFTtoK = np.fft.fftn over the cell axes)
//   k2R:  out[R] = (1/nk) sum_k e^{+i k.R} in[k]       reference: system/fourier.py:168-177 (FFTtoT = ifftn, real part, |Im| checked)
//
// Rounds 1-3 ran these as a dense nk x nk DFT-GEMM (zgemm.hip): 8 nk^2 flop per column where the separable transform needs
// 8 nk (n0 + n1 + n2), and 3 % of the HBM roof at C5 (6 x 6 x 6, 80 000 columns of 216 cells: 0.72 ms per launch for 414 MB).
// This kernel is the HBM-shaped version: a workgroup owns a tile of CT columns, loads all nk cells of the tile into LDS with
// coalesced row segments (CT x 16 B contiguous per cell), runs the three axis passes IN PLACE -- one thread per (line, column):
// the n_d points of a line go to registers, the small DFT is a direct sum against the n_d twiddles of that axis (exact on the
// quarter turns, cosl / sinl of one reduced fraction elsewhere, like the GEMM's table) and goes back to the same LDS cells, so a
// pass has no cross-thread hazards and costs one barrier -- and stores once (k2R: real part scaled by 1 / nk, max |Im| folded
// into the caller's flag word).  Algorithmic bytes: 16 nk per column in + 8 (k2R) or 16 (R2k) out; nothing else touches HBM.
// The dense GEMM form stays for the k-SUBSET partial fold of the multi-rank path (a rank holds a subset of k: not a full mesh)
// and for axes longer than 16 or meshes whose tile does not fit LDS.
#include "common.h"
#include <cmath>
#include <cstdlib>
#include <vector>

namespace {

constexpr int FD_NT = 256;
constexpr int FD_MAXN = 16;
constexpr int FD_LDS_BYTES = 72 * 1024;          // two workgroups per CU inside the 160 KiB LDS

struct FoldArgs {
    const void *in;       // [batch][nk][ncol] c128 (or f64 when IN_REAL)
    void *out;            // [batch][nk][ncol] c128 (or f64 when OUT_REAL)
    const double2 *tw;    // [3][FD_MAXN]: e^{-2 pi i t / n_d} (forward); the inverse conjugates
    long long ncol;
    int n0, n1, n2, nk, ct, tiles;
    double scale;
    double *imag_max;     // OUT_REAL: atomic max of |Im| * scale (may be null)
};

template <int N>
__device__ __forceinline__ void line_dft_static(double2 *x0, int stride, const double2 *tw, bool inverse) {
    double2 x[N], w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x[i] = x0[i * stride];
        w[i] = tw[i];
        if (inverse) w[i].y = -w[i].y;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double re = x[0].x, im = x[0].y;
#pragma unroll
        for (int i = 1; i < N; ++i) {
            const double2 ww = w[(i * j) % N];
            re = fma(x[i].x, ww.x, fma(-x[i].y, ww.y, re));
            im = fma(x[i].x, ww.y, fma(x[i].y, ww.x, im));
        }
        x0[j * stride] = make_double2(re, im);
    }
}

// 9 <= n <= 16 (runtime): inputs in registers, outputs one at a time with a running twiddle index (LDS broadcast reads)
__device__ __forceinline__ void line_dft_rt(int n, double2 *x0, int stride, const double2 *tw, bool inverse) {
    double2 x[FD_MAXN];
#pragma unroll
    for (int i = 0; i < FD_MAXN; ++i)
        if (i < n) x[i] = x0[i * stride];
    const double sg = inverse ? -1.0 : 1.0;
    for (int j = 0; j < n; ++j) {
        double re = x[0].x, im = x[0].y;
        int t = 0;
#pragma unroll
        for (int i = 1; i < FD_MAXN; ++i) {
            if (i < n) {
                t += j;
                if (t >= n) t -= n;
                const double2 ww = tw[t];
                const double wy = sg * ww.y;
                re = fma(x[i].x, ww.x, fma(-x[i].y, wy, re));
                im = fma(x[i].x, wy, fma(x[i].y, ww.x, im));
            }
        }
        x0[j * stride] = make_double2(re, im);
    }
}

__device__ __forceinline__ void line_dft(int n, double2 *x0, int stride, const double2 *tw, bool inverse) {
    switch (n) {
        case 2: line_dft_static<2>(x0, stride, tw, inverse); break;
        case 3: line_dft_static<3>(x0, stride, tw, inverse); break;
        case 4: line_dft_static<4>(x0, stride, tw, inverse); break;
        case 5: line_dft_static<5>(x0, stride, tw, inverse); break;
        case 6: line_dft_static<6>(x0, stride, tw, inverse); break;
        case 7: line_dft_static<7>(x0, stride, tw, inverse); break;
        case 8: line_dft_static<8>(x0, stride, tw, inverse); break;
        default: line_dft_rt(n, x0, stride, tw, inverse); break;
    }
}

template <bool IN_REAL, bool OUT_REAL, bool INVERSE>
__global__ __launch_bounds__(FD_NT, 2) void fold_fft_kernel(const FoldArgs g) {
    extern __shared__ __attribute__((aligned(16))) double2 X[];        // [nk][ct]
    __shared__ double2 tws[3 * FD_MAXN];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / g.tiles, tile = blockIdx.x - b * g.tiles;
    const int ct = g.ct, nk = g.nk;
    const long long c0 = (long long)tile * ct;
    const int cw = (int)((g.ncol - c0) < ct ? (g.ncol - c0) : ct);     // valid columns of this tile
    if (tid < 3 * FD_MAXN) tws[tid] = g.tw[tid];

    // ---- load: cell k, column c0 + c -> X[k][c]; a row segment is ct consecutive elements ----------------------------
    const int c = tid % ct, kr = tid / ct, kstep = FD_NT / ct;
    const int cc = c < cw ? c : cw - 1;                                 // tail tile: clamped columns are never stored
    if (IN_REAL) {
        const double *src = reinterpret_cast<const double *>(g.in) + (long long)b * nk * g.ncol + c0 + cc;
#pragma unroll 4
        for (int k = kr; k < nk; k += kstep) X[k * ct + c] = make_double2(src[(long long)k * g.ncol], 0.0);
    } else {
        const double2 *src = reinterpret_cast<const double2 *>(g.in) + (long long)b * nk * g.ncol + c0 + cc;
#pragma unroll 4
        for (int k = kr; k < nk; k += kstep) X[k * ct + c] = src[(long long)k * g.ncol];
    }
    __syncthreads();

    // ---- three axis passes, in place: item = (line, column).  Axis d has stride s_d in the cell index (s_2 = 1, s_1 = n2,
    //      s_0 = n1 n2); its lines are (outer, inner < s_d) -> first cell outer s_d n_d + inner -----------------------------
    for (int d = 2; d >= 0; --d) {
        const int nd = d == 2 ? g.n2 : d == 1 ? g.n1 : g.n0;
        if (nd == 1) continue;
        const int sd = d == 2 ? 1 : d == 1 ? g.n2 : g.n1 * g.n2;
        const int items = (nk / nd) * ct;
        for (int it = tid; it < items; it += FD_NT) {
            const int line = it / ct, col = it - line * ct;
            const int outer = line / sd, inner = line - outer * sd;
            line_dft(nd, X + (outer * sd * nd + inner) * ct + col, sd * ct, tws + d * FD_MAXN, INVERSE);
        }
        __syncthreads();
    }

    // ---- store ------------------------------------------------------------------------------------------------------------
    if (OUT_REAL) {
        double *dst = reinterpret_cast<double *>(g.out) + (long long)b * nk * g.ncol + c0 + c;
        double im = 0.0;
        if (c < cw) {
#pragma unroll 4
            for (int k = kr; k < nk; k += kstep) {
                const double2 v = X[k * ct + c];
                dst[(long long)k * g.ncol] = v.x * g.scale;
                im = fmax(im, fabs(v.y * g.scale));
            }
        }
        if (g.imag_max) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) im = fmax(im, __shfl_xor(im, off, 64));
            if ((tid & 63) == 0 && im > 0.0)                            // |x| orders like its bit pattern
                atomicMax(reinterpret_cast<unsigned long long *>(g.imag_max), (unsigned long long)__double_as_longlong(fabs(im)));
        }
    } else {
        double2 *dst = reinterpret_cast<double2 *>(g.out) + (long long)b * nk * g.ncol + c0 + c;
        if (c < cw) {
#pragma unroll 4
            for (int k = kr; k < nk; k += kstep) {
                const double2 v = X[k * ct + c];
                dst[(long long)k * g.ncol] = make_double2(v.x * g.scale, v.y * g.scale);
            }
        }
    }
}

}  // namespace

// Forward twiddles of the three axes, cached per mesh in the context (dmk_ctx::phases with nsub = -1).
static int fold_twiddles(dmk_ctx *ctx, const int n[3], const double2 **dev) {
    for (auto &p : ctx->phases)
        if (p.nsub == -1 && p.mesh[0] == n[0] && p.mesh[1] == n[1] && p.mesh[2] == n[2]) {
            *dev = reinterpret_cast<const double2 *>(p.dev);
            return DMK_OK;
        }
    std::vector<double> host((size_t)2 * 3 * FD_MAXN, 0.0);
    for (int d = 0; d < 3; ++d)
        for (int t = 0; t < n[d]; ++t) {
            double cr, sr;
            if ((4 * t) % n[d] == 0) {                                   // quarter turns are exact
                const int q = (4 * t) / n[d];
                const double cq[4] = {1, 0, -1, 0}, sq[4] = {0, -1, 0, 1};
                cr = cq[q]; sr = sq[q];
            } else {
                const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)t / (long double)n[d];
                cr = (double)cosl(ang); sr = (double)sinl(ang);
            }
            host[2 * ((size_t)d * FD_MAXN + t)] = cr;
            host[2 * ((size_t)d * FD_MAXN + t) + 1] = sr;
        }
    void *d = nullptr;
    if (hipMalloc(&d, host.size() * sizeof(double)) != hipSuccess) return dmk_fail(ctx, DMK_ERR_NOMEM, "fold: twiddle allocation failed");
    DMK_HIP(ctx, hipMemcpyAsync(d, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dmk_ctx::Phase ph;
    ph.mesh[0] = n[0]; ph.mesh[1] = n[1]; ph.mesh[2] = n[2];
    ph.dir = 0; ph.nsub = -1; ph.dev = d;
    ctx->phases.push_back(ph);
    *dev = reinterpret_cast<const double2 *>(d);
    return DMK_OK;
}

// Returns 1 if handled, 0 if the caller must use the DFT-GEMM (axis > 16, tile does not fit, DMK_FOLD_FFT=0), < 0 on error.
// inverse: k -> R (conjugate twiddles, scale 1 / nk); out_real only with inverse.
int launch_fold_fft(dmk_ctx *ctx, const int n[3], long long ncol, int batch, const void *in, int in_real, void *out, int out_real,
                    int inverse, double *imag_max) {
    if (const char *e = getenv("DMK_FOLD_FFT")) if (atoi(e) == 0) return 0;
    const long long nk = (long long)n[0] * n[1] * n[2];
    if (n[0] > FD_MAXN || n[1] > FD_MAXN || n[2] > FD_MAXN || (in_real && inverse) || (out_real && !inverse)) return 0;
    int ct = 16;                                                       // 256-B row segments; narrower only when the mesh is large
    while (ct > 2 && nk * ct * 16 > FD_LDS_BYTES) ct >>= 1;
    if (nk * ct * 16 > FD_LDS_BYTES) return 0;
    const long long tiles = (ncol + ct - 1) / ct;
    if (tiles * batch > 0x7fffffffLL) return 0;
    const double2 *tw = nullptr;
    int rc = fold_twiddles(ctx, n, &tw);
    if (rc) return rc;
    FoldArgs g;
    g.in = in; g.out = out; g.tw = tw; g.ncol = ncol;
    g.n0 = n[0]; g.n1 = n[1]; g.n2 = n[2]; g.nk = (int)nk; g.ct = ct; g.tiles = (int)tiles;
    g.scale = inverse ? 1.0 / (double)nk : 1.0;
    g.imag_max = imag_max;
    const size_t lds = (size_t)nk * ct * sizeof(double2);
    const dim3 grid((unsigned)(tiles * batch)), block(FD_NT);
    FamScope fs(ctx, DMK_FAM_FOLD);
#define FOLD_LAUNCH(IR, OR, INV)                                                                                              \
    do {                                                                                                                        \
        DMK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(fold_fft_kernel<IR, OR, INV>),                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)FD_LDS_BYTES));                        \
        hipLaunchKernelGGL((fold_fft_kernel<IR, OR, INV>), grid, block, lds, ctx->stream, g);                                    \
    } while (0)
    if (inverse) {
        if (out_real) FOLD_LAUNCH(false, true, true);
        else FOLD_LAUNCH(false, false, true);
    } else {
        if (in_real) FOLD_LAUNCH(true, false, false);
        else FOLD_LAUNCH(false, false, false);
    }
#undef FOLD_LAUNCH
    DMK_CHECK_LAUNCH(ctx);
    return 1;
}
