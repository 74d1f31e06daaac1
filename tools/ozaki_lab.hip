// LAB (round 5; not part of the product): the contraction  C (N x N) = X^T Y,  X, Y (K x N) f64, K = 1600
// (basis_transform/eri_transform.py:436-485 `_Lij_s4_to_eri`; product kernel: csrc/dgemm_tn.hip at 69-70 TF = 0.89 of the f64
// matrix pipe) as an Ozaki-style sum of INT8 products on the i8 matrix cores (v_mfma_i32_16x16x64_i8, ~50x the f64 MFMA rate).
//
//   x_ka = 2^(e_a) * sum_s d_s[a][k] 2^(-6 - 7 s) + r,   d_s integers in [-64, 64], |r| <= 2^(e_a - 7 S)      (per COLUMN a)
//   C_ab ~ 2^(e_a + e_b) * sum_{s + t < S} 2^(-12 - 7 (s + t)) * sum_k d^X_s[a][k] d^Y_t[b][k]
//
// every inner sum is EXACT in int32 (|d| <= 64, K <= 1600 + padding: < 2^24 per pair, < 2^27 per anti-diagonal), the only
// error is the truncation: |dC_ab| <= (S + 1) K 2^(-7 S) 2^(e_a + e_b) (bound printed; the dropped pairs s + t >= S).  S is
// chosen from the data so that the bound meets an ABSOLUTE tolerance (the north star's 1e-8 on the ERI, split over the kL sum):
// the benchmark's planes need S = 4 (10 slice pairs), data with a 1e5 dynamic range in |eri| S = 7-8 (28-36 pairs: no gain).
//
// What the lab measures: slicing pass (f64 planes -> S int8 slice planes, transposed to k-contiguous rows), the fused slice-pair
// GEMM with f64 recombination per anti-diagonal, accuracy against an f64 reference, and the time against the f64 pipe.
//   hipcc -O3 --offload-arch=gfx950 tools/ozaki_lab.hip -o tools/ozaki_lab && tools/ozaki_lab [N=8192] [K=1600] [S=4]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef double d4_t __attribute__((ext_vector_type(4)));

// ---- inputs: uniform in (-1, 1) * column scale (a mild dynamic range across columns, like tril-packed pair densities) --------
__global__ void fill_kernel(long long n, int N, unsigned long long seed, double *out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
        const int col = (int)(i % N);
        out[i] = u * (0.25 + 0.75 * (double)((col * 2654435761u) >> 24) / 256.0) * 1.0e-2;
    }
}

// ---- column exponents: e_a = smallest integer with max_k |x_ka| <= 2^e_a ------------------------------------------------------
__global__ void colexp_kernel(int K, int N, const double *__restrict__ X, int *__restrict__ ex) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= N) return;
    double m = 0.0;
    for (int k = 0; k < K; ++k) m = fmax(m, fabs(X[(long long)k * N + a]));
    int e = 0;
    if (m > 0.0) {
        (void)frexp(m, &e);                 // m = f 2^e, 0.5 <= f < 1  ->  m < 2^e
    } else {
        e = -1000;
    }
    ex[a] = e;
}

// ---- slicing: Xs[s][a][k] (k contiguous, row pitch KP bytes), 32 x 32 tile transpose through LDS ------------------------------
template <int S>
__global__ __launch_bounds__(256) void slice_kernel(int K, int KP, int N, const double *__restrict__ X, const int *__restrict__ ex,
                                                    int8_t *__restrict__ Xs) {
    __shared__ int8_t tile[S][32][33];
    const int k0 = blockIdx.y * 32, a0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, a = a0 + tx;
        double v = 0.0;
        if (k < K && a < N) v = ldexp(X[(long long)k * N + a], -ex[a]);       // |v| <= 1
        double scale = 64.0;                                                   // 2^6, then 2^13, ...
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double d = rint(v * scale);                                 // |d| <= 64
            tile[s][r][tx] = (int8_t)(int)d;
            v -= d / scale;                                                    // exact: d / scale has <= 7 significant bits
            scale *= 128.0;
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int a = a0 + r, k = k0 + tx;
        if (a < N && k < KP) {
#pragma unroll
            for (int s = 0; s < S; ++s) Xs[((long long)s * N + a) * KP + k] = tile[s][tx][r];
        }
    }
}

// ---- the fused slice-pair GEMM ------------------------------------------------------------------------------------------------
// workgroup tile 128 (a) x 128 (b), 2 x 2 waves of 64 x 64 (4 x 4 MFMA tiles, i32x4 each); a K step is 128 bytes (two MFMA k-steps of
// 64).  For every anti-diagonal d = s + t < S: int32 accumulation over its pairs and all K steps, then acc_f64 += 2^(-12 - 7 d) acc_i32.
// Operands: global -> registers -> LDS (double buffered through registers), LDS rows padded to 144 bytes.
constexpr int BKB = 128;                 // bytes of k per step
constexpr int LDP = BKB + 16;            // LDS row pitch
template <int S>
__global__ __launch_bounds__(256, 2) void ozaki_gemm_kernel(int N, int KP, const int8_t *__restrict__ Xs, const int8_t *__restrict__ Ys,
                                                            const int *__restrict__ exX, const int *__restrict__ exY,
                                                            double *__restrict__ C) {
    __shared__ __attribute__((aligned(16))) int8_t lds[2][2][128 * LDP];      // [buffer][A | B][row][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int a0 = blockIdx.y * 128, b0 = blockIdx.x * 128;
    const int li = lane & 15, lg = lane >> 4;
    // loader: thread t moves 16 bytes: row = t / 8 + 32 * pass (4 passes), k chunk = t % 8
    const int lrow = tid >> 3, lchunk = tid & 7;
    // (first version: f64 accumulators of the whole wave tile in registers next to the int32 ones -- 256 VGPRs + 416 B of scratch,
    // 400 TOPS.  The anti-diagonals are folded into the C tile through L2 instead: one read-modify-write of 128 KB per diagonal.)
    const int T = KP / BKB;
    for (int d = 0; d < S; ++d) {
        i32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
        for (int s = 0; s <= d; ++s) {
            const int t2 = d - s;
            const int8_t *A = Xs + ((long long)s * N + a0) * KP, *B = Ys + ((long long)t2 * N + b0) * KP;
            int4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;      // (scalars: as arrays behind lambdas they stayed in scratch)
            const int8_t *Ag[4], *Bg[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = lrow + 32 * p;
                const int ar = (a0 + row < N) ? row : 0, br = (b0 + row < N) ? row : 0;
                Ag[p] = A + (long long)ar * KP + lchunk * 16;
                Bg[p] = B + (long long)br * KP + lchunk * 16;
            }
#define gload(t) do { const int o_ = (t) * BKB; \
                ra0 = *reinterpret_cast<const int4 *>(Ag[0] + o_); rb0 = *reinterpret_cast<const int4 *>(Bg[0] + o_); \
                ra1 = *reinterpret_cast<const int4 *>(Ag[1] + o_); rb1 = *reinterpret_cast<const int4 *>(Bg[1] + o_); \
                ra2 = *reinterpret_cast<const int4 *>(Ag[2] + o_); rb2 = *reinterpret_cast<const int4 *>(Bg[2] + o_); \
                ra3 = *reinterpret_cast<const int4 *>(Ag[3] + o_); rb3 = *reinterpret_cast<const int4 *>(Bg[3] + o_); } while (0)
#define lstore(buf) do { int8_t *la_ = &lds[buf][0][lrow * LDP + lchunk * 16], *lb_ = &lds[buf][1][lrow * LDP + lchunk * 16]; \
                *reinterpret_cast<int4 *>(la_) = ra0;                *reinterpret_cast<int4 *>(lb_) = rb0; \
                *reinterpret_cast<int4 *>(la_ + 32 * LDP) = ra1;     *reinterpret_cast<int4 *>(lb_ + 32 * LDP) = rb1; \
                *reinterpret_cast<int4 *>(la_ + 64 * LDP) = ra2;     *reinterpret_cast<int4 *>(lb_ + 64 * LDP) = rb2; \
                *reinterpret_cast<int4 *>(la_ + 96 * LDP) = ra3;     *reinterpret_cast<int4 *>(lb_ + 96 * LDP) = rb3; } while (0)
            gload(0);
            __syncthreads();                       // the previous pair's last reads of both buffers are done
            lstore(0);
            for (int t = 0; t < T; ++t) {
                const int buf = t & 1;
                if (t + 1 < T) gload(t + 1);
                __syncthreads();                   // buffer `buf` is complete
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    i32x4 fa[4], fb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        fa[i] = *reinterpret_cast<const i32x4 *>(&lds[buf][0][(wm * 64 + i * 16 + li) * LDP + kk * 64 + lg * 16]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        fb[j] = *reinterpret_cast<const i32x4 *>(&lds[buf][1][(wn * 64 + j * 16 + li) * LDP + kk * 64 + lg * 16]);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
                if (t + 1 < T) lstore(buf ^ 1);    // the other buffer was last read in step t - 1: everyone is past that barrier
            }
#undef gload
#undef lstore
        }
        // fold this anti-diagonal into the C tile (D layout, dtype independent: col = lane & 15, row = (lane >> 4) * 4 + reg);
        // the column / row exponents are applied with the last one
        const double sc = ldexp(1.0, -12 - 7 * d);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = a0 + wm * 64 + i * 16 + lg * 4 + r, b = b0 + wn * 64 + j * 16 + li;
                    if (a < N && b < N) {
                        double *c = C + (long long)a * N + b;
                        double v = sc * (double)acc[i][j][r];
                        if (d > 0) v += *c;
                        if (d == S - 1) v = ldexp(v, exX[a] + exY[b]);
                        *c = v;
                    }
                }
    }
}

// ---- second kernel: 256 x 256 workgroup tile, LDS-DMA ring ------------------------------------------------------------------------
// What bounds the first kernel is the operand feed (a 128 x 128 tile consumes 64 B per clock and CU: 39 TB/s over the chip at the
// int8 rate, above what the L2 delivers) and its one K step of prefetch distance.  Here: workgroup tile 256 x 256 (half the bytes
// per operation), 2 x 2 waves of 128 x 128 (8 x 8 MFMA tiles: 256 accumulator registers, i.e. ONE workgroup per CU with the
// accumulators in AGPRs), K step 64 bytes = one v_mfma_i32_16x16x64_i8 per tile, operands by LDS-DMA into a four-stage ring
// (prefetch distance three K steps, one barrier per step), 64-byte LDS rows with the 16-byte chunk position XOR-swizzled by
// (0, 3, 2, 1)[(row >> 2) & 3] so that the ds_read_b128 lane groups of a fragment read sixteen different bank quads.  The slice
// pairs of ALL anti-diagonals run through the ring as one sequence of K steps; at the end of a diagonal the int32 tile is folded
// into C through L2 as before.  N must be a multiple of 256 (lab).
__device__ __forceinline__ unsigned oz_lds_addr(const void *p) {
    typedef __attribute__((address_space(3))) void lds_void_t;
    return (unsigned)(size_t)(lds_void_t *)p;
}
__device__ __forceinline__ void oz_glds16(const void *gsrc, unsigned lds_base) {
    unsigned keep;
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(base) : "memory");
}
constexpr int OZ_STAGES = 4;
constexpr int OZ_STAGE_BYTES = 2 * 256 * 64;          // A panel | B panel
template <int S, int ABL = 0>       // ABL (ablation, wrong results, timing only): 1 = no LDS-DMA after the prologue, 2 = fragments read once, 4 = no barrier
__global__ __launch_bounds__(256, 1) void ozaki_gemm256_kernel(int N, int KP, const int8_t *__restrict__ Xs, const int8_t *__restrict__ Ys,
                                                               const int *__restrict__ exX, const int *__restrict__ exY,
                                                               double *__restrict__ C) {
    extern __shared__ __attribute__((aligned(1024))) int8_t ring[];           // [OZ_STAGES][A 256 x 64 | B 256 x 64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    // XCD-aware tile order: the 32 workgroups an XCD runs together form 8 x 4 tiles (8 A panels, 4 B panels shared in its L2)
    const int NT = N / 256;
    unsigned bid = blockIdx.x;
    {
        const unsigned nblocks = gridDim.x, q = nblocks >> 3, r = nblocks & 7u;
        const unsigned xcd = bid & 7u, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int cb = (int)(bid / (4u * NT)), rem = (int)(bid % (4u * NT));
    const int width = min(4, NT - 4 * cb);
    const int ta = rem / width, tb = 4 * cb + rem % width;
    const int a0 = ta * 256, b0 = tb * 256;
    const int T = KP / 64;
    constexpr int NPAIR = S * (S + 1) / 2;
    const int nsteps = NPAIR * T;
    const unsigned ring0 = oz_lds_addr(ring);
    // loader: a wave-instruction moves 16 rows x 64 bytes; wave w takes the row groups w, w + 4, ... of both panels (8 instructions)
    const int lrow = lane >> 2, lpos = lane & 3;
    const int lchunk = lpos ^ ((0x6C >> (2 * ((lrow >> 2) & 3))) & 3);       // (0, 3, 2, 1)[(row >> 2) & 3] = bits of 0b01101100
    auto pair_of = [&](int pi, int &s, int &t2) {                            // pairs in anti-diagonal order: d = 0: (0,0); d = 1: (0,1), (1,0); ...
        int d = 0;
        while (pi > d) { pi -= d + 1; ++d; }
        s = pi; t2 = d - pi;
    };
    auto issue = [&](int q) {                                                // K step q of the whole sequence -> ring stage q % OZ_STAGES
        const int pi = q / T, t = q - pi * T;
        int s, t2;
        pair_of(pi, s, t2);
        const int8_t *A = Xs + ((long long)s * N + a0) * KP + t * 64, *B = Ys + ((long long)t2 * N + b0) * KP + t * 64;
        const unsigned st = ring0 + (unsigned)((q % OZ_STAGES) * OZ_STAGE_BYTES);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = wave + 4 * u;                                      // row group 0 .. 15
            oz_glds16(A + (long long)(g * 16 + lrow) * KP + lchunk * 16, st + (unsigned)(g * 1024));
            oz_glds16(B + (long long)(g * 16 + lrow) * KP + lchunk * 16, st + (unsigned)(256 * 64 + g * 1024));
        }
    };
    i32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
    // fragment addresses inside a stage: row r, chunk lg at position lg ^ f(r); tile i of a wave is 16 rows = 1024 bytes further
    // and has the same swizzle (bits 2-3 of the row come from the lane), so ONE base per operand and immediate offsets
    const int ra0 = wm * 128 + li, rb0 = wn * 128 + li;
    const unsigned offA0 = (unsigned)(ra0 * 64 + ((lg ^ ((0x6C >> (2 * ((ra0 >> 2) & 3))) & 3)) * 16));
    const unsigned offB0 = (unsigned)(256 * 64 + rb0 * 64 + ((lg ^ ((0x6C >> (2 * ((rb0 >> 2) & 3))) & 3)) * 16));
    issue(0);
    if (nsteps > 1) issue(1);
    if (nsteps > 2) issue(2);
    int q = 0;
    i32x4 fa[8], fb[8];
    if (ABL & 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    for (int d = 0; d < S; ++d) {
        const int qend = q + (d + 1) * T;                                    // the K steps of this anti-diagonal (the ring runs through)
#pragma unroll 1
        for (; q < qend; ++q) {
            // my pieces of stage q have landed when at most the 8 + 8 of the two later stages are outstanding
            if (!(ABL & 1)) {
                if (q + 2 < nsteps) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (q + 1 < nsteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (!(ABL & 4)) __syncthreads();                                 // everybody's pieces; and stage (q - 1) % 4 is free
            if (!(ABL & 1) && q + 3 < nsteps) issue(q + 3);
            const int8_t *st = ring + (q % OZ_STAGES) * OZ_STAGE_BYTES;
            if (!(ABL & 2) || q == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const i32x4 *>(st + offA0 + i * 1024);
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[j] = *reinterpret_cast<const i32x4 *>(st + offB0 + j * 1024);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        // fold the anti-diagonal into the C tile (D layout: col = lane & 15, row = (lane >> 4) * 4 + reg)
        const double sc = ldexp(1.0, -12 - 7 * d);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = a0 + wm * 128 + i * 16 + lg * 4 + r, b = b0 + wn * 128 + j * 16 + li;
                    double *c = C + (long long)a * N + b;
                    double v = sc * (double)acc[i][j][r];
                    if (d > 0) v += *c;
                    if (d == S - 1) v = ldexp(v, exX[a] + exY[b]);
                    *c = v;
                }
                acc[i][j] = i32x4{0, 0, 0, 0};
            }
    }
}

// ---- f64 reference on sampled entries ------------------------------------------------------------------------------------------
__global__ void ref_kernel(int K, int N, const double *__restrict__ X, const double *__restrict__ Y, int nsamp, const int *__restrict__ sa,
                           const int *__restrict__ sb, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsamp) return;
    double s = 0.0;
    for (int k = 0; k < K; ++k) s = fma(X[(long long)k * N + sa[i]], Y[(long long)k * N + sb[i]], s);
    out[i] = s;
}

// ---- f64 MFMA baseline (register-staged simple kernel is NOT the product's: the product's rate is quoted from profiles) ---------

static int g_kernel = 1;
template <int S>
double run(int N, int K, int KP, const double *X, const double *Y, const int *exX, const int *exY, int8_t *Xs, int8_t *Ys, double *C,
           double *t_slice_ms) {
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const dim3 gs((N + 31) / 32, (KP + 31) / 32);
    const dim3 gg((N + 127) / 128, (N + 127) / 128);
    float best_s = 1e30f, best_g = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(slice_kernel<S>, gs, dim3(256), 0, 0, K, KP, N, X, exX, Xs);
        hipLaunchKernelGGL(slice_kernel<S>, gs, dim3(256), 0, 0, K, KP, N, Y, exY, Ys);
        CK(hipEventRecord(e1));
        if (g_kernel == 2) {
            const size_t lds = (size_t)OZ_STAGES * OZ_STAGE_BYTES;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ozaki_gemm256_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(ozaki_gemm256_kernel<S>, dim3((N / 256) * (N / 256)), dim3(256), lds, 0, N, KP, Xs, Ys, exX, exY, C);
        } else if (g_kernel >= 3 && g_kernel <= 9) {
            const size_t lds = (size_t)OZ_STAGES * OZ_STAGE_BYTES;
            const dim3 g256((N / 256) * (N / 256));
#define OZ_ABL(a) case a: CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ozaki_gemm256_kernel<S, a>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                          hipLaunchKernelGGL((ozaki_gemm256_kernel<S, a>), g256, dim3(256), lds, 0, N, KP, Xs, Ys, exX, exY, C); break;
            switch (g_kernel - 2) { OZ_ABL(1) OZ_ABL(2) OZ_ABL(3) OZ_ABL(4) OZ_ABL(5) OZ_ABL(6) OZ_ABL(7) }
#undef OZ_ABL
        } else {
            hipLaunchKernelGGL(ozaki_gemm_kernel<S>, gg, dim3(256), 0, 0, N, KP, Xs, Ys, exX, exY, C);
        }
        CK(hipEventRecord(e2));
        CK(hipEventSynchronize(e2));
        float ms_s, ms_g;
        CK(hipEventElapsedTime(&ms_s, e0, e1));
        CK(hipEventElapsedTime(&ms_g, e1, e2));
        if (rep > 0) { best_s = fminf(best_s, ms_s); best_g = fminf(best_g, ms_g); }
    }
    *t_slice_ms = best_s;
    return best_g;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 8192, K = argc > 2 ? atoi(argv[2]) : 1600, S = argc > 3 ? atoi(argv[3]) : 4;
    const int KP = ((K + BKB - 1) / BKB) * BKB;
    g_kernel = argc > 4 ? atoi(argv[4]) : (N % 256 == 0 ? 2 : 1);
    if (g_kernel >= 2 && N % 256 != 0) { printf("kernel 2 needs N %% 256 == 0\n"); return 1; }
    printf("ozaki_lab: C = X^T Y, N = %d, K = %d (padded %d), S = %d slices -> %d slice pairs, kernel %d (%s)\n", N, K, KP, S, S * (S + 1) / 2,
           g_kernel, g_kernel == 2 ? "256 x 256 tiles, LDS-DMA ring" : g_kernel > 2 ? "256 x 256 ABLATION (bits of kernel - 2: 1 no LDS-DMA, 2 no fragment reads, 4 no barrier): results wrong, timing only" : "128 x 128 tiles, register staged");
    double *X, *Y, *C;
    int *exX, *exY;
    int8_t *Xs, *Ys;
    CK(hipMalloc(&X, (size_t)K * N * 8)); CK(hipMalloc(&Y, (size_t)K * N * 8)); CK(hipMalloc(&C, (size_t)N * N * 8));
    CK(hipMalloc(&exX, N * 4)); CK(hipMalloc(&exY, N * 4));
    CK(hipMalloc(&Xs, (size_t)S * N * KP)); CK(hipMalloc(&Ys, (size_t)S * N * KP));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (long long)K * N, N, 1234ull, X);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, (long long)K * N, N, 99991ull, Y);
    hipLaunchKernelGGL(colexp_kernel, dim3((N + 255) / 256), dim3(256), 0, 0, K, N, X, exX);
    hipLaunchKernelGGL(colexp_kernel, dim3((N + 255) / 256), dim3(256), 0, 0, K, N, Y, exY);
    CK(hipDeviceSynchronize());
    double t_slice = 0.0, t_gemm = 0.0;
    switch (S) {
        case 2: t_gemm = run<2>(N, K, KP, X, Y, exX, exY, Xs, Ys, C, &t_slice); break;
        case 3: t_gemm = run<3>(N, K, KP, X, Y, exX, exY, Xs, Ys, C, &t_slice); break;
        case 4: t_gemm = run<4>(N, K, KP, X, Y, exX, exY, Xs, Ys, C, &t_slice); break;
        case 5: t_gemm = run<5>(N, K, KP, X, Y, exX, exY, Xs, Ys, C, &t_slice); break;
        case 6: t_gemm = run<6>(N, K, KP, X, Y, exX, exY, Xs, Ys, C, &t_slice); break;
        case 7: t_gemm = run<7>(N, K, KP, X, Y, exX, exY, Xs, Ys, C, &t_slice); break;
        default: t_gemm = run<8>(N, K, KP, X, Y, exX, exY, Xs, Ys, C, &t_slice); break;
    }
    // accuracy on sampled entries + a priori bound
    const int nsamp = 4096;
    std::vector<int> sa(nsamp), sb(nsamp), hx(N), hy(N);
    for (int i = 0; i < nsamp; ++i) { sa[i] = (int)((i * 2654435761ull + 17) % N); sb[i] = (int)((i * 40503ull * 977 + 3) % N); }
    int *dsa, *dsb;
    double *dref;
    CK(hipMalloc(&dsa, nsamp * 4)); CK(hipMalloc(&dsb, nsamp * 4)); CK(hipMalloc(&dref, nsamp * 8));
    CK(hipMemcpy(dsa, sa.data(), nsamp * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), nsamp * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((nsamp + 255) / 256), dim3(256), 0, 0, K, N, X, Y, nsamp, dsa, dsb, dref);
    std::vector<double> ref(nsamp), got(nsamp);
    CK(hipMemcpy(ref.data(), dref, nsamp * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx.data(), exX, N * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy.data(), exY, N * 4, hipMemcpyDeviceToHost));
    double maxerr = 0.0, maxref = 0.0, maxbound = 0.0;
    for (int i = 0; i < nsamp; ++i) {
        CK(hipMemcpy(&got[i], C + (long long)sa[i] * N + sb[i], 8, hipMemcpyDeviceToHost));
        maxerr = fmax(maxerr, fabs(got[i] - ref[i]));
        maxref = fmax(maxref, fabs(ref[i]));
        maxbound = fmax(maxbound, (S + 1.0) * K * ldexp(1.0, -7 * S + hx[sa[i]] + hy[sb[i]]));
    }
    const double flop = 2.0 * (double)K * (double)N * (double)N;
    const double f64_ms_at_product_rate = flop / 69.5e12 * 1e3;            // the product's rectangular launch: 69.3-69.7 TF (profiles/r04_d_*)
    printf("  slicing (both operands): %.3f ms   gemm: %.3f ms   total %.3f ms\n", t_slice, t_gemm, t_slice + t_gemm);
    printf("  f64-equivalent rate: gemm only %.1f TF, with slicing %.1f TF  (product f64 kernel: 69.5 TF = %.3f ms for this shape)\n",
           flop / (t_gemm * 1e-3) / 1e12, flop / ((t_slice + t_gemm) * 1e-3) / 1e12, f64_ms_at_product_rate);
    printf("  speed-up over the product's f64 kernel: gemm only %.2fx, with slicing %.2fx\n", f64_ms_at_product_rate / t_gemm,
           f64_ms_at_product_rate / (t_slice + t_gemm));
    const double i8ops = 2.0 * (double)KP * (double)N * (double)N * (S * (S + 1) / 2);
    printf("  int8 rate: %.0f TOPS (micro-benchmark floor of the pipe: 3944)\n", i8ops / (t_gemm * 1e-3) / 1e12);
    printf("  accuracy on %d sampled entries: max |C - ref| = %.3e (max |ref| %.3e, relative %.2e); a priori bound %.3e\n", nsamp, maxerr,
           maxref, maxerr / maxref, maxbound);
    return 0;
}
