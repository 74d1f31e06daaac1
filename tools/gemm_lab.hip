// Kernel laboratory: ablations of the 512-thread LDS-DMA-ring dgemm (not part of the product).
// MODE 0: full; 1: no global loads after the prologue (compute + LDS + barriers only);
// 2: loads + barriers, no MFMA; 3: full but WITHOUT the per-tile barrier wait coupling (no barrier; WRONG results, timing only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ __forceinline__ unsigned lds_addr_of(const void *p) { return (unsigned)(size_t)(lds_void_t *)p; }
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_base) {
    unsigned keep;
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(base) : "memory");
}
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks >> 3, r = nblocks & 7u;
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

constexpr int GBM = 256, GBN = 128, GBK = 16, GD = 3, GNT = 512;
constexpr int GA_LD = GBM + 16, GB_LD = GBN + 16;
constexpr int GA_STAGE = GBK * GA_LD, GB_STAGE = GBK * GB_LD;
constexpr int G_STAGE = GA_STAGE + GB_STAGE;

template <int MODE, int GROUP>
__global__ __launch_bounds__(GNT, 2) void big_kernel(int M, int N, int K, double alpha, const double *__restrict__ X, int64_t ldx,
    const double *__restrict__ Y, int64_t ldy, double *__restrict__ C, int64_t ldc, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) double lds[GD * G_STAGE];
    const unsigned nblocks = (unsigned)tiles_m * (unsigned)tiles_n;
    const unsigned lid = xcd_remap(blockIdx.x, nblocks);
    const unsigned per_group = GROUP * (unsigned)tiles_n;
    const unsigned g = lid / per_group;
    const unsigned first_m = g * GROUP;
    const unsigned gsize = min((unsigned)tiles_m - first_m, (unsigned)GROUP);
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
        if (MODE != 1 && MODE != 5) {
            if (t + 1 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE != 3) __builtin_amdgcn_s_barrier();
        if (MODE != 1 && MODE != 5 && t + 2 < T) issue(t + 2);
        if (MODE == 2) continue;
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
    if (MODE == 4 || MODE == 5) {       // no real epilogue: keep the accumulators alive with an impossible store
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (alpha == 12345.678) C[tid] = s;
        return;
    }
    if (MODE == 7) {                    // fire-and-forget f64 atomics (single writer per address: deterministic)
#pragma unroll
        for (int i = 0; i < 4; ++i)
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
        return;
    }
    if (MODE == 6) {                    // batched epilogue: 16 loads in flight, then 16 stores
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v[4][4];
            double *ptr[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int row = m0 + wm * 64 + i * 16 + frag_k + 4 * r, col = n0 + wn * 64 + j * 16 + frag_x;
                    const bool ok = row < M && col < N;
                    row = ok ? row : 0; col = ok ? col : 0;
                    ptr[r][j] = ok ? C + (int64_t)row * ldc + col : nullptr;
                    v[r][j] = ok ? *ptr[r][j] : 0.0;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (ptr[r][j]) *ptr[r][j] = v[r][j] + alpha * acc[i][j][r];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
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

template <int MODE, int GROUP>
void run(const char *tag, int N, int K, const double *X, double *C) {
    const int btm = (N + GBM - 1) / GBM, btn = (N + GBN - 1) / GBN;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    big_kernel<MODE, GROUP><<<btm * btn, GNT>>>(N, N, K, 1.0, X, N, X, N, C, N, btm, btn);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        big_kernel<MODE, GROUP><<<btm * btn, GNT>>>(N, N, K, 1.0, X, N, X, N, C, N, btm, btn);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-34s N=%d K=%d: %.3f ms  %.2f TF\n", tag, N, K, best, 2.0 * K * (double)N * N / (best * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32896, K = argc > 2 ? atoi(argv[2]) : 1600;
    double *X, *C;
    CK(hipMalloc(&X, sizeof(double) * (size_t)K * N)); CK(hipMalloc(&C, sizeof(double) * (size_t)N * N));
    std::vector<double> h((size_t)K * N);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
    CK(hipMemcpy(X, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    CK(hipMemset(C, 0, sizeof(double) * (size_t)N * N));
    run<0, 4>("full group4", N, K, X, C);
    run<0, 8>("full group8", N, K, X, C);
    run<0, 16>("full group16", N, K, X, C);
    run<0, 2>("full group2", N, K, X, C);
    run<1, 4>("no global loads (compute only)", N, K, X, C);
    run<2, 4>("loads+barriers only (no MFMA)", N, K, X, C);
    run<3, 4>("full, no barrier (wrong, timing)", N, K, X, C);
    run<4, 4>("full loads, no epilogue", N, K, X, C);
    run<5, 4>("no loads, no epilogue", N, K, X, C);
    run<6, 4>("full, batched epilogue", N, K, X, C);
    run<7, 4>("full, atomic epilogue", N, K, X, C);
    return 0;
}
