// Shader clock under load: a one-wave sampler kernel on its own stream reads s_memtime (shader-clock counter) and
// s_memrealtime (constant 100 MHz) every ~50 us while the contraction kernel of libdmetk occupies the chip on the
// context stream.  Prints the clock the matrix pipe actually ran at -- the denominator the measured TFLOP/s should be
// held against (spec peak assumes 2.4 GHz).
//   hipcc -O3 --offload-arch=gfx950 tools/clock_probe.hip -Llibdmet_preview_amd -ldmetk -Wl,-rpath,$PWD/libdmet_preview_amd -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/libdmetk.h"

__global__ void sampler(unsigned long long *out, int nsamp, unsigned long long gap_rt, volatile int *stop) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < nsamp; ++i) {
        const unsigned long long rt0 = wall_clock64();
        out[2 * i] = clock64();
        out[2 * i + 1] = rt0;
        while (wall_clock64() - rt0 < gap_rt) __builtin_amdgcn_s_sleep(8);
        if (*stop) { for (int j = i + 1; j < nsamp; ++j) out[2 * j] = 0; return; }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32896, K = argc > 2 ? atoi(argv[2]) : 1600, reps = argc > 3 ? atoi(argv[3]) : 6;
    dmk_ctx *ctx;
    if (dmk_init(0, nullptr, &ctx)) return 1;
    double *X, *Y, *C;
    CK(hipMalloc(&X, (size_t)K * N * 8)); CK(hipMalloc(&Y, (size_t)K * N * 8)); CK(hipMalloc(&C, (size_t)N * N * 8));
    CK(hipMemset(X, 0, (size_t)K * N * 8)); CK(hipMemset(Y, 0, (size_t)K * N * 8)); CK(hipMemset(C, 0, (size_t)N * N * 8));
    {   // non-trivial operands (power draw depends on toggling bits)
        std::vector<double> h((size_t)K * 4096);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
        for (int c = 0; c + 4096 <= N; c += 4096) {
            CK(hipMemcpy2D(X + c, (size_t)N * 8, h.data(), 4096 * 8, 4096 * 8, K, hipMemcpyHostToDevice));
            CK(hipMemcpy2D(Y + c, (size_t)N * 8, h.data(), 4096 * 8, 4096 * 8, K, hipMemcpyHostToDevice));
        }
    }
    const int nsamp = 20000;
    unsigned long long *d_s;
    int *d_stop;
    CK(hipMalloc(&d_s, nsamp * 16));
    CK(hipHostMalloc(&d_stop, 4, hipHostMallocMapped));
    *d_stop = 0;
    hipStream_t ss;
    CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    hipLaunchKernelGGL(sampler, dim3(1), dim3(64), 0, ss, d_s, nsamp, 5000ull /* 50 us */, d_stop);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // idle phase 100 ms, then the load
    hipDeviceptr_t dummy; (void)dummy;
    struct timespec ts = {0, 100000000}; nanosleep(&ts, nullptr);
    CK(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r)
        if (dmk_dgemm_tn_acc(ctx, N, K, 1.0, X, Y, N, C, N)) { printf("dgemm failed: %s\n", dmk_last_error(ctx)); return 1; }
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    nanosleep(&ts, nullptr);
    *d_stop = 1;
    CK(hipStreamSynchronize(ss));
    std::vector<unsigned long long> h(2 * nsamp);
    CK(hipMemcpy(h.data(), d_s, nsamp * 16, hipMemcpyDeviceToHost));
    printf("rect dgemm N %d K %d x %d: %.2f ms each = %.1f TFLOP/s\n", N, K, reps, ms / reps, 2.0 * K * (double)N * N * reps / (ms * 1e-3) / 1e12);
    // clock per 5 ms window
    int n = 0;
    while (n < nsamp && h[2 * n]) ++n;
    const int win = 100;
    for (int i = 0; i + win < n; i += win) {
        const double dclk = (double)(h[2 * (i + win)] - h[2 * i]), drt = (double)(h[2 * (i + win) + 1] - h[2 * i + 1]);
        printf("t = %7.1f ms   s_memtime rate %.1f MHz\n", (h[2 * i + 1] - h[1]) / 1e5, dclk / drt * 100.0);
    }
    return 0;
}
