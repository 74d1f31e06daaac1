// How does the f64 MFMA issue rate depend on the number of live accumulator tiles per wave?  (2 waves per SIMD, asm-pinned)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template <int NA, int NB, int MODE>
__global__ __launch_bounds__(256, 2) void k(double *out, int iters, double a0, double b0) {
    d4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = d4{0, 0, 0, 0};
    double a[NA], b[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = a0 + threadIdx.x * 1e-9 + i * 1e-3;
#pragma unroll
    for (int j = 0; j < NB; ++j) b[j] = b0 - j * 1e-3;
    if (MODE == 2) {
        // operands with random mantissas and signs (what real data looks like to the multiplier array): does the
        // sustained rate depend on the data (power management)?
        unsigned long long h = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
        auto rnd = [&]() {
            h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
            return __longlong_as_double((long long)((h & 0x800FFFFFFFFFFFFFull) | 0x3FE0000000000000ull));   // +-[0.5, 1)
        };
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = rnd();
#pragma unroll
        for (int j = 0; j < NB; ++j) b[j] = rnd();
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (MODE == 0 || MODE == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
                else acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NA, int NB, int MODE>
void run(const char *tag) {
    const int blocks = 512 * 8, iters = 2000 / (NA * NB) * 4;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k<NA, NB, MODE><<<blocks, 256>>>(out, 10, 1.0, 1.0);
    CK(hipDeviceSynchronize());
    double best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k<NA, NB, MODE><<<blocks, 256>>>(out, iters, 1.000001, 0.999999);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double tf = (double)blocks * 4 * iters * NA * NB * 2048.0 / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
    }
    printf("%s NA=%d NB=%d (%d accumulator tiles, %d VGPRs): %.2f TF\n", tag, NA, NB, NA * NB, NA * NB * 8, best);
    CK(hipFree(out));
}

template <int NA, int NB, int MODE>
void sustained(double seconds) {
    // the same kernel launched back to back for `seconds`: does the rate hold under sustained load (power / clock management)?
    const int blocks = 512 * 8, iters = 2000 / (NA * NB) * 40;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double elapsed = 0;
    int n = 0;
    while (elapsed < seconds * 1e3) {
        CK(hipEventRecord(e0));
        k<NA, NB, MODE><<<blocks, 256>>>(out, iters, 1.000001, 0.999999);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        elapsed += ms;
        const double tf = (double)blocks * 4 * iters * NA * NB * 2048.0 / (ms * 1e-3) / 1e12;
        if (n % 8 == 0) printf("sustained NA=%d NB=%d: launch %3d at %7.1f ms: %.1f ms per launch, %.2f TF\n", NA, NB, n, elapsed, ms, tf);
        ++n;
    }
    CK(hipFree(out));
}

int main(int argc, char **argv) {
    if (argc > 2) { sustained<3, 8, 2>(atof(argv[1])); return 0; }
    if (argc > 1) { sustained<3, 8, 0>(atof(argv[1])); return 0; }
    run<4, 4, 0>("asm");
    run<4, 6, 0>("asm");
    run<3, 8, 0>("asm");
    run<3, 9, 0>("asm");
    run<2, 4, 0>("asm");
    run<4, 4, 2>("asm, random operands");
    run<3, 8, 2>("asm, random operands");
    run<4, 4, 1>("intrinsic");
    run<3, 8, 1>("intrinsic");
    run<3, 9, 1>("intrinsic");
    return 0;
}
