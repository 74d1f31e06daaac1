// Can a SIMD of gfx950 run f64 MFMAs and f64 vector FMAs AT THE SAME TIME?  Both pipes are specified at 78.6 TF on MI355X; if the
// matrix pipe and the vector ALU were independent units, a kernel that splits a contraction between them could exceed the
// matrix-pipe roofline the ERI kernels are priced against.  Three experiments, all register-only (no memory traffic):
//   same wave   : per iteration NM MFMAs (16x16x4 f64, independent accumulators) interleaved with NV independent v_fma_f64
//   other waves : workgroups of 512 threads (2 waves per SIMD), even waves MFMA only, odd waves v_fma only
//   references  : MFMA only, v_fma only
// Reported: TF of each pipe and their sum, against MFMA-only.  (round 6 lab; result in DESIGN.md section 9)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

constexpr int NACC = 8;          // MFMA accumulator tiles per wave
constexpr int NFMA = 16;         // independent vector accumulators per lane

// role: 0 = this wave does both (NM MFMAs then NV FMAs per iteration, the compiler may not reorder the asm volatile statements,
// the hardware may overlap them), 1 = MFMA only, 2 = FMA only
template <int NM, int NV>
__device__ __forceinline__ void body(int iters, d4 (&acc)[NACC], double (&x)[NFMA], double a, double b, double c, double d) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[(r * NM + m) % NACC]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v)
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[(r * NV + v) % NFMA]) : "v"(c), "v"(d));
        }
    }
}

template <int NM, int NV, int SPLIT>
__global__ __launch_bounds__(512, 1) void k(double *out, int iters, double a0, double b0) {
    d4 acc[NACC];
    double x[NFMA];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NFMA; ++i) x[i] = 1e-3 * i;
    const double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9, c = 1.0 - 1e-9 * (threadIdx.x & 7), d = 1e-12;
    const int wave = threadIdx.x >> 6;
    if (SPLIT == 0) body<NM, NV>(iters, acc, x, a, b, c, d);
    else if ((wave >> 2) == 0) body<NM, 0>(iters, acc, x, a, b, c, d);      // waves 0-3: one per SIMD, MFMA only
    else body<0, NV>(iters, acc, x, a, b, c, d);                             // waves 4-7: the second wave of each SIMD, FMA only
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < NFMA; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NV, int SPLIT>
void run(const char *tag) {
    const int blocks = 256 * 8, iters = 4000;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * blocks * 512));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k<NM, NV, SPLIT><<<blocks, 512>>>(out, 10, 1.0, 1.0);
    CK(hipDeviceSynchronize());
    double best_ms = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k<NM, NV, SPLIT><<<blocks, 512>>>(out, iters, 1.000001, 0.999999);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_ms) best_ms = ms;
    }
    const double waves_m = SPLIT ? 4.0 : 8.0, waves_v = SPLIT ? 4.0 : 8.0;
    const double fm = (double)blocks * waves_m * iters * 4.0 * NM * 2048.0;
    const double fv = (double)blocks * waves_v * iters * 4.0 * NV * 128.0;
    const double s = best_ms * 1e-3;
    printf("%-34s NM=%d NV=%2d  %8.3f ms   mfma %6.2f TF   vfma %6.2f TF   sum %6.2f TF\n", tag, NM, NV, best_ms, fm / s / 1e12, fv / s / 1e12,
           (fm + fv) / s / 1e12);
    CK(hipFree(out));
}

int main() {
    printf("f64 matrix pipe vs f64 vector ALU, register-only, 2 waves per SIMD\n");
    run<2, 0, 0>("mfma only");
    run<0, 16, 0>("vfma only");
    run<1, 2, 0>("same wave");
    run<1, 4, 0>("same wave");
    run<1, 8, 0>("same wave");
    run<1, 12, 0>("same wave");
    run<1, 16, 0>("same wave");
    run<2, 16, 0>("same wave");
    run<2, 0, 1>("split waves (mfma half only)");
    run<0, 16, 1>("split waves (vfma half only)");
    run<2, 8, 1>("split waves");
    run<2, 16, 1>("split waves");
    run<2, 32, 1>("split waves");
    return 0;
}
