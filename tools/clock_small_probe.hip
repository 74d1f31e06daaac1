// LAB: which shader clock does a ONE-workgroup latency-bound kernel run at when it is launched over and over with a host
// synchronisation in between (the shape of the model-lattice step, csrc/small.hip)?  The kernel walks a dependent chain of LDS
// reads (pointer chasing, one wave active) and reads s_memtime (shader clock) and s_memrealtime (100 MHz) around it.
//   hipcc -O3 --offload-arch=gfx950 tools/clock_small_probe.hip -o tools/clock_small_probe && tools/clock_small_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void chase(int hops, unsigned long long *out, int slot) {
    __shared__ int nxt[1024];
    const int tid = threadIdx.x;
    nxt[tid] = (tid * 37 + 11) & 1023;
    __syncthreads();
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    int p = tid;
    for (int i = 0; i < hops; ++i) p = nxt[p];
    __syncthreads();
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    if (tid == 0) { out[3 * slot] = c1 - c0; out[3 * slot + 1] = r1 - r0; out[3 * slot + 2] = (unsigned long long)p; }
}
__global__ void burn(double *x, int iters) {          // every CU busy for a while: what a preceding heavy kernel does to the clock
    double a = x[threadIdx.x & 63], b = 1.0000001;
    for (int i = 0; i < iters; ++i) a = fma(a, b, 1e-9);
    x[(blockIdx.x * blockDim.x + threadIdx.x) & 63] = a;
}

int main() {
    const int reps = 400, hops = 400;
    unsigned long long *d;
    double *x;
    CK(hipMalloc(&d, reps * 24)); CK(hipMalloc(&x, 512)); CK(hipMemset(x, 0, 512));
    std::vector<unsigned long long> h(reps * 3);
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: launch + synchronise; mode 1: back to back, one synchronise at the end; mode 2: a chip-filling kernel before each
        for (int i = 0; i < reps; ++i) {
            if (mode == 2) hipLaunchKernelGGL(burn, dim3(2048), dim3(256), 0, 0, x, 20000);
            hipLaunchKernelGGL(chase, dim3(1), dim3(1024), 0, 0, hops, d, i);
            if (mode == 0) CK(hipDeviceSynchronize());
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, reps * 24, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %d LDS hops per launch\n", mode, mode == 0 ? "launch + sync" : mode == 1 ? "back to back" : "after a chip-filling kernel", hops);
        for (int i : {0, 1, 2, 5, 10, 50, 100, 200, 399}) {
            const double us = h[3 * i + 1] * 0.01, mhz = us > 0 ? h[3 * i] / us : 0.0;
            printf("   launch %3d: %7.2f us, shader clock %6.0f MHz, %.1f shader cycles per hop\n", i, us, mhz, (double)h[3 * i] / hops);
        }
    }
    return 0;
}
