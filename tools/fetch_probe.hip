// What does FETCH_SIZE count for the two LDS-DMA source-address forms?  (profiles/traffic_latest.json applies the on-image
// guide's gfx950 correction "FETCH_SIZE counts 1/2 of a wide coalesced 16 B/lane stream" to every kernel; round 4 moved the ring
// kernels from per-lane 64-bit pointers to scalar base + per-lane offset, and the contraction's FETCH_SIZE doubled while its time
// fell.)  Each kernel streams the same 4 GiB buffer ONCE through LDS-DMA (every byte fetched exactly once, far larger than L2 +
// Infinity Cache), so the counter's reading divided by 4 GiB is the factor for that form.
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_probe.hip -o tools/fetch_probe
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/fetch_probe ; python3 tools/rocprof_summary.py <trace.db> <pmc.db>
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) void *)p; }

template <int FORM>      // 0: per-lane 64-bit pointer, 1: scalar base + per-lane 32-bit offset, 2: plain global_load_dwordx4 to registers
__global__ __launch_bounds__(256) void stream_kernel(const double2 *src, size_t n16, double *sink) {
    __shared__ double2 buf[4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per_wg = 256 * 64;                       // elements per workgroup iteration
    double acc = 0.0;
    for (size_t base = (size_t)blockIdx.x * per_wg; base + per_wg <= n16; base += (size_t)gridDim.x * per_wg) {
#pragma unroll 4
        for (int it = 0; it < 64; ++it) {
            const size_t e = base + (size_t)it * 256 + wave * 64;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr(&buf[wave][0]));
            if (FORM == 0) {
                const double2 *p = src + e + lane;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
            } else if (FORM == 1) {
                const double2 *sb = src + e;
                const unsigned off = lane * 16;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off), "s"(sb), "s"(dst) : "memory");
            } else {
                const double2 v = src[e + lane];
                acc += v.x + v.y;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += buf[wave][lane].x;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

int main() {
    const size_t bytes = 4ull << 30, n16 = bytes / 16;
    double2 *src; double *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 64);
    hipMemset(src, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream_kernel<0>, dim3(2048), dim3(256), 0, 0, src, n16, sink);
        hipLaunchKernelGGL(stream_kernel<1>, dim3(2048), dim3(256), 0, 0, src, n16, sink);
        hipLaunchKernelGGL(stream_kernel<2>, dim3(2048), dim3(256), 0, 0, src, n16, sink);
    }
    hipDeviceSynchronize();
    printf("streamed %zu bytes per launch (stream_kernel<0> per-lane pointers, <1> scalar base + offset, <2> register loads)\n", bytes);
    return 0;
}
