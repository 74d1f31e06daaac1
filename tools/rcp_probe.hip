// Issue cost of the f64 reciprocal variants used by the Sturm-count / twisted-factorisation recurrences (eigh_tripairs.hip):
//   A: v_rcp_f64 + two Newton steps            (what eigh_kernel and tri_eigpairs_kernel used through round 4's first version)
//   B: v_rcp_f32 seed on the mantissa (frexp / ldexp around it) + two Newton steps in f64
//   C: v_rcp_f32 seed directly on (float) q + two Newton steps (no range protection)
//   D: plain IEEE division q_new = a - b / q  (what hipcc emits for '/')
// One dependent chain per lane (the recurrences are sequential in the row index), 1 / 2 / 4 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/rcp_probe.hip -o tools/rcp_probe ; run: tools/rcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double rcpA(double q) {
    double r = __builtin_amdgcn_rcp(q);
    r = r * (2.0 - q * r);
    return r * (2.0 - q * r);
}
__device__ __forceinline__ double rcpB(double q) {
    const int ex = __builtin_amdgcn_frexp_exp(q);
    const double m = __builtin_amdgcn_frexp_mant(q);
    double r = (double)__builtin_amdgcn_rcpf((float)m);
    r = __builtin_amdgcn_ldexp(r, -ex);
    r = r * (2.0 - q * r);
    return r * (2.0 - q * r);
}
__device__ __forceinline__ double rcpC(double q) {
    double r = (double)__builtin_amdgcn_rcpf((float)q);
    r = r * (2.0 - q * r);
    return r * (2.0 - q * r);
}

template <int V>
__global__ void probe(int steps, const double *__restrict__ a, const double *__restrict__ b, double *out) {
    double q = a[threadIdx.x & 63] + 1.0;
    int cnt = 0;
    for (int s = 0; s < steps; ++s) {
        const double ai = a[s & 255], bi = b[s & 255];
        double r;
        if (V == 0) r = rcpA(q);
        else if (V == 1) r = rcpB(q);
        else if (V == 2) r = rcpC(q);
        else r = 1.0 / q;
        q = ai - bi * r;
        if (fabs(q) < 1e-300) q = -1e-300;
        cnt += q < 0.0 ? 1 : 0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = q + cnt;
}

int main() {
    double *a, *b, *out;
    hipMalloc(&a, 256 * 8); hipMalloc(&b, 256 * 8); hipMalloc(&out, 256 * 1024 * 8);
    double ha[256], hb[256];
    for (int i = 0; i < 256; ++i) { ha[i] = 0.3 + 0.01 * i; hb[i] = 0.05 + 0.001 * i; }
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    const int steps = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"A v_rcp_f64 + 2 Newton", "B rcp_f32(mantissa) + ldexp + 2 Newton", "C rcp_f32 + 2 Newton", "D IEEE division"};
    for (int wps = 1; wps <= 4; wps *= 2) {                     // waves per SIMD: 256 CUs x 4 SIMDs x wps waves
        const int threads = 256, blocks = 256 * wps;
        for (int v = 0; v < 4; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (v == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(threads), 0, 0, steps, a, b, out);
                if (v == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(threads), 0, 0, steps, a, b, out);
                if (v == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(threads), 0, 0, steps, a, b, out);
                if (v == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(threads), 0, 0, steps, a, b, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%d wave(s)/SIMD  %-42s %8.3f ms  = %6.1f cycles per step per SIMD (2.4 GHz)\n", wps, names[v], ms,
                   ms * 1e-3 * 2.4e9 / steps);
        }
    }
    double h[4]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("(checksum %g)\n", h[0] + h[1]);
    return 0;
}
