// LAB: why does the k -> R fold of the small-lattice kernel (csrc/small.hip) cost ~0.5-0.8 us per term?  Same LDS layout and loop as
// the product, one 1024-thread workgroup, mesh 6 x 6 x 1, nn = 16; variants remove one ingredient at a time.  Shader cycles per term
// from s_memtime of the slowest wave.
//   hipcc -O3 --offload-arch=gfx950 tools/fold_small_lab.hip -o tools/fold_small_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VAR>
__global__ __launch_bounds__(1024) void fold(int n0, int n1, int n2, int nn, int spin, double *out, unsigned long long *clk) {
    extern __shared__ double dyn[];
    const int nk = n0 * n1 * n2, tid = threadIdx.x;
    double2 *rl = reinterpret_cast<double2 *>(dyn);
    double2 *tw = rl + spin * nk * nn;
    for (int t = tid; t < 3 * 128; t += 1024) { double s, c; sincospi(2.0 * (t % 128) / 7.0, &s, &c); tw[t] = double2{c, s}; }
    for (int t = tid; t < spin * nk * nn; t += 1024) rl[t] = double2{0.001 * t, 1.0 - 0.002 * t};
    __syncthreads();
    const unsigned long long c0 = clock64();
    for (int o = tid; o < spin * nk * nn; o += 1024) {
        const int ij = o % nn, R = (o / nn) % nk, s = o / (nn * nk);
        const int r0 = R / (n1 * n2), r1 = (R / n2) % n1, r2 = R % n2;
        double re = 0.0, im = 0.0;
        const double2 *v = rl + (size_t)s * nk * nn + ij;
        int i0 = 0, i1 = 0, i2 = 0, c1 = 0, c2 = 0;
#pragma unroll 4
        for (int k = 0; k < nk; ++k) {
            double2 w0, w1, w2, x;
            if (VAR == 1) { w0 = double2{0.6, 0.8}; w1 = double2{0.8, 0.6}; w2 = double2{1.0, 0.0}; } else { w0 = tw[i0]; w1 = tw[128 + i1]; w2 = tw[256 + i2]; }
            if (VAR == 2) x = double2{0.5, 0.25}; else x = v[(size_t)k * nn];
            if (VAR == 3) { re += w0.x + w1.x + w2.x + x.x; im += w0.y + w1.y + w2.y + x.y; }
            else {
                const double ar = w0.x * w1.x - w0.y * w1.y, ai = w0.x * w1.y + w0.y * w1.x;
                const double pr = ar * w2.x - ai * w2.y, pi = ar * w2.y + ai * w2.x;
                re += pr * x.x - pi * x.y;
                im += pr * x.y + pi * x.x;
            }
            if (VAR != 4) {
                ++c2;
                i2 += r2; if (i2 >= n2) i2 -= n2;
                if (c2 == n2) {
                    c2 = 0; i2 = 0; ++c1;
                    i1 += r1; if (i1 >= n1) i1 -= n1;
                    if (c1 == n1) { c1 = 0; i1 = 0; i0 += r0; if (i0 >= n0) i0 -= n0; }
                }
            } else { i0 = (i0 + r0) & 7; i1 = (i1 + r1) & 7; }
        }
        out[o] = re + im;
    }
    const unsigned long long c1 = clock64();
    __syncthreads();
    const unsigned long long c2 = clock64();
    if ((tid & 63) == 0) clk[tid >> 6] = c1 - c0;
    if (tid == 0) clk[16] = c2 - c0;
}

int main() {
    double *out; unsigned long long *clk, h[17];
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&clk, 17 * 8));
    const int n0 = 6, n1 = 6, n2 = 1, nn = 16, spin = 1, nk = n0 * n1 * n2;
    const size_t lds = (size_t)spin * nk * nn * 16 + 3 * 128 * 16;
    const char *names[] = {"product loop", "twiddles constant (no LDS reads of tw)", "x constant (no LDS read of rho_k)", "no complex products (reads only)", "power-of-two index advance (no counters / wraps)"};
    for (int var = 0; var < 5; ++var)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(clk, 0, 17 * 8));
            switch (var) {
                case 0: hipLaunchKernelGGL(fold<0>, dim3(1), dim3(1024), lds, 0, n0, n1, n2, nn, spin, out, clk); break;
                case 1: hipLaunchKernelGGL(fold<1>, dim3(1), dim3(1024), lds, 0, n0, n1, n2, nn, spin, out, clk); break;
                case 2: hipLaunchKernelGGL(fold<2>, dim3(1), dim3(1024), lds, 0, n0, n1, n2, nn, spin, out, clk); break;
                case 3: hipLaunchKernelGGL(fold<3>, dim3(1), dim3(1024), lds, 0, n0, n1, n2, nn, spin, out, clk); break;
                case 4: hipLaunchKernelGGL(fold<4>, dim3(1), dim3(1024), lds, 0, n0, n1, n2, nn, spin, out, clk); break;
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
            if (rep == 0 || rep == 2) {
                unsigned long long mx = 0; for (int w = 0; w < 9; ++w) mx = h[w] > mx ? h[w] : mx;
                printf("%s %-52s wave 0 %6llu cycles, slowest of the 9 active waves %6llu (%.0f per term), to the barrier %6llu = %.2f us at 2.39 GHz\n", rep == 0 ? "COLD (first launch)" : "warm (third launch)", names[var], h[0], mx,
                       (double)mx / nk, h[16], h[16] / 2390.0);
            }
        }
    return 0;
}
