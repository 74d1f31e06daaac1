// Ablation lab of the two hot half-transform kernels at C5 shapes: the PRODUCT kernels of
// libdmet_preview_amd/csrc/zhot.hip are compiled into this binary with their LAB template bits, which remove one ingredient
// of the K loop at a time (results are then meaningless -- only the time is looked at):
//     1  no epilogue stores (step 1: Ut; step 2: the tril-pack atomics)      2  no LDS-DMA after the prologue
//     4  no s_barrier
//     3 / 7  combinations: what is left is the MFMA stream, its LDS fragment reads, the 3M operand sums and the loop control
// and the time of every variant is printed next to the flop the launch issues to the matrix pipe (the library's own count).
// Shapes: 8 queued AO blocks x 2 spins, naux 800, nao 200, nemb 256 -- one launch of each step of the C5 bench -- and the
// table-driven step 2 at C4 (16 queued blocks, naux 416, nao 104, nemb 136, one spin).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics tools/zhot_lab.hip -Llibdmet_preview_amd -l:libdmetk.so \
//        -Wl,-rpath,'$ORIGIN/../libdmet_preview_amd' -o tools/zhot_lab            run: tools/zhot_lab
#include "../libdmet_preview_amd/csrc/zhot.hip"
#include "../libdmet_preview_amd/csrc/zhot_tab.hip"
#include "zhot_half1p_lab.h"
#include <cstdio>
#include <vector>

namespace {
__global__ void fill_kernel(double *p, size_t n, double scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z ^= z >> 27;
        p[i] = ((double)(long long)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0) * scale;
    }
}
void *dalloc(size_t bytes, double scale) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc(%zu) failed\n", bytes); exit(1); }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<double *>(p), bytes / 8, scale);
    return p;
}
template <class F> float time_ms(F &&launch, int reps = 4) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();                                           // warm
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}
}  // namespace

int main() {
    const int nL = 800, nao = 200, nemb = 256, nspin = 2, nslot = 8, nk = 16;
    const long long npair = (long long)nemb * (nemb + 1) / 2;
    const size_t blk = (size_t)nL * nao * nao, ut = (size_t)nL * nao * nemb;
    double2 *Lpq = (double2 *)dalloc(blk * nslot * 16, 1.0);
    double2 *C = (double2 *)dalloc((size_t)nspin * nk * nao * nemb * 16, 0.05);
    double2 *Ut = (double2 *)dalloc(ut * nslot * nspin * 16, 0.1);
    double *planes = (double *)dalloc((size_t)nspin * 2 * nL * npair * 8, 0.0);
    hipDeviceSynchronize();

    // ---- step 1: as launch_flat_hot (BM 128, BN 64, both spins, 8 slots) ---------------------------------------------------
    H1Args a;
    a.Lpq = Lpq; a.Ci = C; a.Ut = Ut;
    a.nL = nL; a.nao = nao; a.kdim = nao; a.nemb = nemb; a.mrows = nao; a.nblk = (nao + 15) / 16;
    a.tiles_m = (int)(((long long)nL * nao + 127) / 128); a.tiles_n = nemb / 64;
    a.nspin = nspin; a.b_spin_stride = (long long)nk * nao * nemb; a.out_spin_stride = (long long)nslot * ut;
    a.nslot = nslot; a.a_slot_stride = (long long)blk; a.out_slot_stride = (long long)ut; a.b_k_stride = (long long)nao * nemb;
    for (int i = 0; i < 16; ++i) a.bk[i] = i % nk;
    a.per_slot = (unsigned)(a.tiles_m * a.tiles_n * nspin);
    a.nblocks = a.per_slot * nslot;
    const double f1 = 6.0 * (double)a.nblocks * 128 * 64 * nao;
    printf("step 1 (half1_kernel<conj, 128, 2, wide>): %u workgroups, %.1f GFLOP issued per launch\n", a.nblocks, f1 * 1e-9);
#define RUN1(LABV)                                                                                                            \
    {                                                                                                                           \
        const float ms = time_ms([&] { hipLaunchKernelGGL((half1_kernel<true, 128, 2, false, LABV>), dim3(a.nblocks), dim3(HNT), 0, 0, a); }); \
        printf("   LAB %2d  %8.3f ms  %6.2f TF  (%5.1f %% of 78.6)\n", LABV, ms, f1 / ms * 1e-9, f1 / ms * 1e-9 / 78.6 * 100.0);  \
    }
    RUN1(0) RUN1(1) RUN1(2) RUN1(4) RUN1(3) RUN1(7)
    {
        const float ms = time_ms([&] { hipLaunchKernelGGL((half1nt_kernel<true, 128, 2, false, 0>), dim3(a.nblocks), dim3(HNT), 0, 0, a); }, 8);
        const float ms0 = time_ms([&] { hipLaunchKernelGGL((half1_kernel<true, 128, 2, false, 0>), dim3(a.nblocks), dim3(HNT), 0, 0, a); }, 8);
        printf("   nontemporal Ut stores (lab copy) %8.3f ms  %6.2f TF  (%5.1f %%)   against the product kernel in the same run %8.3f ms (%5.1f %%)\n", ms,
               f1 / ms * 1e-9, f1 / ms * 1e-9 / 78.6 * 100.0, ms0, f1 / ms0 * 1e-9 / 78.6 * 100.0);
    }
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("step 1, persistent (half1p_kernel<conj, 128, 2, wide>): %d workgroups walk the same %u tiles\n", 2 * ncu, a.nblocks);
#define RUN1P(LABV)                                                                                                           \
    {                                                                                                                           \
        const float ms = time_ms([&] { hipLaunchKernelGGL((half1p_kernel<true, 128, 2, false, LABV>), dim3(2 * ncu), dim3(HNT), 0, 0, a); }); \
        printf("   LAB %2d  %8.3f ms  %6.2f TF  (%5.1f %% of 78.6)\n", LABV, ms, f1 / ms * 1e-9, f1 / ms * 1e-9 / 78.6 * 100.0);  \
    }
    RUN1P(0) RUN1P(8) RUN1P(16) RUN1P(32) RUN1P(64) RUN1P(1) RUN1P(2) RUN1P(4) RUN1P(3) RUN1P(7)
    {   // ---- step 1 at C4: 16 queued blocks, naux 416, nao 104, nemb 136, one spin, 48-column tiles ---------------------------
        const int nL4 = 416, nao4 = 104, nemb4 = 136, nslot4 = 16;
        H1Args c = a;
        c.nL = nL4; c.nao = nao4; c.kdim = nao4; c.nemb = nemb4; c.mrows = nao4; c.nblk = (nao4 + 15) / 16;
        c.tiles_m = (int)(((long long)nL4 * nao4 + 127) / 128); c.tiles_n = (nemb4 + 47) / 48;
        c.nspin = 1; c.b_spin_stride = 0; c.out_spin_stride = 0;
        c.nslot = nslot4; c.a_slot_stride = (long long)nL4 * nao4 * nao4; c.out_slot_stride = (long long)nL4 * nao4 * nemb4;
        c.b_k_stride = (long long)nao4 * nemb4;
        c.per_slot = (unsigned)(c.tiles_m * c.tiles_n);
        c.nblocks = c.per_slot * nslot4;
        const double f4 = 6.0 * (double)c.nblocks * 128 * 48 * nao4;
        printf("step 1 at C4 (<conj, 128, 2, narrow>): %u tiles, %.1f GFLOP issued per launch\n", c.nblocks, f4 * 1e-9);
#define RUN4(KERNEL, GRID, LABV, TAG)                                                                                         \
        {                                                                                                                       \
            const float ms = time_ms([&] { hipLaunchKernelGGL((KERNEL<true, 128, 2, true, LABV>), dim3(GRID), dim3(HNT), 0, 0, c); }); \
            printf("   %-10s LAB %2d  %8.3f ms  %6.2f TF  (%5.1f %% of 78.6)\n", TAG, LABV, ms, f4 / ms * 1e-9, f4 / ms * 1e-9 / 78.6 * 100.0); \
        }
        RUN4(half1_kernel, c.nblocks, 0, "per tile") RUN4(half1_kernel, c.nblocks, 1, "per tile") RUN4(half1_kernel, c.nblocks, 2, "per tile")
        RUN4(half1_kernel, c.nblocks, 7, "per tile")
        RUN4(half1p_kernel, 2 * ncu, 0, "persistent") RUN4(half1p_kernel, 2 * ncu, 8, "persistent") RUN4(half1p_kernel, 2 * ncu, 16, "persistent") RUN4(half1p_kernel, 2 * ncu, 32, "persistent") RUN4(half1p_kernel, 2 * ncu, 64, "persistent") RUN4(half1p_kernel, 2 * ncu, 1, "persistent") RUN4(half1p_kernel, 2 * ncu, 2, "persistent")
        RUN4(half1p_kernel, 2 * ncu, 7, "persistent")
        {   // 64 x 48 tiles, three workgroups per CU
            H1Args d = c;
            d.tiles_m = (int)(((long long)nL4 * nao4 + 63) / 64);
            d.per_slot = (unsigned)(d.tiles_m * d.tiles_n);
            d.nblocks = d.per_slot * nslot4;
            const double f5 = 6.0 * (double)d.nblocks * 64 * 48 * nao4;
            for (int rep = 0; rep < 2; ++rep) {
                const float ms = time_ms([&] { hipLaunchKernelGGL((half1_kernel<true, 64, 3, true, 0>), dim3(d.nblocks), dim3(HNT), 0, 0, d); });
                printf("   64x48 occ3 LAB  0  %8.3f ms  %6.2f TF  (%5.1f %% of 78.6)\n", ms, f5 / ms * 1e-9, f5 / ms * 1e-9 / 78.6 * 100.0);
            }
            const float ms7 = time_ms([&] { hipLaunchKernelGGL((half1_kernel<true, 64, 3, true, 7>), dim3(d.nblocks), dim3(HNT), 0, 0, d); });
            printf("   64x48 occ3 LAB  7  %8.3f ms  %6.2f TF  (%5.1f %% of 78.6)\n", ms7, f5 / ms7 * 1e-9, f5 / ms7 * 1e-9 / 78.6 * 100.0);
        }
    }

    // ---- step 2: as launch_half2_hot (all blocks symmetrised -> folded diagonal blocks) ------------------------------------
    H2Args h;
    h.Ut = Ut; h.symmask = (1u << nslot) - 1u;
    for (int i = 0; i < H2_MAXSLOT; ++i) h.Cj[i] = C + (size_t)((i * 3 + 1) % nk) * nao * nemb;
    h.slot_stride = (long long)ut; h.planes = planes; h.naux = nL; h.npair = npair;
    h.nL = nL; h.nao = nao; h.kdim = nao; h.nslot = nslot; h.nspin = nspin;
    h.ut_spin_stride = (long long)nslot * ut; h.cj_spin_stride = (long long)nk * nao * nemb; h.planes_spin_stride = 2LL * nL * npair;
    h.nblocks = (unsigned)(4 * nL * nspin); h.fold_diag = 1;
    const double f2 = 6.0 * (double)nslot * (136.0 + 120.0) * 256.0 * nao * nL * nspin;
    printf("step 2 (half2_kernel): %u workgroups, %.1f GFLOP issued per launch\n", h.nblocks, f2 * 1e-9);
#define RUN2(LABV)                                                                                                            \
    {                                                                                                                           \
        const float ms = time_ms([&] { hipLaunchKernelGGL(half2_kernel<LABV>, dim3(h.nblocks), dim3(HNT), 0, 0, h); });          \
        printf("   LAB %2d  %8.3f ms  %6.2f TF  (%5.1f %% of 78.6)\n", LABV, ms, f2 / ms * 1e-9, f2 / ms * 1e-9 / 78.6 * 100.0);  \
    }
    RUN2(0) RUN2(1) RUN2(2) RUN2(4) RUN2(3) RUN2(7)

    // ---- table-driven step 2 at C4: as launch_half2_tab (three workgroups per CU, all blocks symmetrised) --------------------
    {
        const int nL4 = 416, nao4 = 104, nemb4 = 136, nslot4 = 16;
        const long long npair4 = (long long)nemb4 * (nemb4 + 1) / 2;
        const size_t ut4 = (size_t)nL4 * nao4 * nemb4;
        std::vector<int> tab;
        double useful, slots, folded;
        build_table(nemb4, Cfg3::MAXBLK, Cfg3::SEG, tab, useful, slots, folded);
        int *dtab = nullptr;
        hipMalloc(&dtab, tab.size() * sizeof(int));
        hipMemcpy(dtab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice);
        H2TArgs t;
        t.Ut = Ut; t.symmask = 0xffffu;
        for (int i = 0; i < T_MAXSLOT; ++i) t.Cj[i] = C + (size_t)((i * 3 + 1) % nk) * nao4 * nemb4;
        t.slot_stride = (long long)ut4; t.planes = planes; t.naux = nL4; t.npair = npair4;
        t.nL = nL4; t.nao = nao4; t.kdim = nao4; t.nslot = nslot4; t.nemb = nemb4; t.nspin = 1;
        t.ut_spin_stride = 0; t.cj_spin_stride = 0; t.planes_spin_stride = 0; t.fold_diag = 1;
        t.table = dtab; t.nitems = (int)(tab.size() / T_ITEM);
        t.nsub = 1; t.sub_slots = nslot4; t.planes_sub = nullptr; t.sub_stride = 0;
        t.per_sub = (unsigned)(t.nitems * nL4);
        t.nblocks = t.per_sub;
        const double f3 = 6.0 * ((double)nslot4 * useful + (double)nslot4 * (useful - folded)) * 256.0 * nao4 * nL4;
        printf("step 2, table-driven at C4 (half2_tab_kernel<Cfg3>): %u workgroups of %d items, %.1f GFLOP issued per launch, "
               "%.0f useful block slots of %.0f\n", t.nblocks, t.nitems, f3 * 1e-9, useful, slots);
#define RUN3(LABV)                                                                                                            \
        {                                                                                                                       \
            const float ms = time_ms([&] { hipLaunchKernelGGL((half2_tab_kernel<Cfg3, LABV>), dim3(t.nblocks), dim3(HNT), 0, 0, t); }); \
            printf("   LAB %2d  %8.3f ms  %6.2f TF  (%5.1f %% of 78.6)\n", LABV, ms, f3 / ms * 1e-9, f3 / ms * 1e-9 / 78.6 * 100.0); \
        }
        RUN3(0) RUN3(1) RUN3(2) RUN3(4) RUN3(3) RUN3(7)
        // the same kernel with 2x / 4x / 8x the auxiliary rows per launch: what fusing the step-2 launches of several kL would give
        // (1248 workgroups on 768 resident slots = 1.6 rounds -> 3.25 / 6.5 / 13)
        for (int mult = 2; mult <= 8; mult *= 2) {
            H2TArgs u = t;
            u.nL = nL4 * mult; u.naux = u.nL;
            u.per_sub = (unsigned)(u.nitems * u.nL); u.nblocks = u.per_sub;
            if ((size_t)u.nL * nao4 * nemb4 * nslot4 > ut * nslot * nspin || (size_t)2 * u.nL * npair4 > (size_t)nspin * 2 * nL * npair) break;
            u.slot_stride = (long long)u.nL * nao4 * nemb4;
            const double fm = f3 * mult;
            const float ms0 = time_ms([&] { hipLaunchKernelGGL((half2_tab_kernel<Cfg3, 0>), dim3(u.nblocks), dim3(HNT), 0, 0, u); });
            const float ms7 = time_ms([&] { hipLaunchKernelGGL((half2_tab_kernel<Cfg3, 7>), dim3(u.nblocks), dim3(HNT), 0, 0, u); });
            printf("   x%d rows (%u workgroups)  LAB 0 %8.3f ms %6.2f TF (%5.1f %%)   LAB 7 %8.3f ms %6.2f TF (%5.1f %%)\n", mult, u.nblocks,
                   ms0, fm / ms0 * 1e-9, fm / ms0 * 1e-9 / 78.6 * 100.0, ms7, fm / ms7 * 1e-9, fm / ms7 * 1e-9 / 78.6 * 100.0);
        }
    }
    return 0;
}
