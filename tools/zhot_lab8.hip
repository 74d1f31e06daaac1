// Production-shaped ablation lab: ONE launch of step 1 and of step 2 over 8 queued AO blocks x 2 spins (C5 shapes), compile
// with -DZHOT_ABL=<mask> (bits: 1 no LDS-DMA after the prologue, 2 no epilogue, 4 no barrier, 8 no 3M operand sums,
// 16 no LDS fragment reads).  Results are wrong for any non-zero mask; timing only.
#include "../libdmet_preview_amd/csrc/zhot.hip"
#include <vector>
int dmk_fail(dmk_ctx *ctx, int code, const char *fmt, ...) { return code; }
FamScope::FamScope(dmk_ctx *c, int f) : ctx(c), fam(f) {}
FamScope::~FamScope() {}
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
int main() {
    const int nao = 200, naux = 800, nemb = 256, nslot = 8, nspin = 2, nk = 4;
    const long long npair = (long long)nemb * (nemb + 1) / 2;
    const size_t blk = (size_t)naux * nao * nao, slot_elems = (size_t)naux * nao * nemb;
    dmk_ctx ctx;
    double2 *ring, *C, *Ut; double *planes;
    CK(hipMalloc(&ring, 16ull * blk * nslot)); CK(hipMalloc(&C, 16ull * nspin * nk * nao * nemb));
    CK(hipMalloc(&Ut, 16ull * slot_elems * nslot * nspin)); CK(hipMalloc(&planes, 8ull * nspin * 2 * naux * npair));
    std::vector<double> h(2ull * nspin * nk * nao * nemb);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
    CK(hipMemcpy(C, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
    CK(hipMemset(ring, 0x3c, 16ull * blk * nslot)); CK(hipMemset(planes, 0, 8ull * nspin * 2 * naux * npair));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ki[8] = {0, 1, 2, 3, 0, 1, 2, 3};
    const void *cj[8]; int sy[8];
    for (int i = 0; i < 8; ++i) { cj[i] = C + (size_t)ki[7 - i] * nao * nemb; sy[i] = 1; }
    for (int which = 1; which <= 2; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 1) launch_half1_hot_multi(&ctx, ring, (long long)blk, nslot, ki, C, Ut, (long long)slot_elems, naux, nao, nemb, nspin,
                                                   (long long)nk * nao * nemb, (long long)nslot * slot_elems);
            else launch_half2_hot(&ctx, Ut, (long long)slot_elems, nslot, cj, sy, planes, naux, npair, naux, nao, nemb, nspin,
                                  (long long)nslot * slot_elems, (long long)nk * nao * nemb, 2LL * naux * npair);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double fl = (which == 1 ? 8.0 * naux * nao * (double)nao * nemb : 8.0 * naux * nao * (double)nemb * nemb) * nslot * nspin;
        printf("ABL=%2d step %d: %8.4f ms per launch  %.2f TF (algorithmic)\n", ZHOT_ABL, which, best, fl / (best * 1e-3) / 1e12);
    }
    return 0;
}
