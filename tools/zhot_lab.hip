// Ablation lab for the hot half-transform kernels: compile with -DZHOT_ABL=<mask>.
#include "../libdmet_preview_amd/csrc/zhot.hip"
#include <vector>
// minimal stand-ins for the library internals zhot.hip references
int dmk_fail(dmk_ctx *ctx, int code, const char *fmt, ...) { return code; }
FamScope::FamScope(dmk_ctx *c, int f) : ctx(c), fam(f) {}
FamScope::~FamScope() {}
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
int main() {
    const int nao = 200, naux = 800, nemb = 256;
    const long long npair = (long long)nemb * (nemb + 1) / 2;
    dmk_ctx ctx;
    double2 *Lpq, *C, *Ut; double *planes;
    CK(hipMalloc(&Lpq, 16ull * naux * nao * nao)); CK(hipMalloc(&C, 16ull * nao * nemb));
    CK(hipMalloc(&Ut, 16ull * naux * nao * nemb)); CK(hipMalloc(&planes, 8ull * 2 * naux * npair));
    std::vector<double> h(2ull * nao * nemb);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
    CK(hipMemcpy(C, h.data(), 8 * h.size(), hipMemcpyHostToDevice));
    CK(hipMemset(Lpq, 0x3c, 16ull * naux * nao * nao)); CK(hipMemset(planes, 0, 8ull * 2 * naux * npair));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int which = 1; which <= 2; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 1) launch_half1_hot(&ctx, Lpq, C, Ut, naux, nao, nemb);
            else { const void *cj[1] = {C}; int sy[1] = {1}; launch_half2_hot(&ctx, Ut, 0, 1, cj, sy, planes, naux, npair, naux, nao, nemb, 1, 0, 0, 0); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double fl = which == 1 ? 8.0 * naux * nao * (double)nao * nemb : 8.0 * naux * nao * (double)nemb * nemb;
        printf("ABL=%d half%d: %.4f ms  %.2f TF (algorithmic)\n", ZHOT_ABL, which, best, fl / (best * 1e-3) / 1e12);
    }
    return 0;
}
