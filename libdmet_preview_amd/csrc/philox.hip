// K10 -- procedural density-fitted AO block generator (synthetic configs C4/C5).
//
// Stands in for the HDF5 reads of `sr_loop` (basis_transform/eri_transform.py:195-227):
// config C5 would need 6 TB of AO blocks, so a block L^{(ki,kj)} is regenerated on
// the device every time it is visited.  Counter-based Philox4x32-10 (Salmon et al.,
// SC'11): key = (seed_lo, seed_hi), counter = (c_lo, c_hi, ki, kj) with c = e >> 1,
// e = (L*nao + p)*nao + q; words (0,1) -> element 2c, words (2,3) -> element 2c+1,
// value = (u32 * 2^-31 - 1) / sqrt(nao) for re and im.  Bit-identical to
// oracle/restate.py:df_block_philox (u32*2^-31 - 1 is exact in f64; the scale is one
// rounding).  HBM-write bound: 16 B per element.
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                             uint32_t k0, uint32_t k1) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}

__global__ __launch_bounds__(256) void philox_block_kernel(uint32_t seed_lo, uint32_t seed_hi, uint32_t ki,
                                                           uint32_t kj, long long nelem, double scale,
                                                           double2 *__restrict__ out) {
    const long long npairs = (nelem + 1) >> 1;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < npairs;
         c += (long long)gridDim.x * blockDim.x) {
        uint32_t c0 = (uint32_t)(c & 0xffffffffLL), c1 = (uint32_t)((unsigned long long)c >> 32);
        uint32_t c2 = ki, c3 = kj;
        uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const double s31 = 4.656612873077392578125e-10;   // 2^-31
        const double2 e0 = make_double2(((double)c0 * s31 - 1.0) * scale, ((double)c1 * s31 - 1.0) * scale);
        const double2 e1 = make_double2(((double)c2 * s31 - 1.0) * scale, ((double)c3 * s31 - 1.0) * scale);
        const long long e = c << 1;
        out[e] = e0;
        if (e + 1 < nelem) out[e + 1] = e1;
    }
}

// several blocks in ONE launch (blockIdx.y = block): at C4 a 72 MB block is 11 us of HBM writes behind ~6 us of launch and
// ramp-up -- the 12 - 16 blocks of a step-1 group as one launch write at the streaming rate of the C5 blocks
constexpr int PHILOX_MAXBATCH = 16;
struct PhiloxBatch {
    uint32_t seed_lo, seed_hi;
    uint32_t ki[PHILOX_MAXBATCH], kj[PHILOX_MAXBATCH];
    long long nelem, stride;         // elements per block, distance between consecutive blocks in double2 elements
    double scale;
    double2 *out;
};
__global__ __launch_bounds__(256) void philox_blocks_kernel(const PhiloxBatch g) {
    const int b = blockIdx.y;
    uint32_t ki = g.ki[0], kj = g.kj[0];          // constant-index picks (no scratch copy of the argument arrays)
#pragma unroll
    for (int i = 1; i < PHILOX_MAXBATCH; ++i)
        if (b == i) { ki = g.ki[i]; kj = g.kj[i]; }
    double2 *__restrict__ out = g.out + (long long)b * g.stride;
    const long long nelem = g.nelem, npairs = (nelem + 1) >> 1;
    const double scale = g.scale;
    for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < npairs;
         c += (long long)gridDim.x * blockDim.x) {
        uint32_t c0 = (uint32_t)(c & 0xffffffffLL), c1 = (uint32_t)((unsigned long long)c >> 32);
        uint32_t c2 = ki, c3 = kj;
        uint32_t k0 = g.seed_lo, k1 = g.seed_hi;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const double s31 = 4.656612873077392578125e-10;   // 2^-31
        const double2 e0 = make_double2(((double)c0 * s31 - 1.0) * scale, ((double)c1 * s31 - 1.0) * scale);
        const double2 e1 = make_double2(((double)c2 * s31 - 1.0) * scale, ((double)c3 * s31 - 1.0) * scale);
        const long long e = c << 1;
        out[e] = e0;
        if (e + 1 < nelem) out[e + 1] = e1;
    }
}

}  // namespace

int launch_philox_blocks_on(dmk_ctx *ctx, hipStream_t stream, uint64_t seed, int nblk, const int *ij, int naux, int nao, void *out,
                            long long stride_bytes) {
    const long long nelem = (long long)naux * nao * nao;
    if (nelem <= 0 || nblk <= 0) return DMK_OK;
    const double scale = 1.0 / sqrt((double)nao);
    for (int b0 = 0; b0 < nblk; b0 += PHILOX_MAXBATCH) {
        const int m = std::min(PHILOX_MAXBATCH, nblk - b0);
        PhiloxBatch g;
        g.seed_lo = (uint32_t)(seed & 0xffffffffu); g.seed_hi = (uint32_t)(seed >> 32);
        for (int i = 0; i < PHILOX_MAXBATCH; ++i) {
            g.ki[i] = (uint32_t)ij[2 * (b0 + (i < m ? i : 0))];
            g.kj[i] = (uint32_t)ij[2 * (b0 + (i < m ? i : 0)) + 1];
        }
        g.nelem = nelem; g.stride = stride_bytes / (long long)sizeof(double2); g.scale = scale;
        g.out = reinterpret_cast<double2 *>(static_cast<char *>(out) + (long long)b0 * stride_bytes);
        const long long npairs = (nelem + 1) >> 1;
        long long blocks = (npairs + 255) / 256;
        const long long cap = std::max<long long>(256LL * 32 / m, 512);
        if (blocks > cap) blocks = cap;
        FamScope fs(ctx, DMK_FAM_PHILOX, stream);
        hipLaunchKernelGGL(philox_blocks_kernel, dim3((unsigned)blocks, (unsigned)m), dim3(256), 0, stream, g);
        DMK_CHECK_LAUNCH(ctx);
    }
    return DMK_OK;
}

int launch_philox_block(dmk_ctx *ctx, uint64_t seed, int ki, int kj, int naux, int nao, void *out) {
    return launch_philox_block_on(ctx, ctx->stream, seed, ki, kj, naux, nao, out);
}

// the same launch on a stream of the caller's choice (the producer stream of the ERI pipeline's block ring)
int launch_philox_block_on(dmk_ctx *ctx, hipStream_t stream, uint64_t seed, int ki, int kj, int naux, int nao, void *out) {
    const long long nelem = (long long)naux * nao * nao;
    if (nelem <= 0) return DMK_OK;
    const long long npairs = (nelem + 1) >> 1;
    long long blocks = (npairs + 255) / 256;
    if (blocks > 256LL * 32) blocks = 256LL * 32;
    const double scale = 1.0 / sqrt((double)nao);
    FamScope fs(ctx, DMK_FAM_PHILOX, stream);
    hipLaunchKernelGGL(philox_block_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), (uint32_t)ki, (uint32_t)kj, nelem,
                       scale, reinterpret_cast<double2 *>(out));
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}
