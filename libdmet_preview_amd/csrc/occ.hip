// a4 -- Fermi level and occupation numbers of a mean field on the device.
//
// What the reference computes on the host (routine/mfd.py:887-957 `assignocc`, routine/ftsystem.py:24-105
// `fermi_smearing_occ` / `find_mu`): a stable sort of all spin * nk * nlo eigenvalues, the chemical potential
// (T = 0: the mid-point of the two frontier order statistics unless the caller's mu0 already separates nelec levels
// within the degeneracy window thr_deg; T > 0: the root of sum_i f(e_i; mu) = nelec), and the occupation of every
// level (T = 0: step function with the remaining electrons spread evenly over the window [mu - thr, mu + thr]).
//
// Here none of it needs a sort or a trip over PCIe:
//   * an order statistic is found by a 64-step bisection on the bit pattern of the doubles (IEEE-754 doubles order
//     like sign-flipped unsigned integers): each step counts the keys below a candidate prefix -- exact, and it returns
//     the very element a stable sort would put at that rank, so the T = 0 mid-point is bit-identical to the host value;
//   * the finite-temperature root uses a bracketed Newton iteration on N(mu) = sum_i f_i with the analytic slope
//     beta sum_i f_i (1 - f_i), falling back to a bisection step whenever Newton leaves the bracket; the bracket is
//     grown geometrically from the frontier levels, N is strictly increasing so the root is unique;
//   * all sums run in ONE workgroup with a fixed tree order (deterministic, no atomics).  The eigenvalues of C5 are
//     86 400 doubles (0.7 MB, L2 resident): a pass is ~1 us of loads per step, the whole assignment well under 1 ms
//     against 5 ms for the D2H + numpy mergesort + H2D it replaces.
#include "common.h"
#include <cmath>

#include "occ_body.h"

namespace {

__global__ __launch_bounds__(OCC_NT) void occ_zero_t_kernel(const OccArgs g) { occ_zero_t_body(g); }

// several independent spectra (the spin channels of an embedding problem) in ONE launch: workgroup b takes spectrum b
constexpr int OCC_MAXBATCH = 8;
struct OccBatchArgs {
    OccArgs base;                         // ew / occ / out of spectrum 0; spectrum b at + b * n (ew, occ) and + 8 b (out)
    double nelec[OCC_MAXBATCH], mu0[OCC_MAXBATCH];
};
__global__ __launch_bounds__(OCC_NT) void occ_zero_t_batch_kernel(const OccBatchArgs gb) {
    OccArgs g = gb.base;
    const int b = blockIdx.x;
    g.ew += (long long)b * g.n;
    g.occ += (long long)b * g.n;
    g.out += 8 * b;
    // constant-index picks: a dynamically indexed kernel-argument array would be copied to scratch
    double ne = gb.nelec[0], m0 = gb.mu0[0];
#pragma unroll
    for (int i = 1; i < OCC_MAXBATCH; ++i)
        if (b == i) { ne = gb.nelec[i]; m0 = gb.mu0[i]; }
    g.nelec = ne;
    g.mu0 = m0;
    occ_zero_t_body(g);
}

__global__ __launch_bounds__(OCC_NT) void occ_fermi_kernel(const OccArgs g) { occ_fermi_body(g); }

}  // namespace

// T = 0 occupations of `batch` spectra of n levels each (ew, occ: [batch][n]) in one launch, nothing read back; info_dev
// [batch][8] receives (mu, nerr, spread, window, status) per spectrum.  Internal (csrc/fit.hip: dmk_fit_objective).
int dmk_assign_occ_zero_t_batch(dmk_ctx *ctx, int64_t n, int batch, const double *ew, const double *nelec_host,
                                const double *mu0_host, int flags, double thr_deg, double *occ, double *info_dev) {
    if (!ctx || n <= 0 || batch < 1 || batch > OCC_MAXBATCH || !ew || !occ || !nelec_host || !info_dev) return DMK_ERR_INVALID;
    OccBatchArgs a;
    a.base.ew = ew; a.base.n = n; a.base.nelec = 0.0; a.base.beta = INFINITY; a.base.mu0 = 0.0; a.base.thr = thr_deg;
    a.base.tol = 1e-12;
    a.base.has_mu0 = (flags & 1) ? 1 : 0;
    a.base.fix_mu = (flags & 2) ? 1 : 0;
    a.base.sorted = (flags & 4) ? 1 : 0;
    a.base.occ = occ; a.base.out = info_dev;
    for (int i = 0; i < OCC_MAXBATCH; ++i) {
        a.nelec[i] = nelec_host[i < batch ? i : 0];
        a.mu0[i] = mu0_host ? mu0_host[i < batch ? i : 0] : 0.0;
        const double ne = a.nelec[i];
        if (ne < 0.0 || ne > (double)n || ne != std::floor(ne))
            return dmk_fail(ctx, DMK_ERR_INVALID, "assign_occ: T = 0 needs an integer 0 <= nelec <= %lld levels", (long long)n);
    }
    FamScope fs(ctx, DMK_FAM_MISC);
    hipLaunchKernelGGL(occ_zero_t_batch_kernel, dim3(batch), dim3(OCC_NT), 0, ctx->stream, a);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

extern "C" int dmk_assign_occ(dmk_ctx *ctx, int64_t n, const double *ew, double nelec, double beta, double mu0, int flags,
                              double thr_deg, double fit_tol, double *occ, double *info_host) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n <= 0 || !ew || !occ) return dmk_fail(ctx, DMK_ERR_INVALID, "assign_occ: bad arguments");
    const bool zero_t = !(beta < INFINITY);
    if (!zero_t && !(beta > 0.0)) return dmk_fail(ctx, DMK_ERR_INVALID, "assign_occ: beta must be positive");
    if (!zero_t && !(flags & 2) && !(nelec >= 0.0 && nelec <= (double)n))
        return dmk_fail(ctx, DMK_ERR_INVALID, "assign_occ: %g electrons do not fit %lld levels", nelec, (long long)n);
    if (zero_t && (nelec < 0.0 || nelec > (double)n || nelec != std::floor(nelec)))
        return dmk_fail(ctx, DMK_ERR_INVALID, "assign_occ: T = 0 needs an integer 0 <= nelec <= %lld levels", (long long)n);
    void *scratch;
    int rc = dmk_scratch(ctx, 8 * sizeof(double), &scratch);
    if (rc) return rc;
    OccArgs a;
    a.ew = ew; a.n = n; a.nelec = nelec; a.beta = beta; a.mu0 = mu0; a.thr = thr_deg;
    a.tol = fit_tol > 0.0 ? fit_tol : 1e-12;
    a.has_mu0 = (flags & 1) ? 1 : 0;
    a.fix_mu = (flags & 2) ? 1 : 0;
    a.sorted = (flags & 4) ? 1 : 0;
    a.occ = occ; a.out = reinterpret_cast<double *>(scratch);
    {
        FamScope fs(ctx, DMK_FAM_MISC);
        if (zero_t) hipLaunchKernelGGL(occ_zero_t_kernel, dim3(1), dim3(OCC_NT), 0, ctx->stream, a);
        else hipLaunchKernelGGL(occ_fermi_kernel, dim3(1), dim3(OCC_NT), 0, ctx->stream, a);
        DMK_CHECK_LAUNCH(ctx);
    }
    if (!info_host) return DMK_OK;          // asynchronous use: occupations only, nothing is read back
    DMK_HIP(ctx, hipMemcpyAsync(info_host, scratch, 5 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (info_host[4] == 2.0) return dmk_fail(ctx, DMK_ERR_INVALID, "assign_occ: the eigenvalue list contains NaN / Inf");
    if (info_host[4] != 0.0) return dmk_fail(ctx, DMK_ERR_INVALID, "assign_occ: no chemical potential gives %g electrons", nelec);
    return DMK_OK;
}
