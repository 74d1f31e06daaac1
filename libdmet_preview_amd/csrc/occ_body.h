// Device-side bodies of the occupation assignment (row a4; see occ.hip for the algorithm notes): shared by the stand-alone
// kernels of occ.hip and by the fused small-lattice mean-field kernel of small.hip.  Every function expects a workgroup of
// OCC_NT threads and is called by ALL of them (they contain barriers).
#pragma once
#include "common.h"
#include <cmath>

namespace {

constexpr int OCC_NT = 1024;

__device__ __forceinline__ unsigned long long occ_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double occ_val(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// block-wide sums over OCC_NT threads; every thread receives the total (fixed combination order)
__device__ double block_sum_f64(double v, double *sh) {
    v = dmk_wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < OCC_NT / 64; ++w) t += sh[w];
    return t;
}
__device__ long long block_sum_i64(long long v, long long *sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    long long t = 0;
#pragma unroll
    for (int w = 0; w < OCC_NT / 64; ++w) t += sh[w];
    return t;
}

// number of levels that are NaN or +-Inf: such a spectrum has no order statistics (the reference fails at its sort / index
// step); the kernels report status 2 instead of ranking the bit patterns
__device__ long long count_nonfinite(const double *__restrict__ e, long long n, long long *sh) {
    long long c = 0;
    for (long long i = threadIdx.x; i < n; i += OCC_NT) c += (fabs(e[i]) <= 1.7976931348623157e308) ? 0 : 1;
    return block_sum_i64(c, sh);
}

constexpr int OCC_SMALL = 2048;      // spectra up to this size are ranked directly in LDS (model lattices: 12 - 150 levels)

// element of rank k (0-based) in ascending order
__device__ double kth_smallest(const double *__restrict__ e, long long n, long long k, long long *sh) {
    if (n <= OCC_SMALL) {
        // small spectra: every thread counts the elements ordered before its own (ties by index, like a stable sort) and
        // the one whose count is k publishes itself -- two barriers instead of the 128 of the bit-pattern bisection
        __shared__ double es[OCC_SMALL];
        __shared__ double found;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += OCC_NT) es[i] = e[i];
        if (threadIdx.x == 0) found = __longlong_as_double(0x7ff8000000000000ll);   // no thread matches (k out of range): NaN, never stale LDS
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += OCC_NT) {
            const double x = es[i];
            int before = 0;
            for (int j = 0; j < n; ++j) before += (es[j] < x || (es[j] == x && j < i)) ? 1 : 0;
            if (before == k) found = x;
        }
        __syncthreads();
        return found;
    }
    // Large spectra (C5: 86 400 levels): radix select on the order-preserving bit pattern, ELEVEN bits per pass -- a 2048-bin
    // histogram in LDS of the keys that agree with the digits found so far, a workgroup scan to the bin that holds rank k.  Six
    // passes over the keys instead of the 64 of a bit-by-bit bisection (1.84 ms at C5, round 4); exact like it: the result is the
    // very element a stable sort puts at rank k.
    __shared__ unsigned hist[2048];
    __shared__ unsigned wave_tot[OCC_NT / 64];
    __shared__ unsigned long long pick_prefix;
    __shared__ long long pick_k;
    unsigned long long prefix = 0;
    int decided = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    while (decided < 64) {
        const int bits = (64 - decided) >= 11 ? 11 : (64 - decided);
        const int shift = 64 - decided - bits, nbins = 1 << bits;
        __syncthreads();
        for (int b = tid; b < 2048; b += OCC_NT) hist[b] = 0u;
        __syncthreads();
        for (long long i = tid; i < n; i += OCC_NT) {
            const unsigned long long key = occ_key(e[i]);
            if (decided == 0 || (key >> (64 - decided)) == prefix) atomicAdd(&hist[(unsigned)((key >> shift) & (unsigned long long)(nbins - 1))], 1u);
        }
        __syncthreads();
        // thread t owns bins 2t, 2t + 1; inclusive scan of the per-thread counts over the workgroup
        const unsigned c0 = hist[2 * tid], c1 = hist[2 * tid + 1];
        const unsigned mine = c0 + c1;
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned before = 0u;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        const long long excl = (long long)before + (long long)(incl - mine);
        if (mine > 0u && k >= excl && k < excl + (long long)mine) {       // exactly one thread: counts are non-negative, k < total
            const bool first = k < excl + (long long)c0;
            pick_prefix = (prefix << bits) | (unsigned long long)(2 * tid + (first ? 0 : 1));
            pick_k = first ? k - excl : k - excl - (long long)c0;
        }
        __syncthreads();
        prefix = pick_prefix;
        k = pick_k;
        decided += bits;
    }
    return occ_val(prefix);
}

__device__ __forceinline__ double fermi(double e, double mu, double beta) {
    const double de = beta * (e - mu);
    return de < 100.0 ? 1.0 / (exp(de) + 1.0) : 0.0;            // the reference's cut-off (ftsystem.py:43)
}

struct OccArgs {
    const double *ew;
    long long n;
    double nelec, beta, mu0, thr, tol;
    int has_mu0, fix_mu;
    int sorted;         // the levels are in ascending order: rank k is e[k], no order-statistics search
    double *occ;
    double *out;        // [0] mu, [1] nerr, [2] electrons spread over the window, [3] levels in the window, [4] status
};

__device__ void occ_zero_t_body(const OccArgs &g) {
    __shared__ long long shi[OCC_NT / 64];
    const double *e = g.ew;
    const long long n = g.n, ne = (long long)g.nelec;
    if (count_nonfinite(e, n, shi) != 0) {
        if (threadIdx.x == 0) { g.out[0] = g.out[1] = g.out[2] = g.out[3] = 0.0; g.out[4] = 2.0; }
        return;
    }
    double mu = g.mu0;
    bool keep = false;
    if (g.has_mu0) {
        long long below = 0, upto = 0;
        for (long long i = threadIdx.x; i < n; i += OCC_NT) {
            below += e[i] < g.mu0 - g.thr ? 1 : 0;
            upto += e[i] <= g.mu0 + g.thr ? 1 : 0;
        }
        below = block_sum_i64(below, shi);
        upto = block_sum_i64(upto, shi);
        keep = below <= ne && upto >= ne;
    }
    if (!keep) {
        // ranks ne - 1 and ne of the ascending order; rank -1 wraps to the largest level like the host indexing does
        const long long klo = ne > 0 ? ne - 1 : n - 1, khi = ne < n ? ne : n - 1;
        const double lo = g.sorted ? e[klo] : kth_smallest(e, n, klo, shi);
        const double hi = g.sorted ? e[khi] : kth_smallest(e, n, khi, shi);
        mu = 0.5 * (lo + hi);
    }
    long long filled = 0, window = 0;
    for (long long i = threadIdx.x; i < n; i += OCC_NT) {
        filled += e[i] < mu - g.thr ? 1 : 0;
        window += (e[i] <= mu + g.thr && e[i] >= mu - g.thr) ? 1 : 0;
    }
    filled = block_sum_i64(filled, shi);
    window = block_sum_i64(window, shi);
    const long long remain = ne - filled;
    const double share = (remain > 0 && window > 0) ? (double)remain / (double)window : 0.0;
    for (long long i = threadIdx.x; i < n; i += OCC_NT) {
        double o = e[i] < mu - g.thr ? 1.0 : 0.0;
        if (remain > 0 && e[i] <= mu + g.thr && e[i] >= mu - g.thr) o += share;
        g.occ[i] = o;
    }
    if (threadIdx.x == 0) {
        g.out[0] = mu;
        g.out[1] = 0.0;
        g.out[2] = remain > 0 ? (double)remain : 0.0;
        g.out[3] = remain > 0 ? (double)window : 0.0;
        g.out[4] = 0.0;
    }
}

__device__ void occ_fermi_body(const OccArgs &g) {
    __shared__ long long shi[OCC_NT / 64];
    __shared__ double shd[OCC_NT / 64];
    const double *e = g.ew;
    const long long n = g.n;
    const double beta = g.beta, target = g.nelec;
    if (count_nonfinite(e, n, shi) != 0) {
        if (threadIdx.x == 0) { g.out[0] = g.out[1] = g.out[2] = g.out[3] = 0.0; g.out[4] = 2.0; }
        return;
    }
    double mu = g.mu0, status = 0.0;

    auto count = [&](double x, double &slope) {       // N(x) - target and dN/dx
        double s = 0.0, d = 0.0;
        for (long long i = threadIdx.x; i < n; i += OCC_NT) {
            const double f = fermi(e[i], x, beta);
            s += f;
            d += f * (1.0 - f);
        }
        s = block_sum_f64(s, shd);
        slope = beta * block_sum_f64(d, shd);
        return s - target;
    };

    if (!g.fix_mu) {
        const long long ni = llrint(target);
        const long long rlo = ni - 1 < 0 ? 0 : (ni - 1 > n - 1 ? n - 1 : ni - 1), rhi = ni < 0 ? 0 : (ni > n - 1 ? n - 1 : ni);
        const double width = 1.0 / beta;
        double lo = kth_smallest(e, n, rlo, shi) - width;
        double hi = kth_smallest(e, n, rhi, shi) + width;
        double dummy, flo = count(lo, dummy), fhi = count(hi, dummy);
        double grow = fmax(width, 1.0);
        for (int it = 0; it < 80 && flo > 0.0; ++it) { lo -= grow; grow *= 2.0; flo = count(lo, dummy); }
        grow = fmax(width, 1.0);
        for (int it = 0; it < 80 && fhi < 0.0; ++it) { hi += grow; grow *= 2.0; fhi = count(hi, dummy); }
        if (flo > 0.0 || fhi < 0.0) status = 1.0;          // no sign change: nelec outside (0, n)
        double x = 0.5 * (lo + hi);
        for (int it = 0; it < 200 && status == 0.0; ++it) {
            double slope;
            const double f = count(x, slope);
            if (f == 0.0) break;
            if (f < 0.0) lo = x; else hi = x;
            double xn = slope > 0.0 ? x - f / slope : 0.5 * (lo + hi);
            if (!(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
            const double step = fabs(xn - x);
            x = xn;
            // Newton converges quadratically: once a step is below the tolerance the error is far below it
            if (step <= 0.25 * g.tol * (1.0 + fabs(x)) || hi - lo <= 4.0e-16 * (1.0 + fabs(x))) break;
        }
        mu = x;
    }
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += OCC_NT) {
        const double f = fermi(e[i], mu, beta);
        g.occ[i] = f;
        s += f;
    }
    s = block_sum_f64(s, shd);
    if (threadIdx.x == 0) {
        g.out[0] = mu;
        g.out[1] = fabs(s - target);
        g.out[2] = 0.0;
        g.out[3] = 0.0;
        g.out[4] = status;
    }
}

}  // namespace
