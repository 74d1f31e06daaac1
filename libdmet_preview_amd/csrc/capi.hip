// C ABI of libdmetk: context, memory, k-mesh bookkeeping (host, integer), folds, ERI pipeline.
// Kernel launchers live in the sibling .hip files; this file holds no device code except
// tiny utility kernels (transpose, restore).
#include "common.h"
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <algorithm>

// =============================================================================================
// context / errors / memory
// =============================================================================================

int dmk_fail(dmk_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else fprintf(stderr, "libdmetk: %s\n", buf);
    return code;
}

static thread_local std::string g_noctx_err;

// `a` is recorded exactly once, on the stream the launch goes to (ctx->stream unless the caller names another one)
FamScope::FamScope(dmk_ctx *c, int f, hipStream_t stream) : ctx(c), fam(f), on(stream) {
    on_set = ctx && stream != ctx->stream;
    begin();
}
FamScope::FamScope(dmk_ctx *c, int f) : ctx(c), fam(f), on(nullptr) { begin(); }
void FamScope::begin() {
    if (ctx && ctx->profile) {
        auto get = [&]() {
            hipEvent_t e;
            if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
            return e;
        };
        a = get(); b = get();
        if (a) (void)hipEventRecord(a, on_set ? on : ctx->stream);
    }
    if (ctx) ctx->fam_launches[fam] += 1;
}
FamScope::~FamScope() {
    if (ctx && ctx->profile && a && b) {
        (void)hipEventRecord(b, on_set ? on : ctx->stream);
        ctx->pending.push_back({fam, a, b});
    }
}

static void drain_pending(dmk_ctx *ctx) {
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess)
            ctx->fam_ms[p.fam] += ms;
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

extern "C" {

const char *dmk_version(void) { return "libdmetk 0.1 (gfx950)"; }

int dmk_init(int device, void *stream, dmk_ctx **out) {
    if (!out) return DMK_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        fprintf(stderr, "libdmetk: no HIP device available (%s)\n", hipGetErrorString(e));
        return DMK_ERR_HIP;
    }
    if (device < 0 || device >= ndev) return DMK_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return DMK_ERR_HIP;
    dmk_ctx *c = new dmk_ctx();
    c->device = device;
    c->stream = reinterpret_cast<hipStream_t>(stream);
    if (hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess) {
        delete c;
        return DMK_ERR_HIP;
    }
    *out = c;
    return DMK_OK;
}

int dmk_destroy(dmk_ctx *ctx) {
    if (!ctx) return DMK_OK;
    (void)hipStreamSynchronize(ctx->stream);
    drain_pending(ctx);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto &p : ctx->phases) (void)hipFree(p.dev);
    for (auto &t : ctx->tile_tables) (void)hipFree(t.dev);
    for (auto &t : ctx->step2_tables) (void)hipFree(t.dev);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->scratch2) (void)hipFree(ctx->scratch2);
    for (int w = 0; w < 3; ++w)
        if (ctx->eri_ws[w]) (void)hipFree(ctx->eri_ws[w]);
    (void)hipEventDestroy(ctx->t0);
    (void)hipEventDestroy(ctx->t1);
    delete ctx;
    return DMK_OK;
}

int dmk_set_stream(dmk_ctx *ctx, void *stream) {
    if (!ctx) return DMK_ERR_INVALID;
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    return DMK_OK;
}

int dmk_mem_info(dmk_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return DMK_ERR_INVALID;
    DMK_HIP(ctx, hipMemGetInfo(free_bytes, total_bytes));
    return DMK_OK;
}

int dmk_set_oom_hook(dmk_ctx *ctx, void (*hook)(void *), void *user) {
    if (!ctx) return DMK_ERR_INVALID;
    ctx->oom_hook = hook;
    ctx->oom_user = user;
    return DMK_OK;
}

int dmk_sync(dmk_ctx *ctx) {
    if (!ctx) return DMK_ERR_INVALID;
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DMK_OK;
}

const char *dmk_last_error(const dmk_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int dmk_malloc(dmk_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return DMK_ERR_INVALID;
    *out = nullptr;
    if (bytes == 0) return DMK_OK;
    hipError_t e = dmk_dev_alloc(ctx, out, bytes);
    if (e != hipSuccess) return dmk_fail(ctx, DMK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return DMK_OK;
}
int dmk_free(dmk_ctx *ctx, void *p) {
    if (!ctx) return DMK_ERR_INVALID;
    if (p) {
        // work enqueued on the context stream may still read the buffer: drain it first
        DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        DMK_HIP(ctx, hipFree(p));
    }
    return DMK_OK;
}
int dmk_memset(dmk_ctx *ctx, void *p, int value, size_t bytes) {
    if (!ctx) return DMK_ERR_INVALID;
    if (bytes) DMK_HIP(ctx, hipMemsetAsync(p, value, bytes, ctx->stream));
    return DMK_OK;
}
int dmk_memcpy_h2d(dmk_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return DMK_ERR_INVALID;
    if (bytes) {
        DMK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return DMK_OK;
}
int dmk_memcpy_d2h(dmk_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return DMK_ERR_INVALID;
    if (bytes) {
        DMK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return DMK_OK;
}
int dmk_memcpy_d2d(dmk_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return DMK_ERR_INVALID;
    if (bytes) DMK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return DMK_OK;
}

int dmk_timer_start(dmk_ctx *ctx) {
    if (!ctx) return DMK_ERR_INVALID;
    DMK_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return DMK_OK;
}
int dmk_timer_stop(dmk_ctx *ctx, double *ms_out) {
    if (!ctx || !ms_out) return DMK_ERR_INVALID;
    DMK_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
    DMK_HIP(ctx, hipEventSynchronize(ctx->t1));
    float ms = 0.f;
    DMK_HIP(ctx, hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    *ms_out = ms;
    return DMK_OK;
}

int dmk_profile(dmk_ctx *ctx, int enable) {
    if (!ctx) return DMK_ERR_INVALID;
    if (!enable && ctx->profile) { (void)hipStreamSynchronize(ctx->stream); drain_pending(ctx); }
    ctx->profile = enable != 0;
    return DMK_OK;
}
int dmk_profile_read(dmk_ctx *ctx, double *ms, int64_t *launches, int reset) {
    if (!ctx) return DMK_ERR_INVALID;
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    drain_pending(ctx);
    for (int i = 0; i < DMK_FAM_COUNT; ++i) {
        if (ms) ms[i] = ctx->fam_ms[i];
        if (launches) launches[i] = ctx->fam_launches[i];
        if (reset) { ctx->fam_ms[i] = 0; ctx->fam_launches[i] = 0; }
    }
    return DMK_OK;
}
int dmk_profile_read_flops(dmk_ctx *ctx, double *flops, int reset) {
    if (!ctx) return DMK_ERR_INVALID;
    for (int i = 0; i < DMK_FAM_COUNT; ++i) {
        if (flops) flops[i] = ctx->fam_mfma_flops[i];
        if (reset) ctx->fam_mfma_flops[i] = 0;
    }
    return DMK_OK;
}

}  // extern "C"

hipError_t dmk_dev_alloc(dmk_ctx *ctx, void **out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess || !ctx) return e;
    bool parked = false;
    for (int w = 0; w < 3; ++w) parked = parked || ctx->eri_ws[w] != nullptr;
    if (!parked && !ctx->oom_hook) return e;
    (void)hipGetLastError();
    (void)hipStreamSynchronize(ctx->stream);      // parked blocks may still be read by queued work
    // the workspaces the last ERI pipeline left in the context (plane stack: tens of GB) are only a cache
    for (int w = 0; w < 3; ++w)
        if (ctx->eri_ws[w]) {
            (void)hipFree(ctx->eri_ws[w]);
            ctx->eri_ws[w] = nullptr;
            ctx->eri_ws_bytes[w] = 0;
        }
    if (ctx->oom_hook) ctx->oom_hook(ctx->oom_user);
    e = hipMalloc(out, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}

int dmk_scratch(dmk_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            DMK_HIP(ctx, hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        hipError_t e = dmk_dev_alloc(ctx, &ctx->scratch, bytes);
        if (e != hipSuccess) return dmk_fail(ctx, DMK_ERR_NOMEM, "scratch hipMalloc(%zu) failed", bytes);
        ctx->scratch_bytes = bytes;
    }
    *out = ctx->scratch;
    return DMK_OK;
}

int dmk_scratch2(dmk_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch2_bytes) {
        if (ctx->scratch2) {
            DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            DMK_HIP(ctx, hipFree(ctx->scratch2));
            ctx->scratch2 = nullptr;
            ctx->scratch2_bytes = 0;
        }
        hipError_t e = dmk_dev_alloc(ctx, &ctx->scratch2, bytes);
        if (e != hipSuccess) return dmk_fail(ctx, DMK_ERR_NOMEM, "scratch2 hipMalloc(%zu) failed", bytes);
        ctx->scratch2_bytes = bytes;
    }
    *out = ctx->scratch2;
    return DMK_OK;
}

// =============================================================================================
// a1 / a2 / a15 : integer mesh bookkeeping (host)
// =============================================================================================

namespace {

struct Mesh {
    int n[3];
    int nk;
    explicit Mesh(const int m[3]) { n[0] = m[0]; n[1] = m[1]; n[2] = m[2]; nk = m[0] * m[1] * m[2]; }
    bool ok() const { return n[0] > 0 && n[1] > 0 && n[2] > 0 && (long long)n[0] * n[1] * n[2] < (1LL << 24); }
    void ints(int idx, int a[3]) const {
        a[2] = idx % n[2];
        a[1] = (idx / n[2]) % n[1];
        a[0] = idx / (n[2] * n[1]);
    }
    int index(const int a[3]) const { return (a[0] * n[1] + a[1]) * n[2] + a[2]; }
    static int mod(int x, int m) { int r = x % m; return r < 0 ? r + m : r; }
    int combine(int i, int j, int sign) const {   // idx(a_i + sign*a_j)
        int a[3], b[3], c[3];
        ints(i, a); ints(j, b);
        for (int d = 0; d < 3; ++d) c[d] = mod(a[d] + sign * b[d], n[d]);
        return index(c);
    }
    int minus(int i) const {
        int a[3], c[3];
        ints(i, a);
        for (int d = 0; d < 3; ++d) c[d] = mod(-a[d], n[d]);
        return index(c);
    }
    // fftfreq integer of mesh index a on an axis of length n
    static int freq(int a, int n) { return a <= (n - 1) / 2 ? a : a - n; }
};

void tr_weights(const Mesh &m, int tr, std::vector<int> &w) {
    w.assign(m.nk, 1);
    if (!tr) return;
    for (int i = 0; i < m.nk; ++i) {
        const int mi = m.minus(i);
        w[i] = (mi == i) ? 1 : (mi > i ? 2 : 0);
    }
}

}  // namespace

extern "C" {

int dmk_kmesh_tables(const int mesh[3], int32_t *kint, int32_t *minus_k, int32_t *weights) {
    Mesh m(mesh);
    if (!m.ok()) return DMK_ERR_INVALID;
    std::vector<int> w;
    tr_weights(m, 1, w);
    for (int i = 0; i < m.nk; ++i) {
        if (kint) { int a[3]; m.ints(i, a); kint[3 * i] = a[0]; kint[3 * i + 1] = a[1]; kint[3 * i + 2] = a[2]; }
        if (minus_k) minus_k[i] = m.minus(i);
        if (weights) weights[i] = w[i];
    }
    return DMK_OK;
}

int dmk_kconserv_table(const int mesh[3], int32_t *out) {
    Mesh m(mesh);
    if (!m.ok() || !out) return DMK_ERR_INVALID;
    for (int kL = 0; kL < m.nk; ++kL)
        for (int i = 0; i < m.nk; ++i) out[(size_t)kL * m.nk + i] = m.combine(i, kL, -1);
    return DMK_OK;
}

int dmk_cell_add_table(const int mesh[3], int sign, int32_t *out) {
    Mesh m(mesh);
    if (!m.ok() || !out || (sign != 1 && sign != -1)) return DMK_ERR_INVALID;
    for (int i = 0; i < m.nk; ++i)
        for (int j = 0; j < m.nk; ++j) out[(size_t)i * m.nk + j] = m.combine(i, j, sign);
    return DMK_OK;
}

int dmk_kpts_scaled(const int mesh[3], double *kpts) {
    Mesh m(mesh);
    if (!m.ok() || !kpts) return DMK_ERR_INVALID;
    for (int i = 0; i < m.nk; ++i) {
        int a[3];
        m.ints(i, a);
        for (int d = 0; d < 3; ++d) {
            const double val = 1.0 / ((double)m.n[d] * 1.0);      // numpy fftfreq: results * (1/(n*d))
            kpts[3 * i + d] = (double)Mesh::freq(a[d], m.n[d]) * val;
        }
    }
    return DMK_OK;
}

int dmk_kpt_member(const int mesh[3], const double kpt[3], double tol) {
    Mesh m(mesh);
    if (!m.ok() || !kpt) return DMK_ERR_INVALID;
    std::vector<double> ks((size_t)3 * m.nk);
    dmk_kpts_scaled(mesh, ks.data());
    for (int i = 0; i < m.nk; ++i) {
        double s = 0.0;
        for (int d = 0; d < 3; ++d) {
            double dk = ks[3 * i + d] - kpt[d];
            dk -= nearbyint(dk);
            s += dk * dk;
        }
        if (sqrt(s) < tol) return i;
    }
    return -1;
}

// basis_transform/eri_transform.py:1409-1427 (get_mask_kptij_lst) on mesh indices: pair p' = (-ki, -kj) of pair p.
// mask[p] = index of the time-reversed partner (first later pair), -2 for a pair already claimed, -1 otherwise.
int dmk_kptij_mask(const int mesh[3], int npairs, const int32_t *pairs, int32_t *mask) {
    Mesh m(mesh);
    if (!m.ok() || npairs < 0 || (npairs > 0 && (!pairs || !mask))) return DMK_ERR_INVALID;
    for (int p = 0; p < npairs; ++p) {
        if (pairs[2 * p] < 0 || pairs[2 * p] >= m.nk || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= m.nk) return DMK_ERR_INVALID;
        mask[p] = -1;
    }
    for (int i = 0; i < npairs; ++i) {
        if (mask[i] != -1) continue;
        const int na = m.minus(pairs[2 * i]), nb = m.minus(pairs[2 * i + 1]);
        for (int j = i + 1; j < npairs; ++j) {
            if (pairs[2 * j] == na && pairs[2 * j + 1] == nb) {
                mask[i] = j;
                mask[j] = -2;
                break;
            }
        }
    }
    return DMK_OK;
}

int dmk_eri_plan(const int mesh[3], int tr, int32_t *plan, int64_t capacity, int64_t *nrec) {
    Mesh m(mesh);
    if (!m.ok() || !nrec) return DMK_ERR_INVALID;
    std::vector<int> w;
    tr_weights(m, tr, w);
    std::vector<char> visited(m.nk);
    int64_t n = 0;
    for (int kL = 0; kL < m.nk; ++kL) {
        if (w[kL] <= 0) continue;
        std::fill(visited.begin(), visited.end(), 0);
        for (int i = 0; i < m.nk; ++i) {
            if (visited[i]) continue;
            visited[i] = 1;
            const int j = m.combine(i, kL, -1);
            int jm = -1, sym = 0;
            if (tr) {
                jm = m.minus(j);
                sym = visited[jm] ? 0 : 1;
            }
            if (plan) {
                if (n >= capacity) return DMK_ERR_INVALID;
                int32_t *r = plan + 5 * n;
                r[0] = kL; r[1] = i; r[2] = j; r[3] = jm; r[4] = sym;
            }
            ++n;
            if (tr) visited[jm] = 1;
        }
    }
    *nrec = n;
    return DMK_OK;
}

int dmk_assign_workload(const int mesh[3], int tr, int nranks, int rank, int32_t *kl, int *n_out) {
    Mesh m(mesh);
    if (!m.ok() || nranks <= 0 || rank < 0 || rank >= nranks || !n_out) return DMK_ERR_INVALID;
    std::vector<int> w, idx1, idx2;
    tr_weights(m, tr, w);
    for (int i = 0; i < m.nk; ++i) {
        if (w[i] == 1) idx1.push_back(i);
        else if (w[i] == 2) idx2.push_back(i);
    }
    const int nibz = (int)(idx1.size() + idx2.size());
    const int neach = nibz / nranks, extras = nibz % nranks;
    std::vector<std::vector<int>> kids(nranks);
    for (size_t i = 0; i < idx1.size(); ++i) kids[i % nranks].push_back(idx1[i]);
    size_t start = 0;
    for (int r = 0; r < nranks; ++r) {
        const int ns = neach + (r < extras ? 1 : 0);
        long long want = (long long)ns - (long long)kids[r].size();
        // python slice semantics of idx_2[start:end] (end may fall below start -> empty)
        long long end = (long long)start + want;
        long long s = std::min<long long>((long long)start, (long long)idx2.size());
        long long e2 = std::min<long long>(std::max<long long>(end, 0), (long long)idx2.size());
        for (long long t = s; t < e2; ++t) kids[r].push_back(idx2[(size_t)t]);
        start = (size_t)std::max<long long>(end, 0);
    }
    *n_out = (int)kids[rank].size();
    if (kl) for (size_t t = 0; t < kids[rank].size(); ++t) kl[t] = kids[rank][t];
    return DMK_OK;
}

}  // extern "C"

// =============================================================================================
// a6 : folds -- full meshes through fold.hip (fused mixed-radix pass); the DFT as a complex GEMM against a cached twiddle
//      matrix below serves k subsets (multi-rank partial folds), axes longer than 16 and DMK_FOLD_FFT=0
// =============================================================================================

namespace {

// P[r][k] = exp(-2 pi i sum_d a_d(k) a_d(r) / n_d), symmetric in (r, k); rows optionally
// restricted to `subset`.
int get_phase(dmk_ctx *ctx, const Mesh &m, const int32_t *subset, int nsub, void **dev) {
    for (auto &p : ctx->phases) {
        if (p.mesh[0] == m.n[0] && p.mesh[1] == m.n[1] && p.mesh[2] == m.n[2] && p.nsub == nsub &&
            (nsub == 0 || std::equal(p.subset.begin(), p.subset.end(), subset))) {
            *dev = p.dev;
            return DMK_OK;
        }
    }
    // common denominator D = lcm(n0, n1, n2)
    auto gcd = [](long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return a; };
    long long D = m.n[0];
    D = D / gcd(D, m.n[1]) * m.n[1];
    D = D / gcd(D, m.n[2]) * m.n[2];
    const int rows = nsub > 0 ? nsub : m.nk;
    std::vector<double> tw((size_t)2 * D);
    for (long long t = 0; t < D; ++t) {
        const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)t / (long double)D;
        tw[2 * t] = (double)cosl(ang);
        tw[2 * t + 1] = (double)sinl(ang);
    }
    // exact values on the axes
    for (long long t = 0; t < D; ++t) {
        if ((4 * t) % D == 0) {
            const int q = (int)((4 * t) / D);   // angle = -q*pi/2
            const double c[4] = {1, 0, -1, 0}, s[4] = {0, -1, 0, 1};
            tw[2 * t] = c[q]; tw[2 * t + 1] = s[q];
        }
    }
    std::vector<double> host((size_t)2 * rows * m.nk);
    for (int rr = 0; rr < rows; ++rr) {
        const int r = nsub > 0 ? subset[rr] : rr;
        if (r < 0 || r >= m.nk) return dmk_fail(ctx, DMK_ERR_INVALID, "fold: k subset index out of range");
        int a[3];
        m.ints(r, a);
        for (int k = 0; k < m.nk; ++k) {
            int b[3];
            m.ints(k, b);
            long long t = 0;
            for (int d = 0; d < 3; ++d) t += (long long)a[d] * b[d] % m.n[d] * (D / m.n[d]);
            t %= D;
            host[2 * ((size_t)rr * m.nk + k)] = tw[2 * t];
            host[2 * ((size_t)rr * m.nk + k) + 1] = tw[2 * t + 1];
        }
    }
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, host.size() * sizeof(double));
    if (e != hipSuccess) return dmk_fail(ctx, DMK_ERR_NOMEM, "fold: phase allocation failed");
    DMK_HIP(ctx, hipMemcpyAsync(d, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    dmk_ctx::Phase ph;
    ph.mesh[0] = m.n[0]; ph.mesh[1] = m.n[1]; ph.mesh[2] = m.n[2];
    ph.dir = 0; ph.nsub = nsub;
    if (nsub > 0) ph.subset.assign(subset, subset + nsub);
    ph.dev = d;
    ctx->phases.push_back(ph);
    *dev = d;
    return DMK_OK;
}

}  // namespace

extern "C" {

int dmk_fold_R2k(dmk_ctx *ctx, const int mesh[3], int64_t ncol, int batch, const void *in_R, int in_is_complex,
                 void *out_k) {
    if (!ctx) return DMK_ERR_INVALID;
    Mesh m(mesh);
    if (!m.ok() || ncol <= 0 || batch <= 0 || !in_R || !out_k || ncol > 0x7fffffffLL)
        return dmk_fail(ctx, DMK_ERR_INVALID, "fold_R2k: bad arguments");
    {
        int rf = launch_fold_fft(ctx, m.n, ncol, batch, in_R, in_is_complex ? 0 : 1, out_k, 0, 0, nullptr);
        if (rf != 0) return rf < 0 ? rf : DMK_OK;
    }
    void *P = nullptr;
    int rc = get_phase(ctx, m, nullptr, 0, &P);
    if (rc) return rc;
    ZGemm g;
    g.M = m.nk; g.N = (int)ncol; g.K = m.nk; g.batch = batch; g.nseg = 1;
    g.seg[0].A = P; g.seg[0].lda = m.nk; g.seg[0].strideA = 0; g.seg[0].a_kmajor = 1;   // A[kdim=R][m=k] = P[R][k]
    g.seg[0].B = in_R; g.seg[0].ldb = ncol; g.seg[0].strideB = (int64_t)m.nk * ncol; g.seg[0].b_kmajor = 1;
    g.seg[0].b_real = in_is_complex ? 0 : 1;
    g.alpha = 1.0; g.epi = ZEPI_STORE; g.C = out_k; g.ldc = ncol; g.strideC = (int64_t)m.nk * ncol;
    return launch_zgemm(ctx, g, DMK_FAM_FOLD);
}

static int fold_k2R_impl(dmk_ctx *ctx, const int mesh[3], int64_t ncol, int batch, const void *in_k, void *out,
                         int real_out, double *imag_max, const int32_t *subset, int nsub) {
    if (!ctx) return DMK_ERR_INVALID;
    Mesh m(mesh);
    if (!m.ok() || ncol <= 0 || batch <= 0 || !in_k || !out || ncol > 0x7fffffffLL || nsub < 0 || nsub > m.nk)
        return dmk_fail(ctx, DMK_ERR_INVALID, "fold_k2R: bad arguments");
    if (imag_max) DMK_HIP(ctx, hipMemsetAsync(imag_max, 0, sizeof(double), ctx->stream));
    if (!(subset && nsub > 0)) {                 // a full mesh: the fused mixed-radix kernel; a k subset is not a mesh
        int rf = launch_fold_fft(ctx, m.n, ncol, batch, in_k, 0, out, real_out, 1, imag_max);
        if (rf != 0) return rf < 0 ? rf : DMK_OK;
    }
    void *P = nullptr;
    int rc = get_phase(ctx, m, subset, subset ? nsub : 0, &P);
    if (rc) return rc;
    const int kin = (subset && nsub > 0) ? nsub : m.nk;
    ZGemm g;
    g.M = m.nk; g.N = (int)ncol; g.K = kin; g.batch = batch; g.nseg = 1;
    // out[R] = (1/N) sum_k conj(P[k][R]) in[k]  ->  A[kdim=k][m=R] = conj(P[k][R])
    g.seg[0].A = P; g.seg[0].lda = m.nk; g.seg[0].strideA = 0; g.seg[0].a_kmajor = 1; g.seg[0].conjA = 1;
    g.seg[0].B = in_k; g.seg[0].ldb = ncol; g.seg[0].strideB = (int64_t)kin * ncol; g.seg[0].b_kmajor = 1;
    g.alpha = 1.0 / (double)m.nk;
    g.epi = real_out ? ZEPI_STORE_REAL : ZEPI_STORE;
    g.C = out; g.ldc = ncol; g.strideC = (int64_t)m.nk * ncol; g.imag_max = imag_max;
    return launch_zgemm(ctx, g, DMK_FAM_FOLD);
}

int dmk_fold_k2R(dmk_ctx *ctx, const int mesh[3], int64_t ncol, int batch, const void *in_k, double *out_R,
                 double *imag_max_dev, const int32_t *k_subset_host, int nsub) {
    return fold_k2R_impl(ctx, mesh, ncol, batch, in_k, out_R, 1, imag_max_dev, k_subset_host, nsub);
}
int dmk_fold_k2R_complex(dmk_ctx *ctx, const int mesh[3], int64_t ncol, int batch, const void *in_k, void *out_R) {
    return fold_k2R_impl(ctx, mesh, ncol, batch, in_k, out_R, 0, nullptr, nullptr, 0);
}

// =============================================================================================
// a9 / a10 : generic batched complex product
// =============================================================================================

namespace {
// C[e] = alpha * sum_s part[s][e]: the reduction of a split-K product (fixed order: deterministic)
__global__ void splitk_reduce_kernel(long long nelem, int nsplit, double alpha, const double2 *__restrict__ part, double2 *__restrict__ C) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < nelem; e += (long long)gridDim.x * blockDim.x) {
        double re = 0.0, im = 0.0;
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            const double2 v = part[(long long)sidx * nelem + e];
            re += v.x;
            im += v.y;
        }
        C[e] = make_double2(alpha * re, alpha * im);
    }
}
}  // namespace

int dmk_zgemm_batched(dmk_ctx *ctx, int opA, int opB, int M, int N, int K, int batch, double alpha, const void *A,
                      int64_t strideA, const void *B, int64_t strideB, void *C, int64_t strideC) {
    if (!ctx) return DMK_ERR_INVALID;
    if (opA < 0 || opA > 2 || opB < 0 || opB > 2 || M < 0 || N < 0 || K < 0 || batch < 0)
        return dmk_fail(ctx, DMK_ERR_INVALID, "zgemm_batched: bad arguments");
    if (M == 0 || N == 0 || batch == 0) return DMK_OK;
    // SPLIT-K for the long-K folds (1/nk) sum_k B_k^H T_k written as ONE product with K = nk * nlo (slater.py:682-704): at C5
    // M = N = 256, K = 43 200, two spins -- 32 workgroups of 64 x 64 tiles on 256 CUs, 5.6 ms per call.  K is cut into `ns` equal
    // chunks that become the batch of one launch per original batch element (partial products in the context's second scratch),
    // then summed in a fixed order.  Only for K-major operands (op(A) = T | C, op(B) = N) and few, small output matrices.
    if (opA != 0 && opB == 0 && K >= 4096 && batch <= 4 && (long long)M * N <= (1 << 18)) {
        int ns = 0;
        for (int cand = 128; cand >= 8; --cand)
            if (K % cand == 0 && K / cand >= 128) { ns = cand; break; }
        void *part = nullptr;
        if (ns && dmk_scratch2(ctx, (size_t)ns * M * N * sizeof(double2), &part) == DMK_OK) {
            const int kc = K / ns;
            const long long nelem = (long long)M * N;
            for (int b = 0; b < batch; ++b) {
                ZGemm gs;
                gs.M = M; gs.N = N; gs.K = kc; gs.batch = ns; gs.nseg = 1;
                ZSeg &ss = gs.seg[0];
                ss.A = reinterpret_cast<const double2 *>(A) + (long long)b * strideA; ss.a_kmajor = 1; ss.lda = M; ss.conjA = (opA == 2);
                ss.strideA = (int64_t)kc * M;
                ss.B = reinterpret_cast<const double2 *>(B) + (long long)b * strideB; ss.b_kmajor = 1; ss.ldb = N;
                ss.strideB = (int64_t)kc * N;
                gs.alpha = 1.0; gs.epi = ZEPI_STORE; gs.C = part; gs.ldc = N; gs.strideC = nelem;
                int rc = launch_zgemm(ctx, gs, DMK_FAM_ZGEMM_SMALL);
                if (rc) return rc;
                FamScope fs(ctx, DMK_FAM_ZGEMM_SMALL);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<long long>((nelem + 255) / 256, 1024)), dim3(256), 0,
                                   ctx->stream, nelem, ns, alpha, reinterpret_cast<const double2 *>(part),
                                   reinterpret_cast<double2 *>(C) + (long long)b * strideC);
                DMK_CHECK_LAUNCH(ctx);
            }
            return DMK_OK;
        }
    }
    ZGemm g;
    g.M = M; g.N = N; g.K = K; g.batch = batch; g.nseg = 1;
    ZSeg &s = g.seg[0];
    s.A = A; s.B = B; s.strideA = strideA; s.strideB = strideB;
    if (opA == 0) { s.a_kmajor = 0; s.lda = K; } else { s.a_kmajor = 1; s.lda = M; s.conjA = (opA == 2); }
    if (opB == 0) { s.b_kmajor = 1; s.ldb = N; } else { s.b_kmajor = 0; s.ldb = K; s.conjB = (opB == 2); }
    g.alpha = alpha; g.epi = ZEPI_STORE; g.C = C; g.ldc = N; g.strideC = strideC;
    if (K == 0) {
        DMK_HIP(ctx, hipMemsetAsync(C, 0, (size_t)16 * ((size_t)(batch - 1) * strideC + (size_t)M * N), ctx->stream));
        return DMK_OK;
    }
    return launch_zgemm(ctx, g, DMK_FAM_ZGEMM_SMALL);
}

int dmk_occ_density(dmk_ctx *ctx, int n, int batch, const void *Vt, const double *occ, void *rho) {
    if (!ctx) return DMK_ERR_INVALID;
    if (n <= 0 || batch <= 0 || !Vt || !occ || !rho) return dmk_fail(ctx, DMK_ERR_INVALID, "occ_density: bad arguments");
    // rho[i][j] = sum_m Vt[m][i] occ[m] conj(Vt[m][j])   (routine/mfd.py:357)
    ZGemm g;
    g.M = n; g.N = n; g.K = n; g.batch = batch; g.nseg = 1;
    ZSeg &s = g.seg[0];
    s.A = Vt; s.lda = n; s.strideA = (int64_t)n * n; s.a_kmajor = 1;
    s.B = Vt; s.ldb = n; s.strideB = (int64_t)n * n; s.b_kmajor = 1; s.conjB = 1; s.kscaleB = occ;
    g.alpha = 1.0; g.epi = ZEPI_STORE; g.C = rho; g.ldc = n; g.strideC = (int64_t)n * n;
    return launch_zgemm(ctx, g, DMK_FAM_ZGEMM_SMALL);
}

int dmk_dgemm_tn_acc(dmk_ctx *ctx, int N, int K, double alpha, const double *X, const double *Y, int64_t ldxy,
                     double *C, int64_t ldc) {
    if (!ctx) return DMK_ERR_INVALID;
    if (N < 0 || K < 0 || !X || !Y || !C) return dmk_fail(ctx, DMK_ERR_INVALID, "dgemm_tn_acc: bad arguments");
    return launch_dgemm_tn_acc(ctx, N, N, K, alpha, X, ldxy, Y, ldxy, C, ldc);
}

int dmk_dgemm_tn_acc_rect(dmk_ctx *ctx, int M, int N, int K, double alpha, const double *X, int64_t ldx,
                          const double *Y, int64_t ldy, double *C, int64_t ldc) {
    if (!ctx) return DMK_ERR_INVALID;
    if (M < 0 || N < 0 || K < 0 || !X || !Y || !C) return dmk_fail(ctx, DMK_ERR_INVALID, "dgemm_tn_acc_rect: bad arguments");
    return launch_dgemm_tn_acc(ctx, M, N, K, alpha, X, ldx, Y, ldy, C, ldc);
}

int dmk_df_block_philox(dmk_ctx *ctx, uint64_t seed, int ki, int kj, int naux, int nao, void *out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (naux <= 0 || nao <= 0 || !out || ki < 0 || kj < 0) return dmk_fail(ctx, DMK_ERR_INVALID, "df_block_philox: bad arguments");
    return launch_philox_block(ctx, seed, ki, kj, naux, nao, out);
}

int dmk_df_blocks_philox_on(dmk_ctx *ctx, void *stream, uint64_t seed, int nblk, const int32_t *ij, int naux, int nao, void *out,
                            int64_t stride_bytes) {
    if (!ctx) return DMK_ERR_INVALID;
    if (naux <= 0 || nao <= 0 || !out || !ij || nblk < 0 || stride_bytes < (int64_t)naux * nao * nao * 16 || (stride_bytes & 15))
        return dmk_fail(ctx, DMK_ERR_INVALID, "df_blocks_philox_on: bad arguments");
    for (int b = 0; b < 2 * nblk; ++b)
        if (ij[b] < 0) return dmk_fail(ctx, DMK_ERR_INVALID, "df_blocks_philox_on: negative k-point index");
    return launch_philox_blocks_on(ctx, reinterpret_cast<hipStream_t>(stream), seed, nblk, ij, naux, nao, out, stride_bytes);
}

int dmk_df_block_philox_on(dmk_ctx *ctx, void *stream, uint64_t seed, int ki, int kj, int naux, int nao, void *out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (naux <= 0 || nao <= 0 || !out || ki < 0 || kj < 0) return dmk_fail(ctx, DMK_ERR_INVALID, "df_block_philox_on: bad arguments");
    return launch_philox_block_on(ctx, reinterpret_cast<hipStream_t>(stream), seed, ki, kj, naux, nao, out);
}

}  // extern "C"

// =============================================================================================
// a11 - a14 : ERI pipeline
// =============================================================================================

struct dmk_eri {
    dmk_ctx *ctx;
    Mesh mesh;
    int nao, naux, nemb, spin, tr;
    int64_t npair;
    const double2 *C;     // spin x nk x nao x nemb
    // AO dimensions off the K tile of the hot kernels (8): they loop over kdim = hot_kdim(nao) against Ch, the pipeline's own copy
    // of C with kdim rows per k point, zero beyond nao (Ch == C and kdim == nao when nao is on the tile)
    int kdim = 0;
    const double2 *Ch = nullptr;
    double2 *Cpad = nullptr;
    double *eri;
    size_t ws_bytes[2] = {0, 0};   // capacities of planes / Ut (they may come from the context's cache)
    // PLANE GEOMETRY: a Re or Im plane has `pr` rows (naux rounded up to the K tile of the contraction kernel, 8) of `pl` doubles
    // (npair rounded up to even).  The padding rows and the padding column are never written by the half transform and stay zero
    // from the memset at the start of a kL, so the contraction always runs on the LDS-DMA kernel with its symmetric launch --
    // up to round 5 an auxiliary basis off the tile (naux 411) or an odd pair count (nemb 250) fell to the register-staged kernel
    // without the symmetric saving.  pr == naux and pl == npair for shapes on the tile: the layout of rounds 1 - 5.
    int64_t pr = 0, pl = 0;
    double *planes = nullptr;   // spin x (2 pr) x pl
    double *planes_view = nullptr;      // dmk_eri_planes with a padded geometry: compact (spin, 2, naux, npair) copy
    size_t planes_view_bytes = 0;
    double2 *Ut = nullptr;      // lchunk x nao x nemb
    int lchunk;
    int hot_rows = 0;     // auxiliary rows per hot step-1 launch (half1_hot_max_rows): blocks of 4 GiB and more go in ranges of L
    int use_3m = 1;       // Karatsuba complex product in the generic half transform (DMK_ERI_3M=0 restores 4M)
    // hot path: step-1 outputs of up to `group` consecutive AO blocks are queued and transformed by ONE
    // step-2 launch whose accumulators (and tril-pack epilogue) are shared by all of them
    int group = 1, pending = 0;
    double *imag = nullptr;   // flags & 2 (no time reversal): Im of the contraction, spin_pair x npair^2, for dmk_eri_imag_norm
    bool hot256 = false;      // step 2 by the nemb = 256 kernel (zhot.hip) instead of the table-driven one (zhot_tab.hip)
    int pend_kj[16], pend_sym[16], pend_ki[16];
    // block ring (dmk_eri_block_ring / dmk_eri_push_ring_slot): `group` AO-block buffers owned by the pipeline; blocks
    // written there are queued WITHOUT running step 1, and the flush runs ONE step-1 launch over all of them
    double2 *ring = nullptr;
    size_t ring_bytes = 0;
    int ring_pending = 0;       // queued ring slots whose step 1 has not run yet (they are the first `ring_pending` slots)
    const double2 *resident_src = nullptr;   // dmk_eri_push_resident: the queued group is read in place from here, not from the ring
    // PRODUCER STREAM of the ring (dmk_eri_ring_slot): the ring is double buffered and device-side producers of group g + 1
    // (a generator kernel, a decompressor) run on `gen_stream` while the compute stream transforms group g.  Ordering by events:
    // ev_free[half] = step 1 of the group that last used that half has run (recorded on the compute stream; the producer stream
    // waits on it before the first fill of the half), ev_gen[half] = the fills of the pending group (recorded on the producer
    // stream after every fill; step 1 of that group waits on it).
    hipStream_t gen_stream = nullptr;
    hipEvent_t ev_gen[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    int ring_halves = 1;        // 2 when the ring is double buffered
    int fill_half = 0;          // half of the pending group
    int next_half = 0;          // half the next group of a ring_slot producer will fill
    bool gen_pending = false;   // the pending group was (partly) filled on the producer stream
    int slot_reserved = -1;     // ring slot handed out by dmk_eri_ring_slot and not pushed yet (-1: none)
    int cur_kL = -1;
    double flops_half = 0.0, flops_contract = 0.0;
    // plane STACK (dmk_eri_stack): nslots > 1 defers the contraction -- the planes of up to nslots kL stay resident, weight-2 kL
    // fill slots from the front, weight-1 kL (only their Re halves are contracted) from the back, and one K-stacked GEMM per
    // weight class and spin block contracts them all (dmk_eri_contract, or automatically when the stack is full / at finish)
    int nslots = 1, n_w2 = 0, n_w1 = 0, cur_slot = 0, cur_weight = 1;
    // A kL that is its own time-reversal partner (weight 1) only ever contributes the REAL part of its planes (eri_transform.py:453-455,
    // 464-467), so step 2 of its blocks computes Re S alone -- two real products instead of the three of 3M (zhot_common.h RE).
    // Known when the kL is begun with its weight (dmk_eri_begin_kL_weighted); dmk_eri_begin_kL keeps the full product.
    bool re_only = false;
    double *slot_planes(int slot, int spin_idx) const {
        return planes + ((size_t)spin_idx * nslots + slot) * 2 * (size_t)pr * pl;
    }
    // host feed (dmk_eri_push_block_host): two device staging blocks filled on a copy stream while the compute stream
    // transforms the other one; created on first use
    hipStream_t copy_stream = nullptr;
    double2 *dstage[2] = {nullptr, nullptr};
    double2 *tstage = nullptr;       // conjugate-transposed copy of a block uploaded for the swapped pair
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
    // sub-group plane copies of the table-driven step 2 (zhot_tab.hip H2TArgs): run p >= 1 of a launch accumulates into copy
    // p - 1 ([spin][2 naux][npair] each); they are zeroed when a kL begins and added to its planes, in order, when it ends
    double *sub_planes = nullptr;
    int nsub_max = 1, sub_used = 1;
    // Freivalds probe (dmk_eri_probe, eri_probe.hip): yref[b] += w X_a^T (X_b x) for every kL that is contracted
    const double *probe_x = nullptr;
    double *probe_y = nullptr;
    bool probe_pending = false;      // planes entered the stack since the probe last ran over it
    dmk_eri(dmk_ctx *c, const int m[3]) : ctx(c), mesh(m) {}
};

extern "C" {

int dmk_eri_begin(dmk_ctx *ctx, const int mesh[3], int nao, int naux, int nemb, int spin, int flags,
                  const void *C_ao_emb, double *eri_out, dmk_eri **out) {
    if (!ctx || !out) return DMK_ERR_INVALID;
    *out = nullptr;
    Mesh m(mesh);
    const bool no_out = (flags & 4) != 0;       // rows-only pipeline (dmk_eri_contract_rows): no ERI of its own
    if (!m.ok() || nao <= 0 || naux <= 0 || nemb <= 0 || (spin != 1 && spin != 2) || !C_ao_emb || (!eri_out && !no_out))
        return dmk_fail(ctx, DMK_ERR_INVALID, "eri_begin: bad arguments");
    dmk_eri *h = new dmk_eri(ctx, mesh);
    h->nao = nao; h->naux = naux; h->nemb = nemb; h->spin = spin; h->tr = flags & 1;
    h->npair = (int64_t)nemb * (nemb + 1) / 2;
    h->pr = ((int64_t)naux + 7) / 8 * 8;
    h->pl = h->npair + (h->npair & 1);
    if (const char *e = getenv("DMK_ERI_PLANE_PAD")) if (atoi(e) == 0) { h->pr = naux; h->pl = h->npair; }     // the unpadded layout (labs)
    if ((flags & 2) && !h->tr) {
        const size_t ib = (size_t)(spin == 2 ? 3 : 1) * h->npair * h->npair * sizeof(double);
        if (dmk_dev_alloc(ctx, reinterpret_cast<void **>(&h->imag), ib) != hipSuccess ||
            hipMemsetAsync(h->imag, 0, ib, ctx->stream) != hipSuccess) {
            if (h->imag) (void)hipFree(h->imag);
            delete h;
            return dmk_fail(ctx, DMK_ERR_NOMEM, "eri_begin: imaginary-part buffer allocation failed (%zu bytes)", ib);
        }
    }
    h->C = reinterpret_cast<const double2 *>(C_ao_emb);
    h->Ch = h->C;
    h->kdim = nao;
    h->eri = no_out ? nullptr : eri_out;
    h->lchunk = naux;
    if (const char *e = getenv("DMK_ERI_3M")) h->use_3m = atoi(e) != 0;
    if (const char *e = getenv("DMK_ERI_LCHUNK")) {
        int v = atoi(e);
        if (v > 0 && v < naux) h->lchunk = v;
    }
    // grouped hot path (block queue, multi-slot launches of both half-transform steps): the specialised step-2 kernel
    // for nemb = 256, the table-driven one for every other embedding dimension; only offered when the flattened step-1
    // kernel covers the shape too, so a queued group can never be left without a kernel
    h->hot256 = half2_hot_usable(nao, nemb) != 0;
    if (const char *e = getenv("DMK_ERI_TAB256")) if (atoi(e) != 0) h->hot256 = false;      // route nemb = 256 through the table kernel
    {   // ranges of L of equal length (a short last range would fall under the kernel's minimum launch size)
        const int cap = std::max(1, std::min(naux, half1_hot_max_rows(nao)));
        const int nranges = (naux + cap - 1) / cap;
        h->hot_rows = (naux + nranges - 1) / nranges;
    }
    if ((h->hot256 || half2_tab_usable(nao, nemb)) && half1_hot_usable(h->hot_rows, nao, nemb)) {
        h->lchunk = naux;
        h->group = h->hot256 ? 8 : 16;               // the table kernel cuts its queue into sub-group runs: a longer queue per launch
        if (const char *e = getenv("DMK_ERI_GROUP")) h->group = atoi(e);
        h->group = std::max(1, std::min(h->group, h->hot256 ? half2_hot_maxslot() : half2_tab_maxslot()));
        if (!h->hot256) h->nsub_max = half2_tab_subgroups(ctx, naux, nao, nemb, spin, h->group, 4);      // 1 unless DMK_ERI_TAB_SUB asks
    }
    // K padding of the hot kernels: a zero-padded copy of C (made below) and Ut rows to read past the last auxiliary row
    const bool kpad = half1_hot_usable(h->hot_rows, nao, nemb) && hot_kdim(nao) != nao;
    if (kpad) h->kdim = hot_kdim(nao);
    const size_t plane_bytes = (size_t)spin * 2 * h->pr * h->pl * sizeof(double);
    const size_t ut_bytes = (size_t)h->lchunk * nao * nemb * sizeof(double2) * (h->group > 1 ? (size_t)h->group * spin : 1) +
                            (size_t)(h->kdim - nao) * nemb * sizeof(double2);
    // reuse the workspace parked in the context by the previous pipeline when it is large enough
    const size_t want[2] = {plane_bytes, ut_bytes};
    void *got[2] = {nullptr, nullptr};
    for (int w = 0; w < 2; ++w) {
        if (ctx->eri_ws[w] && ctx->eri_ws_bytes[w] >= want[w]) {
            got[w] = ctx->eri_ws[w];
            h->ws_bytes[w] = ctx->eri_ws_bytes[w];
            ctx->eri_ws[w] = nullptr;
            ctx->eri_ws_bytes[w] = 0;
        } else {
            if (ctx->eri_ws[w]) {
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipFree(ctx->eri_ws[w]);
                ctx->eri_ws[w] = nullptr;
                ctx->eri_ws_bytes[w] = 0;
            }
            if (dmk_dev_alloc(ctx, &got[w], want[w]) != hipSuccess) got[w] = nullptr;
            h->ws_bytes[w] = want[w];
        }
    }
    h->planes = reinterpret_cast<double *>(got[0]);
    h->Ut = reinterpret_cast<double2 *>(got[1]);
    if (!h->planes || !h->Ut) {
        if (h->planes) (void)hipFree(h->planes);
        if (h->Ut) (void)hipFree(h->Ut);
        if (h->imag) (void)hipFree(h->imag);
        delete h;
        return dmk_fail(ctx, DMK_ERR_NOMEM, "eri_begin: workspace allocation failed (%zu + %zu bytes)", plane_bytes, ut_bytes);
    }
    if (kpad) {
        // Step 2 reads kdim - nao rows past every L of Ut against the zero rows of Ch: whatever is there must be FINITE (a queue
        // slot that step 1 has not written yet, the tail of a parked workspace) -- the buffer is zeroed once.
        const size_t cb = (size_t)spin * m.nk * h->kdim * nemb * sizeof(double2);
        bool ok = dmk_dev_alloc(ctx, reinterpret_cast<void **>(&h->Cpad), cb) == hipSuccess;
        ok = ok && hipMemsetAsync(h->Cpad, 0, cb, ctx->stream) == hipSuccess;
        ok = ok && hipMemcpy2DAsync(h->Cpad, (size_t)h->kdim * nemb * sizeof(double2), h->C, (size_t)nao * nemb * sizeof(double2),
                                    (size_t)nao * nemb * sizeof(double2), (size_t)spin * m.nk, hipMemcpyDeviceToDevice,
                                    ctx->stream) == hipSuccess;
        ok = ok && hipMemsetAsync(h->Ut, 0, ut_bytes, ctx->stream) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (h->Cpad) (void)hipFree(h->Cpad);
            (void)hipFree(h->planes);
            (void)hipFree(h->Ut);
            if (h->imag) (void)hipFree(h->imag);
            delete h;
            return dmk_fail(ctx, DMK_ERR_NOMEM, "eri_begin: padded copy of C_ao_emb failed (%zu bytes)", cb);
        }
        h->Ch = h->Cpad;
    }
    if (h->nsub_max > 1) {
        const size_t sb = (size_t)(h->nsub_max - 1) * spin * 2 * h->pr * h->pl * sizeof(double);
        if (dmk_dev_alloc(ctx, reinterpret_cast<void **>(&h->sub_planes), sb) != hipSuccess) {
            (void)hipGetLastError();
            h->sub_planes = nullptr;                 // not fatal: one run per launch, as before
            h->nsub_max = 1;
        }
    }
    *out = h;
    return DMK_OK;
}

static int eri_contract_stack(dmk_eri *h, int band_lo, int band_hi);

// the probe's share of one plane slot (see dmk_eri_probe)
static int eri_probe_slot(dmk_eri *h, int slot, int nrows, double w) {
    void *tw = nullptr;
    int rc = dmk_scratch(h->ctx, (size_t)4 * h->pr * sizeof(double), &tw);
    if (rc) return rc;
    return launch_eri_probe_slot(h->ctx, h->slot_planes(slot, 0), h->spin == 2 ? h->slot_planes(slot, 1) : nullptr, nrows, h->npair, h->pl,
                                 w, h->probe_x, h->probe_y, reinterpret_cast<double *>(tw));
}

static int eri_begin_kL_impl(dmk_eri *h, int kL, int weight) {
    dmk_ctx *ctx = h->ctx;
    if (kL < 0 || kL >= h->mesh.nk) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_begin_kL: kL out of range");
    if (h->cur_kL >= 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_begin_kL: previous kL not ended");
    h->cur_slot = 0;
    const char *re_env = getenv("DMK_ERI_RE_ONLY");             // read per kL (a handful per second): tests toggle it
    h->re_only = !(re_env && atoi(re_env) == 0) && h->tr && weight == 1 && !h->imag;
    if (h->nslots > 1) {
        if (weight != 1 && weight != 2) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_begin_kL: a plane stack needs the weight (1 or 2) of the kL");
        if (h->n_w2 + h->n_w1 == h->nslots) {                 // stack full: contract everything that is resident
            if (!h->eri)
                return dmk_fail(ctx, DMK_ERR_STATE, "eri_begin_kL: the plane stack is full and this pipeline has no ERI of its own "
                                                    "(take the rows with dmk_eri_contract_rows, then dmk_eri_stack_clear)");
            int rc = eri_contract_stack(h, -1, -1);
            if (rc) return rc;
            h->n_w2 = h->n_w1 = 0;
        }
        h->cur_slot = weight == 2 ? h->n_w2 : h->nslots - 1 - h->n_w1;
        h->cur_weight = weight;
    }
    const size_t bytes = (size_t)2 * h->pr * h->pl * sizeof(double);
    for (int s = 0; s < h->spin; ++s) DMK_HIP(ctx, hipMemsetAsync(h->slot_planes(h->cur_slot, s), 0, bytes, ctx->stream));
    if (h->sub_planes) DMK_HIP(ctx, hipMemsetAsync(h->sub_planes, 0, bytes * h->spin * (h->nsub_max - 1), ctx->stream));
    h->sub_used = 1;
    h->cur_kL = kL;
    h->slot_reserved = -1;
    return DMK_OK;
}

int dmk_eri_begin_kL(dmk_eri *h, int kL) {
    if (!h) return DMK_ERR_INVALID;
    int weight = 0;                     // unknown: the full complex product (no stack: the weight only arrives with dmk_eri_end_kL)
    if (h->nslots > 1 && h->tr && kL >= 0 && kL < h->mesh.nk) {      // integer-mesh plan: the weight follows from the mesh
        std::vector<int> w;
        tr_weights(h->mesh, 1, w);
        weight = w[kL];
    }
    return eri_begin_kL_impl(h, kL, weight);
}

int dmk_eri_begin_kL_weighted(dmk_eri *h, int kL, int weight) {
    if (!h) return DMK_ERR_INVALID;
    return eri_begin_kL_impl(h, kL, weight);
}

static int eri_ring_step1(dmk_eri *h) {
    dmk_ctx *ctx = h->ctx;
    if (h->ring_pending == 0) return DMK_OK;
    const int nao = h->nao, naux = h->naux, nemb = h->nemb;
    const size_t slot_elems = (size_t)naux * nao * nemb;
    const bool resident = h->resident_src != nullptr;
    const double2 *src = resident ? h->resident_src : h->ring + (size_t)h->fill_half * h->group * naux * nao * nao;
    if (!resident && h->gen_pending) DMK_HIP(ctx, hipStreamWaitEvent(ctx->stream, h->ev_gen[h->fill_half], 0));      // the producers of this group
    for (int l0 = 0; l0 < naux; l0 += h->hot_rows) {         // (one launch unless an AO block reaches 4 GiB)
        const int nl = std::min(h->hot_rows, naux - l0);
        int rc = launch_half1_hot_multi(ctx, src + (size_t)l0 * nao * nao, (long long)naux * nao * nao, h->ring_pending, h->pend_ki, h->Ch,
                                        h->Ut + (size_t)l0 * nao * nemb, (long long)slot_elems, nl, nao, nemb, h->spin,
                                        (long long)h->mesh.nk * h->kdim * nemb, (long long)h->group * (long long)slot_elems, h->kdim);
        if (rc < 0) return rc;
        if (rc == 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri ring: hot step-1 kernel unavailable for the queued blocks");
    }
    if (!resident && h->ring_halves == 2) DMK_HIP(ctx, hipEventRecord(h->ev_free[h->fill_half], ctx->stream));           // the half may be refilled
    h->resident_src = nullptr;
    h->ring_pending = 0;
    h->gen_pending = false;
    h->fill_half = 0;                                  // a producer on the compute stream (no dmk_eri_ring_slot) always uses half 0
    return DMK_OK;
}

static int eri_flush(dmk_eri *h) {
    dmk_ctx *ctx = h->ctx;
    if (h->pending == 0) return DMK_OK;
    {
        int rc1 = eri_ring_step1(h);
        if (rc1) return rc1;
    }
    const int nao = h->nao, naux = h->naux, nemb = h->nemb;
    const size_t slot_elems = (size_t)naux * nao * nemb;
    // one launch for both spin channels: C, Ut and the planes of spin 1 sit at constant offsets from those of spin 0
    const void *cj[16];
    for (int i = 0; i < h->pending; ++i)
        cj[i] = h->Ch + (size_t)h->pend_kj[i] * h->kdim * nemb;
    int rc;
    if (h->hot256) {
        rc = launch_half2_hot(ctx, h->Ut, (long long)slot_elems, h->pending, cj, h->pend_sym, h->slot_planes(h->cur_slot, 0), h->pr, h->pl,
                              naux, nao, nemb, h->spin, (long long)h->group * (long long)slot_elems, (long long)h->mesh.nk * h->kdim * nemb,
                              (long long)h->nslots * 2LL * h->pr * h->pl, h->kdim, h->re_only ? 1 : 0);
    } else {
        const int nsub = h->sub_planes ? half2_tab_subgroups(ctx, naux, nao, nemb, h->spin, h->pending, h->nsub_max) : 1;
        rc = launch_half2_tab(ctx, h->Ut, (long long)slot_elems, h->pending, cj, h->pend_sym, h->slot_planes(h->cur_slot, 0), h->pr, h->pl,
                              naux, nao, nemb, h->spin, (long long)h->group * (long long)slot_elems, (long long)h->mesh.nk * h->kdim * nemb,
                              (long long)h->nslots * 2LL * h->pr * h->pl, nsub, h->sub_planes, (long long)h->spin * 2LL * h->pr * h->pl,
                              h->kdim, h->re_only ? 1 : 0);
        if (rc == 1) h->sub_used = std::max(h->sub_used, nsub);
    }
    if (rc < 0) return rc;
    if (rc == 0) {
        // the grouped kernel declined (misaligned buffer, a table it cannot build): step 2 of every queued block through the
        // generic c128 GEMM with the same tril-pack epilogue -- slower (one launch per block and spin), never wrong
        for (int i = 0; i < h->pending; ++i)
            for (int s = 0; s < h->spin; ++s) {
                const double2 *ut = h->Ut + ((size_t)s * h->group + i) * slot_elems;
                const double2 *Cj = h->C + ((size_t)s * h->mesh.nk + h->pend_kj[i]) * nao * nemb;
                ZGemm g2;
                g2.M = nemb; g2.N = nemb; g2.K = nao; g2.batch = naux; g2.nseg = h->pend_sym[i] ? 2 : 1;
                g2.seg[0].A = ut; g2.seg[0].lda = nemb; g2.seg[0].strideA = (int64_t)nao * nemb;
                g2.seg[0].B = Cj; g2.seg[0].ldb = nemb; g2.seg[0].strideB = 0;
                g2.seg[1].A = Cj; g2.seg[1].lda = nemb; g2.seg[1].strideA = 0;
                g2.seg[1].B = ut; g2.seg[1].ldb = nemb; g2.seg[1].strideB = (int64_t)nao * nemb;
                g2.epi = ZEPI_PACK_ACC; g2.lower_only = 1; g2.use_3m = h->use_3m;
                g2.planes = h->slot_planes(h->cur_slot, s); g2.naux = h->pr; g2.npair = h->pl;
                int rg = launch_zgemm(ctx, g2, DMK_FAM_ZGEMM_HALF2);
                if (rg) return rg;
            }
    }
    h->pending = 0;
    return DMK_OK;
}

namespace {
__global__ void planes_add_kernel(long long n, double *__restrict__ a, const double *__restrict__ b) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
        a[t] += b[t];
}
}  // namespace

// End of a kL: the sub-group copies (runs p >= 1 of the step-2 launches) are added to the kL's planes, copy by copy in a fixed
// order -- the sum every plane element receives is the same whichever way the launches were cut.
static int eri_fold_subplanes(dmk_eri *h, bool rezero = false) {
    dmk_ctx *ctx = h->ctx;
    if (!h->sub_planes || h->sub_used <= 1) return DMK_OK;
    const long long n = 2LL * h->pr * h->pl;
    for (int p = 1; p < h->sub_used; ++p)
        for (int s = 0; s < h->spin; ++s) {
            FamScope fs(ctx, DMK_FAM_MISC);
            hipLaunchKernelGGL(planes_add_kernel, dim3(4096), dim3(256), 0, ctx->stream, n, h->slot_planes(h->cur_slot, s),
                               h->sub_planes + ((size_t)(p - 1) * h->spin + s) * (size_t)n);
            DMK_CHECK_LAUNCH(ctx);
        }
    // the kL goes on (dmk_eri_planes in the middle of one): what was just added must not be added again at its end
    if (rezero) DMK_HIP(ctx, hipMemsetAsync(h->sub_planes, 0, (size_t)n * sizeof(double) * h->spin * (h->nsub_max - 1), ctx->stream));
    h->sub_used = 1;
    return DMK_OK;
}

int dmk_eri_push_block(dmk_eri *h, int ki, int kj, int symmetrise, const void *Lpq) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    if (h->cur_kL < 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_block: no kL in progress");
    if (ki < 0 || ki >= h->mesh.nk || kj < 0 || kj >= h->mesh.nk || !Lpq)
        return dmk_fail(ctx, DMK_ERR_INVALID, "eri_push_block: bad arguments");
    const int nao = h->nao, naux = h->naux, nemb = h->nemb;
    const double2 *L = reinterpret_cast<const double2 *>(Lpq);
    const size_t slot_elems = (size_t)naux * nao * nemb;
    if (h->group > 1) {
        // hot path: step 1 now (it consumes the caller's block buffer), step 2 when the queue is full
        if (h->ring_pending) {
            int rc1 = eri_ring_step1(h);
            if (rc1) return rc1;
        }
        const int slot = h->pending;
        // both spin channels in one launch (they share the AO block); per-spin generic kernels only if it declines
        int rc_hot = 1;
        for (int l0 = 0; l0 < naux && rc_hot == 1; l0 += h->hot_rows) {
            const int nl = std::min(h->hot_rows, naux - l0);
            rc_hot = launch_half1_hot(ctx, L + (size_t)l0 * nao * nao, h->Ch + (size_t)ki * h->kdim * nemb,
                                      h->Ut + (size_t)slot * slot_elems + (size_t)l0 * nao * nemb, nl, nao, nemb, h->spin,
                                      (long long)h->mesh.nk * h->kdim * nemb, (long long)h->group * (long long)slot_elems, h->kdim);
            if (rc_hot == 0 && l0 > 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_block: hot step-1 kernel declined a later range of L");
        }
        if (rc_hot < 0) return rc_hot;
        for (int s = 0; s < h->spin && rc_hot == 0; ++s) {
            const double2 *Ci = h->C + ((size_t)s * h->mesh.nk + ki) * nao * nemb;
            double2 *ut = h->Ut + ((size_t)s * h->group + slot) * slot_elems;
            int rc = 0;
            {
                ZGemm g1;
                g1.M = nao; g1.N = nemb; g1.K = nao; g1.batch = naux; g1.nseg = 1;
                g1.seg[0].A = L; g1.seg[0].lda = nao; g1.seg[0].strideA = (int64_t)nao * nao;
                g1.seg[0].B = Ci; g1.seg[0].ldb = nemb; g1.seg[0].strideB = 0; g1.seg[0].conjB = 1;
                g1.flatten_m = 1; g1.big_tile = 1; g1.use_3m = h->use_3m;
                g1.epi = ZEPI_STORE; g1.C = ut; g1.ldc = nemb; g1.strideC = (int64_t)nao * nemb;
                rc = launch_zgemm(ctx, g1, DMK_FAM_ZGEMM_HALF1);
                if (rc) return rc;
            }
        }
        h->pend_kj[slot] = kj;
        h->pend_sym[slot] = symmetrise ? 1 : 0;
        h->pending += 1;
        if (h->pending == h->group) {
            int rc = eri_flush(h);
            if (rc) return rc;
        }
        h->flops_half += (double)h->spin * (8.0 * naux * (double)nao * nao * nemb + 8.0 * naux * (double)nao * nemb * nemb);
        return DMK_OK;
    }
    for (int s = 0; s < h->spin; ++s) {
        const double2 *Ci = h->C + ((size_t)s * h->mesh.nk + ki) * nao * nemb;
        const double2 *Cj = h->C + ((size_t)s * h->mesh.nk + kj) * nao * nemb;
        double *planes = h->slot_planes(h->cur_slot, s);
        for (int l0 = 0; l0 < naux; l0 += h->lchunk) {
            const int nl = std::min(h->lchunk, naux - l0);
            // step 1: Ut[L][q][a] = sum_p Lpq[L][p][q] conj(Ci[p][a])
            ZGemm g1;
            g1.M = nao; g1.N = nemb; g1.K = nao; g1.batch = nl; g1.nseg = 1;
            g1.seg[0].A = L + (size_t)l0 * nao * nao; g1.seg[0].lda = nao; g1.seg[0].strideA = (int64_t)nao * nao;
            g1.seg[0].B = Ci; g1.seg[0].ldb = nemb; g1.seg[0].strideB = 0; g1.seg[0].conjB = 1;
            g1.flatten_m = 1; g1.big_tile = 1; g1.use_3m = h->use_3m;
            g1.epi = ZEPI_STORE; g1.C = h->Ut; g1.ldc = nemb; g1.strideC = (int64_t)nao * nemb;
            int rc = launch_half1_hot(ctx, g1.seg[0].A, h->Ch + ((size_t)s * h->mesh.nk + ki) * h->kdim * nemb, h->Ut, nl, nao, nemb, 1, 0, 0,
                                      h->kdim);
            if (rc < 0) return rc;
            if (rc == 0) {
                rc = launch_zgemm(ctx, g1, DMK_FAM_ZGEMM_HALF1);
                if (rc) return rc;
            }
            // step 2: S[a][b] = sum_q Ut[L][q][a] Cj[q][b] (+ sum_q Cj[q][a] Ut[L][q][b]); tril-pack, accumulate
            ZGemm g2;
            g2.M = nemb; g2.N = nemb; g2.K = nao; g2.batch = nl; g2.nseg = symmetrise ? 2 : 1;
            g2.seg[0].A = h->Ut; g2.seg[0].lda = nemb; g2.seg[0].strideA = (int64_t)nao * nemb;
            g2.seg[0].B = Cj; g2.seg[0].ldb = nemb; g2.seg[0].strideB = 0;
            g2.seg[1].A = Cj; g2.seg[1].lda = nemb; g2.seg[1].strideA = 0;
            g2.seg[1].B = h->Ut; g2.seg[1].ldb = nemb; g2.seg[1].strideB = (int64_t)nao * nemb;
            g2.epi = ZEPI_PACK_ACC; g2.lower_only = 1; g2.use_3m = h->use_3m;
            g2.planes = planes + (size_t)l0 * h->pl; g2.naux = h->pr; g2.npair = h->pl;
            rc = launch_zgemm(ctx, g2, DMK_FAM_ZGEMM_HALF2);
            if (rc) return rc;
        }
    }
    h->flops_half += (double)h->spin * (8.0 * naux * (double)nao * nao * nemb + 8.0 * naux * (double)nao * nemb * nemb);
    return DMK_OK;
}

int dmk_eri_end_kL(dmk_eri *h, int weight) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    if (h->cur_kL < 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL: no kL in progress");
    {
        int rcf = eri_flush(h);
        if (rcf) return rcf;
        rcf = eri_fold_subplanes(h);
        if (rcf) return rcf;
    }
    int K, Kalg;                        // rows of the planes that enter (padding rows are zero) / rows that count as work
    double alpha;
    if (h->tr) {
        if (weight != 1 && weight != 2) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_end_kL: weight must be 1 or 2");
        if (h->re_only && weight != 1)
            return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL: weight %d, but the kL was begun with weight 1 (its imaginary planes were not computed)",
                            weight);
        K = (int)(weight == 1 ? h->pr : 2 * h->pr);
        Kalg = weight == 1 ? h->naux : 2 * h->naux;
        alpha = (double)weight;
    } else {
        K = (int)(2 * h->pr);
        Kalg = 2 * h->naux;
        alpha = 1.0;
    }
    const int64_t np = h->npair, pl = h->pl;
    if (h->nslots > 1) {
        // deferred: the planes stay in their slot until the stack is contracted
        if (!h->tr) return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL: the plane stack needs time-reversal symmetry");
        if (weight != h->cur_weight)
            return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL: weight %d, but the slot was chosen for weight %d at begin_kL", weight,
                            h->cur_weight);
        if (weight == 2) h->n_w2 += 1; else h->n_w1 += 1;
        h->probe_pending = true;
        h->flops_contract += (h->spin == 2 ? 3.0 : 1.0) * 2.0 * (double)Kalg * (double)np * (double)np;
        h->cur_kL = -1;
        return DMK_OK;
    }
    if (!h->eri) return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL: a pipeline without an ERI of its own needs a plane stack (dmk_eri_stack)");
    if (h->probe_x) {
        int rcp = eri_probe_slot(h, 0, K, alpha);
        if (rcp) return rcp;
    }
    const double *X0 = h->slot_planes(0, 0);
    const double *X1 = h->slot_planes(0, 1);
    auto gemm = [&](int Kr, double al, const double *A, const double *B, double *Cb) {
        return launch_dgemm_tn_acc_seg(ctx, (int)np, (int)np, Kr, al, A, pl, B, pl, Cb, np, 0, 0, 0, -1, -1, (int)pl, (int)pl);
    };
    int rc = gemm(K, alpha, X0, X0, h->eri);
    if (rc) return rc;
    if (h->spin == 2) {
        rc = gemm(K, alpha, X0, X1, h->eri + (size_t)np * np);
        if (rc) return rc;
        rc = gemm(K, alpha, X1, X1, h->eri + (size_t)2 * np * np);
        if (rc) return rc;
    }
    if (h->imag) {
        // Im (L_a^H L_b) = Re_a^T Im_b - Im_a^T Re_b  (the part eri.real drops, eri_transform.py:385-394)
        const int nb = h->spin == 2 ? 3 : 1;
        for (int b = 0; b < nb; ++b) {
            const double *A = (b == 2) ? X1 : X0, *B = (b == 0) ? X0 : X1;
            const double *Are = A, *Aim = A + (size_t)h->pr * pl, *Bre = B, *Bim = B + (size_t)h->pr * pl;
            double *Cb = h->imag + (size_t)b * np * np;
            rc = gemm((int)h->pr, 1.0, Are, Bim, Cb);
            if (rc) return rc;
            rc = gemm((int)h->pr, -1.0, Aim, Bre, Cb);
            if (rc) return rc;
        }
    }
    h->flops_contract += (h->spin == 2 ? 3.0 : 1.0) * 2.0 * (double)Kalg * (double)np * (double)np;
    h->cur_kL = -1;
    return DMK_OK;
}

// One K-stacked GEMM per weight class and spin block over everything resident in the stack, restricted to the tile band
// [band_lo, band_hi) of the pair index (-1: all).  Weight-2 slots are adjacent from the front: their Re and Im planes form one
// contiguous K range.  Weight-1 slots sit at the back and only their Re halves enter: K segments of naux rows, one slot apart.
static int eri_contract_stack(dmk_eri *h, int band_lo, int band_hi) {
    dmk_ctx *ctx = h->ctx;
    if (!h->eri) return dmk_fail(ctx, DMK_ERR_STATE, "eri contraction: this pipeline was opened without an ERI of its own (flags bit 2)");
    const int64_t np = h->npair, pl = h->pl;
    const int64_t slot_stride = 2LL * h->pr * pl;
    // slots per launch.  Measured at C5 (13 weight-2 kL resident): 1, 2, 4 or all 13 kL per launch run at the same 69.3-69.6 TF on
    // the matrix pipe -- there the contraction is not sensitive to K -- but the HBM traffic is not the same: with K = 1600 the
    // operand panels of the eight XCDs' super-blocks (8 x 16 panels x K x 128 x 8 B = 210 MB) still fit the 256 MB Infinity Cache
    // and a launch fetches 31 GB; with K = 3200 they do not and it fetches 90 GB for twice the work (rocprofv3 FETCH_SIZE).
    // Small pair spaces are different: at C4 (npair 9316, K = 832 per kL) a tile's epilogue -- direct plus mirrored store of
    // 128 x 128 doubles -- is a visible share of its 104 K-tiles: 1 / 2 / 4 kL per launch measured 56.6 / 63.8 / 66.4 TF on the
    // pipe (round 4).  Rule: as many kL per launch (at most 4) as keep the panels of a launch inside the Infinity Cache --
    // C5: 1 (as before), C4: 4.  DMK_ERI_KCHUNK overrides.
    static const int kchunk_env = [] { const char *e = getenv("DMK_ERI_KCHUNK"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    auto kchunk_for = [&](int seg_rows) {
        if (kchunk_env) return kchunk_env;
        const double per_slot = (double)seg_rows * (double)std::min<int64_t>(np, 16384) * 8.0;
        return std::max(1, std::min(4, (int)(268435456.0 / per_slot)));
    };
    if (h->probe_x && h->probe_pending) {
        // once per resident plane set, however many bands the contraction is finished in
        for (int i = 0; i < h->n_w2; ++i) {
            int rc = eri_probe_slot(h, i, (int)(2 * h->pr), 2.0);
            if (rc) return rc;
        }
        for (int i = 0; i < h->n_w1; ++i) {
            int rc = eri_probe_slot(h, h->nslots - h->n_w1 + i, (int)h->pr, 1.0);
            if (rc) return rc;
        }
        h->probe_pending = false;
    }
    for (int w = 2; w >= 1; --w) {
        const int n = w == 2 ? h->n_w2 : h->n_w1;
        const int first = w == 2 ? 0 : h->nslots - h->n_w1;
        const int seg_rows = (int)(w == 2 ? 2 * h->pr : h->pr);
        const int kchunk = kchunk_for(seg_rows);
        for (int s0 = 0; s0 < n; s0 += kchunk) {
            const int K = std::min(kchunk, n - s0) * seg_rows;
            const double *X0 = h->slot_planes(first + s0, 0);
            int rc = launch_dgemm_tn_acc_seg(ctx, (int)np, (int)np, K, (double)w, X0, pl, X0, pl, h->eri, np, seg_rows, slot_stride,
                                             slot_stride, band_lo, band_hi, (int)pl, (int)pl);
            if (rc) return rc;
            if (h->spin == 2) {
                const double *X1 = h->slot_planes(first + s0, 1);
                rc = launch_dgemm_tn_acc_seg(ctx, (int)np, (int)np, K, (double)w, X0, pl, X1, pl, h->eri + (size_t)np * np, np, seg_rows,
                                             slot_stride, slot_stride, band_lo, band_hi, (int)pl, (int)pl);
                if (rc) return rc;
                rc = launch_dgemm_tn_acc_seg(ctx, (int)np, (int)np, K, (double)w, X1, pl, X1, pl, h->eri + (size_t)2 * np * np, np,
                                             seg_rows, slot_stride, slot_stride, band_lo, band_hi, (int)pl, (int)pl);
                if (rc) return rc;
            }
        }
    }
    return DMK_OK;
}

int dmk_eri_stack(dmk_eri *h, int nslots_wanted, int *nslots_granted) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    if (nslots_wanted < 1) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_stack: needs at least one slot");
    if (h->cur_kL >= 0 || h->n_w2 + h->n_w1 > 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_stack: a kL is in progress or the stack is not empty");
    if ((h->imag || !h->tr) && nslots_wanted > 1) nslots_wanted = 1;          // the non-time-reversal branch contracts per kL
    const size_t slot_bytes = (size_t)h->spin * 2 * h->pr * h->pl * sizeof(double);
    int n = nslots_wanted;
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    while (true) {
        if (h->ws_bytes[0] >= slot_bytes * n) break;                            // the buffer at hand is large enough
        void *fresh = nullptr;
        if (dmk_dev_alloc(ctx, &fresh, slot_bytes * n) == hipSuccess) {
            (void)hipFree(h->planes);
            h->planes = reinterpret_cast<double *>(fresh);
            h->ws_bytes[0] = slot_bytes * n;
            break;
        }
        if (n == 1) return dmk_fail(ctx, DMK_ERR_NOMEM, "eri_stack: no memory for a single plane slot");
        n = std::max(1, n / 2);
    }
    h->nslots = n;
    if (nslots_granted) *nslots_granted = n;
    return DMK_OK;
}

int dmk_eri_probe(dmk_eri *h, const double *x, double *yref) {
    if (!h) return DMK_ERR_INVALID;
    if ((x == nullptr) != (yref == nullptr)) return dmk_fail(h->ctx, DMK_ERR_INVALID, "eri_probe: x and yref go together (both NULL: off)");
    if (x && !h->eri) return dmk_fail(h->ctx, DMK_ERR_STATE, "eri_probe: a rows-only pipeline contracts nothing to probe");
    if (h->n_w2 + h->n_w1 > 0 || h->cur_kL >= 0) return dmk_fail(h->ctx, DMK_ERR_STATE, "eri_probe: set it before the first kL of a plane set");
    h->probe_x = x;
    h->probe_y = yref;
    h->probe_pending = false;
    return DMK_OK;
}

int dmk_eri_contract(dmk_eri *h, int band_lo, int band_hi, int done) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    if (h->cur_kL >= 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_contract: a kL is in progress");
    if (h->nslots > 1) {
        int rc = eri_contract_stack(h, band_lo, band_hi);
        if (rc) return rc;
        if (done) h->n_w2 = h->n_w1 = 0;
    }
    return DMK_OK;
}

// Rows [row_lo, row_hi) of every spin block of the contraction of what is resident, ACCUMULATED into `out` ((spin_pair, rows,
// npair) f64, caller-zeroed) instead of into the pipeline's ERI: the out-of-core form of _Lij_s4_to_eri
// (eri_transform.py:486-521 adds ERI_SLICE-row slabs to the file): the full (spin_pair, npair, npair) tensor never has to fit HBM.
int dmk_eri_contract_rows(dmk_eri *h, int64_t row_lo, int64_t row_hi, double *out) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    const int64_t np = h->npair;
    if (h->cur_kL >= 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_contract_rows: a kL is in progress");
    if (h->nslots <= 1) return dmk_fail(ctx, DMK_ERR_STATE, "eri_contract_rows: needs a plane stack (dmk_eri_stack)");
    if (!out || row_lo < 0 || row_hi > np || row_lo >= row_hi || (row_lo & 1))
        return dmk_fail(ctx, DMK_ERR_INVALID, "eri_contract_rows: bad row range [%lld, %lld) (row_lo must be even)", (long long)row_lo, (long long)row_hi);
    const int rows = (int)(row_hi - row_lo);
    const int64_t pl = h->pl;
    const int64_t slot_stride = 2LL * h->pr * pl;
    const size_t blk = (size_t)rows * np;
    const int rows_p = (row_hi == np) ? (int)(pl - row_lo) : rows;      // the last slab may load the padding column
    for (int w = 2; w >= 1; --w) {
        const int n = w == 2 ? h->n_w2 : h->n_w1;
        if (n == 0) continue;
        const int first = w == 2 ? 0 : h->nslots - h->n_w1;
        const int seg_rows = (int)(w == 2 ? 2 * h->pr : h->pr);
        const int K = n * seg_rows;
        const double *X0 = h->slot_planes(first, 0);
        int rc = launch_dgemm_tn_acc_seg(ctx, rows, (int)np, K, (double)w, X0 + row_lo, pl, X0, pl, out, np, seg_rows, slot_stride,
                                         slot_stride, -1, -1, rows_p, (int)pl);
        if (rc) return rc;
        if (h->spin == 2) {
            const double *X1 = h->slot_planes(first, 1);
            rc = launch_dgemm_tn_acc_seg(ctx, rows, (int)np, K, (double)w, X0 + row_lo, pl, X1, pl, out + blk, np, seg_rows, slot_stride,
                                         slot_stride, -1, -1, rows_p, (int)pl);
            if (rc) return rc;
            rc = launch_dgemm_tn_acc_seg(ctx, rows, (int)np, K, (double)w, X1 + row_lo, pl, X1, pl, out + 2 * blk, np, seg_rows,
                                         slot_stride, slot_stride, -1, -1, rows_p, (int)pl);
            if (rc) return rc;
        }
    }
    return DMK_OK;
}

int dmk_eri_stack_clear(dmk_eri *h) {
    if (!h) return DMK_ERR_INVALID;
    if (h->cur_kL >= 0) return dmk_fail(h->ctx, DMK_ERR_STATE, "eri_stack_clear: a kL is in progress");
    h->n_w2 = h->n_w1 = 0;
    return DMK_OK;
}

int dmk_eri_stack_free_slots(const dmk_eri *h, int *free_slots) {
    if (!h || !free_slots) return DMK_ERR_INVALID;
    *free_slots = h->nslots > 1 ? h->nslots - h->n_w2 - h->n_w1 : 0;
    return DMK_OK;
}

int dmk_eri_bands(const dmk_eri *h, int *nbands, int *band_rows) {
    if (!h || !nbands) return DMK_ERR_INVALID;
    *nbands = (int)((h->npair + 127) / 128);
    if (band_rows) *band_rows = 128;
    return DMK_OK;
}

namespace {
__global__ void maxabs_kernel(long long n, const double *__restrict__ a, unsigned long long *__restrict__ out) {
    double m = 0.0;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
        m = fmax(m, fabs(a[t]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(m));   // |x| orders like its bits
}
}  // namespace

int dmk_eri_imag_norm(dmk_eri *h, double *out) {
    if (!h || !out) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    *out = 0.0;
    if (!h->imag) return DMK_OK;                     // time reversal: the contraction is real by construction
    void *scr;
    int rc = dmk_scratch(ctx, 16, &scr);
    if (rc) return rc;
    DMK_HIP(ctx, hipMemsetAsync(scr, 0, 8, ctx->stream));
    const long long n = (long long)(h->spin == 2 ? 3 : 1) * h->npair * h->npair;
    {
        FamScope fs(ctx, DMK_FAM_MISC);
        hipLaunchKernelGGL(maxabs_kernel, dim3(4096), dim3(256), 0, ctx->stream, n, h->imag,
                           reinterpret_cast<unsigned long long *>(scr));
        DMK_CHECK_LAUNCH(ctx);
    }
    unsigned long long bits = 0;
    DMK_HIP(ctx, hipMemcpyAsync(&bits, scr, 8, hipMemcpyDeviceToHost, ctx->stream));
    DMK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, &bits, 8);
    return DMK_OK;
}

int dmk_eri_imag_buffer(dmk_eri *h, double **imag_out, int64_t *elems_out) {
    if (!h || !imag_out) return DMK_ERR_INVALID;
    *imag_out = h->imag;
    if (elems_out) *elems_out = h->imag ? (int64_t)(h->spin == 2 ? 3 : 1) * h->npair * h->npair : 0;
    return DMK_OK;
}

namespace {
__global__ void planes_sub_kernel(long long n, double *__restrict__ a, const double *__restrict__ b) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
        a[t] -= b[t];
}
}  // namespace

// GSO (partial particle-hole) contraction, eri_transform.py:1252-1277: with the two "spin" flavours a, b of the
// half-transformed tensor, aaaa + bbbb - aabb - bbaa = (a - b)^T (a - b): ONE symmetric GEMM on the difference planes.
int dmk_eri_end_kL_gso(dmk_eri *h, int weight) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    if (h->cur_kL < 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL_gso: no kL in progress");
    if (h->spin != 2) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_end_kL_gso: needs the two flavours (spin = 2)");
    if (!h->eri) return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL_gso: this pipeline has no ERI of its own");
    if (h->probe_x) return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL_gso: the contraction probe covers the spin-block contraction only");
    if (h->nslots > 1) return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL_gso: not available with a plane stack");
    {
        int rcf = eri_flush(h);
        if (rcf) return rcf;
        rcf = eri_fold_subplanes(h);
        if (rcf) return rcf;
    }
    int K, Kalg;
    double alpha;
    if (h->tr) {
        if (weight != 1 && weight != 2) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_end_kL_gso: weight must be 1 or 2");
        if (h->re_only && weight != 1)
            return dmk_fail(ctx, DMK_ERR_STATE, "eri_end_kL_gso: weight %d, but the kL was begun with weight 1", weight);
        K = (int)(weight == 1 ? h->pr : 2 * h->pr);
        Kalg = weight == 1 ? h->naux : 2 * h->naux;
        alpha = (double)weight;
    } else {
        K = (int)(2 * h->pr);
        Kalg = 2 * h->naux;
        alpha = 1.0;
    }
    const int64_t np = h->npair, pl = h->pl;
    double *X0 = h->slot_planes(0, 0);
    const double *X1 = h->slot_planes(0, 1);
    const long long nel = (long long)2 * h->pr * pl;
    {
        FamScope fs(ctx, DMK_FAM_MISC);
        hipLaunchKernelGGL(planes_sub_kernel, dim3(8192), dim3(256), 0, ctx->stream, nel, X0, X1);
        DMK_CHECK_LAUNCH(ctx);
    }
    int rc = launch_dgemm_tn_acc_seg(ctx, (int)np, (int)np, K, alpha, X0, pl, X0, pl, h->eri, np, 0, 0, 0, -1, -1, (int)pl, (int)pl);
    if (rc) return rc;
    h->flops_contract += 2.0 * (double)Kalg * (double)np * (double)np;
    h->cur_kL = -1;
    return DMK_OK;
}

int dmk_eri_block_ring(dmk_eri *h, void **ring_out, int *nslots_out) {
    if (!h || !ring_out || !nslots_out) return DMK_ERR_INVALID;
    *ring_out = nullptr;
    *nslots_out = 0;
    if (h->group <= 1) return DMK_OK;                 // generic path: no queue, use dmk_eri_push_block
    if (!h->ring) {
        dmk_ctx *ctx = h->ctx;
        {   // double buffering + producer stream only on request (DMK_ERI_GEN_STREAM=1).  Measured with the Philox generator as the
            // producer (round 4): the overlap LOSES -- C4 258.7 -> 266.4 ms per step, C5 (4 kL) 1884 -> 1892 ms: the MFMA kernels
            // occupy every CU, a concurrent generator only gets slots as their workgroups retire (a 17 us launch takes 95 us) and
            // its 4 TB/s write burst slows the step-2 launch it overlaps by 5-15 %.  A producer that is NOT bandwidth-bound (a
            // decompressor) may do better, so the path stays available and tested.
            const char *e = getenv("DMK_ERI_GEN_STREAM");
            h->ring_halves = (e && atoi(e) != 0) ? 2 : 1;
        }
        const size_t bytes = (size_t)h->ring_halves * h->group * h->naux * h->nao * h->nao * sizeof(double2);
        if (ctx->eri_ws[2] && ctx->eri_ws_bytes[2] >= bytes) {          // parked by the previous pipeline
            h->ring = reinterpret_cast<double2 *>(ctx->eri_ws[2]);
            h->ring_bytes = ctx->eri_ws_bytes[2];
            ctx->eri_ws[2] = nullptr;
            ctx->eri_ws_bytes[2] = 0;
        } else {
            if (ctx->eri_ws[2]) {
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipFree(ctx->eri_ws[2]);
                ctx->eri_ws[2] = nullptr;
                ctx->eri_ws_bytes[2] = 0;
            }
            if (dmk_dev_alloc(ctx, reinterpret_cast<void **>(&h->ring), bytes) != hipSuccess) {
                (void)hipGetLastError();
                h->ring = nullptr;
                return DMK_OK;                        // not fatal: the caller falls back to dmk_eri_push_block
            }
            h->ring_bytes = bytes;
        }
        if (h->ring_halves == 2 && !h->gen_stream) {
            bool ok = hipStreamCreateWithFlags(&h->gen_stream, hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; i < 2 && ok; ++i)
                ok = hipEventCreateWithFlags(&h->ev_gen[i], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&h->ev_free[i], hipEventDisableTiming) == hipSuccess;
            if (!ok) {                                // no second stream: single-buffered ring on the compute stream, as before
                (void)hipGetLastError();
                h->ring_halves = 1;
            }
        }
    }
    *ring_out = h->ring;
    *nslots_out = h->group;
    return DMK_OK;
}

int dmk_eri_ring_slot(dmk_eri *h, int slot, void **ptr_out, void **stream_out) {
    if (!h || !ptr_out || !stream_out) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    if (!h->ring || h->group <= 1) return dmk_fail(ctx, DMK_ERR_STATE, "eri_ring_slot: no block ring (dmk_eri_block_ring)");
    if (slot != h->pending || slot >= h->group)
        return dmk_fail(ctx, DMK_ERR_INVALID, "eri_ring_slot: slots are filled in order (next is %d, asked for %d)", h->pending, slot);
    const size_t blk = (size_t)h->naux * h->nao * h->nao;
    h->slot_reserved = slot;
    if (h->ring_halves < 2) {                          // single buffer: the producer shares the compute stream
        *ptr_out = h->ring + (size_t)slot * blk;
        *stream_out = reinterpret_cast<void *>(ctx->stream);
        return DMK_OK;
    }
    if (slot == 0) {                                   // a new group: the other half; its last consumer must have run
        h->fill_half = h->next_half;
        h->next_half ^= 1;
        DMK_HIP(ctx, hipStreamWaitEvent(h->gen_stream, h->ev_free[h->fill_half], 0));
        h->gen_pending = true;
    }
    *ptr_out = h->ring + ((size_t)h->fill_half * h->group + slot) * blk;
    *stream_out = reinterpret_cast<void *>(h->gen_stream);
    return DMK_OK;
}

int dmk_eri_push_ring_slot(dmk_eri *h, int ki, int kj, int symmetrise) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    if (h->cur_kL < 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_ring_slot: no kL in progress");
    if (!h->ring || h->group <= 1) return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_ring_slot: no block ring (dmk_eri_block_ring)");
    if (ki < 0 || ki >= h->mesh.nk || kj < 0 || kj >= h->mesh.nk)
        return dmk_fail(ctx, DMK_ERR_INVALID, "eri_push_ring_slot: bad arguments");
    if (h->ring_pending != h->pending)
        return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_ring_slot: ring slots and directly pushed blocks cannot share a group");
    const int slot = h->pending;
    h->slot_reserved = -1;
    h->pend_ki[slot] = ki;
    h->pend_kj[slot] = kj;
    h->pend_sym[slot] = symmetrise ? 1 : 0;
    h->pending += 1;
    h->ring_pending += 1;
    if (h->gen_pending) DMK_HIP(ctx, hipEventRecord(h->ev_gen[h->fill_half], h->gen_stream));   // everything produced so far for this group
    h->flops_half += (double)h->spin * (8.0 * h->naux * (double)h->nao * h->nao * h->nemb +
                                        8.0 * h->naux * (double)h->nao * h->nemb * h->nemb);
    if (h->pending == h->group) return eri_flush(h);
    return DMK_OK;
}

int dmk_eri_push_resident(dmk_eri *h, const void *blocks, int nblk, const int32_t *ki, const int32_t *kj, const int32_t *symmetrise) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    if (h->cur_kL < 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_resident: no kL in progress");
    if (!h->ring || h->group <= 1) return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_resident: this shape has no grouped hot path (dmk_eri_block_ring)");
    if (!blocks || !ki || !kj || !symmetrise || nblk < 1 || nblk > h->group)
        return dmk_fail(ctx, DMK_ERR_INVALID, "eri_push_resident: bad arguments (1 <= nblk <= %d queue slots)", h->group);
    if ((reinterpret_cast<uintptr_t>(blocks) & 15) != 0) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_push_resident: blocks must be 16-byte aligned");
    if (h->slot_reserved >= 0)                          // the resident launch resets the ring's producer state: the reservation would be lost
        return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_resident: ring slot %d was handed out (dmk_eri_ring_slot) and not pushed yet",
                        h->slot_reserved);
    if (h->pending != 0) {                              // a resident group is a launch of its own
        int rcf = eri_flush(h);
        if (rcf) return rcf;
    }
    for (int b = 0; b < nblk; ++b) {
        if (ki[b] < 0 || ki[b] >= h->mesh.nk || kj[b] < 0 || kj[b] >= h->mesh.nk)
            return dmk_fail(ctx, DMK_ERR_INVALID, "eri_push_resident: k index out of range");
        h->pend_ki[b] = ki[b];
        h->pend_kj[b] = kj[b];
        h->pend_sym[b] = symmetrise[b] ? 1 : 0;
    }
    h->pending = nblk;
    h->ring_pending = nblk;
    h->resident_src = reinterpret_cast<const double2 *>(blocks);
    h->flops_half += (double)nblk * h->spin * (8.0 * h->naux * (double)h->nao * h->nao * h->nemb +
                                               8.0 * h->naux * (double)h->nao * h->nemb * h->nemb);
    return eri_flush(h);
}

int dmk_eri_flush(dmk_eri *h) {
    if (!h) return DMK_ERR_INVALID;
    if (h->cur_kL < 0) return dmk_fail(h->ctx, DMK_ERR_STATE, "eri_flush: no kL in progress");
    return eri_flush(h);
}

int dmk_eri_planes(dmk_eri *h, double **planes_out, int64_t *elems_out) {
    if (!h || !planes_out) return DMK_ERR_INVALID;
    {
        int rcf = eri_flush(h);      // queued blocks must land before anyone looks at the planes
        if (rcf) return rcf;
        if (h->cur_kL >= 0) {
            rcf = eri_fold_subplanes(h, true);
            if (rcf) return rcf;
        }
    }
    if (h->nslots > 1) return dmk_fail(h->ctx, DMK_ERR_STATE, "eri_planes: with a plane stack the spin planes of a kL are not contiguous");
    if (elems_out) *elems_out = (int64_t)h->spin * 2 * h->naux * h->npair;
    if (h->pr == h->naux && h->pl == h->npair) {
        *planes_out = h->planes;
        return DMK_OK;
    }
    // padded plane geometry: the caller is handed the documented (spin, 2, naux, npair) array, gathered into a buffer of the pipeline
    dmk_ctx *ctx = h->ctx;
    const size_t want = (size_t)h->spin * 2 * h->naux * h->npair * sizeof(double);
    if (h->planes_view_bytes < want) {
        if (h->planes_view) (void)hipFree(h->planes_view);
        h->planes_view = nullptr;
        h->planes_view_bytes = 0;
        if (dmk_dev_alloc(ctx, reinterpret_cast<void **>(&h->planes_view), want) != hipSuccess)
            return dmk_fail(ctx, DMK_ERR_NOMEM, "eri_planes: no memory for the compact copy (%zu bytes)", want);
        h->planes_view_bytes = want;
    }
    for (int s = 0; s < h->spin; ++s)
        for (int ri = 0; ri < 2; ++ri)
            DMK_HIP(ctx, hipMemcpy2DAsync(h->planes_view + ((size_t)s * 2 + ri) * h->naux * h->npair, (size_t)h->npair * sizeof(double),
                                          h->slot_planes(0, s) + (size_t)ri * h->pr * h->pl, (size_t)h->pl * sizeof(double),
                                          (size_t)h->npair * sizeof(double), (size_t)h->naux, hipMemcpyDeviceToDevice, ctx->stream));
    *planes_out = h->planes_view;
    return DMK_OK;
}

namespace {
// out[b][c][r] = conj(in[b][r][c]): a block stored for the swapped k-point pair (kj, ki) becomes the block of (ki, kj)
// (eri_transform.py:213-224 serves it as Lpq.conj().transpose(0, 2, 1) on the host: one more pass over 512 MB per block there)
__global__ void conj_transpose_c128_kernel(int n, const double2 *__restrict__ in, double2 *__restrict__ out) {
    __shared__ double2 tile[32][33];
    const size_t boff = (size_t)blockIdx.z * n * n;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int y = threadIdx.y; y < 32; y += blockDim.y) {
        const int r = r0 + y, c = c0 + threadIdx.x;
        if (r < n && c < n) tile[y][threadIdx.x] = in[boff + (size_t)r * n + c];
    }
    __syncthreads();
    for (int y = threadIdx.y; y < 32; y += blockDim.y) {
        const int c = c0 + y, r = r0 + threadIdx.x;
        if (r < n && c < n) {
            const double2 v = tile[threadIdx.x][y];
            out[boff + (size_t)c * n + r] = make_double2(v.x, -v.y);
        }
    }
}
}  // namespace

int dmk_eri_push_block_host(dmk_eri *h, int ki, int kj, int symmetrise, const void *Lpq_host, int slot) {
    if (!h) return DMK_ERR_INVALID;
    dmk_ctx *ctx = h->ctx;
    const bool swapped = (symmetrise & 2) != 0;        // the host buffer holds the block of the pair (kj, ki)
    symmetrise &= 1;
    if (slot < 0 || slot > 1 || !Lpq_host) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_push_block_host: bad arguments");
    if (h->cur_kL < 0) return dmk_fail(ctx, DMK_ERR_STATE, "eri_push_block_host: no kL in progress");
    const size_t bytes = (size_t)h->naux * h->nao * h->nao * sizeof(double2);
    if (!h->copy_stream) {
        DMK_HIP(ctx, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            DMK_HIP(ctx, dmk_dev_alloc(ctx, reinterpret_cast<void **>(&h->dstage[i]), bytes));
            DMK_HIP(ctx, hipEventCreateWithFlags(&h->ev_copied[i], hipEventDisableTiming));
            DMK_HIP(ctx, hipEventCreateWithFlags(&h->ev_consumed[i], hipEventDisableTiming));
            DMK_HIP(ctx, hipEventRecord(h->ev_consumed[i], ctx->stream));
        }
    }
    // the staging block is free once the step-1 launches that read it have run
    DMK_HIP(ctx, hipStreamWaitEvent(h->copy_stream, h->ev_consumed[slot], 0));
    DMK_HIP(ctx, hipMemcpyAsync(h->dstage[slot], Lpq_host, bytes, hipMemcpyHostToDevice, h->copy_stream));
    DMK_HIP(ctx, hipEventRecord(h->ev_copied[slot], h->copy_stream));
    DMK_HIP(ctx, hipStreamWaitEvent(ctx->stream, h->ev_copied[slot], 0));
    if (swapped) {
        // conjugate-transpose on the device into a third block; the staging slot is free again as soon as that kernel has run
        if (!h->tstage) DMK_HIP(ctx, dmk_dev_alloc(ctx, reinterpret_cast<void **>(&h->tstage), bytes));
        if (h->naux > 65535) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_push_block_host: naux too large for the device transpose");
        {
            FamScope fs(ctx, DMK_FAM_MISC);
            dim3 grid((h->nao + 31) / 32, (h->nao + 31) / 32, h->naux), block(32, 8);
            hipLaunchKernelGGL(conj_transpose_c128_kernel, grid, block, 0, ctx->stream, h->nao, h->dstage[slot], h->tstage);
            DMK_CHECK_LAUNCH(ctx);
        }
        DMK_HIP(ctx, hipEventRecord(h->ev_consumed[slot], ctx->stream));
        return dmk_eri_push_block(h, ki, kj, symmetrise, h->tstage);
    }
    int rc = dmk_eri_push_block(h, ki, kj, symmetrise, h->dstage[slot]);
    if (rc) return rc;
    DMK_HIP(ctx, hipEventRecord(h->ev_consumed[slot], ctx->stream));
    return DMK_OK;
}

int dmk_eri_host_slot_wait(dmk_eri *h, int slot) {
    if (!h) return DMK_ERR_INVALID;
    if (slot < 0 || slot > 1) return dmk_fail(h->ctx, DMK_ERR_INVALID, "eri_host_slot_wait: bad slot");
    if (h->copy_stream) DMK_HIP(h->ctx, hipEventSynchronize(h->ev_copied[slot]));
    return DMK_OK;
}

int dmk_host_alloc(dmk_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return DMK_ERR_INVALID;
    *out = nullptr;
    DMK_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return DMK_OK;
}

int dmk_host_free(dmk_ctx *ctx, void *p) {
    if (!ctx) return DMK_ERR_INVALID;
    if (p) DMK_HIP(ctx, hipHostFree(p));
    return DMK_OK;
}

int dmk_eri_finish(dmk_eri *h) {
    if (!h) return DMK_OK;
    dmk_ctx *ctx = h->ctx;
    int rc_stack = DMK_OK;
    // planes still waiting for their contraction (a rows-only pipeline just drops them: its caller took the rows it wanted)
    if (h->eri && h->nslots > 1 && h->cur_kL < 0 && h->n_w2 + h->n_w1 > 0) {
        rc_stack = eri_contract_stack(h, -1, -1);
        h->n_w2 = h->n_w1 = 0;
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (h->copy_stream) {
        (void)hipStreamSynchronize(h->copy_stream);
        for (int i = 0; i < 2; ++i) {
            (void)hipFree(h->dstage[i]);
            (void)hipEventDestroy(h->ev_copied[i]);
            (void)hipEventDestroy(h->ev_consumed[i]);
        }
        (void)hipStreamDestroy(h->copy_stream);
        if (h->tstage) (void)hipFree(h->tstage);
    }
    if (h->gen_stream) {
        (void)hipStreamSynchronize(h->gen_stream);
        for (int i = 0; i < 2; ++i) {
            if (h->ev_gen[i]) (void)hipEventDestroy(h->ev_gen[i]);
            if (h->ev_free[i]) (void)hipEventDestroy(h->ev_free[i]);
        }
        (void)hipStreamDestroy(h->gen_stream);
    }
    if (h->ring) {
        if (!ctx->eri_ws[2]) {
            ctx->eri_ws[2] = h->ring;
            ctx->eri_ws_bytes[2] = h->ring_bytes;
        } else {
            (void)hipFree(h->ring);
        }
    }
    if (h->imag) (void)hipFree(h->imag);
    if (h->sub_planes) (void)hipFree(h->sub_planes);
    if (h->Cpad) (void)hipFree(h->Cpad);
    if (h->planes_view) (void)hipFree(h->planes_view);
    void *mine[2] = {h->planes, h->Ut};
    for (int w = 0; w < 2; ++w) {
        if (!mine[w]) continue;
        if (!ctx->eri_ws[w]) {                      // park it for the next pipeline
            ctx->eri_ws[w] = mine[w];
            ctx->eri_ws_bytes[w] = h->ws_bytes[w];
        } else {
            (void)hipFree(mine[w]);
        }
    }
    delete h;
    return rc_stack;
}

int dmk_eri_flops(const dmk_eri *h, double f[2]) {
    if (!h || !f) return DMK_ERR_INVALID;
    f[0] = h->flops_half;
    f[1] = h->flops_contract;
    return DMK_OK;
}

}  // extern "C"

// =============================================================================================
// small utility kernels: transpose, restore
// =============================================================================================

namespace {

__global__ void transpose_c128_kernel(int rows, int cols, const double2 *__restrict__ in, double2 *__restrict__ out) {
    __shared__ double2 tile[32][33];
    const size_t boff = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int y = threadIdx.y; y < 32; y += blockDim.y) {
        const int r = r0 + y, c = c0 + threadIdx.x;
        if (r < rows && c < cols) tile[y][threadIdx.x] = in[boff + (size_t)r * cols + c];
    }
    __syncthreads();
    for (int y = threadIdx.y; y < 32; y += blockDim.y) {
        const int c = c0 + y, r = r0 + threadIdx.x;
        if (r < rows && c < cols) out[boff + (size_t)c * rows + r] = tile[threadIdx.x][y];
    }
}

// 4-fold (npair x npair) -> 1-fold (n^4): out[i][j][k][l] = eri4[pair(i,j)][pair(k,l)]
__global__ void restore_4to1_kernel(int n, const double *__restrict__ e4, double *__restrict__ out) {
    const long long total = (long long)n * n * n * n;
    const long long npair = (long long)n * (n + 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(t % n), k = (int)((t / n) % n), j = (int)((t / ((long long)n * n)) % n),
                  i = (int)(t / ((long long)n * n * n));
        const long long ij = i >= j ? (long long)i * (i + 1) / 2 + j : (long long)j * (j + 1) / 2 + i;
        const long long kl = k >= l ? (long long)k * (k + 1) / 2 + l : (long long)l * (l + 1) / 2 + k;
        out[t] = e4[ij * npair + kl];
    }
}

// 4-fold -> 8-fold: out[p(p+1)/2 + q] = eri4[p][q], p >= q
__global__ void restore_4to8_kernel(long long npair, const double *__restrict__ e4, double *__restrict__ out) {
    const long long total = npair * (npair + 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        long long p = (long long)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while (p * (p + 1) / 2 > t) --p;
        while ((p + 1) * (p + 2) / 2 <= t) ++p;
        const long long q = t - p * (p + 1) / 2;
        out[t] = e4[p * npair + q];
    }
}

}  // namespace

extern "C" {

int dmk_transpose_c128(dmk_ctx *ctx, int rows, int cols, int batch, const void *in, void *out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (rows <= 0 || cols <= 0 || batch <= 0 || !in || !out || batch > 65535)
        return dmk_fail(ctx, DMK_ERR_INVALID, "transpose: bad arguments");
    FamScope fs(ctx, DMK_FAM_MISC);
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch), block(32, 8);
    hipLaunchKernelGGL(transpose_c128_kernel, grid, block, 0, ctx->stream, rows, cols,
                       reinterpret_cast<const double2 *>(in), reinterpret_cast<double2 *>(out));
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}

int dmk_eri_restore(dmk_ctx *ctx, int nemb, int symmetry, const double *eri4, double *out) {
    if (!ctx) return DMK_ERR_INVALID;
    if (nemb <= 0 || !eri4 || !out) return dmk_fail(ctx, DMK_ERR_INVALID, "eri_restore: bad arguments");
    const long long npair = (long long)nemb * (nemb + 1) / 2;
    FamScope fs(ctx, DMK_FAM_MISC);
    if (symmetry == 4) {
        DMK_HIP(ctx, hipMemcpyAsync(out, eri4, (size_t)npair * npair * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    } else if (symmetry == 1) {
        hipLaunchKernelGGL(restore_4to1_kernel, dim3(2048), dim3(256), 0, ctx->stream, nemb, eri4, out);
        DMK_CHECK_LAUNCH(ctx);
    } else if (symmetry == 8) {
        hipLaunchKernelGGL(restore_4to8_kernel, dim3(2048), dim3(256), 0, ctx->stream, npair, eri4, out);
        DMK_CHECK_LAUNCH(ctx);
    } else {
        return dmk_fail(ctx, DMK_ERR_INVALID, "eri_restore: symmetry must be 1, 4 or 8");
    }
    return DMK_OK;
}

}  // extern "C"

// ---- row gather / scatter (k-sharded mean field: this rank's k rows of the resident Fock batch; eigenvalues of a shard
//      placed into the all-k table before the all-reduce) ------------------------------------------------------------
namespace {
__global__ void copy_rows_kernel(long long nrows, long long row_len, const int *__restrict__ idx, const double *__restrict__ in,
                                 double *__restrict__ out, int scatter) {
    const long long total = nrows * row_len;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long r = t / row_len, c = t - r * row_len;
        const long long far = (long long)idx[r] * row_len + c;
        if (scatter) out[far] = in[t];
        else out[t] = in[far];
    }
}
}  // namespace

extern "C" int dmk_copy_rows_f64(dmk_ctx *ctx, int64_t nrows, int64_t row_len, const int32_t *idx_dev, const double *in, double *out,
                                 int scatter) {
    if (!ctx) return DMK_ERR_INVALID;
    if (nrows < 0 || row_len < 0 || !idx_dev || !in || !out) return dmk_fail(ctx, DMK_ERR_INVALID, "copy_rows: bad arguments");
    if (nrows == 0 || row_len == 0) return DMK_OK;
    const long long total = (long long)nrows * row_len;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 65536);
    FamScope fs(ctx, DMK_FAM_MISC);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, (long long)nrows, (long long)row_len, idx_dev, in, out,
                       scatter ? 1 : 0);
    DMK_CHECK_LAUNCH(ctx);
    return DMK_OK;
}
