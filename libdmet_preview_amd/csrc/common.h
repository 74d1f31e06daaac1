// Shared internals of libdmetk (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/libdmetk.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

struct dmk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // HIP-event timer (dmk_timer_start/stop)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // per-family profiling
    bool profile = false;
    double fam_ms[DMK_FAM_COUNT] = {0};
    int64_t fam_launches[DMK_FAM_COUNT] = {0};
    double fam_mfma_flops[DMK_FAM_COUNT] = {0};   // flop ISSUED to the matrix pipe (tiles launched x MFMAs per tile x 512)
    struct Pending { int fam; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool;
    // scratch owned by the context (grown on demand)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // second scratch for callers that run the eigensolver (which uses `scratch`) on data they keep alive across it
    void *scratch2 = nullptr;
    size_t scratch2_bytes = 0;
    // cached twiddle matrices for the folds (device), keyed by mesh + direction
    struct Phase { int mesh[3]; int dir; int nsub; std::vector<int32_t> subset; void *dev; };
    std::vector<Phase> phases;
    // workspace of the last ERI pipeline (plane set + Ut slots, several GB): kept across dmk_eri_finish /
    // dmk_eri_begin so that a self-consistency loop does not pay hipMalloc of it (~0.25 s) every iteration
    void *eri_ws[3] = {nullptr, nullptr, nullptr};   // planes, Ut, AO-block ring: parked between pipelines
    size_t eri_ws_bytes[3] = {0, 0, 0};
    // tile visiting orders of the contraction kernel (dgemm_tn.hip), one per (tiles_m, tiles_n, symm)
    struct TileTable { int tiles_m, tiles_n, symm, lo, hi; unsigned count; unsigned *dev; };
    std::vector<TileTable> tile_tables;
    // block-ownership tables of the general-nemb step-2 kernel (zhot_tab.hip), one per embedding dimension
    struct StepTable { int nemb, cfg, nitems; double useful_blocks, folded_blocks; int *dev; };
    std::vector<StepTable> step2_tables;
    // caller-side cache release (dmk_set_oom_hook): the host binding parks freed device blocks in a pool the library cannot
    // see; before any allocation inside the library is reported as failed the hook is asked to give that memory back
    void (*oom_hook)(void *) = nullptr;
    void *oom_user = nullptr;
    // warm-started eigensolver (jacobi_eigh.hip): bookkeeping of the refinement fast path
    unsigned long long fit_seq = 0;          // sequence number of the pinned result record of dmk_fit_objective
    int refine_streak = 0, refine_skip = 0;
    long long refine_ok = 0, refine_failed = 0;
};

// hipMalloc with one retry after the out-of-memory hook; hipSuccess or the error of the second attempt
hipError_t dmk_dev_alloc(dmk_ctx *ctx, void **out, size_t bytes);

int dmk_fail(dmk_ctx *ctx, int code, const char *fmt, ...);

#define DMK_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return dmk_fail((ctx), DMK_ERR_HIP, "%s failed: %s (%s:%d)", #call,         \
                            hipGetErrorString(e__), __FILE__, __LINE__);                \
    } while (0)

#define DMK_CHECK_LAUNCH(ctx)                                                           \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess)                                                          \
            return dmk_fail((ctx), DMK_ERR_HIP, "kernel launch failed: %s (%s:%d)",     \
                            hipGetErrorString(e__), __FILE__, __LINE__);                \
    } while (0)

// RAII bracket that (optionally) times one kernel family launch with HIP events.
struct FamScope {
    dmk_ctx *ctx; int fam; hipEvent_t a = nullptr, b = nullptr;
    hipStream_t on; bool on_set = false;          // events go to ctx->stream unless the launch sits on another stream
    FamScope(dmk_ctx *c, int f);
    FamScope(dmk_ctx *c, int f, hipStream_t stream);
    ~FamScope();
    void begin();
    // flop this launch issues to the f64 matrix pipe (executed, not algorithmic: 3M complex products, padded tiles,
    // the lower tile triangle of a symmetric contraction); read back by dmk_profile_read_flops
    void mfma_flops(double f) { if (ctx) ctx->fam_mfma_flops[fam] += f; }
};

int dmk_scratch(dmk_ctx *ctx, size_t bytes, void **out);
// occ.hip: T = 0 occupations of several spectra in one launch, nothing read back
int dmk_assign_occ_zero_t_batch(dmk_ctx *ctx, int64_t n, int batch, const double *ew, const double *nelec_host,
                                const double *mu0_host, int flags, double thr_deg, double *occ, double *info_dev);
// jacobi_eigh.hip: warm Ogita-Aishima refinement, enqueued without any host read-back (used by dmk_fit_objective)
int dmk_eigh_refine_enqueue(dmk_ctx *ctx, int n, int batch, const double *A, const double *V0, double *w, double *Vt, int npass,
                            int *verdict_dev, int norm_done = 0);
// scratch layout of the refinement (jacobi_eigh.hip): anorm / state / arrive may be produced by the kernel that writes A
struct RfWorkspace {
    double *Vb[2], *T1, *S, *G, *F, *stats, *lam, *anorm, *part, *sqS, *sqG;
    int *state;
    unsigned *arrive;
};
int rf_workspace(dmk_ctx *ctx, int n, int batch, RfWorkspace *W);
int dmk_scratch2(dmk_ctx *ctx, size_t bytes, void **out);

// XCD-aware, bijective remap of a 1-D block id: blocks that the dispatcher places on the
// same XCD (id % 8) receive a contiguous range of logical ids, so that neighbouring tiles
// share that XCD's L2 (cdna_hip_programming.md T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks >> 3, r = nblocks & 7u;
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// LDS-DMA: 16 bytes per lane from `gsrc` (per-lane global address) to LDS byte address
// `lds_base` (wave-uniform) + 16 * lane.  Issued from inline asm on purpose: hipcc's waitcnt pass does
// not see it, so it never drains the counted vmcnt ring with a conservative s_waitcnt vmcnt(0) in front
// of the next ds_read; the caller retires it with its own s_waitcnt vmcnt(N) + s_barrier
// (cdna_hip_programming.md section 5.7: M0 is written and restored inside the same statement).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void dmk_lds_void_t;
__device__ __forceinline__ unsigned lds_addr_of(const void *p) {
    return (unsigned)(size_t)(dmk_lds_void_t *)p;
}
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_base) {
    unsigned keep;
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(base)
                 : "memory");
}
// The same for a burst of pieces of one wave: M0 is saved and restored ONCE around the whole burst and only rewritten between
// the loads (two scalar moves and one readfirstlane less per piece than repeated glds16 calls).
__device__ __forceinline__ void glds16_x4(const void *g0, const void *g1, const void *g2, const void *g3, unsigned b0, unsigned b1,
                                           unsigned b2, unsigned b3) {
    unsigned keep;
    b0 = __builtin_amdgcn_readfirstlane(b0); b1 = __builtin_amdgcn_readfirstlane(b1);
    b2 = __builtin_amdgcn_readfirstlane(b2); b3 = __builtin_amdgcn_readfirstlane(b3);
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(b0), "s"(b1), "s"(b2), "s"(b3)
                 : "memory");
}
__device__ __forceinline__ void glds16_x6(const void *g0, const void *g1, const void *g2, const void *g3, const void *g4, const void *g5,
                                           unsigned b0, unsigned b1, unsigned b2, unsigned b3, unsigned b4, unsigned b5) {
    unsigned keep;
    b0 = __builtin_amdgcn_readfirstlane(b0); b1 = __builtin_amdgcn_readfirstlane(b1); b2 = __builtin_amdgcn_readfirstlane(b2);
    b3 = __builtin_amdgcn_readfirstlane(b3); b4 = __builtin_amdgcn_readfirstlane(b4); b5 = __builtin_amdgcn_readfirstlane(b5);
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
                 "s_mov_b32 m0, %12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "v"(g5), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5)
                 : "memory");
}
// SADDR form of the same bursts: the source address of a piece is a wave-uniform 64-bit base in SGPRs plus a per-lane 32-bit byte
// offset in ONE VGPR (`global_load_lds_dwordx4 voff, s[base:base+1]`).  The offsets of a ring kernel's pieces are loop invariant and
// its running tile pointers are scalar, so a K step issues its LDS-DMA pieces WITHOUT any vector ALU work -- the per-piece 64-bit
// v_add_co / v_addc (and the v_cndmask pair that picked the operand) of the per-lane-pointer form were 8-24 VALU instructions per
// K step and wave in front of the MFMA stream.  `s_nop 4` opens the burst: a base that the compiler produced with v_readfirstlane
// needs five wait states before a VMEM instruction may read it, and nothing pads hazards inside an asm statement.
__device__ __forceinline__ void glds16s_x4(unsigned o0, unsigned o1, unsigned o2, unsigned o3, const void *s0, const void *s1,
                                            const void *s2, const void *s3, unsigned b0, unsigned b1, unsigned b2, unsigned b3) {
    unsigned keep;
    b0 = __builtin_amdgcn_readfirstlane(b0); b1 = __builtin_amdgcn_readfirstlane(b1);
    b2 = __builtin_amdgcn_readfirstlane(b2); b3 = __builtin_amdgcn_readfirstlane(b3);
    asm volatile("s_mov_b32 %0, m0\n\ts_nop 4\n\t"
                 "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
                 "s_mov_b32 m0, %10\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
                 "s_mov_b32 m0, %11\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %7\n\t"
                 "s_mov_b32 m0, %12\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %8\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(b0), "s"(b1), "s"(b2), "s"(b3)
                 : "memory");
}
__device__ __forceinline__ void glds16s_x6(unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned o4, unsigned o5, const void *s0,
                                            const void *s1, const void *s2, const void *s3, const void *s4, const void *s5, unsigned b0,
                                            unsigned b1, unsigned b2, unsigned b3, unsigned b4, unsigned b5) {
    unsigned keep;
    b0 = __builtin_amdgcn_readfirstlane(b0); b1 = __builtin_amdgcn_readfirstlane(b1); b2 = __builtin_amdgcn_readfirstlane(b2);
    b3 = __builtin_amdgcn_readfirstlane(b3); b4 = __builtin_amdgcn_readfirstlane(b4); b5 = __builtin_amdgcn_readfirstlane(b5);
    asm volatile("s_mov_b32 %0, m0\n\ts_nop 4\n\t"
                 "s_mov_b32 m0, %13\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %7\n\t"
                 "s_mov_b32 m0, %14\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %8\n\t"
                 "s_mov_b32 m0, %15\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %9\n\t"
                 "s_mov_b32 m0, %16\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %10\n\t"
                 "s_mov_b32 m0, %17\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %11\n\t"
                 "s_mov_b32 m0, %18\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %12\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "v"(o4), "v"(o5), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(b0),
                   "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5)
                 : "memory");
}
// Wave64 sum through the DPP crossbar (quad_perm, row_ror, row_bcast) instead of six ds_bpermute round trips:
// every lane of the wave gets the total (read back from lane 63).  Fixed combination order: deterministic.
// The whole wave must be active at the call.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dmk_dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dmk_wave_sum(double v) {
    v += dmk_dpp_mov<0xb1, 0xf>(v);      // quad_perm [1,0,3,2]
    v += dmk_dpp_mov<0x4e, 0xf>(v);      // quad_perm [2,3,0,1]
    v += dmk_dpp_mov<0x124, 0xf>(v);     // row_ror 4
    v += dmk_dpp_mov<0x128, 0xf>(v);     // row_ror 8   -> every lane holds its row's total
    v += dmk_dpp_mov<0x142, 0xa>(v);     // row_bcast 15 into rows 1, 3
    v += dmk_dpp_mov<0x143, 0xc>(v);     // row_bcast 31 into rows 2, 3 -> lane 63 holds the wave total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
#else
__device__ unsigned lds_addr_of(const void *p);          // host pass: declarations only
__device__ void glds16(const void *gsrc, unsigned lds_base);
__device__ void glds16_x4(const void *, const void *, const void *, const void *, unsigned, unsigned, unsigned, unsigned);
__device__ void glds16_x6(const void *, const void *, const void *, const void *, const void *, const void *, unsigned, unsigned, unsigned,
                          unsigned, unsigned, unsigned);
__device__ void glds16s_x4(unsigned, unsigned, unsigned, unsigned, const void *, const void *, const void *, const void *, unsigned,
                           unsigned, unsigned, unsigned);
__device__ void glds16s_x6(unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, const void *, const void *, const void *,
                           const void *, const void *, const void *, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned);
__device__ double dmk_wave_sum(double v);
#endif

// ---- launchers implemented in the .hip files ---------------------------------------------

// C (M x N, ldc) += alpha * X^T Y;  X: K x M (ldx), Y: K x N (ldy)
int launch_dgemm_tn_acc(dmk_ctx *ctx, int M, int N, int K, double alpha, const double *X,
                        int64_t ldx, const double *Y, int64_t ldy, double *C, int64_t ldc);
// K as a stack of row segments and / or the output restricted to a band of 128-wide tiles (dgemm_tn.hip)
// Mp / Np (0: M / N): columns of X / Y that may be loaded, zero beyond M / N -- an odd M padded to the even row length of the planes
int launch_dgemm_tn_acc_seg(dmk_ctx *ctx, int M, int N, int K, double alpha, const double *X, int64_t ldx, const double *Y,
                            int64_t ldy, double *C, int64_t ldc, int seg_rows, int64_t seg_stride_x, int64_t seg_stride_y,
                            int band_lo, int band_hi, int Mp = 0, int Np = 0);

struct ZSeg {
    const void *A = nullptr;   // complex (or real if a_real) operand A
    const void *B = nullptr;
    int64_t lda = 0, ldb = 0;          // leading dimension in ELEMENTS
    int64_t strideA = 0, strideB = 0;  // batch stride in elements
    int a_kmajor = 1, b_kmajor = 1;    // 1: element (k, m) at k*ld + m ; 0: at m*ld + k
    int conjA = 0, conjB = 0;
    int b_real = 0;                    // B holds f64 (imaginary part zero)
    const double *kscaleB = nullptr;   // optional per-k real scale of B (batch stride K)
};

enum { ZEPI_STORE = 0, ZEPI_STORE_REAL = 1, ZEPI_PACK_ACC = 2 };

struct ZGemm {
    int M = 0, N = 0, K = 0, batch = 1;
    int nseg = 1;
    ZSeg seg[2];
    double alpha = 1.0;
    int flatten_m = 0;      // batch folded into the M-block list (B must be batch invariant)
    int epi = ZEPI_STORE;
    void *C = nullptr;      // ZEPI_STORE: c128 [batch][M][ldc] ; ZEPI_STORE_REAL: f64
    int64_t ldc = 0, strideC = 0;
    double *imag_max = nullptr;   // ZEPI_STORE_REAL: atomic max |Im|
    // ZEPI_PACK_ACC: planes[(ri*naux + batch)*npair + a(a+1)/2 + b] += value, a >= b
    double *planes = nullptr;
    int64_t naux = 0, npair = 0;
    int lower_only = 0;     // enumerate only tiles that touch a >= b
    int big_tile = 0;       // 128x64 instead of 64x64 workgroup tile
    int use_3m = 0;         // 3 real MFMAs per complex tile step (Karatsuba) instead of 4
};
int launch_zgemm(dmk_ctx *ctx, const ZGemm &g, int fam);

// fused mixed-radix k <-> R fold of a full mesh (fold.hip): 1 = handled, 0 = use the DFT-GEMM, < 0 = error
int launch_fold_fft(dmk_ctx *ctx, const int n[3], long long ncol, int batch, const void *in, int in_real, void *out, int out_real,
                    int inverse, double *imag_max);

// Freivalds probe of the contraction (eri_probe.hip)
int launch_eri_probe_slot(dmk_ctx *ctx, const double *X0, const double *X1, int nrows, long long npair, long long ld, double w, const double *x,
                          double *yref, double *twork);

int launch_philox_block(dmk_ctx *ctx, uint64_t seed, int ki, int kj, int naux, int nao, void *out);
int launch_philox_block_on(dmk_ctx *ctx, hipStream_t stream, uint64_t seed, int ki, int kj, int naux, int nao, void *out);
int launch_philox_blocks_on(dmk_ctx *ctx, hipStream_t stream, uint64_t seed, int nblk, const int *ij, int naux, int nao, void *out,
                            long long stride_bytes);

// hot half-transform kernels (zhot.hip): return 1 if handled, 0 if the generic kernel must be used.
// kdim (0: nao, which must then be a multiple of the K tile) is the K loop bound hot_kdim(nao) for AO dimensions off the K tile:
// the C operands then hold kdim rows per k point, ZERO beyond nao (stride kdim * nemb), and the Ut buffer of step 2 is the
// pipeline's own, initialised one (its rows of the padding are read against those zeros).
int hot_kdim(int nao);
int half1_hot_max_rows(int nao);     // auxiliary rows per step-1 launch (32-bit lane offsets: < 4 GiB of the AO block)
int launch_half1_hot(dmk_ctx *ctx, const void *Lpq, const void *Ci, void *Ut, int nL, int nao, int nemb, int nspin = 1,
                     long long ci_spin_stride = 0, long long ut_spin_stride = 0, int kdim = 0);
int launch_half1_hot_multi(dmk_ctx *ctx, const void *Lpq, long long a_slot_stride, int nslot, const int *ki, const void *C,
                           void *Ut, long long ut_slot_stride, int nL, int nao, int nemb, int nspin, long long ci_spin_stride,
                           long long ut_spin_stride, int kdim = 0);
int launch_half2_hot(dmk_ctx *ctx, const void *Ut, long long slot_stride, int nslot, const void *const *Cj,
                     const int *sym, double *planes, long long naux, long long npair, int nL, int nao, int nemb, int nspin,
                     long long ut_spin_stride, long long cj_spin_stride, long long planes_spin_stride, int kdim = 0, int re_only = 0);
int half2_hot_usable(int nao, int nemb);
int half2_hot_maxslot();
int half1_hot_usable(int nL, int nao, int nemb);
// step 2 for a general embedding dimension (zhot_tab.hip); same arguments and return convention as launch_half2_hot
// nsub > 1: the queue is cut into nsub runs with one workgroup per (L, item, run); run p >= 1 accumulates into
// planes_sub + (p - 1) * sub_stride ([spin][2 naux][npair] each), which the caller adds to `planes` afterwards
int launch_half2_tab(dmk_ctx *ctx, const void *Ut, long long slot_stride, int nslot, const void *const *Cj,
                     const int *sym, double *planes, long long naux, long long npair, int nL, int nao, int nemb, int nspin,
                     long long ut_spin_stride, long long cj_spin_stride, long long planes_spin_stride, int nsub = 1,
                     double *planes_sub = nullptr, long long sub_stride = 0, int kdim = 0, int re_only = 0);
int half2_tab_subgroups(dmk_ctx *ctx, int nL, int nao, int nemb, int nspin, int nslot, int max_sub);
int half2_tab_usable(int nao, int nemb);
int half2_tab_maxslot();
