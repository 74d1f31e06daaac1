/*
 * libdmetk.h -- C ABI of the MI355X-native libDMET embedding-construction path.
 *
 * The reference (gkclab/libdmet_preview) is pure Python and has no FFI of its
 * own: the boundary a maintainer would bind is the set of
 * numpy-level primitives its hot functions reduce to.  Each entry point below
 * names the reference call site it replaces (paths relative to the reference
 * root).  INTEGRATION.md shows the ctypes stubs that rebind the reference's
 * Python entry points (HF, get_emb_basis, get_emb_eri, Lattice.k2R/R2k,
 * multiply_basis) onto these symbols.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types cross the ABI.
 *   - every function returns 0 on success or a negative dmk_status; the message
 *     is available from dmk_last_error(ctx).  Nothing throws across the ABI.
 *   - all array arguments are DEVICE pointers (HBM) unless the name ends in
 *     `_host`; row-major; complex = interleaved (re, im) f64; all work is
 *     enqueued on the context's stream and is asynchronous w.r.t. the host
 *     unless stated.  dmk_malloc/dmk_memcpy_* serve hosts without torch.
 *   - one context per host thread; contexts are not shared.
 *   - index bookkeeping (k-points, cells, time-reversal plan) is integer mesh
 *     arithmetic done on the host side of the library and exposed bit-exactly
 *     through dmk_kmesh_tables / dmk_eri_plan.
 */
#ifndef LIBDMETK_H
#define LIBDMETK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmk_ctx dmk_ctx;

typedef enum {
    DMK_OK = 0,
    DMK_ERR_INVALID = -1,     /* bad argument / shape                        */
    DMK_ERR_HIP = -2,         /* a HIP runtime call failed                   */
    DMK_ERR_NOMEM = -3,       /* device allocation failed                    */
    DMK_ERR_NOCONV = -4,      /* an iterative kernel did not converge        */
    DMK_ERR_STATE = -5        /* call sequence violated (eri begin/push/...) */
} dmk_status;

/* ------------------------------------------------------------------------- */
/* context, memory, diagnostics                                               */
/* ------------------------------------------------------------------------- */

/* Create a context on `device`.  `stream` is a hipStream_t to enqueue on (pass
 * NULL for the legacy default stream); it may be changed with dmk_set_stream. */
int dmk_init(int device, void *stream, dmk_ctx **out);
int dmk_destroy(dmk_ctx *ctx);
int dmk_set_stream(dmk_ctx *ctx, void *stream);
int dmk_sync(dmk_ctx *ctx);
/* Out-of-memory hook: called (with `user`) when a device allocation INSIDE the library fails, after the context stream
 * has been drained; the allocation is retried once when it returns.  The host binding registers the release of its
 * device-block cache here (libdmet_preview_amd/_lib.py Context.trim), memory the library cannot see otherwise. */
int dmk_set_oom_hook(dmk_ctx *ctx, void (*hook)(void *user), void *user);
const char *dmk_last_error(const dmk_ctx *ctx);
const char *dmk_version(void);

/* free / total device memory in bytes (hipMemGetInfo): sizes the plane stack of the ERI pipeline */
int dmk_mem_info(dmk_ctx *ctx, size_t *free_bytes_host, size_t *total_bytes_host);
int dmk_malloc(dmk_ctx *ctx, size_t bytes, void **out);
int dmk_free(dmk_ctx *ctx, void *p);      /* drains the context stream first */
int dmk_memset(dmk_ctx *ctx, void *p, int value, size_t bytes);
int dmk_memcpy_h2d(dmk_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int dmk_memcpy_d2h(dmk_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int dmk_memcpy_d2d(dmk_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);

/* HIP-event timing of everything enqueued on the context stream between the
 * two calls (bench.py's roofline leg).  dmk_timer_stop synchronises. */
int dmk_timer_start(dmk_ctx *ctx);
int dmk_timer_stop(dmk_ctx *ctx, double *ms_out_host);

/* Per-kernel-family accumulated HIP-event time since the last reset
 * (family ids: DMK_FAM_*).  Enabled with dmk_profile(ctx, 1); when enabled
 * every launch of the family is bracketed by events on the stream. */
enum { DMK_FAM_DGEMM = 0, DMK_FAM_ZGEMM_HALF1 = 1, DMK_FAM_ZGEMM_HALF2 = 2,
       DMK_FAM_PHILOX = 3, DMK_FAM_FOLD = 4, DMK_FAM_EIGH = 5, DMK_FAM_BATH = 6,
       DMK_FAM_ZGEMM_SMALL = 7, DMK_FAM_MISC = 8, DMK_FAM_JK = 9, DMK_FAM_FIT = 10, DMK_FAM_COUNT = 11 };
int dmk_profile(dmk_ctx *ctx, int enable);
int dmk_profile_read(dmk_ctx *ctx, double *ms_host /*[DMK_FAM_COUNT]*/,
                     int64_t *launches_host /*[DMK_FAM_COUNT]*/, int reset);
/* flop ISSUED to the f64 matrix pipe by each family since the last reset (tiles launched x MFMA instructions per
 * tile x 512): the numerator of bench.py's roofline fraction.  Differs from the algorithmic count of SURVEY.md
 * section 8d where a kernel restructures the arithmetic (3M complex product, tril-only step 2, symmetric
 * contraction) or pads tiles.  Counted on every launch, independent of dmk_profile. */
int dmk_profile_read_flops(dmk_ctx *ctx, double *flops_host /*[DMK_FAM_COUNT]*/, int reset);

/* ------------------------------------------------------------------------- */
/* a1 / a2 / a15 : k-point, cell and time-reversal bookkeeping (HOST, integer) */
/* replaces system/fourier.py:46-81 (make_kpts_scaled, round_to_FBZ,           */
/* kpt_member), system/lattice.py:194-204 (cell add/subtract),                 */
/* basis_transform/eri_transform.py:142-157 (get_weights_t_reversal),          */
/* routine/mfd_mpi.py:33-54 (get_kpairs_kidx),                                 */
/* basis_transform/eri_transform_mpi.py:27-55 (assign_workload).               */
/* ------------------------------------------------------------------------- */

/* All outputs are host arrays of length nk = mesh[0]*mesh[1]*mesh[2] (any may be
 * NULL): mesh integers of each k / cell (nk x 3), index of -k, TR weights
 * {1,2,0}. */
int dmk_kmesh_tables(const int mesh[3], int32_t *kint_host, int32_t *minus_k_host,
                     int32_t *weights_host);
/* j = conserving partner: k_j = k_i - k_L (mod mesh).  out: nk*nk, [kL*nk + i]. */
int dmk_kconserv_table(const int mesh[3], int32_t *j_of_kL_i_host);
/* cell index arithmetic: out[i*nk + j] = idx(R_i +/- R_j). */
int dmk_cell_add_table(const int mesh[3], int sign, int32_t *out_host);
/* scaled k-points in fftfreq order (nk x 3 f64), bit-identical to make_kpts_scaled. */
int dmk_kpts_scaled(const int mesh[3], double *kpts_host);
/* index of a scaled k-vector in the mesh modulo reciprocal lattice vectors;
 * returns the index or -1 (kpt_member with tol). */
int dmk_kpt_member(const int mesh[3], const double kpt[3], double tol);

/* Visiting plan of get_emb_eri_fast_gdf's double loop
 * (basis_transform/eri_transform.py:338-382).  Records are int32 quintuples
 * (kL, i, j, jm, symmetrise).  Call with plan_host = NULL to get the count. */
/* Time-reversal mask of a list of k-point pairs given as mesh indices (npairs, 2): basis_transform/eri_transform.py:1409-1427
 * (get_mask_kptij_lst).  mask[p] = index of the first later pair (-ki, -kj), -2 for a pair already claimed, -1 otherwise. */
int dmk_kptij_mask(const int mesh[3], int npairs, const int32_t *pairs, int32_t *mask);
int dmk_eri_plan(const int mesh[3], int t_reversal_symm, int32_t *plan_host,
                 int64_t capacity_records, int64_t *nrecords_out);
/* Static partition of irreducible kL over `nranks` (assign_workload).
 * kl_host receives the kL owned by `rank` (capacity nk); returns the count via n_out. */
int dmk_assign_workload(const int mesh[3], int t_reversal_symm, int nranks, int rank,
                        int32_t *kl_host, int *n_out);

/* ------------------------------------------------------------------------- */
/* a6 : k <-> R Fourier folds                                                  */
/* replaces system/fourier.py:160-177 (FFTtoK / FFTtoT), :129-158 (R2k / k2R)  */
/* ------------------------------------------------------------------------- */

/* out_k[k, c] = sum_R exp(-i k.R) in_R[R, c];  in real (in_is_complex=0) or
 * complex; ncol = product of trailing dims; batch = leading (spin) dim. */
int dmk_fold_R2k(dmk_ctx *ctx, const int mesh[3], int64_t ncol, int batch,
                 const void *in_R, int in_is_complex, void *out_k /* c128 */);
/* out_R[R, c] = Re[(1/N) sum_k exp(+i k.R) in_k[k, c]]; max |Im| of the full
 * result is written to *imag_max_dev (device f64, may be NULL) -- the quantity
 * the reference compares with IMAG_DISCARD_TOL (system/fourier.py:174-175).
 * k_subset/nsub (host, may be NULL/0) restrict the sum to a shard of k-points
 * (multi-GPU partial fold). */
int dmk_fold_k2R(dmk_ctx *ctx, const int mesh[3], int64_t ncol, int batch,
                 const void *in_k /* c128 */, double *out_R, double *imag_max_dev,
                 const int32_t *k_subset_host, int nsub);
/* complex -> complex variant of k2R (no real-part projection). */
int dmk_fold_k2R_complex(dmk_ctx *ctx, const int mesh[3], int64_t ncol, int batch,
                         const void *in_k, void *out_R);

/* ------------------------------------------------------------------------- */
/* a3 / a5 : batched Hermitian eigensolver and density build                   */
/* replaces scipy.linalg.eigh at routine/mfd.py:42-106 (DiagRHF/UHF[_symm])    */
/* and the loop rho[s,k] = (ev*occ) ev^H at routine/mfd.py:355-357             */
/* ------------------------------------------------------------------------- */

/* Complex batches with 64 < n <= 200 run as CU-resident kernels (csrc/eigh_tridiag.hip: tridiagonalisation with the
 * matrix in LDS + registers and the rank-2 update on the matrix cores, compact-WY back-transformation); other shapes
 * as one kernel per launch (csrc/eigh.hip).  DMK_ERR_NOCONV when an eigenvector fails its residual test (NaN / Inf).
 * A: batch x n x n c128 (Hermitian; only the LOWER triangle is referenced).  add: optional n x n
 * f64 matrix added to every A (vcor.get(k, True)[s]), add_stride = 0 or n*n per
 * batch group (add_period matrices cycle with period `add_period` batches;
 * pass add=NULL for none).  w: batch x n ascending.  Vt: batch x n x n c128 with
 * ROW m = eigenvector m (i.e. Vt[b][m][i] = ev[b][i][m]); use dmk_transpose_c128
 * for the reference's column layout. */
int dmk_eigh_batched(dmk_ctx *ctx, int n, int batch, const void *A, const double *add,
                     int add_group, double *w, void *Vt);
/* real symmetric variant (routine/slater.py:278, lo/lowdin.py:87): A batch x n x n f64. */
int dmk_eigh_batched_real(dmk_ctx *ctx, int n, int batch, const double *A, double *w,
                          double *Vt);
/* rho[b] = sum_m occ[b,m] v_m v_m^H from Vt (batch x n x n), occ (batch x n). */
/* Low-latency symmetric eigensolver for a FEW small real matrices (batch * n/32 <= 256, n <= 576): parallel block
 * one-sided Jacobi over several CUs (csrc/jacobi_eigh.hip).  A: batch x n x n f64, symmetric, FULL storage when V0 is
 * given (else only the lower triangle is read).  V0 (optional): batch x n x n, rows = approximate eigenvectors
 * (orthonormal) of a nearby matrix -- the warm start of the vcor-fit line search (routine/slater.py:1075, 1098).
 * With V0 the call first refines that basis on the matrix cores (Newton-like correction V <- V + F V, a few n^3 products
 * per pass; accepted only when the residual and the orthogonality of the returned basis were measured below
 * 4 sqrt(n) eps |A| / 16 sqrt(n) eps) and falls back to the sweeps otherwise (DMK_EIGH_REFINE=0: sweeps only).
 * w ascending, Vt rows = eigenvectors (Vt may alias V0).  sweeps_out (optional): sweeps taken by the slowest matrix,
 * 0 when the refinement was accepted.  DMK_ERR_NOCONV for NaN / Inf input. */
int dmk_eigh_jacobi_real(dmk_ctx *ctx, int n, int batch, const double *A, const double *V0, double *w, double *Vt,
                         int *sweeps_out);
int dmk_occ_density(dmk_ctx *ctx, int n, int batch, const void *Vt, const double *occ,
                    void *rho /* c128 batch x n x n */);
/* a4: chemical potential and occupation numbers of `n` levels `ew` (device, any order; all spins and k-points of one
 * particle-number sector) -- replaces routine/mfd.py:887-957 (assignocc, ncore = nvirt = 0 branches) and
 * routine/ftsystem.py:24-105 (fermi_smearing_occ, find_mu) without the host sort.
 *   beta = +inf: T = 0; nelec must be an integer 0 <= nelec <= n.  flags bit 0: mu0 is a preferred value, kept if it
 *     separates nelec levels within the window thr_deg; otherwise mu = mid-point of the levels of rank nelec - 1 and
 *     nelec (bit-identical to the sorted-array expression).  Levels below mu - thr_deg get 1, the remaining electrons
 *     are spread evenly over [mu - thr_deg, mu + thr_deg].
 *   beta finite: Fermi function 1 / (exp(beta (e - mu)) + 1) (0 beyond beta (e - mu) >= 100); flags bit 1: mu = mu0 is
 *     fixed, else mu solves sum occ = nelec to the tolerance fit_tol (bracketed Newton on the device).
 *   flags bit 2: `ew` is in ascending order (the frontier levels are read, not searched; T = 0 only).
 * occ: device, n doubles.  info_host[5]: mu, |sum occ - nelec| (0 at T = 0), electrons spread over the window, levels
 * in the window, 0.  Synchronises the stream (the caller needs mu) -- unless info_host is NULL: then the call only
 * enqueues the kernel (occupations in stream order, nothing read back, NaN / Inf input unreported: for callers whose
 * eigensolver has already rejected it, e.g. the objective of the vcor fit). */
int dmk_assign_occ(dmk_ctx *ctx, int64_t n, const double *ew, double nelec, double beta, double mu0, int flags,
                   double thr_deg, double fit_tol, double *occ, double *info_host /*[5]*/);
int dmk_transpose_c128(dmk_ctx *ctx, int rows, int cols, int batch, const void *in, void *out);

/* ------------------------------------------------------------------------- */
/* a7 : Schmidt bath                                                           */
/* replaces routine/slater.py:117-220 (_get_emb_basis_svd): gather of          */
/* rdm1_env_imp, scipy.linalg.svd (:180), virtual projection + Loewdin         */
/* (:200-202, lo/lowdin.py:83-101), scatter into `basis` (:212-213)            */
/* ------------------------------------------------------------------------- */

/* rdm1: ncells x nlo x nlo f64 stripe (one spin).  env_idx (nenv), bath_col
 * (nb; global site indices of imp_idx_bath, may lie outside cell 0) and
 * virt_mask (nenv, 0/1) are device int32 arrays.  Outputs: sigma (nb, descending),
 * U (nenv x nb, left singular vectors, column j <-> sigma[j]).  */
int dmk_bath_svd(dmk_ctx *ctx, const int mesh[3], int nlo, const double *rdm1,
                 const int32_t *env_idx, int nenv, const int32_t *bath_col, int nb,
                 double *sigma, double *U);
/* The same for `batch` matrices that share the index maps and differ in the density (the spin channels: rdm1 of member b at
 * rdm1 + b * rdm1_stride; sigma batch x nb; U batch x nenv x nb): one chain of launches for all of them. */
int dmk_bath_svd_batched(dmk_ctx *ctx, const int mesh[3], int nlo, int batch, const double *rdm1, int64_t rdm1_stride,
                         const int32_t *env_idx, int nenv, const int32_t *bath_col, int nb, double *sigma, double *U);
/* B = U[:, :nbath]; if orth: B[virt_mask] = 0; B = B (B^T B)^{-1/2} (eigenvalues
 * <= 1e-14 dropped); then basis[imp_idx, :nimp] = I and
 * basis[env_idx, nimp:nimp+nbath] = B, basis: (ncells*nlo) x ncol_basis f64 (zeroed here). */
int dmk_bath_assemble(dmk_ctx *ctx, const double *U, int nenv, int nb, int nbath,
                      const int32_t *virt_mask, int orth, const int32_t *env_idx,
                      const int32_t *imp_idx, int nimp, int nsites, int ncol_basis,
                      double *basis);

/* ------------------------------------------------------------------------- */
/* a9 / a10 : batched small complex products                                   */
/* replaces utils/misc.py:49-59 (kdot) under make_basis.multiply_basis         */
/* (basis_transform/make_basis.py:923-962) and the mdot triple products of     */
/* transform_h1_to_lo / transform_rdm1_to_lo / _to_ao (:524-644)               */
/* ------------------------------------------------------------------------- */

/* C[b] = alpha * op(A[b]) op(B[b]);  op: 0 = N, 1 = T, 2 = C (conj-transpose).
 * A: batch x (rows x cols as stored) c128 row-major with leading dim = cols. */
int dmk_zgemm_batched(dmk_ctx *ctx, int opA, int opB, int M, int N, int K, int batch,
                      double alpha, const void *A, int64_t strideA, const void *B,
                      int64_t strideB, void *C, int64_t strideC);

/* ------------------------------------------------------------------------- */
/* a11 - a14 : density-fitted AO -> EO ERI transform                           */
/* replaces basis_transform/eri_transform.py:235-399 (get_emb_eri_fast_gdf):   */
/* _ao2mo.r_e2 (:433), lib.hermi_sum (:372), lib.pack_tril (:375), the         */
/* Lij_s4 accumulation (:376-378) and _Lij_s4_to_eri's lib.dot calls (:436-485)*/
/* ------------------------------------------------------------------------- */

typedef struct dmk_eri dmk_eri;

/* C_ao_emb: spin x nk x nao x nemb c128, ALREADY scaled by nk^(-3/4)
 * (eri_transform.py:289-300).  eri_out: device f64 (spin*(spin+1)/2) x npair x
 * npair in (aa, ab, bb) order, accumulated into (caller zeroes it, so that
 * shards can be summed).  flags: bit0 = t_reversal_symm, bit1 = also accumulate the
 * imaginary part of the contraction when bit0 is clear (dmk_eri_imag_norm), bit2 = ROWS-ONLY pipeline: eri_out is ignored
 * (may be NULL), the planes are only taken through dmk_eri_contract_rows, and everything that would contract into an
 * internal ERI (dmk_eri_end_kL without a stack, dmk_eri_contract, a full stack at dmk_eri_begin_kL) returns
 * DMK_ERR_STATE; dmk_eri_finish drops planes still resident (eri_transform.py:486-521, the out-of-core branch).
 * Dimensions are free: an nao off the K tile of the hot kernels (8) makes the pipeline keep a zero-padded COPY of C_ao_emb, taken
 * here (change C afterwards and it is not seen; on the tile C is read in place as before); naux off the contraction's K tile and
 * an odd pair count are padded inside the pipeline's own plane buffers; AO blocks of 4 GiB and more are transformed in ranges of
 * auxiliary rows.  None of it changes what the caller passes in or gets back. */
int dmk_eri_begin(dmk_ctx *ctx, const int mesh[3], int nao, int naux, int nemb, int spin,
                  int flags, const void *C_ao_emb, double *eri_out, dmk_eri **out);
/* Start a momentum-transfer index kL (zeroes the Lij_s4 planes). */
int dmk_eri_begin_kL(dmk_eri *h, int kL);
/* Half-transform one AO block L^{(ki,kj)} (naux x nao x nao c128, device) and
 * accumulate its tril-packed (L|ab) into the current kL's Lij_s4;
 * symmetrise != 0 adds the transposed term of the time-reversal partner pair. */
int dmk_eri_push_block(dmk_eri *h, int ki, int kj, int symmetrise, const void *Lpq);
/* Block ring: `*nslots_out` device buffers of one AO block each (naux x nao x nao c128, contiguous: slot s at
 * ring + s * naux*nao*nao) owned by the pipeline, for producers that write blocks on the device (a generator kernel, a
 * device-side reader).  The caller fills slot number (blocks pushed so far in this group) mod nslots and calls
 * dmk_eri_push_ring_slot: the block is only queued; when `nslots` blocks are queued (or the kL ends) ONE step-1 launch
 * transforms all of them and one step-2 launch accumulates them -- one ramp-up and drain of the GPU per group instead of
 * one per block.  *nslots_out == 0: this shape runs the generic kernels, use dmk_eri_push_block.  A slot may be
 * rewritten as soon as the push that completes its group has returned (work is stream-ordered). */
int dmk_eri_block_ring(dmk_eri *h, void **ring_out, int *nslots_out);
int dmk_eri_push_ring_slot(dmk_eri *h, int ki, int kj, int symmetrise);
/* Producer side of the ring on a stream of its own (opt-in: DMK_ERI_GEN_STREAM=1 double-buffers the ring; otherwise the call hands
 * out the single-buffered slot and the compute stream, i.e. the behaviour of dmk_eri_block_ring).  For the next
 * queue slot (`slot` must be the number of blocks queued so far) this returns where to write the block and the HIP stream to
 * launch the producer on -- the pipeline's second stream, on which the producers of group g + 1 (a generator kernel, a
 * decompressor, a device-side format conversion) run while the compute stream still transforms group g.  The pipeline orders
 * the two streams with events: the producer stream waits until step 1 of the group that last used the half has run, and step
 * 1 of a group waits for the producers of that group.  Follow every fill with dmk_eri_push_ring_slot.  (The reference reads
 * the next block from HDF5 while it transforms the current one only through h5py's prefetch, eri_transform.py:358-366.) */
int dmk_eri_ring_slot(dmk_eri *h, int slot, void **ptr_out, void **stream_out);
/* Transform the blocks queued so far NOW (one step-1 and one step-2 launch) instead of when the queue is full: a caller that
 * knows how many blocks a kL has cuts them into launches of equal length (33 blocks: 11 + 11 + 11, not 16 + 16 + 1).  The next
 * ring slot to fill is slot 0 again. */
int dmk_eri_flush(dmk_eri *h);
/* AO blocks that are already RESIDENT in device memory -- a DF tensor (or the part of it this rank's kL shard reads) kept in HBM
 * across the DMET iterations instead of being re-read from the cderi file for every transform (eri_transform.py:358-366 reads each
 * block once per get_emb_eri call): `nblk` (<= the queue length of dmk_eri_block_ring) consecutive blocks starting at `blocks`
 * (stride naux * nao * nao c128, 16-byte aligned) are transformed as ONE group -- step 1 reads them in place, nothing is copied
 * into the ring.  ki / kj / symmetrise: per block, as for dmk_eri_push_ring_slot.  Blocks queued before the call are flushed; a
 * ring slot handed out by dmk_eri_ring_slot and not pushed yet makes the call fail with DMK_ERR_STATE (push it first). */
int dmk_eri_push_resident(dmk_eri *h, const void *blocks, int nblk, const int32_t *ki, const int32_t *kj, const int32_t *symmetrise);
/* The same for an AO block in HOST memory (what sr_loop / _load3c hand over, eri_transform.py:195-227, 358-366): the
 * block is copied to one of two device staging blocks (`slot` 0 | 1) on a separate copy stream and transformed on the
 * compute stream as soon as it has landed, so the copy of block n+1 overlaps the transform of block n.  Returns
 * without waiting.  The host buffer may be refilled once dmk_eri_host_slot_wait(h, slot) has returned (use two pinned
 * buffers from dmk_host_alloc and alternate the slots; pageable memory works but copies synchronously).
 * symmetrise: bit 0 as in dmk_eri_push_block; bit 1 = the buffer holds the block as STORED for the swapped pair (kj, ki)
 * (a cderi container keeps only i >= j, eri_transform.py:213-224): it is conjugate-transposed on the device after the
 * copy instead of by the host before it. */
int dmk_eri_push_block_host(dmk_eri *h, int ki, int kj, int symmetrise, const void *Lpq_host, int slot);
int dmk_eri_host_slot_wait(dmk_eri *h, int slot);
/* Page-locked host memory for the block feed. */
int dmk_host_alloc(dmk_ctx *ctx, size_t bytes, void **out);
int dmk_host_free(dmk_ctx *ctx, void *p);
/* Contract the current kL: eri[blk] += w (Re^T Re [+ Im^T Im]) (TR) or Re(X^H X). */
int dmk_eri_end_kL(dmk_eri *h, int weight);
/* GSO (partial particle-hole) contraction of basis_transform/eri_transform.py:1252-1277 (_Lij_s4_to_eri_gso):
 * eri[0] += w [(a - b)^T (a - b)] over the Re (and, w = 2, Im) planes of the two flavours; needs spin = 2 at begin. */
int dmk_eri_end_kL_gso(dmk_eri *h, int weight);

/* Plane stack (deferred, K-stacked contraction).  By default every dmk_eri_end_kL contracts its kL at once
 * (eri_transform.py:385-386 calls _Lij_s4_to_eri once per kL).  With a stack of `nslots_wanted` slots (each spin x 2 x naux x
 * npair f64: 842 MB at C5) the planes of up to that many kL stay resident and dmk_eri_contract -- or a full stack, or
 * dmk_eri_finish -- contracts them with ONE GEMM per weight class and spin block whose K runs over all of them: the 26 GB
 * result is touched once instead of once per kL.  `nslots_granted` may be smaller than asked (memory).  Time-reversal
 * pipelines only; call it before the first dmk_eri_begin_kL.  With a stack, k lists that are not the integer mesh must use
 * dmk_eri_begin_kL_weighted (the slot region depends on the weight of the kL). */
int dmk_eri_stack(dmk_eri *h, int nslots_wanted, int *nslots_granted);
/* (Also without a stack: a kL begun with weight 1 -- its own time-reversal partner -- only contributes the real part of its planes
 * (eri_transform.py:453-455), so its blocks run a real-part-only step 2, 2/3 of the matrix work; dmk_eri_end_kL must then be given
 * the same weight.  After such a kL the Im half of what dmk_eri_planes returns is zero.  DMK_ERI_RE_ONLY=0 switches it off.) */
int dmk_eri_begin_kL_weighted(dmk_eri *h, int kL, int weight);
/* Contract what is resident, restricted to the band [band_lo, band_hi) of 128-row tiles of the pair index (-1, -1: all;
 * dmk_eri_bands gives their number).  Bands must be issued in increasing order; after band b the ROWS of band b of every
 * spin block are complete, so a kL-sharded job can reduce them over its ranks while later bands are still being computed
 * (eri_transform_mpi.py:203-210 reduces the whole array after the loop).  `done` != 0 empties the stack. */
int dmk_eri_contract(dmk_eri *h, int band_lo, int band_hi, int done);
/* Freivalds probe of the contraction (eri_transform.py:451-478): with a device vector x (npair f64) and a caller-zeroed
 * yref ((spin*(spin+1)/2) x npair f64, blocks aa, ab, bb), every kL the pipeline contracts from now on also adds
 * w X_a^T (X_b x) to yref[b] through two matrix-vector kernels that share nothing with the tiled GEMM (eri_probe.hip).
 * After dmk_eri_finish / the last band, eri[b] x (dmk_dgemv2) must equal yref[b] to rounding: a check of EVERY tile row and
 * column of the contraction for ~0.1 % of its time.  Per kL planes of a time-reversal pipeline (w = 2: Re and Im rows,
 * w = 1: Re rows) or all 2 naux rows without time reversal; not available for the GSO contraction or a rows-only pipeline.
 * x = yref = NULL switches it off.  Call it while no planes are resident. */
int dmk_eri_probe(dmk_eri *h, const double *x, double *yref);
int dmk_eri_bands(const dmk_eri *h, int *nbands, int *band_rows);
/* Out-of-core form (eri_transform.py:486-521 adds ERI_SLICE-row slabs of every kL to the file): rows [row_lo, row_hi) of every
 * spin block of the contraction of the RESIDENT planes, accumulated into `out` ((spin_pair, rows, npair) f64, caller-zeroed)
 * instead of into the pipeline's ERI -- the whole (spin_pair, npair, npair) tensor never has to fit HBM.  The caller empties
 * the stack with dmk_eri_stack_clear when every slab has been taken and asks dmk_eri_stack_free_slots before a new kL
 * (a full stack would otherwise be contracted into the pipeline's ERI). */
int dmk_eri_contract_rows(dmk_eri *h, int64_t row_lo, int64_t row_hi, double *out);
int dmk_eri_stack_clear(dmk_eri *h);
int dmk_eri_stack_free_slots(const dmk_eri *h, int *free_slots);
/* Device pointer of the current kL's Lij_s4 planes (spin x 2 x naux x npair f64:
 * Re plane then Im plane) for inspection / tests: the planes themselves for dimensions on the tile, else a compact copy owned by
 * the pipeline (valid until the next call or dmk_eri_finish) of its padded buffers. */
int dmk_eri_planes(dmk_eri *h, double **planes_out, int64_t *elems_out);
int dmk_eri_finish(dmk_eri *h);
/* ALGORITHMIC flop counters of the work pushed so far:
 * [0] half transform, [1] contraction. */
int dmk_eri_flops(const dmk_eri *h, double flops_host[2]);
/* Without time reversal (dmk_eri_begin flags bit 0 clear) and with flags bit 1 set, the pipeline also accumulates the
 * IMAGINARY part of every contraction, Re_a^T Im_b - Im_a^T Re_b; this returns its max-abs so far -- the
 * `eri_imag_norm = max_abs(eri.imag)` diagnostic of eri_transform.py:385-394 (compared with ERI_IMAG_TOL by the
 * caller).  0 with time reversal. */
int dmk_eri_imag_norm(dmk_eri *h, double *maxabs_host);
/* The accumulator dmk_eri_imag_norm reduces: device f64, (spin*(spin+1)/2) x npair x npair, NULL / 0 elements with time
 * reversal.  A kL-sharded job sums it over ranks BEFORE taking the norm -- the imaginary parts of kL and -kL only cancel
 * in the sum, as in the reference where max_abs(eri.imag) is taken after mpi.reduce (eri_transform_mpi.py:203-215). */
int dmk_eri_imag_buffer(dmk_eri *h, double **imag_out, int64_t *elems_out);

/* Procedural DF block (synthetic configs): Philox4x32-10,
 * key (seed_lo, seed_hi), counter (e>>1 lo, e>>1 hi, ki, kj), e = (L*nao+p)*nao+q. */
int dmk_df_block_philox_on(dmk_ctx *ctx, void *stream, uint64_t seed, int ki, int kj, int naux, int nao, void *out);
/* `nblk` blocks in one launch: block b = pair (ij[2b], ij[2b+1]) (host array) written at out + b * stride_bytes (device; the
 * queue slots of dmk_eri_block_ring are spaced by the block size) on `stream`.  Same values as nblk calls of the above. */
int dmk_df_blocks_philox_on(dmk_ctx *ctx, void *stream, uint64_t seed, int nblk, const int32_t *ij, int naux, int nao, void *out,
                            int64_t stride_bytes);
int dmk_df_block_philox(dmk_ctx *ctx, uint64_t seed, int ki, int kj, int naux, int nao,
                        void *Lpq_out);

/* Modified (pivoted, incomplete) Cholesky vectors of a 4-fold ERI matrix: the arithmetic of convert_eri_to_gdf
 * (eri_transform.py:1483-1535 -> utils/cholesky.py:21-52 modified_cholesky, :54-105 modified_cholesky_uhf), the writer that turns a
 * molecular ERI into a Gamma-point cderi container.  n = norb (norb + 1) / 2 pair columns.  uhf = 0: m_aa (n x n); vecs
 * [max_vecs][n].  uhf = 1: the aa, bb, ab blocks share one pivot sequence over their 2 n diagonal entries; vecs [2][max_vecs][n]
 * (alpha set, then beta set).  The loop stops after the vector whose pivot residual falls below max_error (that vector is kept,
 * as in the reference) or after 2 n + 1 cycles (*exhausted_out = 1: the reference's "does not converge" warning); max_vecs >=
 * 2 n + 2.  Pivots (first maximum of |residual diagonal|) and vectors follow the host loop operation by operation. */
int dmk_modified_cholesky(dmk_ctx *ctx, int n, int uhf, const double *m_aa, const double *m_bb, const double *m_ab, double max_error,
                          int max_vecs, double *vecs, int32_t *nvec_out, int32_t *exhausted_out);
/* Pivots of a column-pivoted QR (largest residual norm first; first index on ties) of the matrix whose COLUMNS are the rows of `cols`
 * (ncol x vlen f64, device): the column selection of SCDM, lo/scdm.py:134 `la.qr(psiT, pivoting=True)` of which scdm_model keeps
 * perm[:nmo] (routine/localizer.py:98-105 localize_bath_scdm).  piv_out: host, npiv indices. */
int dmk_cpqr_pivots(dmk_ctx *ctx, int ncol, int vlen, const double *cols, int npiv, int32_t *piv_out);

/* a14: 4-fold (npair x npair) -> 1-fold (nemb^4) / 8-fold restore
 * (eri_transform.py:523-544 -> pyscf ao2mo.restore). */
int dmk_eri_restore(dmk_ctx *ctx, int nemb, int symmetry, const double *eri4, double *out);

/* Bare real contraction C (N x N, ldc) += alpha * X^T Y with X, Y: K x N row-major
 * (the lib.dot(Lij.T, Lij, alpha, eri, 1) of eri_transform.py:455-476). */
int dmk_dgemm_tn_acc(dmk_ctx *ctx, int N, int K, double alpha, const double *X,
                     const double *Y, int64_t ldxy, double *C, int64_t ldc);

/* Rectangular form of the real contraction: C (M x N, ldc) += alpha * X^T Y, X: K x M (ldx), Y: K x N (ldy).
 * Used for the overlap S = A^T B of dmet/HubPhSymm.py:39 (basisMatching) and the particle weights of
 * routine/bcs.py:92. */
int dmk_dgemm_tn_acc_rect(dmk_ctx *ctx, int M, int N, int K, double alpha, const double *X, int64_t ldx,
                          const double *Y, int64_t ldy, double *C, int64_t ldc);
/* Full SVD of a small square matrix A = U diag(sigma) Vt (one-sided Jacobi, n <= 84; sigma descending;
 * columns of U belonging to exactly zero singular values are returned as zero).  Replaces
 * scipy.linalg.svd at dmet/HubPhSymm.py:42. */
int dmk_svd_small(dmk_ctx *ctx, int n, const double *A, double *sigma, double *U, double *Vt);
/* C (M x N) = A (M x K) op(B), row-major f64, tall-skinny times small (basis rotations A' = A u, B' = B vt^T of
 * dmet/HubPhSymm.py:46-47; slater.py:76-77).  transB != 0: B is stored N x K and used transposed. */
int dmk_dgemm_nn_small(dmk_ctx *ctx, int64_t M, int N, int K, const double *A, const double *B, int transB,
                       double *C);
/* Particle weight of Nambu bath columns, routine/bcs.py:92: U viewed as (ncell, period, nb);
 * w[j] = sum_{c, p < keep} U[c][p][j]^2. */
int dmk_bcs_weight(dmk_ctx *ctx, int ncell, int period, int keep, int nb, const double *U, double *w);
/* BCS embedding basis assembly, routine/bcs.py:88-103: basis (2, ncells, 2n, n+nval) from the left singular
 * vectors U ((ncells-1)*2n x 2nval) and the weight ordering order[2nval] (device int32). */
int dmk_bcs_assemble(dmk_ctx *ctx, int ncells, int n, int nval, const double *U, const int *order, double *basis);
/* out (batch, r_out, c_out) = zero padded in (batch, r_in, c_in): unit ERI -> embedding ERI container,
 * routine/slater_helper.py:494-518 (unit2emb). */
int dmk_pad_block_f64(dmk_ctx *ctx, int batch, int64_t r_in, int64_t c_in, const double *in, int64_t r_out,
                      int64_t c_out, double *out);
/* Row gather (scatter = 0: out[r][:] = in[idx[r]][:]) or scatter (scatter = 1: out[idx[r]][:] = in[r][:]) of rows of
 * `row_len` doubles (a complex row counts twice its length); idx is a DEVICE int32 array.  The k-sharded mean field of
 * the multi-process path (routine/mfd_mpi.py:64-74 scatter_new / :93-94 gather_new): this rank's k rows of the resident
 * Fock batch in one launch, and its eigenvalues placed into the all-k table that is summed over ranks. */
int dmk_copy_rows_f64(dmk_ctx *ctx, int64_t nrows, int64_t row_len, const int32_t *idx_dev, const double *in, double *out,
                      int scatter);

/* ---- ERI x density -------------------------------------------------------
 * Coulomb / exchange matrices from one 4-fold packed ERI block E (npair x npair, leading dimension ld, device),
 * replacing pyscf.scf.hf.dot_eri_dm as called from solver/scf.py:255-335 (_get_jk):
 *     vj_row[i][j] = sum_kl (ij|kl) dm_row[k][l]        J: ijkl,kl->ij
 *     vj_col[k][l] = sum_ij (ij|kl) dm_col[i][j]        the transposed alpha-beta block, solver/scf.py:325-327
 *     vk[j][k]     = sum_il (ij|kl) dm_k[i][l]          K: ijkl,il->jk
 * Any of the three densities may be NULL (its output is then not touched).  Densities and outputs are device
 * (n x n) f64 row-major.  One streaming pass over E for J (both directions) and one for K; no atomics. */
int dmk_jk_s4(dmk_ctx *ctx, int n, const double *eri, int64_t ld, const double *dm_row, const double *dm_col,
              const double *dm_k, double *vj_row, double *vj_col, double *vk);
/* The same restricted to the packed rows in `nranges` host ranges [lo, hi) (lo a multiple of 32): the outputs are the
 * PARTIAL J / K of those rows.  A kL-sharded job keeps the summed ERI row-sharded over its ranks (the reference reduces the
 * whole array to one rank, eri_transform_mpi.py:203-210) and sums only these n x n matrices. */
int dmk_jk_s4_rows(dmk_ctx *ctx, int n, const double *eri, int64_t ld, int nranges, const int64_t *ranges_host,
                   const double *dm_row, const double *dm_col, const double *dm_k, double *vj_row, double *vj_col, double *vk);
/* 1-fold (n^4) or 8-fold (tril of npair x npair) ERI -> 4-fold (npair x npair): ao2mo.restore(4, .) at
 * solver/scf.py:311-321. */
int dmk_eri_to_s4(dmk_ctx *ctx, int n, int from_symmetry, const double *in, double *out);

/* ---- correlation-potential fit (routine/slater.py:851-1329) ------------------
 * Row dots and column sums of a row-major matrix A (M x N, lda) in ONE streaming pass:
 *     yrow[r] = sum_c A[r][c] xrow[c]   (xrow != NULL)      -> grad[p] = <dV_dparam[p], dw_dV>,  slater.py:1141
 *     ycol[c] = sum_r A[r][c] xcol[r]   (xcol != NULL)      -> V_emb = param . dV_dparam,        slater.py:1059-1071
 * Per-block partials are reduced in a fixed order (no atomics). */
int dmk_dgemv2(dmk_ctx *ctx, int64_t M, int64_t N, const double *A, int64_t lda, const double *xrow, const double *xcol,
               double *yrow, double *ycol);
/* out[b] = min(max column sum, Frobenius norm) of the full symmetric matrix A[b] (n x n): the bound on |A|_2 the eigenbasis
 * refinement scales its tolerances with.  A line search H0 + t V1 needs it once per ray (|H0 + t V1| <= |H0| + |t| |V1|:
 * dmk_fit_args.ray_norm) instead of once per trial step. */
int dmk_sym_norm_bound(dmk_ctx *ctx, int n, int batch, const double *A, double *out);
/* C[b] = alpha op(A[b]) op(B[b]) + beta C[b]; real f64, row-major with leading dimensions, op 0 = N, 1 = T.
 * The nemb x nemb algebra of errfunc / gradfunc (np.dot / mdot at slater.py:1088, 1136-1138; ftsystem.py:183-186). */
int dmk_dgemm_batched(dmk_ctx *ctx, int opA, int opB, int M, int N, int K, int batch, double alpha, const double *A,
                      int64_t lda, int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double beta, double *C,
                      int64_t ldc, int64_t strideC);
/* tril[b][pair(k,l)] = full[b][k][l] + full[b][l][k] (k > l), full[b][k][k]: the "x2, diagonal x0.5, tril" packing of
 * slater.py:1138-1141 / ftsystem.py:206-211 (and of the J density, solver/scf.py). */
int dmk_sym_fold(dmk_ctx *ctx, int n, int batch, const double *full, double *tril);
/* full[b] = symmetric unpack of tril[b] (+ add_tril[b] when not NULL): embH1 + V_emb, slater.py:1074. */
int dmk_sym_unpack(dmk_ctx *ctx, int n, int batch, const double *tril, const double *add_tril, double *full);
/* out[r][c] = in[row_idx ? row_idx[r] : r][col_idx ? col_idx[c] : c]  (fit_idx selections, slater.py:1089-1090, 1136). */
int dmk_gather2d_f64(dmk_ctx *ctx, int nrow, int ncol, const int32_t *row_idx, const int32_t *col_idx, const double *in,
                     int64_t ld_in, double *out);
/* mode 0: out = A o B (Hadamard);  mode 1: out[r][c] = A[r][c] * B[r]  (ev * ewocc, slater.py:1088);
 * mode 2: A and out are complex128 viewed as doubles (ncol = 2 x complex columns), B real: out = A o B (ftsystem.py:263). */
int dmk_ewise_mul(dmk_ctx *ctx, int mode, int64_t nrow, int64_t ncol, const double *A, const double *B, double *out);
/* diff = a - b (diff may be NULL), *sumsq_dev = sum (a - b)^2 in a fixed order: la.norm(drho), slater.py:1094. */
int dmk_sub_sumsq(dmk_ctx *ctx, int64_t n, const double *a, const double *b, double *diff, double *sumsq_dev);
/* Divided-difference matrix of the occupations for the fit gradient, K[b][p][q] = (f_p - f_q)/(e_p - e_q) with the
 * degenerate limit -beta f_p (1 - f_q) (routine/ftsystem.py:170-181); beta <= 0: the T = 0 form 1/(e_occ - e_virt) on the
 * occupied-virtual blocks split at nocc (routine/slater.py:1126-1134).  ew, f: batch x n (f unused at T = 0). */
int dmk_fit_kmat(dmk_ctx *ctx, int n, int batch, const double *ew, const double *f, double beta, int nocc, double *K);
/* dst[(idx[a] ld_dst + idx[b]) elem_stride] += alpha src[a][b], src m x m: drho placed at the fitted indices of a zeroed
 * full matrix (elem_stride 2 writes the real parts of a complex128 matrix); ftsystem.py:263 in matrix form. */
int dmk_scatter2d_add_f64(dmk_ctx *ctx, int m, const int32_t *idx, const double *src, double alpha, double *dst, int64_t ld_dst,
                          int elem_stride);
/* y += alpha x */
int dmk_axpy_f64(dmk_ctx *ctx, int64_t n, double alpha, const double *x, double *y);
/* dV_dparam rows from the cell Gram matrix G[(i,p),(j,q)] = sum_c B[c,i,p] B[c,j,q] (row-major, ldg):
 * entry e (one (parameter, spin) pair) = sum over its nonzeros z in [nz_ptr[e], nz_ptr[e+1]) of
 * nz_val[z] * G[(nz_i[z], p), (nz_j[z], q)], written tril-packed at dV + out_off[e].
 * Replaces the loop over transform_local_sparseH at slater.py:868-877 (slater_helper.py:91-100). */
int dmk_vcor_dV_dparam(dmk_ctx *ctx, int nent, int nb, const double *G, int64_t ldg, const int32_t *nz_ptr,
                       const int32_t *nz_i, const int32_t *nz_j, const double *nz_val, const int64_t *out_off, double *dV);

/* SMALL model lattices (Hubbard cells of <= 8 orbitals, spin * nk <= 256 blocks, nk <= 128): one mean-field step of routine/mfd.py
 * HF() -- eigenpairs of every (spin, k) block of Fock_k (+ the shared real shift `add`, as dmk_eigh_batched), occupations and mu of
 * ALL levels as one particle-number sector (flags / mu0 / thr_deg / fit_tol as dmk_assign_occ; beta = INFINITY: T = 0), rho_k =
 * (ev occ) ev^H and its k -> R fold (real part; system/fourier.py:168-177) -- as ONE launch of one workgroup, nothing read back.
 *   ew, occ (spin nk, n); Vt (spin nk, n, n) c128, ROW m = eigenvector m; rho_k (spin nk, n, n) c128; rho_R (spin, nk, n n) f64
 *   info_dev[12]: mu, nerr, electrons spread over the degeneracy window, levels in it, status (0 ok / 1 no mu / 2 non-finite
 *   levels), max |Im| of the folded density, Jacobi not converged (0 / 1), Jacobi sweeps, then four phase clocks in
 *   microseconds (eigensolver, occupations, density, fold)
 * *handled = 0 (and nothing launched) when the shape is outside the limits: the caller then takes dmk_eigh_batched + dmk_assign_occ
 * + dmk_occ_density + dmk_fold_k2R. */
int dmk_small_meanfield(dmk_ctx *ctx, const int mesh[3], int n, int spin, const void *Fock_k, const double *add, int add_group,
                        double nelec, double beta, double mu0, int flags, double thr_deg, double fit_tol, double *ew, double *occ,
                        void *Vt, void *rho_k, double *rho_R, double *info_dev, int *handled);
/* ... and the Schmidt bath of routine/slater.py:117-220 (_get_emb_basis_svd) for nb <= 8 bath columns as ONE launch: the env x imp
 * block of the stripe rdm1 (index maps as dmk_bath_svd), Householder QR + one-sided Jacobi SVD, nbath_s = #(sigma >= tol_bath) per
 * spin, virtual rows zeroed + Loewdin (orth) and the embedding basis [imp identity | bath] of every spin channel with
 * ncol = nimp + min_s nbath_s columns, PACKED with leading dimension ncol in `basis` (room for nimp + nb columns required).
 *   sigma (spin, nb); U (spin, nenv, nb) or NULL; iout_dev: [0] ncol, [1 + s] nbath_s, [1 + spin] SVD not converged (0 / 1). */
int dmk_small_bath(dmk_ctx *ctx, const int mesh[3], int nlo, int spin, const double *rdm1, int64_t rdm1_stride, const int32_t *env_idx,
                   int nenv, const int32_t *bath_col, int nb, const int32_t *virt_mask, int orth, const int32_t *imp_idx, int nimp,
                   int nsites, double tol_bath, double *sigma, double *U, double *basis, int32_t *iout_dev, int *handled);

/* The whole T = 0 objective of FitVcorEmb along a ray of the parameter space as ONE call (errfunc, routine/slater.py:1059-1124:
 * V_emb = sum_p param_p dV_dparam_p is linear in the parameters, so a line search x + t p evaluates V_emb = v0 + t v1):
 *   H = unpack(v0 + t v1 + H1)  ->  eigenpairs (warm refinement of the previous basis, enqueued without host read-back)
 *   ->  T = 0 occupations of every spin channel (mfd.assignocc, routine/mfd.py:887-957)  ->  rho = ev occ ev^T on the fitted
 *   block, W o rho - target, its sum of squares  ->  a pinned host record the call polls.
 * All pointers are device pointers except nelec / mu0 (host, one per spin) and slot (pinned host, >= 256 bytes, dmk_host_alloc).
 *   v0, v1, H1   (spin, npair) tril-packed; v1 NULL: V_emb = v0
 *   H            (spin, nb, nb) work: the matrix that is diagonalised
 *   Vp           (spin, nb, nb) in: eigenvector rows of the previous evaluation (the warm start); out: this evaluation's
 *   w, occ       (spin, nb) out: levels ascending, occupations
 *   fit_idx      (nidx) int32; W, target, drho (spin, nidx, nidx); work: >= 2112 doubles of device scratch owned by the caller
 *   npass        refinement passes to enqueue (a pass that finds its matrix settled costs a few microseconds)
 *   ray_norm     see the struct
 * *status: 0 ok -- *f2 = sum of squares (errfunc = sqrt(f2 / spin)); 1 the refinement did not verify its basis within npass
 * passes: NOTHING of w / occ / drho is valid, Vp still holds the previous basis, the caller falls back to dmk_eigh_jacobi_real
 * + the separate calls above; 2 non-finite levels.  *settle_pass: the measurement pass (0-based) that settled the slower matrix. */
typedef struct {
    int nb, spin, nidx, npass, has_mu0;
    double t, tol_deg;
    const double *v0, *v1, *H1;
    double *H, *Vp, *w, *occ;
    const double *nelec, *mu0;
    const int32_t *fit_idx;
    const double *W, *target;
    double *drho, *work;
    void *slot;
    const double *ray_norm;     /* optional, (2, spin) device: dmk_sym_norm_bound of unpack(v0 + H1) and of unpack(v1) -- constant along
                                 * a ray, so the refinement's bound |H| <= ray_norm[0] + |t| ray_norm[1] costs nothing per trial step
                                 * (NULL: bounded from H itself in every call, one more launch) */
} dmk_fit_args;
int dmk_fit_objective(dmk_ctx *ctx, const dmk_fit_args *args, double *f2, int *status, int *settle_pass);

#ifdef __cplusplus
}
#endif
#endif /* LIBDMETK_H */
