/*
 * ORACLE -- test infrastructure only.  Never imported, linked or executed by the product path
 * (libdmet_preview_amd/); only tests/, __graft_entry__.smoke() and bench.py's parity / cpu_baseline
 * legs call it, and only as the checker.
 *
 * A plain-C restatement of two pieces of the reference's density-fitted ERI transform, written so that
 * ONE momentum transfer kL at production size (C5: 108 AO blocks of 800 x 200 x 200 complex = 55 GB if
 * materialised) can be checked on the host without ever holding a block:
 *
 *   orc_philox_rows    the synthetic DF block of SURVEY.md section 8d (Philox4x32-10, Salmon et al. SC'11:
 *                      element e = (L nao + p) nao + q uses counter (e >> 1, 0, ki, kj), key = seed, words
 *                      (2 (e & 1), 2 (e & 1) + 1) -> (re, im) = (u32 2^-31 - 1) / sqrt(nao)); the same
 *                      recipe as oracle/restate.py:df_block_philox, against which it is pinned bit for bit
 *                      (tests/test_oracle_sample.py), which in turn is pinned by the Random123 known answers.
 *   orc_half_sample    (L|ab) = sum_pq conj(C_i[p,a]) Lpq[L,p,q] C_j[q,b]   for all L but only for a, b in a
 *                      SAMPLE of embedding orbitals, plus the transposed term of lib.hermi_sum when the
 *                      time-reversal partner is folded in -- reference basis_transform/eri_transform.py:403-434
 *                      (transform_ao_to_emb / PySCF _ao2mo.r_e2), :368-378 (hermi_sum, accumulate into Lij_s4).
 *
 * The sampled (L|ab) are exact restrictions of the full tensor, so pack_tril / the w (Re^T Re + Im^T Im)
 * contraction of eri_transform.py:436-485 on them (done in numpy by oracle/eri_sample.py) gives exact entries
 * of the full ERI.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CLONES __attribute__((target_clones("default", "avx2", "avx512f")))

/* ten rounds on n independent counters (structure of arrays of 64-bit lanes holding 32-bit values, so that the
 * 32 x 32 -> 64 products and the rounds vectorise) */
CLONES static void philox_soa(int64_t n, uint64_t *restrict c0, uint64_t *restrict c1, uint64_t *restrict c2,
                              uint64_t *restrict c3, uint32_t k0, uint32_t k1) {
    const uint64_t M32 = 0xFFFFFFFFull;
    for (int r = 0; r < 10; ++r) {
        const uint64_t kk0 = k0, kk1 = k1;
        for (int64_t t = 0; t < n; ++t) {
            const uint64_t p0 = 0xD2511F53ull * c0[t];
            const uint64_t p1 = 0xCD9E8D57ull * c2[t];
            const uint64_t n0 = (p1 >> 32) ^ c1[t] ^ kk0;
            const uint64_t n2 = (p0 >> 32) ^ c3[t] ^ kk1;
            c0[t] = n0;
            c1[t] = p1 & M32;
            c2[t] = n2;
            c3[t] = p0 & M32;
        }
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

/* elements [e0, e1) of block (ki, kj) into split re / im arrays */
static void philox_range(uint64_t seed, int ki, int kj, int nao, int64_t e0, int64_t e1, double *re, double *im,
                         uint64_t *w /* 4 * ((e1 - e0) / 2 + 2) words */) {
    const int64_t t0 = e0 >> 1, t1 = (e1 + 1) >> 1, n = t1 - t0;
    uint64_t *c0 = w, *c1 = w + n, *c2 = w + 2 * n, *c3 = w + 3 * n;
    for (int64_t t = 0; t < n; ++t) {
        const uint64_t ctr = (uint64_t)(t0 + t);
        c0[t] = ctr & 0xFFFFFFFFull;
        c1[t] = ctr >> 32;
        c2[t] = (uint32_t)ki;
        c3[t] = (uint32_t)kj;
    }
    philox_soa(n, c0, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double scale = 1.0 / sqrt((double)nao), two31 = 1.0 / 2147483648.0;
    for (int64_t t = 0; t < n; ++t) {
        const int64_t e = 2 * (t0 + t);
        if (e >= e0 && e < e1) {
            re[e - e0] = ((double)c0[t] * two31 - 1.0) * scale;
            im[e - e0] = ((double)c1[t] * two31 - 1.0) * scale;
        }
        if (e + 1 >= e0 && e + 1 < e1) {
            re[e + 1 - e0] = ((double)c2[t] * two31 - 1.0) * scale;
            im[e + 1 - e0] = ((double)c3[t] * two31 - 1.0) * scale;
        }
    }
}

/* rows [L0, L0 + nL) of the block, interleaved complex, (nL, nao, nao) */
void orc_philox_rows(uint64_t seed, int ki, int kj, int nao, int L0, int nL, double *out) {
    const int64_t row = (int64_t)nao * nao;
#pragma omp parallel
    {
        double *re = (double *)malloc(sizeof(double) * row), *im = (double *)malloc(sizeof(double) * row);
        uint64_t *w = (uint64_t *)malloc(sizeof(uint64_t) * 4 * (row / 2 + 2));
#pragma omp for schedule(static)
        for (int l = 0; l < nL; ++l) {
            const int64_t e0 = (int64_t)(L0 + l) * row;
            philox_range(seed, ki, kj, nao, e0, e0 + row, re, im, w);
            double *o = out + 2 * (int64_t)l * row;
            for (int64_t t = 0; t < row; ++t) {
                o[2 * t] = re[t];
                o[2 * t + 1] = im[t];
            }
        }
        free(re);
        free(im);
        free(w);
    }
}

/* T[x][q] += conj(c[x][p]) * L[p][q] over p, split re / im, unit stride along q */
CLONES static void first_index(int nao, int nA, const double *cr, const double *ci, const double *Lr, const double *Li,
                               double *Tr, double *Ti) {
    for (int x = 0; x < nA; ++x) {
        double *tr = Tr + (int64_t)x * nao, *ti = Ti + (int64_t)x * nao;
        for (int q = 0; q < nao; ++q) tr[q] = ti[q] = 0.0;
        for (int p = 0; p < nao; ++p) {
            const double ar = cr[(int64_t)x * nao + p], ai = -ci[(int64_t)x * nao + p];     /* conj(C_i[p, a]) */
            const double *lr = Lr + (int64_t)p * nao, *li = Li + (int64_t)p * nao;
            for (int q = 0; q < nao; ++q) {
                tr[q] += ar * lr[q] - ai * li[q];
                ti[q] += ar * li[q] + ai * lr[q];
            }
        }
    }
}

/*
 * S[s][L][x][y] += V[x][y] (+ V[y][x] if sym),  V[x][y] = sum_pq conj(Ci[s][p][A[x]]) Lpq[L][p][q] Cj[s][q][A[y]]
 * for the Philox block (seed, ki, kj); Ci / Cj: (spin, nao, nemb) interleaved complex with the given spin strides
 * (in complex elements); S: (spin, naux, nA, nA) interleaved complex, accumulated.
 * L_list (optional, nLs entries): restrict to these auxiliary rows, S is then (spin, nLs, nA, nA).
 */
void orc_half_sample(uint64_t seed, int ki, int kj, int naux, int nao, int nemb, int spin, const double *Ci, int64_t ci_spin,
                     const double *Cj, int64_t cj_spin, int nA, const int *A, int sym, const int *L_list, int nLs, double *S) {
    const int64_t row = (int64_t)nao * nao;
    const int nrows = L_list ? nLs : naux;
    /* gathered coefficient columns: ca[s][x][p] = Ci[s][p][A[x]] (re, im split), cb[s][y][q] = Cj[s][q][A[y]] */
    double *car = (double *)malloc(sizeof(double) * spin * nA * nao), *cai = (double *)malloc(sizeof(double) * spin * nA * nao);
    double *cbr = (double *)malloc(sizeof(double) * spin * nA * nao), *cbi = (double *)malloc(sizeof(double) * spin * nA * nao);
    for (int s = 0; s < spin; ++s)
        for (int x = 0; x < nA; ++x)
            for (int p = 0; p < nao; ++p) {
                const int64_t o = ((int64_t)s * nA + x) * nao + p;
                const double *a = Ci + 2 * ((int64_t)s * ci_spin + (int64_t)p * nemb + A[x]);
                const double *b = Cj + 2 * ((int64_t)s * cj_spin + (int64_t)p * nemb + A[x]);
                car[o] = a[0]; cai[o] = a[1];
                cbr[o] = b[0]; cbi[o] = b[1];
            }
    /* ONE scratch allocation for all threads: per-thread malloc / free of these megabyte buffers inside the parallel
     * region serialises on the process address-space lock (mmap / munmap) once there are hundreds of threads */
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    const int64_t wwords = 4 * (row / 2 + 2);
    const int64_t per = 2 * row + wwords + 2 * (int64_t)nA * nao + 2 * (int64_t)nA * nA + 64;     /* in 8-byte words */
    double *pool = (double *)malloc(sizeof(double) * per * nth);
#pragma omp parallel num_threads(nth)
    {
        int me = 0;
#ifdef _OPENMP
        me = omp_get_thread_num();
#endif
        double *base = pool + per * me;
        double *Lr = base, *Li = Lr + row;
        uint64_t *w = (uint64_t *)(Li + row);
        double *Tr = (double *)(w + wwords), *Ti = Tr + (int64_t)nA * nao;
        double *Vr = Ti + (int64_t)nA * nao, *Vi = Vr + nA * nA;
#pragma omp for schedule(dynamic, 1)
        for (int l = 0; l < nrows; ++l) {
            const int L = L_list ? L_list[l] : l;
            const int64_t e0 = (int64_t)L * row;
            philox_range(seed, ki, kj, nao, e0, e0 + row, Lr, Li, w);
            for (int s = 0; s < spin; ++s) {
                first_index(nao, nA, car + (int64_t)s * nA * nao, cai + (int64_t)s * nA * nao, Lr, Li, Tr, Ti);
                for (int x = 0; x < nA; ++x)
                    for (int y = 0; y < nA; ++y) {
                        const double *tr = Tr + (int64_t)x * nao, *ti = Ti + (int64_t)x * nao;
                        const double *br = cbr + ((int64_t)s * nA + y) * nao, *bi = cbi + ((int64_t)s * nA + y) * nao;
                        double vr = 0.0, vi = 0.0;
                        for (int q = 0; q < nao; ++q) {
                            vr += tr[q] * br[q] - ti[q] * bi[q];
                            vi += tr[q] * bi[q] + ti[q] * br[q];
                        }
                        Vr[x * nA + y] = vr;
                        Vi[x * nA + y] = vi;
                    }
                double *o = S + 2 * (((int64_t)s * nrows + l) * nA * nA);
                for (int x = 0; x < nA; ++x)
                    for (int y = 0; y < nA; ++y) {
                        double vr = Vr[x * nA + y], vi = Vi[x * nA + y];
                        if (sym) {
                            vr += Vr[y * nA + x];
                            vi += Vi[y * nA + x];
                        }
                        o[2 * (x * nA + y)] += vr;
                        o[2 * (x * nA + y) + 1] += vi;
                    }
            }
        }
    }
    free(pool);
    free(car); free(cai); free(cbr); free(cbi);
}

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
