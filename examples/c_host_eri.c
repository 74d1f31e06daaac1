/* A compiled host of the C ABI (include/libdmetk.h) with no Python, torch or HIP header in sight: the density-fitted AO -> embedding
 * ERI transform of a small synthetic system, driven exactly as get_emb_eri_fast_gdf drives it (reference:
 * basis_transform/eri_transform.py:235-399) -- visiting plan, one DF block per (ki, kj), half transform + tril-pack accumulation per
 * block, one contraction per irreducible kL -- and checked on the device with the Freivalds probe of the contraction.
 *
 *   gcc -O2 -Iinclude examples/c_host_eri.c -Llibdmet_preview_amd -l:libdmetk.so -Wl,-rpath,$PWD/libdmet_preview_amd -lm -o examples/c_host_eri
 *   examples/c_host_eri [C_ao_emb.bin eri_out.bin]
 *
 * With two arguments the coefficients are read from the first file (spin x nk x nao x nemb complex128, already scaled by nk^(-3/4))
 * and the 4-fold ERI ((npair x npair) f64) is written to the second: tests/test_gpu_c_host.py runs the ctypes host on the same
 * inputs and requires the two results to be bit-identical.  Sizes: mesh 2 x 2 x 1, nao 16, naux 40, nemb 32, one spin channel,
 * time-reversal symmetry on, Philox DF blocks (seed 2026). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libdmetk.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != 0) {                                                                              \
            fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, ctx ? dmk_last_error(ctx) : "no context"); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

static uint64_t lcg_state = 0x243F6A8885A308D3ull;
static double lcg_uniform(void) {       /* (-1, 1) */
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(int64_t)(lcg_state >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

int main(int argc, char **argv) {
    const int mesh[3] = {2, 2, 1};
    const int nk = 4, nao = 16, naux = 40, nemb = 32, spin = 1;
    const uint64_t seed = 2026;
    const int64_t npair = (int64_t)nemb * (nemb + 1) / 2;
    const size_t c_elems = (size_t)spin * nk * nao * nemb;
    dmk_ctx *ctx = NULL;
    CHECK(dmk_init(0, NULL, &ctx));
    printf("%s\n", dmk_version());

    /* ---- C_ao_emb: from the file, or real cell coefficients folded to the mesh (C(-k) = conj C(k)) ---- */
    double *Ck = (double *)malloc(c_elems * 16);
    if (argc >= 3) {
        FILE *f = fopen(argv[1], "rb");
        if (!f || fread(Ck, 16, c_elems, f) != c_elems) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
        fclose(f);
    } else {
        int32_t kint[12];
        double *CR = (double *)malloc((size_t)nk * nao * nemb * sizeof(double));
        CHECK(dmk_kmesh_tables(mesh, kint, NULL, NULL));
        for (size_t i = 0; i < (size_t)nk * nao * nemb; ++i) CR[i] = lcg_uniform() * (i < (size_t)nao * nemb ? 1.0 : 0.3);
        const double scale = pow((double)nk, -0.75), two_pi = 6.283185307179586476925;
        for (int k = 0; k < nk; ++k)
            for (int e = 0; e < nao * nemb; ++e) {
                double re = 0.0, im = 0.0;
                for (int R = 0; R < nk; ++R) {
                    double ph = 0.0;
                    for (int d = 0; d < 3; ++d) ph += (double)kint[3 * k + d] * (double)kint[3 * R + d] / (double)mesh[d];
                    re += cos(two_pi * ph) * CR[(size_t)R * nao * nemb + e];
                    im -= sin(two_pi * ph) * CR[(size_t)R * nao * nemb + e];
                }
                Ck[2 * ((size_t)k * nao * nemb + e)] = re * scale;
                Ck[2 * ((size_t)k * nao * nemb + e) + 1] = im * scale;
            }
        free(CR);
    }

    /* ---- integer bookkeeping of the k-mesh: TR weights and the visiting plan (eri_transform.py:338-382) ---- */
    int32_t weights[4];
    CHECK(dmk_kmesh_tables(mesh, NULL, NULL, weights));
    int64_t nrec = 0;
    CHECK(dmk_eri_plan(mesh, 1, NULL, 0, &nrec));
    int32_t *plan = (int32_t *)malloc((size_t)nrec * 5 * sizeof(int32_t));
    CHECK(dmk_eri_plan(mesh, 1, plan, nrec, &nrec));

    /* ---- device buffers ---- */
    void *dC = NULL, *dEri = NULL, *dX = NULL, *dYref = NULL, *dY = NULL, *dBlk = NULL;
    CHECK(dmk_malloc(ctx, c_elems * 16, &dC));
    CHECK(dmk_memcpy_h2d(ctx, dC, Ck, c_elems * 16));
    CHECK(dmk_malloc(ctx, (size_t)npair * npair * 8, &dEri));
    CHECK(dmk_memset(ctx, dEri, 0, (size_t)npair * npair * 8));
    double *x = (double *)malloc((size_t)npair * 8), *y = (double *)malloc((size_t)npair * 8), *yref = (double *)malloc((size_t)npair * 8);
    for (int64_t i = 0; i < npair; ++i) x[i] = lcg_uniform();
    CHECK(dmk_malloc(ctx, (size_t)npair * 8, &dX));
    CHECK(dmk_memcpy_h2d(ctx, dX, x, (size_t)npair * 8));
    CHECK(dmk_malloc(ctx, (size_t)npair * 8, &dYref));
    CHECK(dmk_memset(ctx, dYref, 0, (size_t)npair * 8));
    CHECK(dmk_malloc(ctx, (size_t)npair * 8, &dY));

    /* ---- the transform ---- */
    dmk_eri *h = NULL;
    CHECK(dmk_eri_begin(ctx, mesh, nao, naux, nemb, spin, 1 /* t_reversal_symm */, dC, (double *)dEri, &h));
    CHECK(dmk_eri_probe(h, (const double *)dX, (double *)dYref));
    void *ring = NULL;
    int ring_slots = 0;
    CHECK(dmk_eri_block_ring(h, &ring, &ring_slots));
    const size_t blk_bytes = (size_t)naux * nao * nao * 16;
    if (ring_slots == 0) CHECK(dmk_malloc(ctx, blk_bytes, &dBlk));
    int nblocks = 0;
    for (int kL = 0; kL < nk; ++kL) {
        if (weights[kL] <= 0) continue;                          /* the time-reversal image of an earlier kL */
        CHECK(dmk_eri_begin_kL(h, kL));
        int pos = 0;
        for (int64_t r = 0; r < nrec; ++r) {
            const int32_t *rec = plan + 5 * r;                   /* kL, i, j, jm, symmetrise */
            if (rec[0] != kL) continue;
            if (ring_slots > 0) {
                CHECK(dmk_df_block_philox(ctx, seed, rec[1], rec[2], naux, nao, (char *)ring + (size_t)pos * blk_bytes));
                CHECK(dmk_eri_push_ring_slot(h, rec[1], rec[2], rec[4]));
                pos = (pos + 1) % ring_slots;
            } else {
                CHECK(dmk_df_block_philox(ctx, seed, rec[1], rec[2], naux, nao, dBlk));
                CHECK(dmk_eri_push_block(h, rec[1], rec[2], rec[4], dBlk));
            }
            ++nblocks;
        }
        CHECK(dmk_eri_end_kL(h, weights[kL]));
    }
    double flops[2];
    CHECK(dmk_eri_flops(h, flops));
    CHECK(dmk_eri_finish(h));

    /* ---- Freivalds: eri x against the probe's sum_kL w X^T (X x), both on the device ---- */
    CHECK(dmk_dgemv2(ctx, npair, npair, (const double *)dEri, npair, (const double *)dX, NULL, (double *)dY, NULL));
    CHECK(dmk_memcpy_d2h(ctx, y, dY, (size_t)npair * 8));
    CHECK(dmk_memcpy_d2h(ctx, yref, dYref, (size_t)npair * 8));
    double err = 0.0, ymax = 0.0;
    for (int64_t i = 0; i < npair; ++i) {
        if (fabs(y[i] - yref[i]) > err) err = fabs(y[i] - yref[i]);
        if (fabs(yref[i]) > ymax) ymax = fabs(yref[i]);
    }
    double *eri = (double *)malloc((size_t)npair * npair * 8);
    CHECK(dmk_memcpy_d2h(ctx, eri, dEri, (size_t)npair * npair * 8));
    double asym = 0.0, emax = 0.0, sum = 0.0;
    for (int64_t a = 0; a < npair; ++a)
        for (int64_t b = 0; b < npair; ++b) {
            const double v = eri[a * npair + b];
            if (fabs(v - eri[b * npair + a]) > asym) asym = fabs(v - eri[b * npair + a]);
            if (fabs(v) > emax) emax = fabs(v);
            sum += v;
        }
    printf("%d DF blocks over %lld plan records, ring of %d slots; algorithmic GFLOP: half transform %.3f, contraction %.3f\n", nblocks,
           (long long)nrec, ring_slots, flops[0] * 1e-9, flops[1] * 1e-9);
    printf("eri: max |.| %.6e, sum %.15e, max |eri - eri^T| %.3e\n", emax, sum, asym);
    printf("Freivalds: max |eri x - sum_kL w X^T (X x)| = %.3e (|yref| max %.3e)\n", err, ymax);
    if (argc >= 3) {
        FILE *f = fopen(argv[2], "wb");
        if (!f || fwrite(eri, 8, (size_t)npair * npair, f) != (size_t)npair * npair) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
        fclose(f);
    }
    /* ---- the same transform with the DF blocks RESIDENT in device memory (what a DMET run does with a tensor that fits: loaded once,
     *      every later get_emb_eri reads it in place): the shard's blocks in plan order, one dmk_eri_push_resident per group ---- */
    int resident_same = 1;
    if (ring_slots > 0) {
        void *dRes = NULL, *dEri2 = NULL;
        CHECK(dmk_malloc(ctx, (size_t)nblocks * blk_bytes, &dRes));
        CHECK(dmk_malloc(ctx, (size_t)npair * npair * 8, &dEri2));
        CHECK(dmk_memset(ctx, dEri2, 0, (size_t)npair * npair * 8));
        size_t at = 0;
        for (int kL = 0; kL < nk; ++kL) {                        /* load once */
            if (weights[kL] <= 0) continue;
            for (int64_t r = 0; r < nrec; ++r)
                if (plan[5 * r] == kL) { CHECK(dmk_df_block_philox(ctx, seed, plan[5 * r + 1], plan[5 * r + 2], naux, nao, (char *)dRes + at)); at += blk_bytes; }
        }
        dmk_eri *h2 = NULL;
        CHECK(dmk_eri_begin(ctx, mesh, nao, naux, nemb, spin, 1, dC, (double *)dEri2, &h2));
        void *ring2 = NULL; int slots2 = 0;
        CHECK(dmk_eri_block_ring(h2, &ring2, &slots2));
        int32_t gi[16], gj[16], gs[16];
        at = 0;
        for (int kL = 0; kL < nk && slots2 > 0; ++kL) {
            if (weights[kL] <= 0) continue;
            CHECK(dmk_eri_begin_kL(h2, kL));
            int n = 0;
            for (int64_t r = 0; r <= nrec; ++r) {
                const int mine = r < nrec && plan[5 * r] == kL;
                if (mine) { gi[n] = plan[5 * r + 1]; gj[n] = plan[5 * r + 2]; gs[n] = plan[5 * r + 4]; ++n; }
                if (n > 0 && (n == slots2 || n == 16 || r == nrec)) {
                    CHECK(dmk_eri_push_resident(h2, (char *)dRes + at, n, gi, gj, gs));
                    at += (size_t)n * blk_bytes;
                    n = 0;
                }
            }
            CHECK(dmk_eri_end_kL(h2, weights[kL]));
        }
        CHECK(dmk_eri_finish(h2));
        double *eri2 = (double *)malloc((size_t)npair * npair * 8);
        CHECK(dmk_memcpy_d2h(ctx, eri2, dEri2, (size_t)npair * npair * 8));
        resident_same = slots2 > 0 && memcmp(eri, eri2, (size_t)npair * npair * 8) == 0;
        printf("resident DF blocks (%d blocks, %.1f MB, read in place): ERI %s\n", nblocks, nblocks * (double)blk_bytes * 1e-6,
               resident_same ? "bit-identical" : "DIFFERS");
        free(eri2);
        CHECK(dmk_free(ctx, dEri2)); CHECK(dmk_free(ctx, dRes));
    }
    const int ok = ymax > 0.0 && err <= 1e-12 * (ymax > 1.0 ? ymax : 1.0) && asym <= 1e-12 * (emax > 1.0 ? emax : 1.0) && resident_same;
    if (dBlk) CHECK(dmk_free(ctx, dBlk));
    CHECK(dmk_free(ctx, dY)); CHECK(dmk_free(ctx, dYref)); CHECK(dmk_free(ctx, dX)); CHECK(dmk_free(ctx, dEri)); CHECK(dmk_free(ctx, dC));
    CHECK(dmk_destroy(ctx));
    free(eri); free(x); free(y); free(yref); free(plan); free(Ck);
    printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 2;
}
