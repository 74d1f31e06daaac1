"""Randomised differential campaign of the WHOLE device-resident iteration (libdmet_preview_amd/pipeline.py: diag -> occupations ->
rho_k -> fold -> Schmidt bath -> C_ao_emb -> DF half transform + contraction -> J / K -> H1_emb), i.e. of the hand-over between the
stages, against the oracle evaluated on the pipeline's own intermediate products:
    ERI      vs oracle/restate.py get_emb_eri_fast_gdf on the pipeline's basis and Philox blocks        (1e-8)
    H1, JK   vs oracle/restate_ham.py embHam1e on the pipeline's basis, ERI and density                  (1e-10 relative)
    C_ao_emb vs the R -> k fold of the basis;  the basis orthonormal;  the Freivalds probe of the contraction
on random systems: meshes with axes 1 .. 4, 2 .. 24 (STRESS_NLO_HALF) orbitals per cell, 2 .. 40 (STRESS_NAUX) auxiliary functions, one and two spin channels.
    STRESS_SEED=1 STRESS_TRIALS=40 python tools/iteration_stress.py          (test infrastructure: imports the oracle)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from oracle import restate_ham as H
from libdmet_preview_amd import _lib, pipeline
from libdmet_preview_amd.basis_transform import eri_transform as et

ctx = _lib.get_ctx()
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "40"))
worst = {"eri": 0.0, "H1": 0.0, "JK": 0.0, "C": 0.0, "probe": 0.0}
t0, done = time.time(), 0
for trial in range(trials):
    while True:
        mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.45, 0.3, 0.15, 0.1]))
        nk = mesh[0] * mesh[1] * mesh[2]
        nlo = 2 * int(rng.integers(1, int(os.environ.get('STRESS_NLO_HALF', '13'))))   # even: half filling is an integer number of levels
        naux = int(rng.integers(2, int(os.environ.get('STRESS_NAUX', '41'))))
        nval = int(rng.integers(1, nlo + 1))
        spin = int(rng.integers(1, 3))
        nemb_max = nlo + nval
        cost = nk * nk * naux * nlo * nemb_max * (nlo + nemb_max) * 8e-6 * spin + nk * naux * (nemb_max * (nemb_max + 1) // 2) ** 2 * 6e-6
        if 2 <= nk <= 27 and cost < float(os.environ.get('STRESS_ORACLE_MFLOP', '6e4')):
            break
    seed = int(rng.integers(1, 1 << 30))
    sysm = pipeline.SyntheticSystem(ctx, mesh, nlo, naux, nval, spin, seed=seed, name="iter_stress")
    if trial % 2 == 1:                          # every other system with its DF blocks resident in HBM (et.GDFResident, dmk_eri_push_resident)
        sysm.make_df_resident()
    npair_max = nemb_max * (nemb_max + 1) // 2
    out = pipeline.iteration(ctx, sysm)
    nemb = out["nemb"]
    npair = nemb * (nemb + 1) // 2
    basis = out["basis"].get().reshape(spin, nk, nlo, nemb)
    rhoR = out["rho_R"].get().reshape(spin, nk, nlo, nlo)
    eri = out["eri"].get()
    ks = R.make_kpts_scaled(mesh)
    # basis: orthonormal columns; C_ao_emb = C_ao_lo . R2k(basis) / nk^(3/4)
    for s in range(spin):
        B = basis[s].reshape(nk * nlo, nemb)
        assert np.abs(B.T @ B - np.eye(nemb)).max() < 1e-11, (trial, mesh, nlo, nval, spin, "basis not orthonormal")
    Cref = R.make_C_ao_emb(mesh, ks, C_ao_lo=sysm.C_ao_lo, basis=basis, nao=nlo)
    e_C = float(np.abs(out["C_ao_emb"].get().reshape(Cref.shape) - Cref).max())
    assert e_C < 1e-12, (trial, mesh, nlo, nval, spin, e_C)
    ref = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: R.df_block_philox(seed + 2, i, j, naux, nlo), naux, nlo, C_ao_lo=sysm.C_ao_lo,
                                 basis=basis)
    ref = np.asarray(ref).reshape(eri.shape)
    e_eri = float(np.abs(eri - ref).max())
    assert e_eri < 1e-8, (trial, mesh, nlo, naux, nval, spin, e_eri)
    Fk = R.R2k(sysm.Fock_R, mesh)
    H2 = eri[[0, 2, 1]] if spin == 2 else eri
    rdm1_k = R.R2k(rhoR, mesh) * (2.0 if spin == 1 else 1.0)
    Sk = np.asarray([np.eye(nlo)] * nk)
    H1, _, JKc = H.embHam1e(mesh, basis, H2, 0.5 * Fk, Fk, Sk, rdm1_k)
    ham = out["emb_ham"]
    sc = max(1.0, float(np.abs(H1).max()))
    e_H1, e_JK = float(np.abs(ham["H1"] - H1).max()) / sc, float(np.abs(ham["JK_core"] - JKc).max()) / sc
    assert e_H1 < 1e-10 and e_JK < 1e-10, (trial, mesh, nlo, naux, nval, spin, e_H1, e_JK)
    # Freivalds probe of the contraction on a second pass over the same system
    spin_pair = spin * (spin + 1) // 2
    d_x = ctx.to_device(rng.standard_normal(npair))
    d_yref = ctx.zeros((spin_pair, npair), np.float64)
    eri2 = ctx.zeros((spin_pair, npair, npair), np.float64)
    pipeline.eri_stage(ctx, sysm, out["C_ao_emb"], nemb, eri2, probe=(d_x, d_yref))
    y = et.eri_times_vector_dev(ctx, eri2, spin_pair, npair, d_x).get()
    yref = d_yref.get()
    e_p = float(np.abs(y - yref).max()) / max(1.0, float(np.abs(yref).max()))
    assert e_p < 1e-11, (trial, mesh, nlo, naux, nval, spin, e_p)
    assert np.array_equal(eri2.get(), eri), (trial, "second pass over the same system is not bit-identical")
    for k, v in (("eri", e_eri), ("H1", e_H1), ("JK", e_JK), ("C", e_C), ("probe", e_p)):
        worst[k] = max(worst[k], v)
    done += 1
print("iteration stress ok: %d systems in %.0f s, worst |dERI| %.1e, |dH1| %.1e, |dJK_core| %.1e (relative), |dC_ao_emb| %.1e, Freivalds %.1e"
      % (done, time.time() - t0, worst["eri"], worst["H1"], worst["JK"], worst["C"], worst["probe"]))
