"""
oracle/pin_campaign.py -- TEST INFRASTRUCTURE, THIS CONTAINER ONLY (imports the reference from /root/reference through oracle/shim.py;
nothing here travels to the GPU box or is imported by the product).

Randomised PINNING campaign of the oracle: the numpy restatements of oracle/restate.py against the REFERENCE ITSELF, run side by side on
random inputs -- beyond the fixed cases of tests/golden (G1 - G17), which the same generator script produces.  Covered: the k-mesh
bookkeeping (bit-exact), mfd.HF (diagonalisation, occupations at T = 0 and T > 0, density, fold; with and without the k / -k symmetry),
slater.get_emb_basis (Schmidt bath), get_emb_eri_fast_gdf (with and without time reversal, 1- / 4- / 8-fold, unit_eri, C_ao_eo).
    python oracle/pin_campaign.py [seed] [trials]
"""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import shim


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    shim.install()
    shim.quiet()
    from oracle import restate as R
    from oracle import gen_golden as GG
    from libdmet_preview_amd import synth
    from libdmet.routine import mfd, slater
    from libdmet.system import fourier as rf
    from libdmet.basis_transform import eri_transform as ret_plain
    et = shim.patch_eri_transform()
    rng = np.random.default_rng(seed)
    worst = {"hf": 0.0, "bath": 0.0, "eri": 0.0}
    t0, skipped = time.time(), 0
    sink = io.StringIO()
    for trial in range(trials):
        mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4, 5], size=3, p=[0.4, 0.3, 0.15, 0.1, 0.05]))
        nk = int(np.prod(mesh))
        if nk < 2:
            mesh, nk = (2, 1, 1), 2
        if nk > 40:
            mesh, nk = (mesh[0], mesh[1], 1), mesh[0] * mesh[1]
        # ---- k bookkeeping, bit-exact ----
        ks = rf.make_kpts_scaled(mesh)
        assert np.array_equal(ks, R.make_kpts_scaled(mesh)), (mesh, "kpts_scaled")
        cell = shim.FakeCell(3)
        assert np.array_equal(ret_plain.get_weights_t_reversal(cell, cell.get_abs_kpts(ks)), R.get_weights_t_reversal(ks)), (mesh, "weights")
        assert np.array_equal(rf.round_to_FBZ(ks + 0.5, tol=1e-10), R.round_to_FBZ(ks + 0.5, tol=1e-10)), (mesh, "round_to_FBZ")
        # ---- mean field ----
        nlo = int(rng.integers(2, 13))
        spin = int(rng.integers(1, 3))
        beta = np.inf if rng.random() < 0.6 else float(rng.uniform(5.0, 60.0))
        FR = synth.make_fock_R(mesh, nlo, spin=spin, seed=int(rng.integers(1, 1 << 30)))
        Fk = synth.fold_R2k(FR, mesh)
        H1R = 0.5 * FR
        L = GG._duck_lattice(mesh, nlo)
        L.fock_lo_k, L.fock_lo_R = (Fk[0], FR[0]) if spin == 1 else (Fk, FR)
        L.hcore_lo_k = synth.fold_R2k(H1R, mesh)[0] if spin == 1 else synth.fold_R2k(H1R, mesh)
        L.hcore_lo_R = H1R[0] if spin == 1 else H1R
        v = rng.standard_normal((2, nlo, nlo)) * 0.1
        v = 0.5 * (v + v.transpose(0, 2, 1))
        if spin == 1:
            v[1] = v[0]
        ew_all = np.sort(np.concatenate([np.linalg.eigvalsh(Fk[s, k] + v[s]) for s in range(spin) for k in range(nk)]))
        gaps = np.where(np.diff(ew_all) > 1e-4)[0]
        gaps = gaps[(gaps > len(ew_all) // 8) & (gaps < 7 * len(ew_all) // 8)]
        if len(gaps) == 0:
            skipped += 1
            continue
        filling = (int(rng.choice(gaps)) + 1) / float(spin * nk * nlo)
        symm = bool(rng.random() < 0.4)
        with contextlib.redirect_stdout(sink):
            rhoT, mu, E, res = mfd.HF(L, GG._Vcor(v), filling, spin == 1, mu0=None, beta=beta, ires=True, symm=symm)
        rr, mur, Er, resr = R.HF(mesh, Fk, FR, H1R, v, filling, spin == 1, beta=beta, ires=True, symm=symm)
        e = max(float(np.abs(res["e"] - resr["e"]).max()), float(np.abs(rhoT - rr).max()), abs(float(np.asarray(E)) - float(np.asarray(Er))) /
                max(1.0, abs(float(np.asarray(Er)))))
        assert e < 1e-11, (trial, mesh, nlo, spin, beta, symm, e)
        assert np.abs(np.asarray(res["mo_occ"]) - np.asarray(resr["mo_occ"])).max() < 1e-11, (trial, "occupations")
        worst["hf"] = max(worst["hf"], e)
        # ---- Schmidt bath ----
        nval = int(rng.integers(1, nlo + 1))
        perm = rng.permutation(nlo)
        val, virt = sorted(int(x) for x in perm[:nval]), sorted(int(x) for x in perm[nval:])
        Lb = GG._duck_lattice(mesh, nlo, val=val, virt=virt)
        rdm = rhoT if spin == 2 else rhoT[0]
        vb = bool(rng.random() < 0.7)
        with contextlib.redirect_stdout(sink):
            b = slater.get_emb_basis(Lb, rdm, valence_bath=vb)
        ref, info = R.get_emb_basis(mesh, nlo, rdm, imp_idx=list(range(nlo)), val_idx=val, valence_bath=vb, return_info=True)
        assert b.shape == ref.shape, (trial, mesh, nlo, val, vb, b.shape, ref.shape)
        for s in range(b.shape[0]):
            B, Bref = b[s].reshape(-1, b.shape[-1]), ref[s].reshape(-1, ref.shape[-1])
            # same LAPACK on both sides: the columns agree up to a sign; the projector distance is only defined for orthonormal columns
            # (the reference's own basis is not orthonormal to better than 1e-9 when nbath is cut to the other spin's count)
            csd = max(min(np.abs(B[:, j] - Bref[:, j]).max(), np.abs(B[:, j] + Bref[:, j]).max()) for j in range(B.shape[1]))
            d = float(csd) if csd < 1e-10 else float(np.sqrt(2.0) * np.linalg.norm(B - Bref @ (Bref.T @ B)))
            sg = np.sort(np.asarray(info["sigma"][s]))[::-1]
            nb = min(info["nbath_s"])
            ok = not (np.any((sg > 1e-11) & (sg < 1e-7)) or (1 <= nb < len(sg) and sg[nb - 1] - sg[nb] < 1e-6 and sg[nb] > 1e-9))
            if ok:
                smin = float(sg[:nb].min()) if nb >= 1 else 1.0          # singular vectors are determined to eps / sigma
                assert d < 1e-9 + 1e-15 / max(smin, 1e-300), (trial, mesh, nlo, val, vb, d, smin)
                worst["bath"] = max(worst["bath"], d)
        # ---- DF ERI transform ----
        mesh2 = tuple(int(x) for x in rng.choice([1, 2, 3], size=3, p=[0.5, 0.35, 0.15]))
        nk2 = int(np.prod(mesh2))
        if nk2 < 2 or nk2 > 9:
            mesh2, nk2 = (2, 2, 1), 4
        nao, naux, nemb, sp = int(rng.integers(2, 7)), int(rng.integers(2, 9)), int(rng.integers(2, 8)), int(rng.integers(1, 3))
        ks2 = rf.make_kpts_scaled(mesh2)
        cell2 = shim.FakeCell(nao)
        kpts2 = cell2.get_abs_kpts(ks2)
        W0 = synth.make_W0(mesh2, naux, nao, seed=int(rng.integers(1, 1 << 30)))
        blocks = synth.df_blocks_from_W0(W0, mesh2)
        mydf = shim.FakeGDF(cell2, kpts2, lambda i, j, bb=blocks: bb[i, j], naux=naux, blockdim=max(1, naux // 2 + 1))
        C = synth.make_C_ao_lo(mesh2, nao, nao, spin=sp, seed=int(rng.integers(1, 1000)))
        basis = rng.standard_normal((sp, nk2, nao, nemb))
        get = lambda i, j: blocks[i, j]
        runs = [dict(t_reversal_symm=True), dict(t_reversal_symm=False), dict(symmetry=1), dict(unit_eri=True)]
        if sp == 1:
            runs.append(dict(symmetry=8))
        for kw in runs:
            with contextlib.redirect_stdout(sink):
                e_ref = et.get_emb_eri_fast_gdf(cell2, mydf, C_ao_lo=C, basis=basis, max_memory=1, **kw)
            e_or = R.get_emb_eri_fast_gdf(mesh2, ks2, get, naux, nao, C_ao_lo=C, basis=basis, **kw)
            assert np.asarray(e_ref).shape == np.asarray(e_or).shape, (trial, kw)
            d = float(np.abs(np.asarray(e_ref) - np.asarray(e_or)).max()) / max(1.0, float(np.abs(e_ref).max()))
            assert d < 1e-11, (trial, mesh2, nao, naux, nemb, sp, kw, d)
            worst["eri"] = max(worst["eri"], d)
    # ---- vcor fit in the embedding space: the reference's errfunc / gradfunc closures (captured by wrapping slater.minimize, as gen_G9
    #      does) against oracle/restate_fit.py at random parameter vectors ----
    from oracle import restate_fit as F
    from libdmet.dmet import Hubbard
    shim.patch_scf()
    captured = {}
    real_minimize = slater.minimize

    def spy(fn, x0, MaxIter=300, fgrad=None, **kw):
        captured["fn"], captured["fgrad"] = fn, fgrad
        return real_minimize(fn, x0, MaxIter, fgrad, **kw)
    slater.minimize = spy
    worst_fit, nfit = 0.0, 0
    try:
        for trial in range(max(1, trials // 3)):
            mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.45, 0.3, 0.15, 0.1]))
            if int(np.prod(mesh)) < 2:
                mesh = (2, 1, 1)
            nlo, spin = int(rng.integers(2, 8)), int(rng.integers(1, 3))
            lo = int(rng.integers(0, nlo))
            hi = int(rng.integers(lo, nlo))
            val = list(range(lo, hi + 1))
            with contextlib.redirect_stdout(sink):
                L, FR, basis, target = GG._fit_case("x", mesh, nlo, spin, val, int(rng.integers(1, 1 << 20)))
            nb = basis.shape[-1]
            if nb <= nlo - lo or lo + len(val) >= nb:
                continue                                    # no bath orbital, or ncore + nval electrons do not fit the embedding space:
                                                            # the reference's own closure indexes out of bounds there (slater.py:1070)
            Fk = R.R2k(FR, mesh)
            Sk = np.asarray([np.eye(nlo, dtype=complex)] * basis.shape[1])
            ncore = min(val)
            nelec = ncore + len(val) if spin == 1 else [ncore + len(val)] * 2
            runs = [(np.inf, dict()), (float(rng.uniform(8.0, 30.0)), dict()), (float(rng.uniform(8.0, 30.0)), dict(fix_mu=True, mu0=0.1)),
                    (np.inf, dict(imp_fit=True)), (np.inf, dict(det=True)), (np.inf, dict(remove_diag_grad=True))]
            beta, kw = runs[int(rng.integers(0, len(runs)))]
            v = Hubbard.VcorLocal(spin == 1, False, nlo, idx_range=val)
            with contextlib.redirect_stdout(sink):
                slater.FitVcorEmb(target, L, basis, v, beta, MaxIter=1, **kw)
            ov = F.VcorLocal(spin == 1, False, nlo, idx_range=val)
            nimp = nlo - min(val)
            imp_idx, det_idx = (list(range(nimp)), []) if kw.get("imp_fit") else (([], list(range(nimp))) if kw.get("det") else (None, None))
            fit = F.EmbFit(target, mesh, basis, ov, beta, Fk if spin == 2 else Fk[0], Sk, nelec, imp_idx=imp_idx, det_idx=det_idx,
                           mu0=kw.get("mu0"), fix_mu=kw.get("fix_mu", False), remove_diag_grad=kw.get("remove_diag_grad", False))
            grad = fit.gradfunc if beta == np.inf else fit.gradfunc_ft
            for _ in range(2):
                p = 0.1 * rng.standard_normal(v.length())
                with contextlib.redirect_stdout(sink):
                    e_ref, g_ref = captured["fn"](p.copy()), captured["fgrad"](p.copy())
                d = max(abs(fit.errfunc(p) - e_ref), float(np.abs(grad(p) - g_ref).max()) / max(1.0, float(np.abs(g_ref).max())))
                assert d < 1e-9, ("fit", trial, mesh, nlo, spin, val, beta, kw, d)
                worst_fit = max(worst_fit, d)
            nfit += 1
    finally:
        slater.minimize = real_minimize
    print("oracle pin campaign, fit: %d random embedding problems, errfunc / gradfunc closures of the reference against restate_fit.py: worst %.1e"
          % (nfit, worst_fit))
    # ---- exit of the path: the reference's get_emb_Ham / _get_jk / one-body folds against oracle/restate_ham.py ----
    from oracle import restate_ham as H
    from libdmet.routine import slater_helper as sh
    from libdmet.solver import scf as rscf
    slater._get_jk, slater._get_veff = rscf._get_jk, rscf._get_veff
    worst_ham, nham = 0.0, 0
    for trial in range(max(1, trials // 3)):
        mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.45, 0.3, 0.15, 0.1]))
        nk = int(np.prod(mesh))
        if nk < 2:
            mesh, nk = (2, 1, 1), 2
        nlo, spin = int(rng.integers(2, 8)), int(rng.integers(1, 3))
        lo = int(rng.integers(0, nlo))
        val = list(range(lo, int(rng.integers(lo, nlo)) + 1))
        L = GG._duck_lattice(mesh, nlo, val=val, virt=[i for i in range(nlo) if i > max(val)], core=[i for i in range(nlo) if i < min(val)])
        FR = synth.make_fock_R(mesh, nlo, spin=spin, seed=int(rng.integers(1, 1 << 30)))
        Fk = synth.fold_R2k(FR, mesh)
        HR = float(rng.uniform(0.3, 0.8)) * FR
        Hk = synth.fold_R2k(HR, mesh)
        SR = np.zeros((nk, nlo, nlo))
        SR[0] = np.eye(nlo)
        SR = SR + synth.make_fock_R(mesh, nlo, spin=1, seed=int(rng.integers(1, 1000)))[0] * 0.02
        Sk = synth.fold_R2k(SR[None], mesh)[0]
        sq = (lambda x: x[0]) if spin == 1 else (lambda x: x)
        L.fock_lo_k, L.fock_lo_R, L.hcore_lo_k, L.hcore_lo_R = sq(Fk), sq(FR), sq(Hk), sq(HR)
        L.vhf_lo_k, L.ovlp_lo_k, L.JK_imp, L.Ham, L.H0 = sq(Fk - Hk), Sk, None, None, 0.75
        v = rng.standard_normal((2, nlo, nlo)) * 0.1
        v = 0.5 * (v + v.transpose(0, 2, 1))
        if spin == 1:
            v[1] = v[0]
        vc = GG._Vcor(v)
        with contextlib.redirect_stdout(sink):
            rhoT, mu, E, res = mfd.HF(L, vc, 0.5, spin == 1, beta=np.inf, ires=True)
            L.rdm1_lo_k = res["rho_k"] * (2.0 if spin == 1 else 1.0)
            L.rdm1_lo_R = rhoT
            basis = slater.get_emb_basis(L, rhoT)
        nb = basis.shape[-1]
        H2 = GG._psd_eri(rng, nb, 7, spin)
        JK2 = rng.standard_normal((nlo, nlo))
        JK2 = JK2 + JK2.T
        JK3 = np.asarray([JK2, 0.5 * JK2])[:spin]
        runs = [dict(), dict(add_vcor=True), dict(add_vcor=True, fitting=True), dict(int_bath=False), dict(int_bath=False, JK_imp=JK2),
                dict(int_bath=False, JK_imp=JK3), dict(int_bath=False, hcore=True)]
        rdm1_k = L.rdm1_lo_k
        for kw in runs:
            kw = dict(kw)
            okw = {k: w for k, w in kw.items() if k not in ("hcore",)}
            if kw.get("hcore"):
                okw["use_hcore_as_emb_ham"] = True
            L.JK_imp = kw.pop("JK_imp", None)
            L.use_hcore_as_emb_ham = kw.pop("hcore", False)
            L.JK_core = "unset"
            with contextlib.redirect_stdout(sink):
                Himp, _ = slater.get_emb_Ham(L, basis, vc, H2_given=H2, **kw)
            H1, ovlp, JKc = H.embHam1e(mesh, basis, H2, Hk, Fk, Sk, rdm1_k, vcor_mat=v, **okw)
            d = max(float(np.abs(Himp.H1["cd"] - H1).max()), float(np.abs(np.asarray(Himp.ovlp) - ovlp).max()))
            if JKc is not None:
                d = max(d, float(np.abs(np.asarray(L.JK_core) - JKc).max()))
            else:
                assert L.JK_core is None
            assert d < 1e-11 * max(1.0, float(np.abs(H1).max())), ("ham", trial, mesh, nlo, spin, val, sorted(okw), d)
            worst_ham = max(worst_ham, d)
        L.JK_imp, L.use_hcore_as_emb_ham = None, False
        with contextlib.redirect_stdout(sink):
            dm = slater.foldRho_k(L.rdm1_lo_k, L.R2k_basis(basis))
            for eri in (H2, np.asarray([shim.restore(1, h, nb) for h in H2])):
                vj, vk = rscf._get_jk(dm, eri)
                oj, ok_ = H.get_jk(dm, eri)
                d = max(float(np.abs(np.asarray(vj) - np.asarray(oj)).max()), float(np.abs(np.asarray(vk) - np.asarray(ok_)).max()))
                assert d < 1e-11 * max(1.0, float(np.abs(np.asarray(oj)).max())), ("jk", trial, d)
                worst_ham = max(worst_ham, d)
            for s_ in range(spin):
                for a, b in ((sh.transform_trans_inv(basis[s_], L, FR[s_]), H.transform_trans_inv(basis[s_], mesh, FR[s_])),
                             (sh.transform_trans_inv(basis[s_], L, FR[s_], symmetric=False), H.transform_trans_inv(basis[s_], mesh, FR[s_], False)),
                             (sh.transform_local(basis[s_], L, v[s_]), H.transform_local(basis[s_], v[s_])),
                             (sh.transform_imp(basis[s_], L, v[s_]), H.transform_imp(basis[s_], v[s_])),
                             (sh.transform_imp_env(basis[s_], L, FR[s_]), H.transform_imp_env(basis[s_], FR[s_]))):
                    d = float(np.abs(a - b).max())
                    assert d < 1e-11 * max(1.0, float(np.abs(b).max())), ("fold", trial, d)
                    worst_ham = max(worst_ham, d)
        nham += 1
    print("oracle pin campaign, ham: %d random lattices, the reference's get_emb_Ham (7 option sets) / _get_jk / one-body folds against "
          "restate_ham.py: worst %.1e" % (nham, worst_ham))
    # ---- BCS / Nambu twin: the reference's DiagBdG / DiagGHF, bcs.embBasis and bcs_helper folds against oracle/restate_bcs.py ----
    from oracle import restate_bcs as B
    from libdmet.routine import bcs, bcs_helper as bh
    from libdmet.system import lattice as rl
    worst_bcs, nbcs = 0.0, 0
    for trial in range(max(1, trials // 3)):
        mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.45, 0.3, 0.15, 0.1]))
        nk = int(np.prod(mesh))
        if nk < 2:
            mesh, nk = (3, 1, 1), 3
        n = int(rng.integers(1, 8))
        val = sorted(int(x) for x in rng.permutation(n)[: int(rng.integers(1, n + 1))])
        L = GG._duck_lattice(mesh, n, val=val)
        FR = synth.make_fock_R(mesh, n, spin=2, seed=int(rng.integers(1, 1 << 30)))
        Fk = synth.fold_R2k(FR, mesh)
        v = rng.standard_normal((3, n, n)) * 0.2
        v[0], v[1] = 0.5 * (v[0] + v[0].T), 0.5 * (v[1] + v[1].T)
        mu = float(rng.uniform(-0.4, 0.4))
        vc = GG._Vcor(v)
        with contextlib.redirect_stdout(sink):
            ew, ev = mfd.DiagBdG(Fk, vc, mu)
            ews, evs = mfd.DiagBdGsymm(Fk, vc, mu, L)
        ewo, evo = B.DiagBdG(Fk, v, mu)
        ewso, evso = B.DiagBdG(Fk, v, mu, kmesh=mesh)
        d = max(float(np.abs(ew - ewo).max()), float(np.abs(ews - ewso).max()))
        assert d < 1e-12, ("bdg", trial, mesh, n, d)
        GRho_k = np.einsum("kpm,km,kqm->kpq", ev, (ew < 0).astype(float), ev.conj())
        GRho = rl.FFTtoT(GRho_k, mesh)
        D_R = synth.make_fock_R(mesh, n, spin=1, seed=int(rng.integers(1, 1000)))[0] * 0.3
        GF_R = np.zeros((nk, 2 * n, 2 * n))
        GF_R[:, :n, :n], GF_R[:, n:, n:], GF_R[:, :n, n:] = FR[0], -FR[1], D_R
        GF_R[:, n:, :n] = rl.Lattice.transpose(L, D_R)
        GFk = synth.fold_R2k(GF_R[None], mesh)[0]
        with contextlib.redirect_stdout(sink):
            gw, gv = mfd.DiagGHF(GFk, vc, mu)
            gws, gvs = mfd.DiagGHF_symm(GFk, vc, mu, L)
        d = max(d, float(np.abs(gw - B.DiagGHF(GFk, v, mu)[0]).max()), float(np.abs(gws - B.DiagGHF(GFk, v, mu, kmesh=mesh)[0]).max()))
        assert d < 1e-12, ("ghf", trial, mesh, n, d)
        with contextlib.redirect_stdout(sink):
            basis = bcs.embBasis(L, GRho)
        ob, osig, oB, ow = B.embBasis_proj(np.asarray(GRho).real, n, val)
        assert basis.shape == ob.shape
        csd = max(min(np.abs(basis.reshape(-1, basis.shape[-1])[:, j] - ob.reshape(-1, ob.shape[-1])[:, j]).max(),
                      np.abs(basis.reshape(-1, basis.shape[-1])[:, j] + ob.reshape(-1, ob.shape[-1])[:, j]).max()) for j in range(basis.shape[-1]))
        if np.sort(osig)[0] > 1e-7 and np.diff(np.sort(ow)).min(initial=1.0) > 1e-6:
            assert csd < 1e-9, ("bcs basis", trial, mesh, n, val, csd)
            d = max(d, float(csd))
        H3 = np.asarray([FR[0], FR[1], D_R])
        with contextlib.redirect_stdout(sink):
            for fn, fo, Hm in ((bh.transform_trans_inv, B.transform_trans_inv, H3), (bh.transform_trans_inv, B.transform_trans_inv, FR),
                               (bh.transform_local, B.transform_local, v), (bh.transform_imp, B.transform_imp, v),
                               (bh.transform_imp_env, B.transform_imp_env, H3)):
                (hA, hB), hD, e0 = fn(basis, L, Hm)
                (rA, rB), rD, r0 = fo(basis, mesh, Hm)
                df = max(float(np.abs(hA - rA).max()), float(np.abs(hB - rB).max()), float(np.abs(hD - rD).max()), abs(e0 - r0))
                assert df < 1e-12 * max(1.0, float(np.abs(rA).max())), ("bcs fold", trial, df)
                d = max(d, df)
            dV = bh.get_dV_dparam(basis, L, vc)
        d = max(d, float(np.abs(dV - B.get_dV_dparam(basis, vc.length())).max()))
        assert d < 1e-9, ("bcs", trial, d)
        worst_bcs = max(worst_bcs, d)
        nbcs += 1
    print("oracle pin campaign, bcs: %d random lattices, the reference's DiagBdG(symm) / DiagGHF(_symm) / bcs.embBasis / bcs_helper folds and "
          "dV_dparam against restate_bcs.py: worst %.1e" % (nbcs, worst_bcs))
    # ---- GSO twins and the cderi layout: the reference's spinless.get_emb_basis / get_emb_eri_gso against oracle/restate_gso.py, its
    #      transform_gdf_to_lo (writer) and sr_loop (reader) against oracle/restate_cderi.py ----
    from oracle import restate_gso as G
    from oracle import restate_cderi as Cd
    from libdmet.routine import spinless
    worst_gso, worst_cd, ngso = 0.0, 0.0, 0
    for trial in range(max(1, trials // 3)):
        mesh = tuple(int(x) for x in rng.choice([1, 2, 3], size=3, p=[0.5, 0.35, 0.15]))
        nk = int(np.prod(mesh))
        if nk < 2 or nk > 9:
            mesh, nk = (2, 2, 1), 4
        # bath of a generalised density
        n = int(rng.integers(2, 7))
        lo = int(rng.integers(0, n))
        val = list(range(lo, int(rng.integers(lo, n)) + 1))
        FRb = synth.make_fock_R(mesh, n, spin=2, seed=int(rng.integers(1, 1 << 30)))
        vb_ = rng.standard_normal((3, n, n)) * 0.2
        vb_[0], vb_[1] = 0.5 * (vb_[0] + vb_[0].T), 0.5 * (vb_[1] + vb_[1].T)
        ewo, evo = B.DiagBdG(synth.fold_R2k(FRb, mesh), vb_, float(rng.uniform(-0.3, 0.3)))
        GRho = R.FFTtoT(np.einsum("kpm,km,kqm->kpq", evo, (ewo < 0).astype(float), evo.conj()), mesh).real
        Lg = GG._duck_lattice(mesh, n, val=val, virt=[i for i in range(n) if i > max(val)], core=[i for i in range(n) if i < min(val)])
        imp = val + [i for i in range(n) if i > max(val)]
        for vbath in (True, False):
            try:
                ref, sigma, w = G.get_emb_basis_gso(GRho, n, val, imp, valence_bath=vbath)
            except AssertionError:
                continue                                        # odd nbath: the reference refuses as well (spinless.py:113)
            with contextlib.redirect_stdout(sink):
                b = spinless.get_emb_basis(Lg, GRho, valence_bath=vbath)
            assert b.shape == ref.shape, ("gso bath", trial, b.shape, ref.shape)
            nimp = 2 * len(imp)
            sg = np.sort(sigma)[::-1]
            nb = ref.shape[-1] - nimp
            if nb >= 1 and sg[nb - 1] > 1e-7 and (nb == len(sg) or sg[nb] < 1e-11):
                a2, r2 = b.reshape(-1, b.shape[-1])[:, nimp:], ref.reshape(-1, ref.shape[-1])[:, nimp:]
                d = max(min(np.abs(a2[:, j] - r2[:, j]).max(), np.abs(a2[:, j] + r2[:, j]).max()) for j in range(nb))
                if d > 1e-9:
                    d = float(np.sqrt(2.0) * np.linalg.norm(r2 - a2 @ (a2.T @ r2)))
                assert d < 1e-9, ("gso bath", trial, mesh, n, val, vbath, d)
                worst_gso = max(worst_gso, float(d))
        # GSO ERI and the cderi layout on a physical DF tensor
        nao, naux, nemb, sp = int(rng.integers(2, 6)), int(rng.integers(2, 8)), int(rng.integers(2, 8)), int(rng.integers(1, 3))
        ks2 = rf.make_kpts_scaled(mesh)
        cell2 = shim.FakeCell(nao)
        kpts2 = cell2.get_abs_kpts(ks2)
        W0 = synth.make_W0(mesh, naux, nao, seed=int(rng.integers(1, 1 << 30)))
        blocks = synth.df_blocks_from_W0(W0, mesh)
        mydf = shim.FakeGDF(cell2, kpts2, lambda i, j, bb=blocks: bb[i, j], naux=naux, blockdim=max(1, naux // 2 + 1))
        C = synth.make_C_ao_lo(mesh, nao, nao, spin=sp, seed=int(rng.integers(1, 1000)))
        Cg = C[0] if sp == 1 else C
        basis = rng.standard_normal((nk, 2 * nao, nemb))
        get = lambda i, j: blocks[i, j]
        for kw in (dict(t_reversal_symm=True), dict(t_reversal_symm=False), dict(symmetry=1), dict(unit_eri=True)):
            with contextlib.redirect_stdout(sink):
                e_ref = et.get_emb_eri_gso(cell2, mydf, C_ao_lo=Cg, basis=basis, max_memory=1, **kw)
            e_or = G.get_emb_eri_gso(mesh, ks2, get, naux, nao, Cg, basis, **kw)
            d = float(np.abs(np.asarray(e_ref) - np.asarray(e_or)).max()) / max(1.0, float(np.abs(e_ref).max()))
            assert np.asarray(e_ref).shape == np.asarray(e_or).shape and d < 1e-11, ("gso eri", trial, mesh, nao, naux, nemb, sp, kw, d)
            worst_gso = max(worst_gso, d)
        class _H5Mod(object):
            File = GG._FakeH5
        et.h5py = _H5Mod

        class _GDF(shim.FakeGDF):
            def __init__(self, cell_, kpts_, bb=blocks, na=naux):
                shim.FakeGDF.__init__(self, cell_, kpts_, lambda i, j: bb[i, j], naux=na, blockdim=na)
                self.kpts_band, self._j_only = None, False
        mydf2 = _GDF(cell2, kpts2)
        src = "src_pin_%d_%d.h5" % (seed, trial)
        GG._FakeH5(src, "w")["j3c-kptij"] = np.asarray([(ki, kpts2[j]) for i, ki in enumerate(kpts2) for j in range(i + 1)])
        mydf2._cderi = src
        nlo2 = int(rng.integers(max(1, nao - 1), nao + 1))
        C2 = synth.make_C_ao_lo(mesh, nao, nlo2, spin=1, seed=int(rng.integers(1, 1000)))[0]
        for tr in (True, False):
            dst = "dst_pin_%d_%d_%d.h5" % (seed, trial, tr)
            with contextlib.redirect_stdout(sink):
                et.transform_gdf_to_lo(mydf2, C2, fname=dst, t_reversal_symm=tr)
            f = GG._FakeH5.registry[dst]
            ref, mask = Cd.transform_gdf_to_lo(get, ks2, kpts2, naux, C2, t_reversal_symm=tr)
            assert sorted(f.keys()) == sorted(ref.keys()), ("cderi keys", trial, tr, sorted(f.keys()), sorted(ref.keys()))
            for k in ref:
                d = float(np.abs(np.asarray(f[k]) - ref[k]).max())
                assert np.asarray(f[k]).shape == ref[k].shape and d < 1e-12, ("cderi", trial, k, d)
                worst_cd = max(worst_cd, d)
        ngso += 1
    print("oracle pin campaign, gso / cderi: %d random cases, the reference's spinless.get_emb_basis / get_emb_eri_gso against restate_gso.py: worst %.1e; "
          "its transform_gdf_to_lo against restate_cderi.py: worst %.1e" % (ngso, worst_gso, worst_cd))
    print("oracle pin campaign ok: %d random cases against the reference itself in %.0f s (%d without a gap at the Fermi level skipped), worst: "
          "HF %.1e, bath projector %.1e, ERI %.1e (relative)" % (trials, time.time() - t0, skipped, worst["hf"], worst["bath"], worst["eri"]))


if __name__ == "__main__":
    main()
