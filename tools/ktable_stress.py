"""Randomised BIT-EXACT campaign of the integer k-mesh bookkeeping of the C ABI (dmk_kpts_scaled,
dmk_kmesh_tables, dmk_kconserv_table, dmk_cell_add_table, dmk_kpt_member, dmk_eri_plan, dmk_assign_workload, the host planner for
arbitrary k lists and get_kpairs_kidx) against the oracle's restatement of the reference's floating-point loops (oracle/restate.py;
reference: system/lattice.py, utils/misc.py kpt_member / round_to_FBZ, basis_transform/eri_transform.py:262-382,
eri_transform_mpi.py:27-55, routine/mfd_mpi.py:33-54) on random meshes with axes 1 .. 9, permuted and shifted k lists.
Host-only: runs without a GPU.        STRESS_SEED=1 STRESS_TRIALS=60 python tools/ktable_stress.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import restate as R
from libdmet_preview_amd.system import fourier, lattice
from libdmet_preview_amd.basis_transform import eri_transform as et
from libdmet_preview_amd.routine import mfd_mpi

rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
trials = int(os.environ.get("STRESS_TRIALS", "60"))
t0, nplans = time.time(), 0
cell = type("Cell", (), {"get_scaled_kpts": staticmethod(lambda k: np.asarray(k))})()
for trial in range(trials):
    while True:
        mesh = tuple(int(x) for x in rng.choice(np.arange(1, 10), size=3, p=[0.25, 0.18, 0.15, 0.12, 0.1, 0.08, 0.05, 0.04, 0.03]))
        nk = mesh[0] * mesh[1] * mesh[2]
        if nk <= 150:
            break
    ks = np.asarray(R.make_kpts_scaled(mesh), dtype=float)
    assert np.array_equal(fourier.make_kpts_scaled(mesh), ks), (mesh, "kpts_scaled")
    kint, minus_k, w = fourier.kmesh_tables(mesh)
    assert np.array_equal(kint, R.make_cells(mesh)), (mesh, "cells")
    assert np.array_equal(minus_k, R.minus_k_index(ks)), (mesh, "minus_k")
    assert np.array_equal(w, R.get_weights_t_reversal(ks)), (mesh, "weights")
    L = lattice.Lattice(1, mesh)
    ca = R.CellArith(mesh)
    idx = rng.integers(0, nk, size=(min(nk * nk, 400), 2))
    for i, j in idx:
        assert L.add(int(i), int(j)) == ca.add(int(i), int(j)) and L.subtract(int(i), int(j)) == ca.subtract(int(i), int(j)), (mesh, i, j)
    # k-point membership modulo reciprocal lattice vectors, with noise below / above the tolerance
    for _ in range(20):
        k = int(rng.integers(0, nk))
        G = rng.integers(-2, 3, size=3)
        q = ks[k] + G + rng.uniform(-1, 1, 3) * 1e-9
        assert fourier.kpt_member_mesh(q, mesh) == k and list(fourier.kpt_member(q, ks)) == [k], (mesh, k, G)
        far = ks[k] + G + np.array([0.37 / max(mesh), 0.0, 0.0]) * (1 if mesh[0] > 0 else 0)
        ref = R.kpt_member(far, ks)
        got = fourier.kpt_member_mesh(far, mesh)
        assert (got == -1 and len(ref) == 0) or (len(ref) == 1 and got == int(ref[0])), (mesh, far, got, ref)
    # visiting plan of the ERI double loop: integer planner == restated reference loop, with and without time reversal
    if nk <= 64:
        for tr in (True, False):
            w_ref, plan = R.tr_block_plan(ks, tr)
            w_int, rec = et.eri_plan(mesh, tr)
            assert np.array_equal(w_int, w_ref), (mesh, tr, "plan weights")
            assert len(rec) == len(plan), (mesh, tr, len(rec), len(plan))
            ref = np.asarray([(p[0], p[1], p[2], int(p[4])) for p in plan], dtype=np.int64).reshape(-1, 4)
            assert np.array_equal(rec[:, [0, 1, 2, 4]], ref), (mesh, tr, "plan records")
            if tr:
                assert np.array_equal(rec[:, 3], [p[3] for p in plan]), (mesh, "jm")
            nplans += 1
        # arbitrary k lists: a permuted list and a shifted mesh with its centre
        perm = rng.permutation(nk)
        shift = np.array([0.5 / mesh[0], 0.5 / mesh[1], 0.0])
        for kl, center in ((ks[perm], None), (ks + shift, shift), ((ks + shift)[perm], shift)):
            for tr in (True, False):
                w2, rec2 = et.general_plan(kl, center, tr)
                kc = kl if center is None else kl - center
                if tr:
                    w_ref = R.get_weights_t_reversal(kl)
                    plan = R.tr_block_plan_weights(kc, w_ref)
                else:
                    w_ref, plan = R.tr_block_plan(kc, False)
                assert np.array_equal(w2, w_ref), (mesh, tr, "general weights")
                assert [tuple(int(x) for x in r) for r in rec2] == [(p[0], p[1], p[2], p[3], int(p[4])) for p in plan], (mesh, tr, "general plan")
                nplans += 1
    # static partition of the irreducible kL over ranks (assign_workload) and the +-k pairs of the sharded diagonalisation
    for n in (1, 2, 3, 4, 5, 8):
        kids = et.assign_workload(mesh, n)
        ref = R.assign_workload(R.get_weights_t_reversal(ks), n)
        assert [list(map(int, x)) for x in ref] == kids, (mesh, n, "assign_workload")
    for perm2, sh in ((np.arange(nk), 0.0), (rng.permutation(nk), 0.0), (rng.permutation(nk), np.array([0.1, 0.0, 0.0]))):
        k2 = ks[perm2] + sh
        kp, kidx = mfd_mpi.get_kpairs_kidx(cell, k2)
        rp, ridx = R.get_kpairs_kidx(k2)
        assert [tuple(int(y) for y in x) for x in rp] == kp and np.array_equal(kidx, ridx), (mesh, "kpairs")
print("k-table stress ok: %d meshes, %d visiting plans in %.0f s -- every table bit-identical to the restated reference loops" % (trials, nplans, time.time() - t0))
