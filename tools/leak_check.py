"""Does repeated use of the entry points give its device memory back?  Each entry runs 10 warm-up + 40 measured rounds; after every
round the Python-side pool is trimmed and the free device memory read (hipMemGetInfo).  A steady loss per round is a leak in the
library or in the mirror; growth is blocks parked earlier being handed back.    python tools/leak_check.py"""
import gc, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libdmet_preview_amd import _lib, pipeline, synth
from libdmet_preview_amd.basis_transform import eri_transform as et
from libdmet_preview_amd.routine import mfd, slater, bcs, spinless
from libdmet_preview_amd.system.lattice import Lattice
from libdmet_preview_amd.dmet import Hubbard

ctx = _lib.get_ctx()


def measure(tag, fn, rounds=40, warm=10):
    mark = None
    for it in range(warm + rounds):
        fn(it)
        gc.collect()
        ctx.sync()
        ctx.trim()
        free, _ = ctx.mem_info()
        if it == warm - 1:
            mark = free
    lost = (mark - free) / 1e6
    print("%-58s %8.2f MB lost over %d rounds%s" % (tag, lost, rounds, "   <-- LEAK?" if lost > 1.5 else ""), flush=True)
    return lost


def lattice(mesh, n, spin, seed):
    L = Lattice(n, mesh)
    L.val_idx, L.virt_idx, L.core_idx = list(range(n)), [], []
    FR = synth.make_fock_R(mesh, n, spin=spin, seed=seed)
    Fk = synth.fold_R2k(FR, mesh)
    L.fock_lo_k = L.hcore_lo_k = Fk if spin == 2 else Fk[0]
    L.fock_lo_R = L.hcore_lo_R = FR if spin == 2 else FR[0]
    L.H0, L.is_model, L.use_hcore_as_emb_ham = 0.0, True, False
    return L


worst = 0.0
mesh = (3, 2, 1)
L = lattice(mesh, 9, 2, 5)
vc = Hubbard.VcorLocal(False, False, 9)
state = {}


def hf(it):
    state["rho"] = mfd.HF(L, vc, 0.5, False, beta=np.inf)[0]
worst = max(worst, measure("mfd.HF (general chain, 6 k x 9 orbitals, UHF)", hf))
worst = max(worst, measure("mfd.HF finite T", lambda it: mfd.HF(L, vc, 0.5, False, beta=20.0)))
worst = max(worst, measure("slater.get_emb_basis svd", lambda it: slater.get_emb_basis(L, state["rho"])))
worst = max(worst, measure("slater.get_emb_basis svd + scdm", lambda it: slater.get_emb_basis(L, state["rho"], localize_bath="scdm")))
Ls = lattice((4, 2, 1), 4, 2, 6)
worst = max(worst, measure("mfd.HF (small-lattice kernel)", lambda it: mfd.HF(Ls, Hubbard.VcorLocal(False, False, 4), 0.5, False)))

# ERI transform through the numpy entry point with a provider, on and off the tiles
for nao, naux, nemb in ((24, 48, 40), (27, 45, 41)):
    nk = 6
    rng = np.random.default_rng(nao)
    Ce = (rng.standard_normal((2, nk, nao, nemb)) + 1j * rng.standard_normal((2, nk, nao, nemb))) / np.sqrt(nao)
    df = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=3)

    def eri_round(it, Ce=Ce, df=df, nao=nao, naux=naux, nemb=nemb):
        npair = nemb * (nemb + 1) // 2
        eri = ctx.zeros((3, npair, npair), np.float64)
        d_C = ctx.to_device(Ce)
        eng = et.EriEngine(ctx, mesh, nao, naux, nemb, 2, d_C, eri, True)
        eng.set_stack(nslots=3)
        eng.run(df)
        eng.contract()
        eng.close()
        eri.free()
        d_C.free()
    worst = max(worst, measure("EriEngine + plane stack, nao %d naux %d nemb %d" % (nao, naux, nemb), eri_round))

sysm = pipeline.SyntheticSystem.from_workload(ctx, "C4", mesh=(3, 2, 1), nlo=42, naux=45, nval=23, spin=2)


def iteration(it):
    out = pipeline.iteration(ctx, sysm)
    for v in out.values():
        if hasattr(v, "free"):
            v.free()
worst = max(worst, measure("pipeline.iteration (off-tile system, emb Hamiltonian)", iteration, rounds=25))


def resident(it):
    sysm.make_df_resident(None, 0.45)
    pipeline.iteration(ctx, sysm, emb_ham=False)
    sysm.df_resident.free()
    sysm.df_resident = None
worst = max(worst, measure("GDFResident made, used and freed", resident, rounds=25))

# vcor fit
Lf = lattice((2, 3, 1), 4, 2, 11)
rho = mfd.HF(Lf, Hubbard.VcorLocal(False, False, 4), 0.5, False)[0]
basis = slater.get_emb_basis(Lf, rho)
tgt = slater.foldRho(rho, Lf, basis)
tgt = tgt + 0.02 * np.eye(tgt.shape[-1])


def fit(it):
    v = Hubbard.VcorLocal(False, False, 4)
    slater.FitVcorEmb(tgt, Lf, basis, v, np.inf, MaxIter=5)
    slater.FitVcorEmb.last_fit = None
worst = max(worst, measure("slater.FitVcorEmb (T = 0, fused objective)", fit, rounds=25))
worst = max(worst, measure("slater.FitVcorEmb finite T + drho_dparam", lambda it: slater.FitVcorEmb(tgt, Lf, basis, Hubbard.VcorLocal(False, False, 4), 12.0,
                                                                                                   return_drho_dparam=True), rounds=25))
print("leak check %s: worst %.2f MB" % ("ok" if worst <= 1.5 else "FOUND A LOSS", worst))
