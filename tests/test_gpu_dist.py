"""
Two ranks on ONE GPU: the whole multi-process pipeline (k-sharded diagonalisation + partial k->R fold + sum of rho_R,
replicated bath, kL-sharded ERI transform + sum of the partial ERI, embedding Hamiltonian) with real HIP kernels in
every rank and the gloo backend for the exchanges (RCCL refuses two ranks on one device; on a multi-GPU node the
same code runs with backend "nccl").  The result must equal the single-process pipeline.
"""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(workload, dist_on, rank=0, world=1, over=None):
    from libdmet_preview_amd import _lib, pipeline
    from libdmet_preview_amd.basis_transform import eri_transform as et
    ctx = _lib.Context(0)
    _lib.set_ctx(ctx)
    sysm = pipeline.SyntheticSystem.from_workload(ctx, workload, **(over or {}))
    kl = et.assign_workload(sysm.mesh, world)[rank] if dist_on else None
    out = pipeline.iteration(ctx, sysm, kL_list=kl, allreduce_eri=True, emb_ham=True)
    ctx.sync()
    n, nk, spin, nemb = sysm.nlo, sysm.nk, sysm.spin, out["nemb"]
    B = out["basis"].get().reshape(spin, nk * n, nemb)
    return {"eri": out["eri"].get(), "rho_R": out["rho_R"].get(), "proj_diag": np.einsum("spa,spa->sp", B, B),
            "H1": np.asarray(out["emb_ham"]["H1"]), "nemb": nemb, "kl": kl,
            "stages": sorted(out["timers"].keys())}


def _worker(rank, world, port, workload, over, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _run(workload, True, rank, world, over)
        q.put((rank, res))
        td.barrier()
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("workload,over,world", [("C3", None, 2), ("C3", dict(mesh=(3, 2, 2), spin=2, nval=4, nlo=8, naux=12), 3)])
def test_ranks_on_one_gpu_match_single_process(workload, over, world):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29600 + (os.getpid() % 1500)
    procs = [mpc.Process(target=_worker, args=(r, world, port, workload, over, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _run(workload, False, over=over)
    allkl = sum((res[r]["kl"] for r in range(world)), [])
    assert len(allkl) == len(set(allkl))                                  # disjoint shards
    assert "allreduce_rho" in res[0]["stages"] and "allreduce_eri" in res[0]["stages"]
    scale = np.abs(single["eri"]).max()
    for r in range(world):
        assert res[r]["nemb"] == single["nemb"]
        assert np.abs(res[r]["rho_R"] - single["rho_R"]).max() < 1e-12
        assert np.abs(res[r]["proj_diag"] - single["proj_diag"]).max() < 1e-10
        assert np.abs(res[r]["eri"] - single["eri"]).max() < 1e-11 * scale
        assert np.abs(res[r]["H1"] - single["H1"]).max() < 1e-9
    assert np.array_equal(res[0]["eri"], res[1]["eri"])                   # both ranks hold the same sum
