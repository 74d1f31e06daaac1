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


def _run(workload, dist_on, rank=0, world=1, over=None, device=0):
    from libdmet_preview_amd import _lib, pipeline
    from libdmet_preview_amd.basis_transform import eri_transform as et
    ctx = _lib.Context(device)
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


def _worker(rank, world, port, workload, over, q, backend="gloo"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as td
    device = 0
    if backend == "nccl":
        import torch
        device = rank                                  # one process per GPU (RCCL over xGMI)
        torch.cuda.set_device(device)
        td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _run(workload, True, rank, world, over, device=device)
        res["world_size"], res["backend"] = td.get_world_size(), td.get_backend()
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


def test_two_ranks_over_rccl_match_single_process():
    """One process per GPU, backend "nccl" (= RCCL): the zero-copy tensor view of the libdmetk buffers and the two
    all-reduces of the path (rho_R, ERI) across two devices.  Skipped on a box with fewer than two GPUs
    (reference: basis_transform/eri_transform_mpi.py:151-223, routine/mfd_mpi.py:56-114)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (torch.cuda.device_count() = %d)" % torch.cuda.device_count())
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 31100 + (os.getpid() % 1500)
    over = dict(mesh=(3, 2, 2), spin=2, nval=4, nlo=8, naux=12)
    procs = [mpc.Process(target=_worker, args=(r, 2, port, "C3", over, q, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _run("C3", False, over=over)
    scale = np.abs(single["eri"]).max()
    for r in range(2):
        assert res[r]["world_size"] == 2 and res[r]["backend"] == "nccl"
        assert np.abs(res[r]["rho_R"] - single["rho_R"]).max() < 1e-12
        assert np.abs(res[r]["eri"] - single["eri"]).max() < 1e-11 * scale
        assert np.abs(res[r]["H1"] - single["H1"]).max() < 1e-9
    assert np.array_equal(res[0]["eri"], res[1]["eri"])
