"""
world_size-2 gloo test of the multi-GPU structure on CPU: the static kL partition
(eri_transform_mpi.py:35-55 twin) plus ONE sum all-reduce reproduces the serial ERI.  The per-rank
compute is played by the oracle here (no GPU in this container); on the GPU box the same
partition + all-reduce wraps the HIP pipeline (tests/test_gpu_parity.py::test_eri_sharded_sum).
"""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import restate as R
    from libdmet_preview_amd.parallel import dist
    from libdmet_preview_amd.basis_transform import eri_transform as et
    g = np.load(os.path.join(ROOT, "tests", "golden", "G6_eri.npz"))
    name, spin = "m231", 2
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    W0 = g[name + "/W0"]
    naux, _, nao = W0.shape[:3]
    st = "%s/s%d" % (name, spin)
    ks = R.make_kpts_scaled(mesh)
    blocks = R.df_blocks_from_W0(W0, mesh, ks)
    mine = et.assign_workload(mesh, dist.world_size())[dist.rank()]
    part = R.get_emb_eri_fast_gdf(mesh, ks, lambda i, j: blocks[(i, j)], naux, nao, C_ao_lo=g[st + "/C_ao_lo"],
                                  basis=g[st + "/basis"], kL_list=mine)
    total = dist.all_reduce_sum_numpy(part)
    # density partial fold: each rank folds its k-subset, the all-reduce completes k2R
    rng = np.random.default_rng(3)
    rho_k = rng.standard_normal((6, 2, 2)) + 1j * rng.standard_normal((6, 2, 2))
    mine_k = [k for k in range(6) if k % world == rank]
    ph = R.get_phase_R2k(mesh, ks)                      # (R, k) = exp(-ikR)
    partR = np.einsum("Rk,kij->Rij", ph[:, mine_k].conj(), rho_k[mine_k]).real / 6
    fullR = dist.all_reduce_sum_numpy(partR)
    err_rho = np.abs(fullR - R.FFTtoT(rho_k, mesh)).max()
    err = np.abs(total - g[st + "/eri_tr"]).max() / np.abs(g[st + "/eri_tr"]).max()
    out_q.put((rank, float(err), float(err_rho), mine))
    td.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_kl_shard_allreduce_world2(world):
    """world 5 > 4 irreducible kL: one rank owns no kL and only takes part in the sums."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kls = sorted(sum((r[3] for r in res), []))
    assert kls == [0, 1, 3, 4]          # irreducible kL of the (2,3,1) mesh, each owned exactly once
    for r in res:
        assert r[1] < 1e-12 and r[2] < 1e-13
