"""Randomised campaign of the MULTI-PROCESS path (SURVEY.md section 8 row e): `world` ranks on one GPU (gloo for the exchanges; RCCL refuses
two ranks on one device, on a multi-GPU node the same code runs with backend "nccl"), each with real HIP kernels -- k-sharded
diagonalisation + partial k -> R fold + sum of rho_R, replicated bath, kL-sharded ERI transform, all-reduce or ROW-SHARDED sum of the ERI
(bands of finished rows reduced to their owners), J / K from the owned rows, H1_emb -- against the single-process pipeline on random
systems, including the corner cases of the partition: fewer irreducible kL than ranks, fewer ERI bands than ranks, odd k counts.
    STRESS_SEED=1 STRESS_TRIALS=6 python tools/dist_stress.py          (test infrastructure)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def configs(seed, trials):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(trials):
        mesh = tuple(int(x) for x in rng.choice([1, 2, 3, 4], size=3, p=[0.45, 0.3, 0.15, 0.1]))
        if mesh[0] * mesh[1] * mesh[2] < 2:
            mesh = (2, 1, 1)
        nlo = 2 * int(rng.integers(1, 21))
        out.append(dict(mesh=mesh, nlo=nlo, naux=int(rng.integers(2, 25)), nval=int(rng.integers(1, nlo + 1)), spin=int(rng.integers(1, 3))))
    return out


def run_all(cfgs, dist_on, rank, world, exchange):
    from tests.test_gpu_dist import _run
    res = []
    for over in cfgs:
        r = _run("C3", dist_on, rank, world, over=over, exchange=exchange)
        res.append({k: r[k] for k in ("eri", "rho_R", "proj_diag", "H1", "nemb", "kl", "table")})
    return res


def worker(rank, world, port, cfgs, exchange, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["DMK_ERI_BAND_GROUP"] = "1"                 # one band per reduction: several owners even at these sizes
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, run_all(cfgs, True, rank, world, exchange)))
        td.barrier()
    finally:
        td.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    seed, trials = int(os.environ.get("STRESS_SEED", "1")), int(os.environ.get("STRESS_TRIALS", "6"))
    cfgs = configs(seed, trials)
    t0 = time.time()
    single = run_all(cfgs, False, 0, 1, "none")
    worst, groups, short = 0.0, 0, 0
    mpc = mp.get_context("spawn")
    for world in (2, 3, 4):
        for exchange in ("allreduce", "row_sharded"):
            q = mpc.Queue()
            port = 29700 + (os.getpid() + 17 * groups) % 1200
            procs = [mpc.Process(target=worker, args=(r, world, port, cfgs, exchange, q)) for r in range(world)]
            for p in procs:
                p.start()
            res, deadline = {}, time.time() + 600
            while len(res) < world:
                try:
                    r, v = q.get(timeout=2)
                    res[r] = v
                except Exception:
                    dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                    if dead or time.time() > deadline:
                        for p in procs:                      # the other ranks wait in a collective for the one that died: end them
                            if p.is_alive():
                                p.terminate()
                        raise AssertionError((world, exchange, "a rank failed" if dead else "timeout", dead))
            for p in procs:
                p.join(timeout=120)
                assert p.exitcode == 0, (world, exchange, "a rank failed")
            for c, over in enumerate(cfgs):
                s = single[c]
                allkl = sum((res[r][c]["kl"] for r in range(world)), [])
                assert len(allkl) == len(set(allkl)), (over, world, "kL shards overlap")
                if min(len(res[r][c]["kl"]) for r in range(world)) == 0:
                    short += 1                                          # fewer irreducible kL than ranks: a rank with an empty shard
                scale = max(1.0, float(np.abs(s["eri"]).max()))
                for r in range(world):
                    g = res[r][c]
                    assert g["nemb"] == s["nemb"], (over, world, exchange, r)
                    e = max(float(np.abs(g["rho_R"] - s["rho_R"]).max()), float(np.abs(g["proj_diag"] - s["proj_diag"]).max()),
                            float(np.abs(g["eri"] - s["eri"]).max()) / scale, float(np.abs(g["H1"] - s["H1"]).max()) / max(1.0, float(np.abs(s["H1"]).max())))
                    assert e < 1e-9, (over, world, exchange, r, e)
                    worst = max(worst, e)
                if exchange == "row_sharded":
                    assert all(res[r][c]["table"] == res[0][c]["table"] for r in range(world)), (over, world, "ownership tables differ")
            groups += 1
    print("dist stress ok: %d systems x %d (world, exchange) groups in %.0f s (%d runs with an empty kL shard), worst deviation from the single-process "
          "pipeline %.1e" % (len(cfgs), groups, time.time() - t0, short, worst))
