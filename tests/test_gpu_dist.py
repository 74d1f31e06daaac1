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


def _run(workload, dist_on, rank=0, world=1, over=None, device=0, exchange="allreduce", own_stream=False):
    # The ERI and H1 compared below live in the GAUGE of the bath vectors.  The fused small-lattice kernels (csrc/small.hip) only
    # serve single-process runs and return an equally valid but different gauge than the general chain the ranks run: both sides
    # of the comparison take the general chain.
    os.environ["DMK_SMALL"] = "0"
    from libdmet_preview_amd import _lib, pipeline
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd.parallel import dist
    keep = None
    if own_stream:
        # the library on a stream of the caller's (not the legacy default stream): the exchanges must order themselves
        # against it through explicit events (dist.order_after_library), not through null-stream semantics
        import torch
        keep = torch.cuda.Stream(device)
        ctx = _lib.Context(device, stream=keep.cuda_stream)
        assert not ctx.default_stream and ctx.stream_ptr == keep.cuda_stream
    else:
        ctx = _lib.Context(device)
    _lib.set_ctx(ctx)
    sysm = pipeline.SyntheticSystem.from_workload(ctx, workload, **(over or {}))
    kl = et.assign_workload(sysm.mesh, world)[rank] if dist_on else None
    out = pipeline.iteration(ctx, sysm, kL_list=kl, emb_ham=True, eri_exchange=exchange if dist_on else "none")
    ctx.sync()
    n, nk, spin, nemb = sysm.nlo, sysm.nk, sysm.spin, out["nemb"]
    B = out["basis"].get().reshape(spin, nk * n, nemb)
    eri = out["eri"].get()
    table = out.get("eri_rows")
    if table is not None:
        # row-sharded sum: only the owner of a band of rows holds its total; assemble the whole array over ranks to compare
        npair = nemb * (nemb + 1) // 2
        eri = dist.gather_rows_numpy(out["eri"], eri.shape[0], npair, list(range(npair)), table)
    return {"eri": eri, "rho_R": out["rho_R"].get(), "proj_diag": np.einsum("spa,spa->sp", B, B),
            "H1": np.asarray(out["emb_ham"]["H1"]), "nemb": nemb, "kl": kl, "table": table,
            "stages": sorted(out["timers"].keys())}


def _worker(rank, world, port, workload, over, q, backend="gloo", exchange="allreduce", own_stream=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["DMK_ERI_BAND_GROUP"] = "1"                 # one band per reduction: several owners even at test sizes
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
        res = _run(workload, True, rank, world, over, device=device, exchange=exchange, own_stream=own_stream)
        res["world_size"], res["backend"] = td.get_world_size(), td.get_backend()
        q.put((rank, res))
        td.barrier()
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("workload,over,world,exchange", [
    ("C3", None, 2, "allreduce"), ("C3", dict(mesh=(3, 2, 2), spin=2, nval=4, nlo=8, naux=12), 3, "allreduce"),
    # row-sharded sum: bands of ERI rows reduced to their owners, J / K from the owned rows, only n x n matrices all-reduced
    ("C3", dict(mesh=(3, 2, 2), spin=2, nval=8, nlo=24, naux=16), 2, "row_sharded"),
    ("C3", dict(mesh=(4, 2, 1), spin=1, nval=20, nlo=40, naux=8), 3, "row_sharded"),
    # more ranks than +-k groups and than irreducible kL (found by tools/dist_stress.py): rank 2 owns no k-point and no kL and only
    # takes part in the sums
    ("C3", dict(mesh=(2, 1, 1), spin=2, nval=3, nlo=6, naux=8), 3, "row_sharded"),
    ("C3", dict(mesh=(2, 1, 1), spin=1, nval=2, nlo=4, naux=5), 4, "allreduce")])
def test_ranks_on_one_gpu_match_single_process(workload, over, world, exchange):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29600 + (os.getpid() % 1500)
    procs = [mpc.Process(target=_worker, args=(r, world, port, workload, over, q, "gloo", exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _run(workload, False, over=over)
    allkl = sum((res[r]["kl"] for r in range(world)), [])
    assert len(allkl) == len(set(allkl))                                  # disjoint shards
    assert "allreduce_rho" in res[0]["stages"]
    if exchange == "allreduce":
        assert "allreduce_eri" in res[0]["stages"] and res[0]["table"] is None
    else:
        owners = {o for (_, _, o) in res[0]["table"]}
        assert res[0]["table"] == res[1]["table"] and owners <= set(range(world)) and len(owners) == min(world, len(res[0]["table"]))
    scale = np.abs(single["eri"]).max()
    for r in range(world):
        assert res[r]["nemb"] == single["nemb"]
        assert np.abs(res[r]["rho_R"] - single["rho_R"]).max() < 1e-12
        assert np.abs(res[r]["proj_diag"] - single["proj_diag"]).max() < 1e-10
        assert np.abs(res[r]["eri"] - single["eri"]).max() < 1e-11 * scale
        assert np.abs(res[r]["H1"] - single["H1"]).max() < 1e-9
    assert np.array_equal(res[0]["eri"], res[1]["eri"])                   # both ranks hold the same sum


def _run_light(workload, dist_on, rank=0, world=1, over=None, rows=48, seed=5):
    """Like _run for FULL-SIZE configs: instead of the whole ERI (C4: 694 MB per rank through a queue) the result carries
    `rows` sampled pair rows assembled from their owners and eri x for a seeded vector x (every row and column takes part),
    plus rho_R, the bath projector diagonal, H1, the kL shard, the per-stage seconds and the bytes this rank put on the wire."""
    from libdmet_preview_amd import _lib, pipeline
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd.parallel import dist
    os.environ["DMK_SMALL"] = "0"                  # same gauge on both sides (see _run)
    ctx = _lib.Context(0)
    _lib.set_ctx(ctx)
    sysm = pipeline.SyntheticSystem.from_workload(ctx, workload, **(over or {}))
    kl = et.assign_workload(sysm.mesh, world)[rank] if dist_on else None
    out = pipeline.iteration(ctx, sysm, kL_list=kl, emb_ham=True, eri_exchange="row_sharded" if dist_on else "none")
    ctx.sync()
    n, nk, spin, nemb = sysm.nlo, sysm.nk, sysm.spin, out["nemb"]
    npair, spin_pair = nemb * (nemb + 1) // 2, spin * (spin + 1) // 2
    rng = np.random.default_rng(seed)
    pick = np.sort(rng.choice(npair, size=rows, replace=False))
    x = rng.uniform(-1.0, 1.0, npair)
    y = et.eri_times_vector_dev(ctx, out["eri"], spin_pair, npair, ctx.to_device(x)).get()
    table = out.get("eri_rows")
    sent = 0
    if table is not None:
        keep = np.zeros_like(y)
        for (lo, hi) in dist.owned_ranges(table):
            keep[:, lo:hi] = y[:, lo:hi]
        y = dist.all_reduce_sum_numpy(keep)
        got = dist.gather_rows_numpy(out["eri"], spin_pair, npair, [int(r) for r in pick], table)
        # bytes this rank contributes to the exchanges of a step: its partial of every ERI row band it does not own (reduce to
        # the owner), rho_R and the eigenvalue table (all-reduce), the n x n J / K partials
        sent = sum((hi - lo) for (lo, hi, o) in table if o != rank) * npair * 8 * spin_pair
        sent += spin * nk * n * n * 8 + spin * nk * n * 8 + 3 * spin * nemb * nemb * 8
    else:
        got = np.stack([np.stack([out["eri"].offset((b * npair + int(r)) * npair, (npair,)).get() for r in pick])
                        for b in range(spin_pair)])
    B = out["basis"].get().reshape(spin, nk * n, nemb)
    return {"rows": got, "eri_x": y, "rho_R": out["rho_R"].get(), "proj_diag": np.einsum("spa,spa->sp", B, B),
            "H1": np.asarray(out["emb_ham"]["H1"]), "nemb": nemb, "kl": kl, "table": table, "nblocks": out["nblocks"],
            "stage_seconds": {k: round(v, 5) for k, v in out["timers"].items()}, "bytes_sent": int(sent)}


def _worker_light(rank, world, port, workload, over, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _run_light(workload, True, rank, world, over)))
        td.barrier()
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("workload,over,world", [
    # BASELINE config 4 AS STATED: diamond-like 4x4x4 mesh, nao 104, naux 416, nemb 136, RHF, k-points / kL sharded over FOUR ranks
    # (4.2 GB per rank) -- on the one GPU of this box, host-staged (gloo) exchanges, real kernels
    ("C4", None, 4),
    # BASELINE config 5's 8-rank partition with its real 6x6x6 mesh (112 irreducible kL -> 14 per rank, 27 +-k groups of k-points
    # per rank), the orbital spaces shrunk so that eight ranks fit one GPU: nlo 96, nval 32 -> nemb 128, naux 192, UHF
    ("C5", dict(nlo=96, nval=32, naux=192), 8)])
def test_baseline_partition_full_mesh_on_one_gpu(workload, over, world):
    """The partition BASELINE.json states for configs 4 and 5 (reference: basis_transform/eri_transform_mpi.py:27-55, 151-157
    kL shards by assign_workload; routine/mfd_mpi.py:56-114 k-sharded diagonalisation + reduce of rho_R): k-sharded diag +
    all-reduce(rho_R) + kL-sharded ERI with the row-sharded sum + J / K from the owned rows, against the single-process result."""
    import json
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 31200 + (os.getpid() % 1500)
    procs = [mpc.Process(target=_worker_light, args=(r, world, port, workload, over, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=1500) for _ in procs)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    single = _run_light(workload, False, over=over)
    allkl = sum((res[r]["kl"] for r in range(world)), [])
    assert len(allkl) == len(set(allkl)) and sum(res[r]["nblocks"] for r in range(world)) == single["nblocks"]
    owners = {o for (_, _, o) in res[0]["table"]}
    assert owners == set(range(world))                                    # every rank owns ERI rows
    scale = np.abs(single["rows"]).max()
    yscale = max(1.0, np.abs(single["eri_x"]).max())
    for r in range(world):
        assert res[r]["nemb"] == single["nemb"] and res[r]["table"] == res[0]["table"]
        assert np.abs(res[r]["rho_R"] - single["rho_R"]).max() < 1e-12
        assert np.abs(res[r]["proj_diag"] - single["proj_diag"]).max() < 1e-10
        assert np.abs(res[r]["rows"] - single["rows"]).max() < 1e-11 * scale
        assert np.abs(res[r]["eri_x"] - single["eri_x"]).max() < 1e-11 * yscale          # every pair row and column
        assert np.abs(res[r]["H1"] - single["H1"]).max() < 1e-9
    try:                          # builder's record (scratch; the judged copy is profiles/r05_*_partition_one_gpu.json)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump({"workload": workload, "override": over, "world": world,
                   "note": "all ranks on ONE GPU, host-staged (gloo) exchanges: a correctness run, not a scaling number",
                   "per_rank": {str(r): {"kL": len(res[r]["kl"]), "blocks": res[r]["nblocks"], "bytes_sent": res[r]["bytes_sent"],
                                         "stage_seconds": res[r]["stage_seconds"]} for r in range(world)},
                   "single_process_stage_seconds": single["stage_seconds"],
                   "max_rel_eri_rows": float(max(np.abs(res[r]["rows"] - single["rows"]).max() for r in range(world)) / scale),
                   "max_abs_rho_R": float(max(np.abs(res[r]["rho_R"] - single["rho_R"]).max() for r in range(world))),
                   "max_abs_H1": float(max(np.abs(res[r]["H1"] - single["H1"]).max() for r in range(world)))},
                  open(os.path.join(ROOT, "gpurun_out", "partition_%s_x%d.json" % (workload, world)), "w"), indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("exchange,own_stream", [("row_sharded", False), ("row_sharded", True), ("allreduce", True)])
def test_one_rank_real_rccl_matches_local(exchange, own_stream):
    """The RCCL code path itself on the 1-GPU box: ONE rank, backend "nccl", the whole iteration with the row-sharded exchange
    (zero-copy tensor views of libdmetk buffers, asynchronous ncclReduce per band group on the process group's stream underneath
    the remaining contraction launches, Work.wait, event edges between the library's stream and torch's, J / K from the owned
    rows) or the whole-array all-reduce -- with the library on the legacy default stream and on a stream of its own
    (Context(stream=...)).  A one-rank sum is the identity, so the result must equal the non-distributed pipeline bit for bit
    where no reduction order changes, and to rounding elsewhere.  reference: eri_transform_mpi.py:151-223."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 32600 + (os.getpid() % 1500)
    over = dict(mesh=(3, 2, 2), spin=2, nval=8, nlo=24, naux=16)
    p = mpc.Process(target=_worker, args=(0, 1, port, "C3", over, q, "nccl", exchange, own_stream))
    p.start()
    _, res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    single = _run("C3", False, over=over)
    assert res["world_size"] == 1 and res["backend"] == "nccl"
    if exchange == "row_sharded":
        assert res["table"] is not None and len(res["table"]) > 1 and {o for (_, _, o) in res["table"]} == {0}
    scale = np.abs(single["eri"]).max()
    assert res["nemb"] == single["nemb"]
    assert np.abs(res["rho_R"] - single["rho_R"]).max() < 1e-12
    assert np.abs(res["eri"] - single["eri"]).max() < 1e-11 * scale
    assert np.abs(res["H1"] - single["H1"]).max() < 1e-9


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


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_script_two_ranks_gloo_one_gpu(scaling):
    """bench.py ITSELF with two ranks (torch.distributed.run, gloo exchanges, both ranks on GPU 0): shards, k-sharded mean
    field with the device-side eigenvalue exchange, kL-sharded ERI, the full-config pass after the timed steps (weak) or
    the full config as the timed step (strong), the rank-summed oracle samples and the per-stage oracle check.  The JSON
    line must carry the contract keys, every parity flag must be green and the exit status 0.  Workload C4-shaped but
    small enough for a test (mesh 3x2x2: four weight-1 and four weight-2 kL)."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update({"DMK_BENCH_BACKEND": "gloo", "DMK_BENCH_ONE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                "DMK_BENCH_OVERRIDE": json.dumps({"mesh": [3, 2, 2], "nlo": 24, "naux": 16, "nval": 8, "spin": 2})})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "C3", "--kl-per-gpu", "1", "--scaling", scaling, "--fit-iters", "0", "--cpu-seconds", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in res, key
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == scaling and res["dtype"] == "f64"
    assert res["parity_ok"] is True and res["parity_maxabs"] <= 1e-8
    assert res["parity_stages_ok"] is True and res["parity_occ_equal"] is True
    assert res["roofline"]["bound"] == "mfma" and 0.0 < res["roofline"]["frac"] <= 1.0
    assert res["cpu_baseline"]["kind"] == "port" and res["cpu_baseline"]["cores"] >= 1
    if scaling == "weak":
        assert res["full_config"]["parity_ok"] is True and res["full_config"]["n_gpus"] == 2
        assert res["full_config_iteration_wall_s"] == res["full_config"]["iteration_wall_s"]
    else:
        assert "full_config" not in res and res["full_config_iteration_wall_s"] > 0


def _fit_run(dist_on, device=0, shard=None):
    os.environ["DMK_SMALL"] = "0"                  # same gauge on both sides (see _run)
    from libdmet_preview_amd import _lib, pipeline
    ctx = _lib.Context(device)
    _lib.set_ctx(ctx)
    sysm = pipeline.SyntheticSystem.from_workload(ctx, "C3", mesh=(3, 2, 1), spin=2, nval=5, nlo=9, naux=6)
    out = pipeline.iteration(ctx, sysm, emb_ham=True, eri_exchange="allreduce" if dist_on else "none")
    fit = pipeline.vcor_fit_stage(ctx, sysm, out["basis"], out["nemb"], out["emb_ham"]["rdm1_emb"], MaxIter=60, shard=shard)
    return {"param": np.array(fit["vcor"].param), "err_end": fit["err_end"], "rows": fit["table_rows_per_rank"], "nparam": fit["nparam"],
            "nfev": fit["objective_evals"], "ngev": fit["gradient_evals"]}


def _fit_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _fit_run(True)))
        td.barrier()
    finally:
        td.destroy_process_group()


def _local_fit_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    out = None
    os.environ["DMK_SMALL"] = "0"                  # same gauge on both sides (see _run)
    if rank == 0:                    # the single-process pipeline up to the fit, BEFORE a process group exists
        from libdmet_preview_amd import _lib, pipeline
        ctx = _lib.Context(0)
        _lib.set_ctx(ctx)
        sysm = pipeline.SyntheticSystem.from_workload(ctx, "C3", mesh=(3, 2, 1), spin=2, nval=5, nlo=9, naux=6)
        out = pipeline.iteration(ctx, sysm, emb_ham=True, eri_exchange="none")
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # only rank 0 fits (a driver that fits on the root and broadcasts the result): the default -- no `shard` -- must not
        # contain a collective, or rank 0 would wait for rank 1 forever
        q.put((rank, _fit_after(ctx, pipeline, sysm, out) if rank == 0 else None))
        td.barrier()
    finally:
        td.destroy_process_group()


def _fit_after(ctx, pipeline, sysm, out):
    from libdmet_preview_amd.routine import slater
    fit = pipeline.vcor_fit_stage(ctx, sysm, out["basis"], out["nemb"], out["emb_ham"]["rdm1_emb"], MaxIter=60, shard=False)
    assert slater.FitVcorEmb.last_fit._dist is None
    return {"param": np.array(fit["vcor"].param), "rows": fit["table_rows_per_rank"], "nparam": fit["nparam"]}


def test_vcor_fit_is_local_unless_sharding_is_asked_for():
    """FitVcorEmb is purely local in the reference (routine/slater.py:909-1329); with a process group initialised a
    fit on ONE rank must neither hang nor change its result."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 35500 + (os.getpid() % 1500)
    procs = [mpc.Process(target=_local_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _fit_run(False)
    assert res[0]["rows"] == res[0]["nparam"] == single["nparam"]
    assert np.array_equal(res[0]["param"], single["param"])


@pytest.mark.parametrize("world", [2, 3])
def test_rank_sharded_vcor_fit_matches_single_rank(world):
    """FitVcorEmb with the dV_dparam table sharded row-wise over the ranks (every rank contracts its slice in both table passes;
    V_emb and the gradient slices are summed over ranks; the nemb x nemb algebra is replicated): the fitted parameters equal the
    single-rank fit to 1e-10, with the same number of objective and gradient evaluations -- the ranks stay in lock-step because
    the summed quantities are bit-identical on all of them.  reference: routine/mfd_mpi.py:117-162 (local slice + reduce)."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 33900 + (os.getpid() % 1500)
    procs = [mpc.Process(target=_fit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _fit_run(False)
    assert single["rows"] == single["nparam"]
    assert sum(res[r]["rows"] for r in range(world)) == single["nparam"] and max(res[r]["rows"] for r in range(world)) < single["nparam"]
    for r in range(world):
        assert np.abs(res[r]["param"] - single["param"]).max() < 1e-10, np.abs(res[r]["param"] - single["param"]).max()
        assert abs(res[r]["err_end"] - single["err_end"]) < 1e-12
        assert (res[r]["nfev"], res[r]["ngev"]) == (single["nfev"], single["ngev"])
        assert np.array_equal(res[r]["param"], res[0]["param"])
