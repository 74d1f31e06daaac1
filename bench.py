#!/usr/bin/env python
"""
bench.py -- one embedding-construction iteration per step on synthetic k-sampled tensors.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process starts N ranks itself (torch.distributed.run as a
child process, one rank per GPU over RCCL) BEFORE it touches the GPU, waits, and exits with the child's code.
Launched by `python -m torch.distributed.run ... bench.py --gpus N ...` it is one of the N ranks.

Workload (BASELINE.json config 5, "C5"): cuprate-like cell, 6x6x6 k-mesh, nao = nlo = 200, naux = 800,
UHF, 56 valence orbitals -> nemb = 256.  A step = diag (432 Hermitian 200x200) + occupations + rho_k +
k->R fold + Schmidt bath (43144 x 56 SVD per spin) + C_ao_emb + density-fitted ERI transform of this
rank's share of the irreducible momentum transfers kL + embedding Hamiltonian.  WEAK scaling: every GPU
transforms the same number of kL (default 14 = 112 irreducible kL / 8, so N = 8 is exactly the full C5
iteration); the DF blocks are regenerated on the device (Philox) inside the timed region, standing in for the
reference's HDF5 reads.  All other inputs are resident in HBM before the clock starts.

Prints ONE JSON line (rank 0):
  value            = ALGORITHMIC FP64 flop of the ERI transform (8 flop per complex
                     multiply-add, full npair^2 contraction) of all ranks / wall-clock of the whole step (max over ranks)
  roofline         = the dominant kernel family: flop ISSUED to the f64 matrix pipe (counted by the library at launch:
                     3M complex products, padded tiles, lower tile triangle of the symmetric contraction) per launch /
                     its average HIP-event duration, against the FP64 MFMA peak -- a true fraction (<= 1)
  parity_maxabs    = max |device - oracle| over a sample of entries of the TIMED ERI (all auxiliary rows, all AO blocks
                     of this job's kL, sampled embedding-orbital pairs; oracle/eri_sample.py), asserted <= 1e-8
  cpu_baseline     = the oracle (numpy / OpenBLAS port of the reference's loop) on a bounded sample, this host's cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # AMD MI355X FP64 matrix spec (= 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz);
                                 # the on-image microarch guide lists no f64 MFMA row -- tools/mfma_acc_probe.hip
                                 # measures 78.1 sustained on the box
PARITY_TOL = 1e-8                # north star: <= 1e-8 max-abs on the transformed ERI


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default="C5")
    p.add_argument("--kl-per-gpu", type=int, default=14, help="irreducible kL transformed per GPU per step")
    p.add_argument("--max-blocks-per-kl", type=int, default=0, help="debug: truncate the i-loop (0 = all)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-parity", action="store_true", help="skip the sampled oracle check of the timed ERI")
    p.add_argument("--parity-budget-s", type=float, default=150.0,
                   help="host-time budget of the oracle check; if the estimate for the whole timed shard exceeds it (few host "
                        "CPUs per rank), an UN-timed re-run of the first kL of every shard is checked instead, and the line says so")
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    p.add_argument("--fit-iters", type=int, default=300,
                   help="MaxIter of the vcor fit measured after the timed steps (reference default 300, "
                        "routine/slater.py:909; 0 = skip); reported under \"vcor_fit\", never part of `value`")
    return p.parse_args()


def spawn_ranks(n):
    """Start n ranks of this script under torch.distributed.run as a CHILD process.  The parent has not imported
    torch or the HIP library, never touches the GPU and never re-execs; a failed rank makes the child -- and this
    process -- exit non-zero."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def host_threads(world):
    """CPU threads this process may usefully run: the affinity mask, capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs but grants a 16-CPU quota: more threads than that only time-slice and spin), split
    over the ranks of the job."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / float(per) + 0.5)))
        except Exception:
            pass
    return max(1, n // max(1, world))


def cpu_baseline(sysm, nemb, budget_s, threads):
    """Oracle (numpy restatement of the reference, same BLAS entry points) on a bounded sample of the same
    workload, on this host's cores.  Returns (half TFLOP/s, contraction TFLOP/s, description, threads)."""
    from oracle import restate as R
    from oracle import eri_sample as ES
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    nao, naux, spin = sysm.nao, sysm.naux, sysm.spin
    rng = np.random.default_rng(0)
    Cemb = (rng.standard_normal((spin, 2, nao, nemb)) + 1j * rng.standard_normal((spin, 2, nao, nemb))) / nao
    # half transform on an L-slice of one AO block (eri_transform.py:403-434, 368-378)
    lsl = max(8, min(naux, 64))
    blk = ES.philox_rows(1, 0, 1, nao, 0, lsl).reshape(lsl, -1)
    R.transform_ao_to_emb(blk[:2], Cemb, 0, 1)
    t0 = time.perf_counter()
    reps = 0
    while True:
        Lij = R.transform_ao_to_emb(blk, Cemb, 0, 1)
        Lij = Lij + Lij.transpose(0, 1, 3, 2)
        R.pack_tril(Lij)
        reps += 1
        if time.perf_counter() - t0 > budget_s * 0.5:
            break
    t_half = (time.perf_counter() - t0) / reps
    f_half = spin * (8.0 * lsl * nao * nao * nemb + 8.0 * lsl * nao * nemb * nemb)
    # contraction on a column sample of the pair space (eri_transform.py:455-476)
    npair = nemb * (nemb + 1) // 2
    ncs = min(npair, 4096)
    X = rng.standard_normal((2 * naux, ncs))
    t0 = time.perf_counter()
    reps = 0
    while True:
        np.dot(X.T, X)
        reps += 1
        if time.perf_counter() - t0 > budget_s * 0.5:
            break
    t_con = (time.perf_counter() - t0) / reps
    f_con = 2.0 * (2 * naux) * ncs * ncs
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    return f_half / t_half / 1e12, f_con / t_con / 1e12, \
        "oracle: half transform (+ hermi_sum, pack_tril) of a %d-row slice of one %s AO block (both spins) + dgemm %dx%dx%d, " \
        "numpy/OpenBLAS, combined in the workload's flop proportions" % (lsl, sysm.name, ncs, 2 * naux, ncs), threads


def parity_sample(nemb):
    """Embedding orbitals whose pairs are checked: every workgroup type of the step-2 kernels and both ends."""
    cand = [0, 17, nemb // 2 - 1, nemb // 2, (3 * nemb) // 4 - 1, (3 * nemb) // 4, nemb - 1]
    return sorted({min(max(int(c), 0), nemb - 1) for c in cand})


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" in os.environ and a.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; using %d ranks\n" % (a.gpus, world, world))
    os.environ.setdefault("DMK_DEVICE", str(local))
    # DMK_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL plumbing on a 1-GPU box)
    distributed = world > 1 or os.environ.get("DMK_FORCE_DIST", "0") == "1"
    if distributed and "MASTER_ADDR" not in os.environ:
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517", "RANK": "0", "WORLD_SIZE": "1"})
    if distributed:
        import torch
        import torch.distributed as td
        # DMK_BENCH_BACKEND=gloo with DMK_BENCH_ONE_GPU=1 runs every rank on GPU 0 with host-staged exchanges: the whole
        # multi-rank code path of this script (shards, exchanges, parity sum) on a 1-GPU box; never used for reported numbers
        backend = os.environ.get("DMK_BENCH_BACKEND", "nccl")
        if os.environ.get("DMK_BENCH_ONE_GPU", "0") == "1":
            local = 0
            os.environ["DMK_DEVICE"] = "0"
        if local >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d needs GPU %d but only %d are visible" % (rank, local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend)
    from libdmet_preview_amd import _lib, pipeline, synth
    from libdmet_preview_amd.basis_transform import eri_transform as et
    from libdmet_preview_amd.parallel import dist

    ctx = _lib.Context(local)
    _lib.set_ctx(ctx)
    sysm = pipeline.SyntheticSystem.from_workload(ctx, a.workload)
    # weak scaling: the irreducible kL list is cut into shards of --kl-per-gpu; rank r takes shard r
    w, _ = et.eri_plan(sysm.mesh, True)
    irr1 = [k for k in range(len(w)) if w[k] == 1]
    irr2 = [k for k in range(len(w)) if w[k] == 2]
    n_irr = len(irr1) + len(irr2)
    nshards = max(1, (n_irr + a.kl_per_gpu - 1) // a.kl_per_gpu)
    shards = [[] for _ in range(nshards)]
    for i, k in enumerate(irr1):                       # weight-1 kL round-robin first (assign_workload rule)
        shards[i % nshards].append(k)
    it2 = iter(irr2)
    for s in shards:
        while len(s) < a.kl_per_gpu:
            k = next(it2, None)
            if k is None:
                break
            s.append(k)
    kl_mine = shards[rank % nshards]
    maxblk = a.max_blocks_per_kl or None

    nemb_guess = sysm.nlo + sysm.nval
    npair = nemb_guess * (nemb_guess + 1) // 2
    spin_pair = sysm.spin * (sysm.spin + 1) // 2
    eri_dev = ctx.zeros((spin_pair, npair, npair), np.float64)

    def step(timers):
        eri_dev.zero_()
        return pipeline.iteration(ctx, sysm, eri_dev=eri_dev, kL_list=kl_mine, timers=timers,
                                  max_blocks_per_kL=maxblk, allreduce_eri=True)

    for _ in range(a.warmup):
        out = step({})
    ctx.sync()
    if distributed:
        dist.barrier()
    ctx.profile(True)
    ctx.profile_read(reset=True)
    ctx.profile_read_flops(reset=True)
    timers = {}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step(timers)
    ctx.sync()
    if distributed:
        import torch
        torch.cuda.synchronize()
        dist.barrier()
    t1 = time.perf_counter()
    fam = ctx.profile_read(reset=True)
    fam_exec = ctx.profile_read_flops(reset=True)
    ctx.profile(False)
    elapsed = t1 - t0
    if sysm.naux == 0:
        # model lattices (BASELINE configs 1-2: Hubbard): no DF tensor, the step is diag + occupations + density + fold + bath
        if rank == 0:
            eigh_ms, eigh_n = fam.get("eigh", (0.0, 0))
            n, nk, spin = sysm.nlo, sysm.nk, sysm.spin
            alg_bytes = 2.0 * 16 * spin * nk * n * n + 8.0 * spin * nk * n            # SURVEY.md section 8d, diag row
            res = {"metric": "DMET embedding-construction iteration (diag+bath) wall-clock", "value": round(elapsed / a.steps, 6),
                   "unit": "s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
                   "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": "%s: mesh %s nlo %d nemb %d spin %d (model lattice, no DF tensor)"
                                          % (a.workload, "x".join(map(str, sysm.mesh)), n, out["nemb"], spin)},
                   "stage_seconds_per_step": {k: round(v / a.steps, 6) for k, v in timers.items()},
                   "roofline": {"bound": "hbm", "kernel": "eigh", "achieved": round(alg_bytes / max(eigh_ms / max(eigh_n, 1), 1e-9) / 1e6, 3),
                                "peak": 8000.0, "unit": "GB/s", "frac": round(alg_bytes / max(eigh_ms / max(eigh_n, 1), 1e-9) / 1e6 / 8000.0, 6),
                                "traffic": None, "note": "launch/latency bound at this size (SURVEY.md section 8d): %d matrices of %dx%d"
                                                         % (spin * nk, n, n)}}
            print(json.dumps(res), flush=True)
        if distributed:
            import torch.distributed as td
            dist.barrier()
            td.destroy_process_group()
        return
    flops = (out["flops_half"] + out["flops_contract"]) * a.steps
    exec_mine = sum(fam_exec.get(k, 0.0) for k in ("zgemm_half1", "zgemm_half2", "dgemm"))
    if distributed:
        agg = dist.all_reduce_sum_numpy(np.array([flops, exec_mine]))
        flops_all, exec_all = float(agg[0]), float(agg[1])
        slots = np.zeros(world)
        slots[rank] = elapsed
        elapsed = float(dist.all_reduce_sum_numpy(slots).max())         # max over ranks
    else:
        flops_all, exec_all = flops, exec_mine

    # ---- sampled oracle check of the TIMED result (every rank evaluates the oracle on its own kL shard, the tiny
    #      samples are summed like the ERI itself) -------------------------------------------------------------------
    nemb = out["nemb"]
    npair = nemb * (nemb + 1) // 2
    parity = None
    if not a.no_parity:
        from oracle import eri_sample as ES
        threads = host_threads(world)
        ES.set_threads(threads)
        tp = time.perf_counter()
        A = parity_sample(nemb)
        C_host = out["C_ao_emb"].get().reshape(sysm.spin, sysm.nk, sysm.nao, nemb)
        # cost of the oracle: the Philox regeneration of every visited block, ~0.16 core-seconds per C5 block
        est = out["nblocks"] * 0.16 * (sysm.naux * sysm.nao ** 2 / (800.0 * 200 ** 2)) / threads
        if distributed:       # shards differ slightly in their block counts: every rank must take the same branch below
            est = float(dist.all_reduce_sum_numpy(np.array([est]))[0]) / world
        check_kl, check_dev, scope = kl_mine, eri_dev, "timed ERI"
        if est > a.parity_budget_s:
            keep = max(1, int(len(kl_mine) * a.parity_budget_s / est))
            check_kl = kl_mine[:keep]
            check_dev = ctx.zeros((spin_pair, npair, npair), np.float64)
            pipeline.eri_stage(ctx, sysm, out["C_ao_emb"], nemb, check_dev, check_kl, {}, maxblk)
            if distributed:
                dist.all_reduce_sum_dev(check_dev)
            scope = "UN-timed re-run (oracle budget %.0f s < %.0f s for the timed shard)" % (a.parity_budget_s, est)
        ref, idx, _ = ES.eri_sample(sysm.mesh, sysm.df.seed, C_host, sysm.naux, A, check_kl, max_blocks_per_kL=maxblk)
        if distributed:
            ref = dist.all_reduce_sum_numpy(ref)
        if rank == 0:
            got = np.stack([np.stack([check_dev.offset((b * npair + int(r)) * npair, (npair,)).get()[idx] for r in idx])
                            for b in range(spin_pair)])
            err = float(np.abs(got - ref).max())
            parity = {"parity_maxabs": err, "parity_ref_maxabs": float(np.abs(ref).max()),
                      "parity_entries": int(ref.size), "parity_orbitals": A, "parity_seconds": round(time.perf_counter() - tp, 2),
                      "parity_threads_per_rank": threads,
                      "parity_scope": "%s of all %d ranks: %d kL, all AO blocks, all %d auxiliary rows, %d sampled pair columns"
                                      % (scope, world, len(check_kl) * world, sysm.naux, len(idx))}
        del check_dev

    rc = 0
    if rank == 0:
        nblk = out["nblocks"]
        # algorithmic flop of each ERI kernel family over the timed region (DESIGN.md
        # section 5); a launch of the half transform covers up to DMK_ERI_GROUP queued AO blocks: rates are totals / totals
        fam_flops = {
            "zgemm_half1": 8.0 * sysm.naux * sysm.nao * sysm.nao * nemb * sysm.spin * nblk * a.steps,
            "zgemm_half2": 8.0 * sysm.naux * sysm.nao * nemb * nemb * sysm.spin * nblk * a.steps,
            "dgemm": out["flops_contract"] * a.steps,
        }
        fam_out = {}
        for k, (ms, n) in fam.items():
            if n:
                fam_out[k] = {"ms_total": round(ms, 3), "launches": n, "ms_avg": round(ms / n, 4)}
        for k in fam_flops:
            if k in fam_out:
                sec = fam_out[k]["ms_total"] * 1e-3
                fam_out[k]["algorithmic_tflops"] = round(fam_flops[k] / sec / 1e12, 2)
                fam_out[k]["executed_mfma_tflops"] = round(fam_exec.get(k, 0.0) / sec / 1e12, 2)
                fam_out[k]["executed_gflop_per_launch"] = round(fam_exec.get(k, 0.0) / fam_out[k]["launches"] / 1e9, 2)
        tinfo = {}
        tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tj):
            try:
                tinfo = json.load(open(tj))
            except Exception:
                tinfo = {}
        dom = max([k for k in ("zgemm_half1", "zgemm_half2", "dgemm") if k in fam_out],
                  key=lambda k: fam_out[k]["ms_total"])
        achieved = fam_out[dom]["executed_mfma_tflops"]
        traffic = tinfo.get(dom, {}).get("hbm_bytes_per_launch")
        eri_sec = sum(fam_out[k]["ms_total"] for k in fam_flops if k in fam_out) * 1e-3
        roofline = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_note": "HBM bytes per launch from rocprofv3 PMC (profiles/traffic_latest.json), collected offline",
                    "avg_launch_ms": fam_out[dom]["ms_avg"],
                    "executed_gflop_per_launch": fam_out[dom]["executed_gflop_per_launch"],
                    "algorithmic_tflops": fam_out[dom]["algorithmic_tflops"],
                    "note": "achieved = flop ISSUED to the f64 matrix pipe (library launch accounting: 3M complex product = 6 flop per "
                            "complex MAC, padded tiles, lower tile triangle of the symmetric contraction) / HIP-event time; "
                            "algorithmic_tflops counts 8 flop per complex MAC and the full npair^2 contraction (SURVEY.md 8d)",
                    "mfma_ceiling_measured": 78.1,
                    "half1_executed_tflops": fam_out.get("zgemm_half1", {}).get("executed_mfma_tflops"),
                    "half2_executed_tflops": fam_out.get("zgemm_half2", {}).get("executed_mfma_tflops"),
                    "contraction_executed_tflops": fam_out.get("dgemm", {}).get("executed_mfma_tflops"),
                    "eri_kernels_executed_tflops": round(sum(fam_exec.get(k, 0.0) for k in fam_flops) / max(eri_sec, 1e-9) / 1e12, 2),
                    "families": fam_out}
        n_mine = len(kl_mine)
        res = {
            "metric": "DMET embedding-construction iteration (diag+bath+ERI-transform): ERI-transform algorithmic TFLOP/s over the whole step",
            "value": round(flops_all / elapsed / 1e12, 3),
            "unit": "TFLOP/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: mesh %s nao %d naux %d nemb %d spin %d; %d irreducible kL per GPU "
                                   "(%d of %d in total over %d GPUs), %d DF blocks per GPU per step, Philox DF blocks "
                                   "regenerated on device inside the timed region"
                                   % (a.workload, "x".join(map(str, sysm.mesh)), sysm.nao, sysm.naux, nemb, sysm.spin,
                                      n_mine, min(n_mine * world, n_irr), n_irr, world, nblk),
                       "parallelism": "kL-sharded x%d, k-sharded diag, 1 all-reduce(rho_R) + 1 all-reduce(ERI)" % world},
            "value_executed_mfma_tflops": round(exec_all / elapsed / 1e12, 3),
            "value_executed_frac_of_peak": round(exec_all / elapsed / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world), 4),
            "iteration_wall_s": round(elapsed / a.steps, 4),
            "stage_seconds_per_step": {k: round(v / a.steps, 5) for k, v in timers.items()},
            "eri_only_tflops": round(flops / a.steps / (timers.get("eri", 1e-9) / a.steps) / 1e12, 3),
            "roofline": roofline,
        }
        if parity is not None:
            res.update(parity)
            res["parity_ok"] = bool(parity["parity_maxabs"] <= PARITY_TOL)
            res["config"]["parity"] = "max|device - oracle| = %.3e (max|ref| %.3e) on %d sampled entries of the timed ERI, tol %.0e" \
                                      % (parity["parity_maxabs"], parity["parity_ref_maxabs"], parity["parity_entries"], PARITY_TOL)
            if not res["parity_ok"]:
                rc = 3
        # ERI x density inside the step: two J passes + one J(both directions) pass + two K passes over 8.66 GB blocks
        if "jk" in fam_out and sysm.spin == 2:
            gb = 5 * 8.0 * npair * npair / 1e9
            res["emb_ham"] = {"jk_ms_per_step": round(fam_out["jk"]["ms_total"] / a.steps, 3),
                              "jk_algorithmic_GB_per_step": round(gb, 2),
                              "jk_GBps": round(gb * a.steps / (fam_out["jk"]["ms_total"] * 1e-3), 1), "hbm_peak_GBps": 8000.0}
        if a.fit_iters > 0:
            # vcor least-squares fit of the BASELINE target (config 5) at the reference's defaults (MaxIter = 300 and its
            # convergence criteria, routine/slater.py:909): measured once, outside the timed region
            fit = pipeline.vcor_fit_stage(ctx, sysm, out["basis"], nemb, out["emb_ham"]["rdm1_emb"], MaxIter=a.fit_iters)
            fit.pop("vcor")
            passes_bytes = 2.0 * fit["dV_dparam_bytes"]
            fit["note"] = ("FitVcorEmb, VcorLocal on the valence orbitals, CG with analytic gradient, run to the reference's "
                           "convergence criteria; one objective = one pass over dV_dparam + one eigh(nemb) per spin + nemb^3 algebra; "
                           "objective+gradient = two passes")
            fit["dV_stream_GBps_if_only_cost"] = round(passes_bytes / (fit["ms_per_objective_plus_gradient"] * 1e-3) / 1e9, 1)
            res["vcor_fit"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in fit.items()}
            res["iteration_plus_fit_wall_s"] = round(elapsed / a.steps + fit["seconds_total"], 4)
        if not a.no_cpu_baseline:
            th, tc, desc, threads = cpu_baseline(sysm, nemb, a.cpu_seconds, host_threads(1))
            fh, fc = out["flops_half"], out["flops_contract"]
            v = (fh + fc) / (fh / th + fc / tc)
            res["cpu_baseline"] = {"value": round(v, 4), "unit": "TFLOP/s", "cores": threads, "kind": "port",
                                   "sample": desc, "half_transform_tflops": round(th, 4),
                                   "contraction_tflops": round(tc, 4),
                                   "gpu_over_cpu": round(res["value"] / max(v, 1e-12), 1)}
        print(json.dumps(res), flush=True)
    if distributed:
        import torch.distributed as td
        dist.barrier()           # rank 0 may still have been measuring the fit / CPU baseline: tear down together
        td.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
