#!/usr/bin/env python
"""
bench.py -- one embedding-construction iteration per step on synthetic k-sampled tensors.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1 without a torch.distributed environment: this process starts N ranks itself (torch.distributed.run as a
child process, one rank per GPU over RCCL) BEFORE it touches the GPU, waits, and exits with the child's code.
Launched by `python -m torch.distributed.run ... bench.py --gpus N ...` it is one of the N ranks.

Workload (BASELINE.json config 5, "C5"): cuprate-like cell, 6x6x6 k-mesh, nao = nlo = 200, naux = 800,
UHF, 56 valence orbitals -> nemb = 256.  A step = diag (432 Hermitian 200x200) + occupations + rho_k +
k->R fold + Schmidt bath (43144 x 56 SVD per spin) + C_ao_emb + density-fitted ERI transform of this
rank's share of the irreducible momentum transfers kL + embedding Hamiltonian.

  --scaling strong (default): every timed step IS the full config -- all 112 irreducible kL / 12 152 DF blocks of C5 -- sharded
                  over the N ranks by the reference's assign_workload rule: the config the metric is quoted on, at every N
                  (N = 1: ~52 s per step; N = 8: 14 kL per GPU).  --steps K / --warmup W are honoured EXACTLY whenever the whole
                  run (W + K steps, then the checks, the fit and the CPU baseline) fits --max-total-s (default 1500 s: at
                  N = 1, 52 s per step, that is up to 19 + 5); only otherwise the timed steps are cut to
                  min(K, max(3, floor(--max-timed-s / step seconds))) = 6 after two warm-up steps (the decision is taken on the second:
                  the first also pays the one-off allocations), which keeps the run near ten minutes with the GPU busy 70 % of it;
                  --max-total-s 1800 runs all 20 + 5 (25 minutes).  The line reports
                  the counts that were run ("steps", "warmup"), the ones asked for and why they differ ("steps_requested",
                  "warmup_requested", "steps_note": top level AND inside "config").  One extra pass over a 14-kL shard per GPU
                  (the 8-GPU share) is reported under "shard_pass" as a secondary rate.
  --scaling weak: every GPU transforms --kl-per-gpu irreducible kL per step (14 = 112 / 8, so N = 8 is exactly the full C5
                  iteration); after the timed region ONE pass over the full config is reported under "full_config".

The DF blocks of the timed kL shard are RESIDENT in HBM when they fit (--df auto: C3, C4 -- 85 GB; loaded once before the timed
region, what a DMET run does with a DF tensor that fits) and otherwise regenerated on the device (Philox) inside every step,
standing in for the reference's HDF5 reads (C5: 6.2 TB).  All other inputs are resident in HBM before the clock starts; the line
says which under "input".

Prints ONE JSON line (rank 0):
  value            = ALGORITHMIC FP64 flop of the ERI transform (8 flop per complex
                     multiply-add, full npair^2 contraction) of all ranks / wall-clock of the whole step (max over ranks)
  roofline         = the dominant kernel family: flop ISSUED to the f64 matrix pipe (counted by the library at launch:
                     3M complex products, padded tiles, lower tile triangle of the symmetric contraction) per launch /
                     its average HIP-event duration, against the FP64 MFMA peak -- a true fraction (<= 1)
  full_config      = one untimed-in-`value` pass over the whole config (see above) with its own stage times and rate
  parity_*         = (a) every stage upstream of the ERI at full size against the oracle on the same seeded inputs
                     (oracle/stage_check.py: eigenvalues, occupations, mu, rho_R, bath projector, C_ao_emb), (b) sampled
                     entries of the TIMED ERI against the C oracle (oracle/eri_sample.py: all auxiliary rows, all AO blocks,
                     pairs of embedding orbitals drawn per run from --parity-seed) and (c) a Freivalds check of the
                     contraction over ALL pair rows and columns: eri[b] x against sum_kL w X_a^T (X_b x) accumulated from
                     the resident planes by kernels independent of the tiled GEMM (dmk_eri_probe); any of them above
                     tolerance makes the run exit non-zero
  cpu_baseline     = the oracle (numpy / OpenBLAS port of the reference's loop) on a bounded sample of the same workload:
                     whole AO blocks of one weight-1 and one weight-2 kL + their contractions, extrapolated by the exact
                     block counts, this host's cores
"""
import argparse
import hashlib
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
HBM_PEAK_GBPS = 8000.0
PARITY_TOL = 1e-8                # north star: <= 1e-8 max-abs on the transformed ERI
# where the DF blocks of the timed region come from, stated next to `value` (PCIe note)
INPUT_NOTE = "device-generated (Philox) inside the timed region; host-fed blocks are PCIe-bound: est. 2.6x the step time at C5"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None, help="timed steps (default: 2 for the DF workloads C3 - C5, 300 for the model "
                                                            "lattices C1 / C2, whose step is 0.1 - 0.2 ms)")
    p.add_argument("--warmup", type=int, default=None, help="untimed steps first (default 1 / 30)")
    p.add_argument("--workload", default="C5")
    p.add_argument("--scaling", choices=("weak", "strong"), default="strong")
    p.add_argument("--max-timed-s", type=float, default=330.0,
                   help="cap of the timed region when the requested counts do not fit --max-total-s: timed steps = "
                        "min(--steps, max(3, floor(this / seconds of one step)))")
    p.add_argument("--max-total-s", type=float, default=1500.0,
                   help="wall-clock this whole process may take (the driver's window is 1800 s): --steps / --warmup are honoured "
                        "EXACTLY whenever warm-up + timed steps + the checks after them fit it")
    p.add_argument("--parity-seed", type=int, default=-1, help="seed of the sampled embedding orbitals (-1: from the clock)")
    p.add_argument("--df", choices=("auto", "resident", "regenerate"), default="auto",
                   help="AO DF blocks of the timed kL shard: resident in HBM (loaded once before the timed region, read in place "
                        "by every step: what a DMET run does when the DF tensor fits -- C4: 85 GB) or regenerated by the device "
                        "generator inside every step (C5: 6.2 TB does not fit); auto = resident when the shard fits in 45 %% of "
                        "the free device memory")
    p.add_argument("--no-shard-pass", action="store_true", help="strong scaling: skip the extra 14-kL-per-GPU pass")
    p.add_argument("--kl-per-gpu", type=int, default=14, help="weak scaling: irreducible kL transformed per GPU per step")
    p.add_argument("--max-blocks-per-kl", type=int, default=0, help="debug: truncate the i-loop (0 = all)")
    p.add_argument("--no-full-config", action="store_true", help="skip the pass over the whole config after the timed steps")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-parity", action="store_true", help="skip every oracle check")
    p.add_argument("--parity-budget-s", type=float, default=260.0,
                   help="host-time budget of the ERI oracle; the full-config check is dropped first, then the timed check is cut "
                        "to an UN-timed re-run of the first kL of every shard -- the line says which")
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


def blas_build():
    try:
        from threadpoolctl import threadpool_info
        return "; ".join("%s %s (%s threads)" % (i.get("internal_api"), i.get("version"), i.get("num_threads")) for i in threadpool_info())
    except Exception:
        return "unknown"


def cpu_baseline(sysm, nemb, flops_half_full, flops_contract_full, budget_s, threads):
    """SURVEY.md section 8d / BASELINE.md section 3: the oracle (numpy restatement of the reference's loop, same BLAS entry
    points) on 2 sampled kL -- the first weight-1 and the first weight-2 kL of the plan -- with WHOLE AO blocks (all naux rows,
    both spins): per visited block the Philox block, transform_ao_to_emb (r_e2 restated), hermi_sum where the plan says
    so, pack_tril and the accumulation into Lij_s4 (eri_transform.py:338-378); then _Lij_s4_to_eri (:451-478) of that
    kL on a column sample of the pair space.  As many blocks as fit the time budget are run; the config's ERI time is
    extrapolated by the exact block counts and contraction flop."""
    from oracle import restate as R
    from oracle import eri_sample as ES
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    ES.set_threads(threads)
    nao, naux, spin, nk = sysm.nao, sysm.naux, sysm.spin, sysm.nk
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(0)
    Cemb = (rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb))) / nao
    weights, by = ES.plan_records(sysm.mesh)
    irr = sorted(by)
    picks = [next(k for k in irr if weights[k] == 1)] + [k for k in irr if weights[k] == 2][:1]
    t_block, n_block, t_con_full = [], 0, 0.0
    ncs = min(npair, 4096)
    cols = np.linspace(0, npair - 1, ncs).astype(int)
    per_kl = budget_s / (len(picks) + 1.0)
    detail = []
    for kL in picks:
        Lij_s4 = np.zeros((spin, naux, npair), dtype=np.complex128)
        t0 = time.perf_counter()
        done = 0
        for (_, i, j, jm, sym) in by[kL]:
            tb = time.perf_counter()
            blk = ES.df_block_philox(sysm.df.seed, i, j, naux, nao).reshape(naux, -1)
            Lij = R.transform_ao_to_emb(blk, Cemb, i, j)                          # (spin, naux, nemb, nemb)
            if sym:
                Lij = Lij + Lij.transpose(0, 1, 3, 2)                                 # lib.hermi_sum, no conjugation
            Lij_s4 += R.pack_tril(Lij)
            t_block.append(time.perf_counter() - tb)
            done += 1
            if time.perf_counter() - t0 > per_kl:
                break
        n_block += done
        # contraction of this kL on a column sample (area-extrapolated): the .real / .imag copies and lib.dot calls of :451-478
        w = int(weights[kL])
        sub = np.ascontiguousarray(Lij_s4[:, :, cols])
        eri = np.zeros((spin * (spin + 1) // 2, ncs, ncs))
        tc = time.perf_counter()
        R.Lij_s4_to_eri(sub, eri, weight=w, t_reversal_symm=True)
        tc = time.perf_counter() - tc
        full = tc * (float(npair) / ncs) ** 2
        n_w = int(sum(1 for k in irr if weights[k] == w))
        t_con_full += full * n_w
        detail.append({"kL": int(kL), "weight": w, "blocks_run": done, "blocks_in_kL": len(by[kL]),
                       "contraction_sample_s": round(tc, 3), "contraction_full_extrapolated_s": round(full, 2)})
    if limiter is not None and hasattr(limiter, "restore_original_limits"):
        limiter.restore_original_limits()
    nblocks_cfg = int(sum(len(v) for v in by.values()))
    sec_block = float(np.mean(t_block))
    t_half_full = sec_block * nblocks_cfg
    total_s = t_half_full + t_con_full
    f_block = spin * (8.0 * naux * nao * nao * nemb + 8.0 * naux * nao * nemb * nemb)
    return {"value": round((flops_half_full + flops_contract_full) / total_s / 1e12, 4), "unit": "TFLOP/s", "cores": threads,
            "kind": "port",
            "sample": "oracle (oracle/restate.py: numpy restatement of get_emb_eri_fast_gdf, %s) on 2 sampled kL of %s (kL %s: weight "
                      "1 and weight 2) with WHOLE AO blocks (all %d auxiliary rows, both spins): %d blocks run in the time budget "
                      "(%.2f s per block incl. the Philox block, hermi_sum, pack_tril, accumulation) + _Lij_s4_to_eri of each kL on %d of "
                      "%d pair columns; extrapolated to the config by its exact counts: %d blocks and %d + %d contractions"
                      % (blas_build(), sysm.name, picks, naux, n_block, sec_block, ncs, npair, nblocks_cfg,
                         int(sum(1 for k in irr if weights[k] == 1)), int(sum(1 for k in irr if weights[k] == 2))),
            "seconds_per_block": round(sec_block, 3), "half_transform_tflops": round(f_block / sec_block / 1e12, 4),
            "contraction_tflops": round(flops_contract_full / t_con_full / 1e12, 4),
            "extrapolated_eri_transform_s": round(total_s, 1), "kL_detail": detail}


_JSON_FD = None


def emit(res):
    """The one JSON line, on the process's ORIGINAL stdout."""
    line = (json.dumps(res) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def parity_sample(nemb, seed):
    """Embedding orbitals whose pairs are checked against the oracle: both ends plus three drawn from `seed` -- two from the
    lower and one from the upper half of the orbital range (15 pair columns), so that every workgroup type of the step-2 kernels
    (diagonal triangles, off-diagonal rectangles) is hit whatever the draw.  The position-dependent part of the contraction is
    covered by the Freivalds check, not by this sample.  (Seven orbitals until round 4; the oracle's time is mostly the Philox
    regeneration of every visited block, the sampled columns add 6 % each.)"""
    rng = np.random.default_rng(seed)
    half = max(1, nemb // 2)
    lo = rng.choice(np.arange(1, half), size=min(2, max(0, half - 1)), replace=False) if half > 1 else []
    hi = rng.choice(np.arange(half, max(half + 1, nemb - 1)), size=min(1, max(1, nemb - 1 - half)), replace=False)
    return sorted({0, nemb - 1} | {int(x) for x in lo} | {int(x) for x in hi})


def kernel_source_sha():
    """Fingerprint of the HIP sources: profiles/traffic_latest.json records the one its PMC passes were collected on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "libdmet_preview_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def fetch_rows(eri_dev, spin_pair, npair, idx, table=None):
    """Sampled entries eri[b][P, Q], P, Q in idx.  `table`: ownership of a row-sharded ERI -- every rank contributes the rows
    it owns and the sample is assembled over ranks (all ranks must call)."""
    if table is None:
        return np.stack([np.stack([eri_dev.offset((b * npair + int(r)) * npair, (npair,)).get()[idx] for r in idx])
                         for b in range(spin_pair)])
    from libdmet_preview_amd.parallel import dist
    return dist.gather_rows_numpy(eri_dev, spin_pair, npair, idx, table)[:, :, idx]


def main():
    t_process_start = time.perf_counter()
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the JSON of rank 0.  Native libraries write there too (RCCL prints a five-line version
    # banner through C stdio when the process exits, i.e. AFTER anything Python printed): from here on file descriptor 1 is
    # stderr for everybody, and the JSON goes to the saved descriptor.
    sys.stdout.flush()
    global _JSON_FD
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
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
    # DMK_BENCH_OVERRIDE (JSON: mesh / nlo / naux / nval / spin) shrinks the named workload for the tests that drive this script
    over = json.loads(os.environ.get("DMK_BENCH_OVERRIDE", "{}"))
    sysm = pipeline.SyntheticSystem.from_workload(ctx, a.workload, **over)
    maxblk = a.max_blocks_per_kl or None
    model = sysm.naux == 0
    if a.steps is None:
        a.steps = 300 if model else 2
    if a.warmup is None:
        a.warmup = 30 if model else 1

    # ---- shards -------------------------------------------------------------------------------------------------
    kl_full_mine, n_irr, kl_mine = [], 0, []
    if not model:
        w, _ = et.eri_plan(sysm.mesh, True)
        irr1 = [k for k in range(len(w)) if w[k] == 1]
        irr2 = [k for k in range(len(w)) if w[k] == 2]
        n_irr = len(irr1) + len(irr2)
        # the reference's static partition of ALL irreducible kL over the ranks (eri_transform_mpi.py:35-55)
        kl_full_mine = et.assign_workload(sysm.mesh, world, True)[rank]
        if a.scaling == "strong":
            kl_mine = kl_full_mine
        else:
            # weak scaling: the irreducible kL list is cut into shards of --kl-per-gpu; rank r takes shard r
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

    nemb_guess = sysm.nlo + sysm.nval
    npair = nemb_guess * (nemb_guess + 1) // 2
    spin_pair = sysm.spin * (sysm.spin + 1) // 2
    eri_dev = None if model else ctx.zeros((spin_pair, npair, npair), np.float64)

    # AO DF blocks of the timed shard: resident in HBM when they fit (--df), else regenerated inside every step
    df_resident_bytes = 0
    if not model and a.df != "regenerate":
        df_resident_bytes = sysm.make_df_resident(kl_mine, 0.45 if a.df == "auto" else 0.8)
        if distributed:          # every rank takes the same path
            mn = dist.all_reduce_sum_numpy(np.array([1.0 if df_resident_bytes > 0 else 0.0]))[0]
            if mn < world and sysm.df_resident is not None:
                sysm.df_resident.free()
                sysm.df_resident, df_resident_bytes = None, 0
        if a.df == "resident" and df_resident_bytes == 0:
            raise SystemExit("--df resident: the DF blocks of the shard do not fit in device memory")
    input_note = INPUT_NOTE if df_resident_bytes == 0 else \
        "DF blocks of the timed shard RESIDENT in HBM (%.1f GB per GPU, generated once by the device Philox generator before the " \
        "timed region, read in place by every step)" % (df_resident_bytes / 1e9)

    # seed of everything the checks draw per run (sampled orbitals, probe vector): the same on every rank
    pseed = a.parity_seed if a.parity_seed >= 0 else int(time.time()) % (1 << 31)
    if distributed:
        sl = np.zeros(world)
        sl[0] = pseed if rank == 0 else 0
        pseed = int(dist.all_reduce_sum_numpy(sl)[0])
    # Freivalds probe of the contraction: x random in [-1, 1], yref accumulated by the pipeline itself (dmk_eri_probe)
    d_px = d_py = None
    if eri_dev is not None and not a.no_parity:
        d_px = ctx.to_device(np.random.default_rng(pseed + 1).uniform(-1.0, 1.0, npair))
        d_py = ctx.zeros((spin_pair, npair), np.float64)

    def step(timers, kls):
        if eri_dev is not None:
            eri_dev.zero_()
        if d_py is not None:
            d_py.zero_()
        # multi-rank: finished bands of ERI rows are reduced to their owners underneath the contraction, the sum stays row-sharded
        return pipeline.iteration(ctx, sysm, eri_dev=eri_dev, kL_list=kls, timers=timers, max_blocks_per_kL=maxblk,
                                  eri_exchange="row_sharded" if distributed else "none",
                                  eri_probe=None if d_px is None else (d_px, d_py))

    def fence():
        ctx.sync()
        if distributed:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    def max_over_ranks(x):
        if not distributed:
            return x
        slots = np.zeros(world)
        slots[rank] = x
        return float(dist.all_reduce_sum_numpy(slots).max())

    def freivalds(o):
        """eri[b] x (streaming row dots on the finished ERI) against the yref the pipeline accumulated from its planes: covers
        every row and column of every spin block.  Row-sharded ERI: every rank contributes the rows it owns; yref is summed
        over the kL shards like the ERI itself.  Returns (max |eri x - yref|, max |yref|) on every rank."""
        y = et.eri_times_vector_dev(ctx, eri_dev, spin_pair, npair, d_px).get()
        yref = d_py.get()
        if distributed:
            keep = np.zeros_like(y)
            for (lo, hi) in dist.owned_ranges(o["eri_rows"]):
                keep[:, lo:hi] = y[:, lo:hi]
            y = dist.all_reduce_sum_numpy(keep)
            yref = dist.all_reduce_sum_numpy(yref)
        return float(np.abs(y - yref).max()), float(np.abs(yref).max())

    # ---- warm-up and the number of timed steps -------------------------------------------------------------------------
    nsteps, nwarm = a.steps, a.warmup
    if a.warmup > 0:
        fence()
        tw0 = time.perf_counter()
        out = step({}, kl_mine)
        fence()
        t_w = max_over_ranks(time.perf_counter() - tw0)
        warm_done = 1
        if not model and a.warmup >= 2:
            # the first step also pays the one-off allocations (ERI workspace, plane stack): the decision below is taken on the
            # SECOND warm-up step, which costs what a timed step costs
            fence()
            tw0 = time.perf_counter()
            out = step({}, kl_mine)
            fence()
            t_w = max_over_ranks(time.perf_counter() - tw0)
            warm_done = 2
        if not model:
            # A whole-config step at N = 1 takes ~52 s.  The counts asked for are honoured EXACTLY whenever the rest of the
            # warm-up, the timed steps and what follows them (shard pass, oracle checks within --parity-budget-s, fit, CPU
            # baseline) fit --max-total-s; only otherwise are they cut to what fits --max-timed-s (never below 3 + 1)
            after = 0.0 if a.no_parity else min(a.parity_budget_s, 110.0) + 20.0
            after += (0.0 if a.no_cpu_baseline else a.cpu_seconds + 5.0) + (10.0 if a.fit_iters > 0 else 0.0) + 0.2 * t_w + 30.0
            left = a.max_total_s - (time.perf_counter() - t_process_start) - after
            if (a.steps + a.warmup - warm_done) * t_w * 1.03 > left:
                nsteps = min(a.steps, max(3, int(a.max_timed_s // max(t_w, 1e-6))))
                nwarm = max(warm_done, min(a.warmup, max(1, int(0.3 * a.max_timed_s // max(t_w, 1e-6)))))
            if distributed:          # every rank must run the same counts
                c = dist.all_reduce_sum_numpy(np.array([float(nsteps), float(nwarm)]) * (1.0 if rank == 0 else 0.0))
                nsteps, nwarm = int(round(c[0])), int(round(c[1]))
        for _ in range(nwarm - warm_done):
            out = step({}, kl_mine)
    fence()
    ctx.profile(True)
    ctx.profile_read(reset=True)
    ctx.profile_read_flops(reset=True)
    timers = {}
    t0 = time.perf_counter()
    for _ in range(nsteps):
        out = step(timers, kl_mine)
    fence()
    t1 = time.perf_counter()
    fam = ctx.profile_read(reset=True)
    fam_exec = ctx.profile_read_flops(reset=True)
    ctx.profile(False)
    elapsed = max_over_ranks(t1 - t0)
    nemb = out["nemb"]
    npair = nemb * (nemb + 1) // 2

    if model:
        model_line(a, ctx, sysm, out, fam, timers, elapsed, world, rank, distributed)
        return
    a.steps_requested, a.warmup_requested = a.steps, a.warmup
    a.steps, a.warmup = nsteps, nwarm                        # from here on: the counts that were run

    # ---- per-rank view of the timed region (N > 1): stage seconds, DF blocks and bytes put on the wire per step ------------
    per_rank = None
    if distributed:
        keys = sorted(timers)
        sl = np.zeros((world, len(keys) + 2))
        sl[rank, :len(keys)] = [timers[k] / nsteps for k in keys]
        sl[rank, len(keys)] = out["nblocks"]
        n_, nk_, sp_ = sysm.nlo, sysm.nk, sysm.spin
        sent = sp_ * nk_ * n_ * n_ * 8 + sp_ * nk_ * n_ * 8 + 3 * sp_ * nemb * nemb * 8        # rho_R, eigenvalue table, J / K partials
        if out.get("eri_rows"):
            sent += sum(hi - lo for (lo, hi, o) in out["eri_rows"] if o != rank) * npair * 8 * spin_pair   # partial row bands -> owners
        sl[rank, len(keys) + 1] = sent
        sl = dist.all_reduce_sum_numpy(sl)
        per_rank = [{"rank": r, "kL": None, "blocks": int(sl[r, len(keys)]), "bytes_sent_per_step": int(sl[r, len(keys) + 1]),
                     "stage_seconds_per_step": {k: round(float(sl[r, i]), 5) for i, k in enumerate(keys)}} for r in range(world)]
        for r in range(world):
            per_rank[r]["kL"] = len(et.assign_workload(sysm.mesh, world, True)[r]) if a.scaling == "strong" else len(kl_mine)

    fh_timed, fc_timed = out["flops_half"], out["flops_contract"]
    flops = (fh_timed + fc_timed) * a.steps
    exec_mine = sum(fam_exec.get(k, 0.0) for k in ("zgemm_half1", "zgemm_half2", "dgemm"))
    if distributed:
        agg = dist.all_reduce_sum_numpy(np.array([flops, exec_mine]))
        flops_all, exec_all = float(agg[0]), float(agg[1])
    else:
        flops_all, exec_all = flops, exec_mine
    nblk_timed = out["nblocks"]
    timed_is_full = (a.scaling == "strong") or (len(kl_mine) * world >= n_irr and maxblk is None)

    # sampled rows of the TIMED ERI and its Freivalds residual, taken before the buffer is reused by a later pass
    A = parity_sample(nemb, pseed)
    from oracle import eri_sample as ES                      # checker only
    pidx = np.asarray([p[2] for p in ES.sample_pairs(A)])
    got_timed = None
    frv_timed = frv_full = None
    if not a.no_parity:
        got_timed = fetch_rows(eri_dev, spin_pair, npair, pidx, out.get("eri_rows"))
        frv_timed = freivalds(out)

    # ---- weak scaling: ONE pass over the full config (all irreducible kL over all ranks), timed on its own -------------
    full = None
    got_full = None
    if not timed_is_full and not a.no_full_config:
        ftimers = {}
        fence()
        tf0 = time.perf_counter()
        fout = step(ftimers, kl_full_mine)
        fence()
        tf = max_over_ranks(time.perf_counter() - tf0)
        fl = fout["flops_half"] + fout["flops_contract"]
        nb = fout["nblocks"]
        if distributed:
            agg = dist.all_reduce_sum_numpy(np.array([fl, float(nb)]))
            fl, nb = float(agg[0]), int(agg[1])
        full = {"workload": "%s whole config: all %d irreducible kL, %d DF blocks, sharded over %d GPU(s) by assign_workload"
                            % (a.workload, n_irr, nb, world),
                "n_gpus": world, "iteration_wall_s": round(tf, 3), "eri_algorithmic_flop": fl,
                "iteration_tflops": round(fl / tf / 1e12, 3),
                "eri_only_tflops": round(fl / world / max(ftimers.get("eri", 1e-9), 1e-9) / 1e12, 3),
                "stage_seconds": {k: round(v, 5) for k, v in ftimers.items()}}
        out = fout
        if not a.no_parity:
            got_full = fetch_rows(eri_dev, spin_pair, npair, pidx, fout.get("eri_rows"))
            frv_full = freivalds(fout)

    # ---- strong scaling: one extra pass over the 14-kL-per-GPU shard (the share of one GPU of an 8-GPU run), a secondary rate
    shard = None
    if a.scaling == "strong" and not a.no_shard_pass and maxblk is None and a.kl_per_gpu * world < n_irr:
        nshards = max(1, (n_irr + a.kl_per_gpu - 1) // a.kl_per_gpu)
        shards = [[] for _ in range(nshards)]
        for i, k in enumerate(irr1):
            shards[i % nshards].append(k)
        it2 = iter(irr2)
        for sh in shards:
            while len(sh) < a.kl_per_gpu:
                k = next(it2, None)
                if k is None:
                    break
                sh.append(k)
        stimers = {}
        fence()
        ts0 = time.perf_counter()
        sout = step(stimers, shards[rank % nshards])
        fence()
        tsh = max_over_ranks(time.perf_counter() - ts0)
        fl = sout["flops_half"] + sout["flops_contract"]
        if distributed:
            fl = float(dist.all_reduce_sum_numpy(np.array([fl]))[0])
        shard = {"kl_per_gpu": a.kl_per_gpu, "iteration_wall_s": round(tsh, 4), "iteration_tflops": round(fl / tsh / 1e12, 3),
                 "stage_seconds": {k: round(v, 5) for k, v in stimers.items()},
                 "note": "ONE pass (after the timed steps, not part of `value`) over %d irreducible kL per GPU: the per-GPU share of "
                         "the 8-GPU target configuration" % a.kl_per_gpu}

    # ---- oracle checks ---------------------------------------------------------------------------------------------------
    parity, stage_parity = None, None
    if not a.no_parity:
        threads = host_threads(world)
        ES.set_threads(threads)
        if rank == 0:
            # (a) everything upstream of the ERI, at full size, against an independent computation on the same seeded inputs
            from oracle import stage_check as SC
            n, nk, spin = sysm.nlo, sysm.nk, sysm.spin
            got = {"ew": out["ew"].get().reshape(spin, nk, n), "occ": out["occ"].get().reshape(spin, nk, n), "mu": out["mu"],
                   "rho_R": out["rho_R"].get().reshape(spin, nk, n, n), "basis": out["basis"].get().reshape(spin, nk, n, nemb),
                   "sigma": out["sigma"], "C_ao_emb": out["C_ao_emb"].get().reshape(spin, nk, sysm.nao, nemb)}
            stage_parity = SC.compare(sysm.mesh, sysm.Fock_R, sysm.vcor, sysm.filling, sysm.restricted, sysm.imp_idx, sysm.val_idx,
                                      sysm.C_ao_lo, got)
            del got
        # (b) sampled entries of the ERI: every rank evaluates the oracle on its own kL shard, the tiny samples are summed like
        #     the ERI itself.  Cost = the Philox regeneration of every visited block, ~0.16 core-seconds per C5 block
        tp = time.perf_counter()
        C_host = out["C_ao_emb"].get().reshape(sysm.spin, sysm.nk, sysm.nao, nemb)
        per_block = 0.16 * (sysm.naux * sysm.nao ** 2 / (800.0 * 200 ** 2)) / threads
        blocks_of = lambda kls: sum(len(v) for v in ES.plan_records(sysm.mesh, set(kls))[1].values()) if kls else 0
        extra = [k for k in kl_full_mine if k not in set(kl_mine)]
        nested = full is not None and set(kl_mine) <= set(kl_full_mine)         # N = 1: the timed shard is part of the full one
        est_timed = blocks_of(kl_mine) * per_block
        est_full = (blocks_of(extra) if nested else blocks_of(kl_full_mine)) * per_block if full is not None else 0.0
        if distributed:       # shards differ slightly in their block counts: every rank must take the same branch below
            e = dist.all_reduce_sum_numpy(np.array([est_timed, est_full])) / world
            est_timed, est_full = float(e[0]), float(e[1])
        do_full = full is not None and est_timed + est_full <= a.parity_budget_s
        check_kl, check_got, scope = kl_mine, got_timed, "timed ERI"
        check_dev = None
        if est_timed > a.parity_budget_s:
            keep = max(1, int(len(kl_mine) * a.parity_budget_s / est_timed))
            check_kl = kl_mine[:keep]
            check_dev = ctx.zeros((spin_pair, npair, npair), np.float64) if full is None else eri_dev
            check_dev.zero_()
            pipeline.eri_stage(ctx, sysm, out["C_ao_emb"], nemb, check_dev, check_kl, {}, maxblk,
                               exchange="allreduce" if distributed else None)
            check_got = fetch_rows(check_dev, spin_pair, npair, pidx)
            scope = "UN-timed re-run (oracle budget %.0f s < %.0f s for the timed shard)" % (a.parity_budget_s, est_timed)
        ref_t, idx, _ = ES.eri_sample(sysm.mesh, sysm.df.seed, C_host, sysm.naux, A, check_kl, max_blocks_per_kL=maxblk)
        assert np.array_equal(idx, pidx)
        ref_f = None
        if do_full:
            rest = extra if nested else kl_full_mine
            ref_f = ES.eri_sample(sysm.mesh, sysm.df.seed, C_host, sysm.naux, A, rest, max_blocks_per_kL=maxblk)[0] if rest \
                else np.zeros_like(ref_t)
            if nested and check_kl is kl_mine:
                ref_f = ref_f + ref_t
            elif nested:                        # the timed check was cut: the full reference needs the whole timed shard
                ref_f = ref_f + ES.eri_sample(sysm.mesh, sysm.df.seed, C_host, sysm.naux, A, kl_mine, max_blocks_per_kL=maxblk)[0]
        if distributed:
            ref_t = dist.all_reduce_sum_numpy(ref_t)
            if ref_f is not None:
                ref_f = dist.all_reduce_sum_numpy(ref_f)
        if rank == 0:
            err = float(np.abs(check_got - ref_t).max())
            parity = {"parity_maxabs": err, "parity_ref_maxabs": float(np.abs(ref_t).max()),
                      "parity_entries": int(ref_t.size), "parity_orbitals": A, "parity_threads_per_rank": threads,
                      "parity_scope": "%s of all %d ranks: %d kL, all AO blocks, all %d auxiliary rows, %d sampled pair columns"
                                      % (scope, world, len(check_kl) * world, sysm.naux, len(idx))}
            if full is not None:
                if ref_f is not None:
                    ferr = float(np.abs(got_full - ref_f).max())
                    full.update({"parity_maxabs": ferr, "parity_ref_maxabs": float(np.abs(ref_f).max()),
                                 "parity_entries": int(ref_f.size), "parity_ok": bool(ferr <= PARITY_TOL),
                                 "parity_scope": "full-config ERI: all %d kL, all AO blocks, all %d auxiliary rows, %d sampled pair columns"
                                                 % (n_irr, sysm.naux, len(idx))})
                else:
                    full["parity_scope"] = "not checked: oracle estimate %.0f s over the budget of %.0f s" \
                                           % (est_timed + est_full, a.parity_budget_s)
            parity["parity_seconds"] = round(time.perf_counter() - tp, 2)
            parity["parity_seed"] = pseed
            # (c) Freivalds residual of the contraction: ALL pair rows and columns of every spin block
            ftol = lambda ymax: 1e-10 * max(1.0, ymax)
            parity["parity_freivalds_maxabs"], parity["parity_freivalds_ref_maxabs"] = frv_timed
            parity["parity_freivalds_ok"] = bool(frv_timed[0] <= ftol(frv_timed[1]))
            parity["parity_tile_rows_covered"] = "all"
            parity["parity_freivalds_scope"] = ("timed ERI: |eri[b] x - sum_kL w X_a^T (X_b x)| over all %d pair rows x %d spin blocks, x uniform "
                                                "in [-1, 1] from seed %d, yref from the resident planes by streaming kernels independent of the "
                                                "tiled contraction (dmk_eri_probe); tol 1e-10 max(1, |yref|max)" % (npair, spin_pair, pseed + 1))
            if full is not None and frv_full is not None:
                full["parity_freivalds_maxabs"], full["parity_freivalds_ref_maxabs"] = frv_full
                full["parity_freivalds_ok"] = bool(frv_full[0] <= ftol(frv_full[1]))

    fit = None
    if a.fit_iters > 0:
        fit = pipeline.vcor_fit_stage(ctx, sysm, out["basis"], nemb, out["emb_ham"]["rdm1_emb"], MaxIter=a.fit_iters)
    rc = 0
    if rank == 0:
        # algorithmic flop of each ERI kernel family over the timed region (DESIGN.md
        # section 5); a launch of the half transform covers up to DMK_ERI_GROUP queued AO blocks: rates are totals / totals
        fam_flops = {
            "zgemm_half1": 8.0 * sysm.naux * sysm.nao * sysm.nao * nemb * sysm.spin * nblk_timed * a.steps,
            "zgemm_half2": 8.0 * sysm.naux * sysm.nao * nemb * nemb * sysm.spin * nblk_timed * a.steps,
            "dgemm": fc_timed * a.steps,
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
        # HBM bytes per launch come from separate rocprofv3 --pmc passes (they cannot run inside this process); the file
        # records the fingerprint of the kernel sources it was collected on: a stale file yields null, not an old number
        # one section per workload: bytes per launch of C5 launches say nothing about another shape
        if "workloads" in tinfo:
            tinfo = tinfo["workloads"].get(a.workload, {})
        coll = tinfo.get("_collected")
        coll = coll if isinstance(coll, dict) else {}
        sha_now, sha_rec = kernel_source_sha(), coll.get("kernel_source_sha")
        same_shape = coll.get("workload", "C5") == a.workload
        fresh = sha_rec == sha_now and same_shape
        traffic = tinfo.get(dom, {}).get("hbm_bytes_per_launch") if fresh else None
        eri_sec = sum(fam_out[k]["ms_total"] for k in fam_flops if k in fam_out) * 1e-3
        roofline = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_note": ("HBM bytes per launch from rocprofv3 PMC (profiles/traffic_latest.json: FETCH_SIZE x2 + WRITE_SIZE, "
                                     "separate --pmc passes), collected on commit %s, kernel sources %s"
                                     % (coll.get("commit"), sha_rec)) if fresh else
                                    ("null: profiles/traffic_latest.json was collected on kernel sources %s and workload %s, this run "
                                     "is %s on %s" % (sha_rec, coll.get("workload", "C5"), sha_now, a.workload)),
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
        # the HBM-bound stages the north star names, per launch, against the 8 TB/s roof (algorithmic bytes of SURVEY.md 8d)
        n, nk, spin = sysm.nlo, sysm.nk, sysm.spin
        hbm = {}
        alg = {"eigh": 2.0 * 16 * spin * nk * n * n + 8.0 * spin * nk * n,
               "bath": 8.0 * spin * nk * n * (sysm.nval + nemb),
               "fold": 2.0 * 16 * spin * nk * n * n}
        for k, by in alg.items():
            if k in fam_out:
                ms_step = fam_out[k]["ms_total"] / a.steps
                hbm[k] = {"ms_per_step": round(ms_step, 4), "launches_per_step": round(fam_out[k]["launches"] / a.steps, 1),
                          "algorithmic_GB_per_step": round(by / 1e9, 4), "GBps": round(by / (ms_step * 1e-3) / 1e9, 1),
                          "frac_of_hbm_peak": round(by / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}
        if "eigh" in hbm:
            # the eigensolver is flop / latency bound, not bandwidth bound (PMC: 3.45x the algorithmic bytes at 1 % of the HBM roof):
            # its own rate against the (10/3) n^3 complex-Hermitian flop model (x 4 real flop per complex multiply-add pair)
            fl = spin * nk * (10.0 / 3.0) * 4.0 * float(n) ** 3
            hbm["eigh"]["flop_model"] = "(10/3) n^3 x 4 per matrix (tridiagonalisation + eigenvectors + back-transformation)"
            hbm["eigh"]["tflops_of_flop_model"] = round(fl / (hbm["eigh"]["ms_per_step"] * 1e-3) / 1e12, 3)
        if fresh:
            # measured HBM bytes of the same stages (rocprofv3 PMC, profiles/traffic_latest.json): traffic / algorithmic = re-reads
            # per STEP: a family's bytes per launch x its launches per step of the profiled run (older files: one launch per step)
            fam_bytes = lambda *ks: sum(tinfo.get(k, {}).get("hbm_bytes_per_step", tinfo.get(k, {}).get("hbm_bytes_per_launch", 0.0))
                                        for k in ks)
            fam_launch_bytes = lambda *ks: sum(tinfo.get(k, {}).get("hbm_bytes_per_launch", 0.0) for k in ks)
            eb = fam_bytes("eigh_tridiag", "eigh_tripairs", "eigh", "eigh_tfactor", "eigh_backtransform")
            if "eigh" in hbm and eb > 0:
                hbm["eigh"]["pmc_traffic_GB_per_step"] = round(eb / 1e9, 3)
                hbm["eigh"]["pmc_over_algorithmic"] = round(eb / alg["eigh"], 2)
            fb = fam_launch_bytes("fold_k2R")
            if "fold" in hbm and fb > 0:
                hbm["fold"]["k2R_pmc_traffic_GB_per_launch"] = round(fb / 1e9, 3)
                hbm["fold"]["k2R_algorithmic_GB_per_launch"] = round((16.0 + 8.0) * spin * nk * n * n / 1e9, 3)
        roofline["hbm_stages"] = hbm
        n_mine = len(kl_mine)
        steps_note = "as requested" if (a.steps, a.warmup) == (a.steps_requested, a.warmup_requested) else \
            "cut: %.0f s/step x (%d + %d) does not fit --max-total-s %g" % (elapsed / a.steps, a.steps_requested,
                                                                            a.warmup_requested, a.max_total_s)
        res = {
            "metric": "DMET embedding-construction iteration (diag+bath+ERI-transform): ERI-transform algorithmic TFLOP/s over the whole step",
            "value": round(flops_all / elapsed / 1e12, 3),
            "unit": "TFLOP/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "steps_requested": a.steps_requested, "warmup_requested": a.warmup_requested,
            "steps_note": steps_note, "input": input_note,
            "df_mode": "regenerate" if df_resident_bytes == 0 else "resident", "df_resident_GB_per_gpu": round(df_resident_bytes / 1e9, 2),
            "ms_per_step": round(elapsed / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "vs_baseline_note": "null by contract: the reference publishes no number for this metric (BASELINE.md section 1); "
                                "the measured CPU path is under cpu_baseline, its ratio under vs_cpu_baseline",
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: mesh %s nao %d naux %d nemb %d spin %d; timed step = %d irreducible kL per GPU "
                                   "(%d of %d in total over %d GPUs), %d DF blocks per GPU per step, %s%s"
                                   % (a.workload, "x".join(map(str, sysm.mesh)), sysm.nao, sysm.naux, nemb, sysm.spin,
                                      n_mine, min(n_mine * world, n_irr) if a.scaling == "weak" else n_irr, n_irr, world, nblk_timed,
                                      "Philox DF blocks regenerated on device inside the timed region" if df_resident_bytes == 0 else
                                      "DF blocks resident in HBM (%.1f GB per GPU, generated once before the timed region)" % (df_resident_bytes / 1e9),
                                      "; + ONE pass over the whole config after the timed steps (full_config)" if full is not None else
                                      ("; the timed step IS the whole config" if timed_is_full else "")),
                       "scaling_note": "strong: every timed step is the WHOLE config sharded over the ranks" if a.scaling == "strong"
                                       else "weak: --kl-per-gpu irreducible kL per GPU per step",
                       "steps_requested": a.steps_requested, "warmup_requested": a.warmup_requested, "steps_note": steps_note,
                       "input": input_note,
                       "parallelism": "kL-sharded x%d, k-sharded diag, all-reduce(ew) + all-reduce(rho_R); ERI: K-stacked contraction finished "
                                      "band by band, every finished band of rows reduced to its owner underneath the remaining GEMMs "
                                      "(row-sharded sum), all-reduce of the n x n J / K only" % world},
            "value_executed_mfma_tflops": round(exec_all / elapsed / 1e12, 3),
            "value_executed_frac_of_peak": round(exec_all / elapsed / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world), 4),
            "iteration_wall_s": round(elapsed / a.steps, 4),
            "stage_seconds_per_step": {k: round(v / a.steps, 5) for k, v in timers.items()},
            "eri_only_tflops": round(flops / a.steps / (timers.get("eri", 1e-9) / a.steps) / 1e12, 3),
            "roofline": roofline,
        }
        if full is not None:
            res["full_config"] = full
            res["full_config_iteration_wall_s"] = full["iteration_wall_s"]
            if full.get("parity_ok") is False:
                rc = 3
        elif timed_is_full:
            res["full_config_iteration_wall_s"] = round(elapsed / a.steps, 4)
        if shard is not None:
            res["shard_pass"] = shard
        if per_rank is not None:
            res["per_rank"] = per_rank
            if os.environ.get("DMK_BENCH_ONE_GPU", "0") == "1":
                res["one_gpu_note"] = ("ALL %d ranks on ONE GPU with host-staged (%s) exchanges: a correctness / partition run, "
                                       "NOT a scaling number" % (world, os.environ.get("DMK_BENCH_BACKEND", "nccl")))
        if parity is not None:
            res.update(parity)
            res["parity_ok"] = bool(parity["parity_maxabs"] <= PARITY_TOL and parity["parity_freivalds_ok"])
            if full is not None and full.get("parity_freivalds_ok") is False:
                rc = 3
            res["config"]["parity"] = ("max|device - oracle| = %.3e (max|ref| %.3e) on %d sampled entries of the timed ERI, tol %.0e; Freivalds "
                                       "residual of the contraction over all pair rows %.3e (|yref|max %.3e)"
                                       % (parity["parity_maxabs"], parity["parity_ref_maxabs"], parity["parity_entries"], PARITY_TOL,
                                          parity["parity_freivalds_maxabs"], parity["parity_freivalds_ref_maxabs"]))
            if not res["parity_ok"]:
                rc = 3
        if stage_parity is not None:
            res.update(stage_parity)
            if not stage_parity["parity_stages_ok"]:
                rc = 3
        # ERI x density inside the step: two J passes + one J(both directions) pass + two K passes over 8.66 GB blocks
        if "jk" in fam_out and sysm.spin == 2:
            gb = 5 * 8.0 * npair * npair / 1e9
            res["emb_ham"] = {"jk_ms_per_step": round(fam_out["jk"]["ms_total"] / a.steps, 3),
                              "jk_algorithmic_GB_per_step": round(gb, 2),
                              "jk_GBps": round(gb * a.steps / (fam_out["jk"]["ms_total"] * 1e-3), 1), "hbm_peak_GBps": HBM_PEAK_GBPS}
        if fit is not None:
            # vcor least-squares fit of the BASELINE target (config 5): measured once, outside the timed region, on ALL ranks
            # (the dV_dparam table is sharded row-wise over them); run to convergence (tolerances in the entry)
            fit.pop("vcor")
            res["vcor_fit"] = {k: (float("%.6e" % v) if isinstance(v, float) else v) for k, v in fit.items()}    # (err_end ~ 1e-10: no fixed decimals)
            res["iteration_plus_fit_wall_s"] = round(elapsed / a.steps + fit["seconds_total"], 4)
            if full is not None:
                res["full_config_iteration_plus_fit_wall_s"] = round(full["iteration_wall_s"] + fit["seconds_total"], 3)
        if not a.no_cpu_baseline:
            # flop of the WHOLE config (all kL), which is what the CPU time is extrapolated to
            fh_b = sysm.spin * (8.0 * sysm.naux * sysm.nao * sysm.nao * nemb + 8.0 * sysm.naux * sysm.nao * nemb * nemb)
            wts, by = ES.plan_records(sysm.mesh)
            fh_full = fh_b * sum(len(v) for v in by.values())
            fc_full = sum((3.0 if sysm.spin == 2 else 1.0) * (1 if wts[k] == 1 else 2) * 2.0 * sysm.naux * npair * npair for k in by)
            cb = cpu_baseline(sysm, nemb, fh_full, fc_full, a.cpu_seconds, host_threads(1))
            cb["gpu_over_cpu"] = round(res["value"] / max(cb["value"], 1e-12), 1)
            res["cpu_baseline"] = cb
            res["vs_cpu_baseline"] = cb["gpu_over_cpu"]
        emit(res)
    if distributed:
        import torch.distributed as td
        dist.barrier()           # rank 0 may still have been measuring the fit / CPU baseline: tear down together
        td.destroy_process_group()
    if rc:
        sys.exit(rc)


def model_line(a, ctx, sysm, out, fam, timers, elapsed, world, rank, distributed):
    """Model lattices (BASELINE configs 1-2: Hubbard): no DF tensor, the step is diag + occupations + density + fold + bath.
    HBM-bound stages: every family is reported as algorithmic bytes / HIP-event time against the 8 TB/s roof."""
    from libdmet_preview_amd.parallel import dist
    if rank == 0:
        n, nk, spin = sysm.nlo, sysm.nk, sysm.spin
        nemb = out["nemb"]
        alg = {"eigh": 2.0 * 16 * spin * nk * n * n + 8.0 * spin * nk * n,            # SURVEY.md section 8d, diag row
               "bath": 8.0 * spin * nk * n * (sysm.nval + nemb), "fold": 2.0 * 16 * spin * nk * n * n}
        stages = {}
        for k, by in alg.items():
            ms, cnt = fam.get(k, (0.0, 0))
            if cnt:
                ms_step = ms / a.steps
                stages[k] = {"ms_per_step": round(ms_step, 4), "launches_per_step": round(cnt / a.steps, 1),
                             "algorithmic_bytes_per_step": by, "GBps": round(by / (ms_step * 1e-3) / 1e9, 3),
                             "frac_of_hbm_peak": round(by / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)}
        dom = max(stages, key=lambda k: stages[k]["ms_per_step"]) if stages else "eigh"
        d = stages.get(dom, {"GBps": 0.0, "frac_of_hbm_peak": 0.0})
        stage_parity = None
        rc = 0
        if not a.no_parity:
            from oracle import stage_check as SC            # checker only
            got = {"ew": out["ew"].get().reshape(spin, nk, n), "occ": out["occ"].get().reshape(spin, nk, n), "mu": out["mu"],
                   "rho_R": out["rho_R"].get().reshape(spin, nk, n, n), "basis": out["basis"].get().reshape(spin, nk, n, nemb),
                   "sigma": out["sigma"], "C_ao_emb": None}
            ref = SC.reference_chain(sysm.mesh, sysm.Fock_R, sysm.vcor, sysm.filling, sysm.restricted, sysm.imp_idx, sysm.val_idx)
            stage_parity = {"parity_ew_maxabs": float(np.abs(got["ew"] - ref["ew"]).max()),
                            "parity_occ_equal": bool(np.array_equal(got["occ"], ref["occ"])),
                            "parity_rho_maxabs": float(np.abs(got["rho_R"] - ref["rho_R"]).max()),
                            "parity_nbath_equal": bool(got["basis"].shape == ref["basis"].shape)}
            if stage_parity["parity_nbath_equal"]:
                stage_parity["parity_bath_frob"] = max(SC.projector_distance(got["basis"][s], ref["basis"][s]) for s in range(spin))
            ok = stage_parity["parity_ew_maxabs"] <= 1e-10 and stage_parity["parity_rho_maxabs"] <= 1e-10 and \
                stage_parity["parity_nbath_equal"] and stage_parity.get("parity_bath_frob", 1.0) <= 1e-10
            stage_parity["parity_stages_ok"] = bool(ok)
            rc = 0 if ok else 3
        res = {"metric": "DMET embedding-construction iteration (diag+bath) wall-clock", "value": round(elapsed / a.steps, 6),
               "unit": "s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
               "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s: mesh %s nlo %d nemb %d spin %d (model lattice, no DF tensor)"
                                      % (a.workload, "x".join(map(str, sysm.mesh)), n, nemb, spin)},
               "stage_seconds_per_step": {k: round(v / a.steps, 6) for k, v in timers.items()},
               "roofline": {"bound": "hbm", "kernel": dom, "achieved": d["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": d["frac_of_hbm_peak"], "traffic": None, "stages": stages,
                            "note": "launch/latency bound at this size (SURVEY.md section 8d): %d matrices of %dx%d, bath %d x %d"
                                    % (spin * nk, n, n, len(sysm.env_idx), sysm.nval)}}
        if stage_parity is not None:
            res.update(stage_parity)
        if not a.no_cpu_baseline:
            # the model configs run IN FULL on the host -- the oracle's restatement of the same step
            # (R2k of the Fock operator, nk x eigh, occupations, rho_k, k->R fold, Schmidt bath; numpy / LAPACK) on this box's cores
            from oracle import stage_check as SC            # checker / baseline only
            threads = host_threads(1)
            try:
                from threadpoolctl import threadpool_limits
                limiter = threadpool_limits(limits=threads)
            except Exception:
                limiter = None
            chain = lambda: SC.reference_chain(sysm.mesh, sysm.Fock_R, sysm.vcor, sysm.filling, sysm.restricted, sysm.imp_idx, sysm.val_idx)
            chain()
            ts = []
            for _ in range(5):
                tc = time.perf_counter()
                chain()
                ts.append(time.perf_counter() - tc)
            if limiter is not None and hasattr(limiter, "restore_original_limits"):
                limiter.restore_original_limits()
            res["cpu_baseline"] = {"value": round(min(ts), 6), "unit": "s", "cores": threads, "kind": "port",
                                   "sample": "the whole step in full (oracle/restate.py: HF -> get_emb_basis, %s), best of 5 after one "
                                             "warm-up; includes the R2k of the real-space Fock operator, which the device step has resident"
                                             % blas_build()}
            res["vs_cpu_baseline"] = round(min(ts) / max(elapsed / a.steps, 1e-12), 3)
            res["vs_cpu_baseline_note"] = "CPU seconds / GPU seconds per step (> 1: the GPU step is faster)"
        emit(res)
    else:
        rc = 0
    if distributed:
        import torch.distributed as td
        dist.barrier()
        td.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
