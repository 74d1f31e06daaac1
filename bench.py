#!/usr/bin/env python
"""
bench.py -- one embedding-construction iteration per step on synthetic k-sampled tensors.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json config 5, "C5"): cuprate-like cell, 6x6x6 k-mesh, nao = nlo = 200, naux = 800,
UHF, 56 valence orbitals -> nemb = 256.  A step = diag (432 Hermitian 200x200) + occupations + rho_k +
k->R fold + Schmidt bath (43144 x 56 SVD per spin) + C_ao_emb + density-fitted ERI transform of this
rank's share of the irreducible momentum transfers kL.  WEAK scaling: every GPU transforms the same
number of kL (default 14 = 112 irreducible kL / 8, so N = 8 is exactly the full C5 iteration); the DF
blocks are regenerated on the device (Philox) inside the timed region, standing in for the reference's
HDF5 reads.  All other inputs are resident in HBM before the clock starts.

Prints ONE JSON line (rank 0): value = algorithmic FP64 flop of the ERI transform performed by all ranks
divided by the wall-clock of the whole step (max over ranks), in TFLOP/s.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6     # AMD MI355X FP64 matrix spec (= 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz);
                                 # the on-image microarch guide lists no f64 MFMA row -- tools/mfma_f64_probe.hip
                                 # measures the sustained ceiling on the box


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default="C5")
    p.add_argument("--kl-per-gpu", type=int, default=14, help="irreducible kL transformed per GPU per step")
    p.add_argument("--max-blocks-per-kl", type=int, default=0, help="debug: truncate the i-loop (0 = all)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0)
    p.add_argument("--fit-iters", type=int, default=3, help="CG iterations of the vcor fit measured after the timed steps "
                                                            "(0 = skip); reported under \"vcor_fit\", never part of `value`")
    return p.parse_args()


def cpu_baseline(sysm, nemb, budget_s):
    """Oracle (numpy restatement of the reference, same BLAS entry points) on a bounded sample of the same
    workload, on this host's cores.  Returns (TFLOP/s, description, threads)."""
    from oracle import restate as R
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    nao, naux, spin = sysm.nao, sysm.naux, sysm.spin
    rng = np.random.default_rng(0)
    Cemb = (rng.standard_normal((spin, 2, nao, nemb)) + 1j * rng.standard_normal((spin, 2, nao, nemb))) / nao
    # half transform on an L-slice of one AO block (eri_transform.py:403-434)
    lsl = max(8, min(naux, 64))
    blk = R.df_block_philox(1, 0, 1, lsl, nao).reshape(lsl, -1)
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
    return f_half / t_half / 1e12, f_con / t_con / 1e12, \
        "oracle: half transform of a %d-row slice of one C5 AO block (both spins) + dgemm %dx%dx%d, numpy/OpenBLAS" \
        % (lsl, ncs, 2 * naux, ncs), threads


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("DMK_DEVICE", str(local))
    # DMK_FORCE_DIST=1 initialises the process group even for one rank (exercises the RCCL plumbing on a 1-GPU box)
    distributed = world > 1 or os.environ.get("DMK_FORCE_DIST", "0") == "1"
    if distributed and "MASTER_ADDR" not in os.environ:
        os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517", "RANK": "0", "WORLD_SIZE": "1"})
    if distributed:
        import torch
        import torch.distributed as td
        torch.cuda.set_device(local)
        td.init_process_group("nccl", device_id=torch.device("cuda", local))
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
    nshards = max(1, (len(irr1) + len(irr2) + a.kl_per_gpu - 1) // a.kl_per_gpu)
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
    if distributed:
        agg = dist.all_reduce_sum_numpy(np.array([flops, 0.0]))
        flops_all = float(agg[0])
        import torch
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        import torch.distributed as td
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax.item())
    else:
        flops_all = flops

    if rank == 0:
        nemb = out["nemb"]
        npair = nemb * (nemb + 1) // 2
        nblk = out["nblocks"]
        # algorithmic flop of each ERI kernel family over the timed region (DESIGN.md
        # section 5); a step-2 launch covers up to DMK_ERI_GROUP queued AO blocks, so rates are totals / totals
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
                fam_out[k]["tflops"] = round(fam_flops[k] / (fam_out[k]["ms_total"] * 1e-3) / 1e12, 2)
                fam_out[k]["gflop_per_launch"] = round(fam_flops[k] / fam_out[k]["launches"] / 1e9, 2)
        # executed MFMA flop per algorithmic flop: the hot half-transform kernels use the 3M complex product
        # (0.75x) and pad to 16-row blocks (step 1: ceil(nao/16)*16/nao; step 2: 136 of 128.5 blocks); DESIGN.md section 4
        hot = (nemb == 256 and sysm.nao % 8 == 0)
        pad1 = (-(-sysm.nao // 16) * 16) / float(sysm.nao)
        tl = -(-npair // 128)                                   # contraction tiles per side
        symm = (tl * (tl + 1) / 2.0) / float(tl * tl)           # aa / bb launches compute the lower tile triangle only
        dg = (symm + 1.0 + symm) / 3.0 if sysm.spin == 2 else symm
        exec_ratio = {"zgemm_half1": 0.75 * pad1 if hot else 1.0, "zgemm_half2": 0.75 * (136 * 256.0 / npair) if hot else 1.25,
                      "dgemm": dg if (sysm.naux % 16 == 0 and npair % 2 == 0) else 1.0}
        for k, rr in exec_ratio.items():
            if k in fam_out:
                fam_out[k]["executed_mfma_tflops"] = round(fam_out[k]["tflops"] * rr, 2)
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
        tinfo = {}
        if os.path.exists(tj):
            try:
                tinfo = json.load(open(tj))
            except Exception:
                tinfo = {}
        dom = max([k for k in ("zgemm_half1", "zgemm_half2", "dgemm") if k in fam_out],
                  key=lambda k: fam_out[k]["ms_total"])
        achieved = fam_out[dom]["tflops"]
        if dom in tinfo and "hbm_bytes_per_launch" in tinfo[dom]:
            traffic = tinfo[dom]["hbm_bytes_per_launch"]
        roofline = {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_note": "HBM bytes per launch from rocprofv3 PMC (profiles/traffic_latest.json), collected offline",
                    "avg_launch_ms": fam_out[dom]["ms_avg"],
                    "note": "achieved = ALGORITHMIC flop (8 per complex multiply-add) / HIP-event time; the kernel executes "
                            "executed_mfma_tflops on the matrix pipe (3M complex product), ceiling measured 78.1 TFLOP/s (tools/mfma_acc_probe.hip)",
                    "mfma_ceiling_measured": 78.1,
                    "families": fam_out}
        res = {
            "metric": "DMET embedding-construction iteration (diag+bath+ERI-transform): ERI-transform TFLOP/s over the whole step",
            "value": round(flops_all / elapsed / 1e12, 3),
            "unit": "TFLOP/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: mesh %s nao %d naux %d nemb %d spin %d; %d irreducible kL per GPU "
                                   "(%d of 112 in total; N=8 x 14 = full C5), %d DF blocks per GPU per step, Philox DF blocks "
                                   "regenerated on device inside the timed region"
                                   % (a.workload, "x".join(map(str, sysm.mesh)), sysm.nao, sysm.naux, nemb, sysm.spin,
                                      len(kl_mine), len(kl_mine) * world, nblk),
                       "parallelism": "kL-sharded x%d, k-sharded diag, 1 all-reduce(rho_R) + 1 all-reduce(ERI)" % world},
            "iteration_wall_s": round(elapsed / a.steps, 4),
            "stage_seconds_per_step": {k: round(v / a.steps, 5) for k, v in timers.items()},
            "eri_only_tflops": round(flops / a.steps / (timers.get("eri", 1e-9) / a.steps) / 1e12, 3),
            "roofline": roofline,
        }
        # ERI x density inside the step: two J passes + one J(both directions) pass + two K passes over 8.66 GB blocks
        if "jk" in fam_out and sysm.spin == 2:
            gb = 5 * 8.0 * npair * npair / 1e9
            res["emb_ham"] = {"jk_ms_per_step": round(fam_out["jk"]["ms_total"] / a.steps, 3),
                              "jk_algorithmic_GB_per_step": round(gb, 2),
                              "jk_GBps": round(gb * a.steps / (fam_out["jk"]["ms_total"] * 1e-3), 1), "hbm_peak_GBps": 8000.0}
        if a.fit_iters > 0:
            # vcor least-squares fit of the BASELINE target (config 5): measured once, outside the timed region
            fit = pipeline.vcor_fit_stage(ctx, sysm, out["basis"], nemb, out["emb_ham"]["rdm1_emb"], MaxIter=a.fit_iters)
            fit.pop("vcor")
            passes_bytes = 2.0 * fit["dV_dparam_bytes"]
            fit["note"] = ("FitVcorEmb, VcorLocal on the valence orbitals, CG with analytic gradient; one objective = one pass over "
                           "dV_dparam + one eigh(nemb) per spin (multi-CU block Jacobi, warm started from the previous evaluation) "
                           "+ nemb^3 algebra; objective+gradient = two passes; the eigensolve of two %dx%d matrices "
                           "is still most of it" % (nemb, nemb))
            fit["dV_stream_GBps_if_only_cost"] = round(passes_bytes / (fit["ms_per_objective_plus_gradient"] * 1e-3) / 1e9, 1)
            res["vcor_fit"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in fit.items()}
            res["iteration_plus_fit_wall_s"] = round(elapsed / a.steps + fit["seconds_total"], 4)
        if not a.no_cpu_baseline:
            th, tc, desc, threads = cpu_baseline(sysm, nemb, a.cpu_seconds)
            fh, fc = out["flops_half"], out["flops_contract"]
            v = (fh + fc) / (fh / th + fc / tc)
            res["cpu_baseline"] = {"value": round(v, 4), "unit": "TFLOP/s", "cores": threads, "kind": "port",
                                   "sample": desc, "half_transform_tflops": round(th, 4),
                                   "contraction_tflops": round(tc, 4)}
        print(json.dumps(res), flush=True)
    if distributed:
        import torch.distributed as td
        dist.barrier()           # rank 0 may still have been measuring the fit / CPU baseline: tear down together
        td.destroy_process_group()


if __name__ == "__main__":
    main()
