#!/usr/bin/env python
"""Turn rocprofv3 (rocpd sqlite) outputs into the text summaries committed under profiles/.

usage: python tools/rocprof_summary.py <trace.db> [<pmc.db> ...] > profiles/rNN_summary.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                       "order by total_duration desc").fetchall()
    print("== kernel stats (rocprofv3 --kernel-trace --stats): %s" % db)
    print("%-92s %8s %14s %14s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "%"))
    for n, c, t, a, p in rows:
        print("%-92s %8d %14.3f %14.4f %7.2f" % (short(n), c, t / 1e3, a / 1e3, p))   # view is in microseconds
    print()


def pmc(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, counter_name, count(*), avg(counter_value), sum(counter_value) from pmc_events "
                       "group by name, counter_name order by sum(counter_value) desc").fetchall()
    print("== PMC (rocprofv3 --kernel-trace --pmc): %s" % db)
    print("(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 1/2 of a wide coalesced 16 B/lane stream --")
    print(" MI355X_MICROARCH.md section HBM -- so the MB column doubles it; WRITE_SIZE matched the known store bytes 1:1)")
    print("%-92s %-12s %8s %16s %14s" % ("kernel", "counter", "calls", "avg KiB/dispatch", "avg MB (corr.)"))
    for n, cn, c, a, s in rows:
        corr = 2.0 if cn == "FETCH_SIZE" else 1.0
        print("%-92s %-12s %8d %16.1f %14.1f" % (short(n), cn, c, a, a * 1024 * corr / 1e6))
    print()


# first match wins: the multi-CU Jacobi solver must not be counted as the batched QL eigensolver, and only the LDS-DMA
# contraction kernel is the "dgemm" of the ERI transform (dgemm_tn_acc_kernel also serves small Gram matrices)
FAMILY = [("dgemm_tn_acc_dma_kernel", "dgemm"), ("half1_kernel", "zgemm_half1"), ("half2_kernel", "zgemm_half2"), ("half2_tab_kernel", "zgemm_half2"),
          ("philox_block", "philox"), ("jacobi_eigh_kernel", "jacobi_eigh"), ("tridiag_resident_kernel", "eigh_tridiag"),
          ("tridiag_tiles_kernel", "eigh_tridiag"), ("backtransform_kernel", "eigh_backtransform"),
          ("backtransform_wy_kernel", "eigh_backtransform"), ("wy_tfactor_kernel", "eigh_tfactor"), ("tri_eigpairs_kernel", "eigh_tripairs"), ("eigh_kernel", "eigh"),
          ("fold_fft_kernel", None), ("jk_j_kernel", "jk_j"),
          ("jk_k_kernel", "jk_k"), ("gemv2_kernel", "gemv2")]


def family_of(name):
    """Kernel family of a demangled kernel name (first FAMILY key it contains).  fold_fft_kernel<IN_REAL, OUT_REAL, INVERSE>: the
    direction is the THIRD template argument -- INVERSE = true is k -> R whatever the output type (the complex-output k -> R fold
    of dmk_fold_k2R_complex is <false, false, true>), false is R -> k."""
    for key, fam in FAMILY:
        if key in name:
            if fam is not None:
                return fam
            import re
            m = re.search(r"fold_fft_kernel<\s*(\w+)\s*,\s*(\w+)\s*,\s*(\w+)\s*>", name)
            inverse = m is not None and m.group(3) in ("true", "1")
            return "fold_k2R" if inverse else "fold_R2k"
    return None


def traffic_json(dbs, out, workload="C5", how=None, steps_profiled=1):
    """HBM bytes per launch per kernel family: FETCH_SIZE (x2 gfx950 correction for wide coalesced streams) + WRITE_SIZE.
    The file holds one section per workload ({"workloads": {"C5": {...}, "C4": {...}}}); a run replaces only its own."""
    import json
    acc = {}
    for db in dbs:
        con = sqlite3.connect(db)
        per = {}
        for n, cn, c, tot in con.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                                         "group by name, counter_name"):
            if cn not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            fam = family_of(n)
            if fam is not None:
                e = per.setdefault((fam, cn), [0, 0.0])              # template variants of one family are pooled
                e[0] += c
                e[1] += tot
        for (fam, cn), (c, tot) in per.items():
            e = acc.setdefault(fam, {"launches_sampled": 0})
            e[cn + "_KiB_per_launch"] = tot / c
            e["launches_sampled"] = max(e["launches_sampled"], c)
    for fam, e in acc.items():
        f = e.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024 * 2.0
        w = e.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024
        e["hbm_bytes_per_launch"] = f + w
        # a family may launch more than once per step: bytes per STEP = bytes per launch x launches in the profiled run / its steps
        e["steps_profiled"] = int(steps_profiled)
        e["hbm_bytes_per_step"] = (f + w) * e["launches_sampled"] / max(1, int(steps_profiled))
        e["note"] = "FETCH_SIZE x2 (gfx950 counts 1/2 of a wide coalesced stream) + WRITE_SIZE, separate --pmc passes"
    # provenance: bench.py reports `roofline.traffic` only while the kernel sources still are the ones these passes ran on
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from bench import kernel_source_sha
    try:
        commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        commit = None
    acc["_collected"] = {"kernel_source_sha": kernel_source_sha(), "commit": commit, "workload": workload,
                         "how": how or ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python3 bench.py "
                                        "--workload %s ... --steps 1 --warmup 0 --no-parity; tools/rocprof_summary.py --traffic-json"
                                        % workload)}
    doc = {"workloads": {}}
    if os.path.exists(out):
        try:
            old = json.load(open(out))
            if "workloads" in old:
                doc = old
            elif "_collected" in old:                       # round-3 layout: one flat section
                doc["workloads"][old["_collected"].get("workload", "C5")] = old
        except Exception:
            pass
    doc["workloads"][workload] = acc
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--traffic-json":             # --traffic-json OUT [--workload NAME] db ...
        rest = args[2:]
        wl = "C5"
        if rest and rest[0] == "--workload":
            wl, rest = rest[1], rest[2:]
        traffic_json(rest, args[1], wl)
        sys.exit(0)
    stats(args[0])
    for d in args[1:]:
        pmc(d)
