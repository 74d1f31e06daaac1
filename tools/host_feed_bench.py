"""
PCIe-inclusive rate of the ERI transform: C5-shaped AO blocks (800 x 200 x 200 c128 = 512 MB) fed
 (a) from HBM (Philox regenerated on the device: the bench default),
 (b) from two pinned host buffers through dmk_eri_push_block_host (copy stream overlapped with the transform; the
     buffers are pre-filled, i.e. the provider's own read cost is excluded),
 (c) like (b) but every block is first copied by the CPU from pageable memory into the pinned buffer (what an HDF5
     reader that cannot read into pinned memory pays).
 (d) through CderiProvider.load_block_host on a cderi-layout mapping ("j3c-kptij", "j3c/<pair>/0": only pairs i >= j stored, every
     dataset served from one shared 512 MB array so that the box does not need 55 GB of host memory): the reader's single pass
     into the pinned buffer + the device-side conjugate-transpose of blocks stored for the swapped pair.
 (e) the same blocks made RESIDENT once (et.GDFResident over the provider of (d): loaded through the same reader, outside the timed
     pass) and transformed in place (dmk_eri_push_resident): what every DMET iteration after the first pays for a tensor that fits.
One irreducible kL with time-reversal weight 2 (all its blocks + the contraction).
usage: python tools/host_feed_bench.py [C5 | C4]      (C4: 416 x 104 x 104 blocks of 72 MB, nemb 136, one spin)
"""
import json
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    from libdmet_preview_amd._lib import get_ctx
    from libdmet_preview_amd.basis_transform import eri_transform as et
    ctx = get_ctx()
    shape = sys.argv[1] if len(sys.argv) > 1 else "C5"
    mesh, nao, naux, nemb, spin = ((6, 6, 6), 200, 800, 256, 2) if shape == "C5" else ((4, 4, 4), 104, 416, 136, 1)
    nk = int(np.prod(mesh))
    npair = nemb * (nemb + 1) // 2
    rng = np.random.default_rng(1)
    C_dev = ctx.to_device((rng.standard_normal((spin, nk, nao, nemb)) + 1j * rng.standard_normal((spin, nk, nao, nemb)))
                          * nk ** -0.75 / np.sqrt(nao))
    eri_dev = ctx.zeros((spin * (spin + 1) // 2, npair, npair), np.float64)
    eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
    kL = [k for k in eng.irreducible_kL() if eng.weights[k] == 2][0]
    nblk = len(eng.by_kL[kL])
    block_bytes = naux * nao * nao * 16
    pageable = (rng.standard_normal((naux, nao, nao)) + 1j * rng.standard_normal((naux, nao, nao))) / np.sqrt(nao)

    class Dev(object):
        def __init__(self):
            self.p = et.GDFPhilox(np.zeros((nk, 3)), naux, nao, seed=3)

        def load_block(self, c, i, j, o):
            self.p.load_block(c, i, j, o)

    class HostPrefilled(object):
        def load_block_host(self, i, j, out):
            pass

    class HostMemcpy(object):
        def load_block_host(self, i, j, out):
            np.copyto(out, pageable)

    class SharedCderi(object):
        """cderi-layout mapping whose every pair dataset is the same array (the layout and the read path are what is measured)."""
        def __init__(self, kpts):
            nkp = len(kpts)
            self.kptij = np.asarray([(kpts[i], kpts[j]) for i in range(nkp) for j in range(i + 1)])
            self.block = pageable.reshape(naux, nao * nao)

        def __getitem__(self, key):
            if key == "j3c-kptij":
                return self.kptij
            parts = key.split("/")
            if len(parts) == 3 and parts[0] == "j3c" and parts[2] == "0" and int(parts[1]) < len(self.kptij):
                return self.block
            raise KeyError(key)

    from libdmet_preview_amd.system import fourier
    from libdmet_preview_amd.system.lattice import _UnitCell
    kabs = _UnitCell(nao).get_abs_kpts(fourier.make_kpts_scaled(mesh))
    cderi = et.CderiProvider(SharedCderi(kabs), kabs, nao)
    swaps = sum(1 for r in eng.by_kL[kL] if (int(r[1]), int(r[2])) not in cderi.pair_of)

    t_load = time.perf_counter()
    resident = et.GDFResident(ctx, cderi, mesh, nao, naux, kL_list=[kL])
    t_load = time.perf_counter() - t_load
    res = {}
    for name, prov in (("device_philox", Dev()), ("host_pinned_prefilled", HostPrefilled()), ("host_cpu_fill", HostMemcpy()),
                       ("host_cderi_provider", cderi), ("resident_in_hbm", resident)):
        if name not in ("device_philox", "resident_in_hbm") and eng.host_buf is not None:
            for b in eng.host_buf:
                b.a[...] = pageable
        for rep in range(2):                       # first pass warms up (allocations, pinned buffers)
            if name not in ("device_philox", "resident_in_hbm") and eng.host_buf is None:
                eng.run_kL(kL, prov, max_blocks=2)
                for b in eng.host_buf:
                    b.a[...] = pageable
            f0 = eng.flops()
            ctx.sync()
            t0 = time.perf_counter()
            n = eng.run_kL(kL, prov)
            ctx.sync()
            dt = time.perf_counter() - t0
            f1 = eng.flops()
        fl = (f1[0] - f0[0]) + (f1[1] - f0[1])
        res[name] = {"blocks": n, "seconds": round(dt, 4), "ms_per_block": round(1e3 * dt / n, 3),
                     "tflops": round(fl / dt / 1e12, 2),
                     "host_GBps": None if name in ("device_philox", "resident_in_hbm") else round(n * block_bytes / dt / 1e9, 1)}
    eng.close()
    res["host_cderi_provider"]["blocks_stored_for_the_swapped_pair"] = swaps
    res["resident_in_hbm"]["one_off_load_seconds"] = round(t_load, 3)
    res["resident_in_hbm"]["resident_GB"] = round(resident.nblocks * block_bytes / 1e9, 2)
    res["resident_in_hbm"]["speedup_over_host_cderi_provider"] = round(res["host_cderi_provider"]["seconds"] / res["resident_in_hbm"]["seconds"], 2)
    print(json.dumps({"workload": "%s blocks, one kL (w=2), %d blocks of %.0f MB" % (shape, nblk, block_bytes / 1e6), "results": res}))


if __name__ == "__main__":
    main()
