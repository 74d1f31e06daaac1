"""
Multi-process twin of the DF ERI transform (reference: libdmet/basis_transform/eri_transform_mpi.py), one process per
GPU over torch.distributed instead of mpi4pyscf.

  _task_location, assign_workload   (eri_transform_mpi.py:27-55)  static partition of the irreducible kL over ranks:
                                     weight-1 kL round-robin first, then contiguous runs of weight-2 kL so that rank r
                                     ends up with the segment length _task_location gives it
  get_emb_eri_fast_gdf              (eri_transform_mpi.py:57-223)  identical body per rank + ONE sum of the partial ERI
                                     (the reference reduces to the root; here every rank receives the sum)
The same partition is computed inside libdmetk from the integer mesh (dmk_assign_workload, golden G1); this module is
the reference-signature view of it and is checked against it in tests/test_host_abi.py.
"""
import numpy as np

from libdmet_preview_amd.parallel import dist
from libdmet_preview_amd.routine.mfd_mpi import _task_location        # same rule (mfd_mpi.py:25-31)
from libdmet_preview_amd.basis_transform import eri_transform as _et


def assign_workload(weights, n):
    """Per-rank lists of irreducible kL indices for time-reversal weights {1, 2, 0} and n ranks."""
    weights = np.asarray(weights)
    w1, w2 = np.where(weights == 1)[0], np.where(weights == 2)[0]
    sizes = [b - a for a, b in (_task_location(len(w1) + len(w2), r, n) for r in range(n))]
    kids = [[] for _ in range(n)]
    for c, k in enumerate(w1):
        kids[c % n].append(k)
    at = 0
    for r in range(n):
        take = sizes[r] - len(kids[r])
        kids[r].extend(w2[at:at + take])
        at += take
    return kids


def get_emb_eri_fast_gdf(cell, cderi, kpts=None, C_ao_lo=None, basis=None, feri=None, kscaled_center=None, symmetry=4,
                         max_memory=None, kconserv_tol=1e-6, unit_eri=False, swap_idx=None, t_reversal_symm=True,
                         incore=True, fout="H2.h5", C_ao_eo=None, **kwargs):
    """kL-sharded get_emb_eri_fast_gdf with the reference's signature (eri_transform_mpi.py:57-62): `cderi` is the DF
    CONTAINER -- what `mydf._cderi` holds: the path of the HDF5 file or an open mapping with its layout -- and `kpts` the
    absolute k-points; every rank rebuilds its reader from the two (the reference's `mydf = df.GDF(cell, kpts);
    mydf._cderi = cderi`, :80-81), transforms its kL shard and takes part in ONE sum of the partial ERIs (every rank
    receives it; the reference reduces to the root, :203-210).  Also accepted, as before: a DF object / block provider in
    place of `cderi` with `kpts` left out.  Needs an initialised torch.distributed process group (one rank per GPU)."""
    if not dist.is_initialized():
        raise RuntimeError("eri_transform_mpi.get_emb_eri_fast_gdf needs torch.distributed to be initialised "
                           "(use eri_transform.get_emb_eri_fast_gdf on one GPU)")
    if kpts is None:
        if not hasattr(cderi, "kpts"):
            raise TypeError("get_emb_eri_fast_gdf(cell, cderi, kpts, ...): `kpts` is required with a bare cderi container")
        mydf = cderi
    else:
        mydf = _et.resolve_df(cell, cderi, kpts=np.asarray(kpts))
    try:
        return _et.get_emb_eri_fast_gdf(cell, mydf, C_ao_lo=C_ao_lo, basis=basis, feri=feri, kscaled_center=kscaled_center,
                                        symmetry=symmetry, max_memory=max_memory, kconserv_tol=kconserv_tol,
                                        unit_eri=unit_eri, swap_idx=swap_idx, t_reversal_symm=t_reversal_symm, incore=incore,
                                        fout=fout, C_ao_eo=C_ao_eo, use_mpi=True)
    finally:
        _et._release_df(mydf, cderi)
