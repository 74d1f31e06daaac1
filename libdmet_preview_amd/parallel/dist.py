"""
Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests).  Replaces the mpi4pyscf collectives of the reference
(basis_transform/eri_transform_mpi.py:203-210 `mpi.reduce_inplace(eri)`, routine/mfd_mpi.py:93-94).

Exchanges on the path:
  * the eigenvalue table (0.7 MB) and the partial R-space density (138 MB at C5): one all-reduce each;
  * the kL-sharded embedding ERI (26 GB at C5): the contraction of the resident plane stack is finished band by band of
    the pair index and every finished band of rows is REDUCED TO ITS OWNER RANK while the next bands are still being
    computed (`reduce_eri_bands`): (N-1)/N x 26 GB on the wire per step in total, hidden behind the contraction, and
    the summed ERI stays ROW-SHARDED -- its consumer (J / K of the embedding Hamiltonian) streams row blocks anyway and
    only its n x n results are summed.  `all_reduce_sum_dev` of the whole array remains for callers that ask for the
    full tensor on every rank (get_emb_eri(use_mpi=True), like the reference's reduce to rank 0).
"""
import numpy as np


def _td():
    import torch.distributed as td
    return td


def is_initialized():
    try:
        td = _td()
        return td.is_available() and td.is_initialized()
    except Exception:
        return False


def rank():
    return _td().get_rank() if is_initialized() else 0


def world_size():
    return _td().get_world_size() if is_initialized() else 1


def all_reduce_sum_numpy(x):
    """Sum a host array over ranks (gloo path / small host-side quantities)."""
    if not is_initialized():
        return x
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x))
    if _td().get_backend() == "nccl":
        t = t.cuda()
    _td().all_reduce(t)
    return t.cpu().numpy()


def all_reduce_sum_dev(dev_array):
    """In-place sum of a libdmetk device array over ranks.  With the RCCL backend the buffer is
    handed to torch zero-copy (a torch tensor aliasing the HBM pointer); with gloo it is staged
    through the host."""
    if not is_initialized():
        return dev_array
    import torch
    td = _td()
    if td.get_backend() == "nccl":
        order_after_library(dev_array.ctx)
        t = tensor_view(dev_array)
        # large buffers in <= 2 GiB slices keep RCCL's staging bounded
        flat = t.view(-1)
        step = 1 << 28
        for o in range(0, flat.numel(), step):
            td.all_reduce(flat[o:o + step])          # synchronous op: the current stream waits for it on return
        order_library_after_current(dev_array.ctx)
    else:
        host = dev_array.get()
        t = torch.from_numpy(host)
        td.all_reduce(t)
        dev_array.set(t.numpy())
    return dev_array


def broadcast_dev(dev_array, src=0):
    """Overwrite a libdmetk device array on every rank with rank `src`'s copy (RCCL: zero-copy view; gloo: through the host)."""
    if not is_initialized() or world_size() == 1:
        return dev_array
    import torch
    td = _td()
    if td.get_backend() == "nccl":
        order_after_library(dev_array.ctx)
        td.broadcast(tensor_view(dev_array).view(-1), src=src)
        order_library_after_current(dev_array.ctx)
    else:
        t = torch.from_numpy(dev_array.get())
        td.broadcast(t, src=src)
        dev_array.set(t.numpy())
    return dev_array


def broadcast_numpy(x, src=0):
    """Rank `src`'s host array on every rank (same shape / dtype everywhere)."""
    if not is_initialized() or world_size() == 1:
        return x
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x).copy())
    if _td().get_backend() == "nccl":
        t = t.cuda()
    _td().broadcast(t, src=src)
    return t.cpu().numpy()


class _CudaArrayInterface(object):
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def tensor_view(dev_array):
    """torch tensor aliasing a DevArray (no copy)."""
    import torch
    typestr = {"float64": "<f8", "complex128": "<c16", "int32": "<i4"}[dev_array.dtype.name]
    holder = _CudaArrayInterface(dev_array.address, dev_array.shape, typestr)
    t = torch.as_tensor(holder, device="cuda:%d" % dev_array.ctx.device)
    t._dmk_keep = dev_array
    return t


class _Pending(object):
    """Outstanding asynchronous reductions (keeps the aliased tensors alive until they are done)."""

    def __init__(self):
        self.items = []

    def add(self, work, keep):
        self.items.append((work, keep))

    def wait(self):
        for work, _ in self.items:
            if work is not None:
                work.wait()
        self.items = []


def library_stream(ctx):
    """The stream libdmetk launches on, as a torch stream: the device's legacy default stream, or the caller's stream
    (Context(stream=...) / Context.set_stream) wrapped without taking ownership."""
    import torch
    if ctx.default_stream:
        return torch.cuda.default_stream(ctx.device)
    return torch.cuda.ExternalStream(ctx.stream_ptr, device=ctx.device)


def order_after_library(ctx):
    """Make torch's CURRENT stream wait for everything queued so far on the library's stream, through an explicit event
    (no reliance on null-stream semantics, no host synchronisation).  A collective issued next is ordered after the kernels
    that produced its input: ProcessGroupNCCL's own stream waits for the current stream at the call."""
    import torch
    lib_s, cur = library_stream(ctx), torch.cuda.current_stream(ctx.device)
    if lib_s.cuda_stream == cur.cuda_stream:
        return                                          # same queue: already ordered
    ev = torch.cuda.Event()
    ev.record(lib_s)
    cur.wait_event(ev)


def order_library_after_current(ctx):
    """The reverse edge: the library's stream waits for what torch's current stream has queued (e.g. after Work.wait() has
    made the current stream wait for a collective), so library kernels launched next see the reduced data."""
    import torch
    lib_s, cur = library_stream(ctx), torch.cuda.current_stream(ctx.device)
    if lib_s.cuda_stream == cur.cuda_stream:
        return
    ev = torch.cuda.Event()
    ev.record(cur)
    lib_s.wait_event(ev)


def reduce_rows_to(dev_rows, owner, pending):
    """Sum the device block `dev_rows` over ranks INTO rank `owner` (other ranks keep their partial values).
    RCCL: asynchronous on the process group's own stream, ordered after the kernels already queued on the library's
    stream by an explicit event (`order_after_library`) -- whichever stream the library runs on -- while kernels launched
    afterwards overlap with it.  gloo: staged through the host, synchronous."""
    import torch
    td = _td()
    if td.get_backend() == "nccl":
        order_after_library(dev_rows.ctx)
        t = tensor_view(dev_rows)
        pending.add(td.reduce(t.view(-1), dst=owner, async_op=True), t)
    else:
        host = torch.from_numpy(dev_rows.get())
        td.reduce(host, dst=owner)
        if td.get_rank() == owner:
            dev_rows.set(host.numpy())


def reduce_eri_bands(eng, eri_dev, spin_pair, npair, bands_per_group=None):
    """Finish the contraction of `eng`'s plane stack band by band and reduce every finished group of bands (all spin
    blocks) to its owner while the next group is computed.  Returns the ownership table [(row_lo, row_hi, owner)] of
    the pair rows; rows a rank does not own hold its partial sums only."""
    import os
    world, me = world_size(), rank()
    nb, band_rows = eng.nbands()
    if bands_per_group is None:          # 8 bands = 1024 pair rows = 270 MB per reduction at C5
        bands_per_group = max(1, int(os.environ.get("DMK_ERI_BAND_GROUP", "8")))
    pending = _Pending()
    table = []
    for gi, b0 in enumerate(range(0, nb, bands_per_group)):
        b1 = min(nb, b0 + bands_per_group)
        eng.contract(b0, b1, done=(b1 == nb))
        owner = gi % world
        lo, hi = b0 * band_rows, min(npair, b1 * band_rows)
        table.append((lo, hi, owner))
        if is_initialized():           # also with ONE rank (DMK_FORCE_DIST=1): the same RCCL calls as on a multi-GPU node
            for blk in range(spin_pair):
                reduce_rows_to(eri_dev.offset((blk * npair + lo) * npair, (hi - lo, npair)), owner, pending)
    pending.wait()                       # torch's current stream now waits for every reduction
    if is_initialized() and _td().get_backend() == "nccl":
        order_library_after_current(eri_dev.ctx)       # ... and so does whatever the library launches next (J / K on the owned rows)
    return table


def owned_ranges(table, who=None):
    who = rank() if who is None else who
    return [(lo, hi) for (lo, hi, owner) in table if owner == who]


def gather_rows_numpy(eri_dev, spin_pair, npair, rows, table):
    """Host copy of the pair rows `rows` of every spin block of a row-sharded ERI, assembled over ranks:
    (spin_pair, len(rows), npair) on every rank."""
    me = rank()
    out = np.zeros((spin_pair, len(rows), npair))
    for i, r in enumerate(rows):
        owner = next(o for (lo, hi, o) in table if lo <= r < hi)
        if owner == me:
            for b in range(spin_pair):
                out[b, i] = eri_dev.offset((b * npair + int(r)) * npair, (npair,)).get()
    return all_reduce_sum_numpy(out) if world_size() > 1 else out


def barrier():
    if is_initialized():
        _td().barrier()
