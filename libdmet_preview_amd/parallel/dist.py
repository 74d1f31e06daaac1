"""
Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests).  Replaces the mpi4pyscf collectives of the reference
(basis_transform/eri_transform_mpi.py:203-210 `mpi.reduce_inplace(eri)`, routine/mfd_mpi.py:93-94).

Only two exchanges exist on the path: the sum of the partial R-space density
(138 MB at C5) and the sum of the kL-sharded embedding ERI (26 GB at C5), each ONE all-reduce after
the local work -- never per kL, so the per-link-bound ring cost is paid once.
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
        dev_array.ctx.sync()
        t = tensor_view(dev_array)
        # large buffers in <= 2 GiB slices keep RCCL's staging bounded
        flat = t.view(-1)
        step = 1 << 28
        for o in range(0, flat.numel(), step):
            td.all_reduce(flat[o:o + step])
        torch.cuda.synchronize()
    else:
        host = dev_array.get()
        t = torch.from_numpy(host)
        td.all_reduce(t)
        dev_array.set(t.numpy())
    return dev_array


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


def barrier():
    if is_initialized():
        _td().barrier()
