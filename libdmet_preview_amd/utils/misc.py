"""numpy helper semantics of the reference's utils/misc.py:34-86 (shape plumbing only)."""
from functools import reduce
import numpy as np


def max_abs(x):
    x = np.asarray(x)
    if x.size == 0:
        return 0.0
    if np.iscomplexobj(x):
        return np.abs(x).max()
    return max(np.max(x), abs(np.min(x)))


def mdot(*args):
    """Small host-side chain product (bookkeeping sizes only; the hot products run on the GPU)."""
    return reduce(np.dot, args)


def kdot(a, b):
    """utils/misc.py:49-59 -- routed through the batched HIP zgemm."""
    from libdmet_preview_amd.basis_transform.make_basis import _bgemm
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape[0] == b.shape[0]
    return _bgemm("N", "N", a, b)


def get_spin_dim(arrays, non_spin_dim=3):
    spin = 1
    for a in arrays:
        a = np.asarray(a)
        if a.ndim == non_spin_dim:
            continue
        elif a.ndim == non_spin_dim + 1:
            spin = max(spin, a.shape[0])
        else:
            raise ValueError
    return spin


def add_spin_dim(H, spin, non_spin_dim=3):
    H = np.asarray(H)
    if H.ndim == non_spin_dim:
        H = H[None]
    assert H.ndim == (non_spin_dim + 1)
    if H.shape[0] < spin:
        H = np.asarray((H[0],) * spin)
    return H
