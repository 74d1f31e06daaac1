"""
The hot path's exit type: the embedding Hamiltonian handed to an impurity solver.

Interface contract (attribute names and accepted shapes) = libdmet/system/integral.py:60-128 (`Integral`) and
:883-927 (`get_eri_format`); the solvers read `norb, restricted, bogoliubov, H0, H1["cd"], H2["ccdd"], ovlp`.
File formats (FCIDUMP, HDF5 save / load) are outside the path (out of scope).

Own organisation: the storage classes of an ERI are one table (`_ERI_LAYOUTS`: ndim with / without a spin axis and
the element count as a function of nao) that both the constructor's shape check and `get_eri_format` read.
"""
import numpy as np

from libdmet_preview_amd.utils import logger as log


def _npair(n):
    return n * (n + 1) // 2


# symmetry label -> (elements per spin block, ndim without spin axis, ndim with spin axis)
_ERI_LAYOUTS = (
    ("s1", lambda n: n ** 4, 4, 5),
    ("s4", lambda n: _npair(n) ** 2, 2, 3),
    ("s8", lambda n: _npair(_npair(n)), 1, 2),
)


def _as_table(x, key):
    """A bare array is shorthand for the one-entry table the solvers index."""
    return {key: x} if isinstance(x, np.ndarray) else x


class Integral(object):
    def __init__(self, norb, restricted, bogoliubov, H0, H1, H2, ovlp=None):
        """H1: (spin, norb, norb) arrays keyed "cd"; H2: arrays (or open dataset handles) keyed "ccdd" that carry a
        spin axis in front of a 1-fold (5-d), 4-fold (3-d) or 8-fold (2-d) ERI."""
        self.norb, self.restricted, self.bogoliubov, self.H0 = norb, restricted, bogoliubov, H0
        self.H1 = _as_table(H1, "cd")
        self.H2 = _as_table(H2, "ccdd")
        want = "(spin, %d, %d)" % (norb, norb)
        for name, h in self.H1.items():
            ok = h is None or (np.ndim(h) == 3 and np.shape(h)[-1] == norb)
            log.eassert(ok, "Integral: one-body term %r has shape %s, expected %s", name, np.shape(h), want)
        spin_ndims = tuple(l[3] for l in _ERI_LAYOUTS)
        for name, g in self.H2.items():
            log.eassert(g is None or g.ndim in spin_ndims,
                        "Integral: two-body term %r has shape %s; a spin axis plus s1 / s4 / s8 storage is required",
                        name, None if g is None else g.shape)
        self.ovlp = ovlp if ovlp is not None else np.eye(norb)

    # orbital-pair enumerations in the order the reference's FCIDUMP writers walk them (integral.py:119-126)
    def pairNoSymm(self):
        return [(p, q) for p in range(self.norb) for q in range(self.norb)]

    def pairSymm(self):
        return [(p, q) for p in range(self.norb) for q in range(p + 1)]

    def pairAntiSymm(self):
        return [(p, q) for p in range(self.norb) for q in range(p)]


def get_eri_format(eri, nao):
    """-> (storage symmetry in {'s1', 's4', 's8'}, spin_dim): spin_dim is 0 for an array without a spin axis, else the
    length of that axis."""
    eri = np.asarray(eri)
    for label, count, bare_ndim, spin_ndim in _ERI_LAYOUTS:
        per_block = count(nao)
        if eri.ndim == bare_ndim and eri.size == per_block:
            return label, 0
    for label, count, bare_ndim, spin_ndim in _ERI_LAYOUTS:
        per_block = count(nao)
        if eri.ndim == spin_ndim and (label != "s8" or eri.size == per_block):
            nspin, rest = divmod(eri.size, per_block)
            log.eassert(rest == 0, "get_eri_format: %s storage of nao = %d holds %d elements per spin block, array has shape %s",
                        label, nao, per_block, eri.shape)
            return label, nspin
    raise ValueError("get_eri_format: shape %s is no s1 / s4 / s8 ERI of nao = %d" % (eri.shape, nao))
