"""
Shape plumbing shared by the host mirrors.  Names and meanings follow libdmet/utils/misc.py:34-86 (`max_abs`, `mdot`,
`kdot`, `get_spin_dim`, `add_spin_dim`) because callers of the reference import them from here; the bodies are this
package's own (kdot runs on the device).
"""
import numpy as np


def max_abs(x):
    """Largest magnitude of any entry; 0.0 for an empty array."""
    x = np.asarray(x)
    return float(np.abs(x).max()) if x.size else 0.0


def mdot(*factors):
    """Left-to-right matrix chain product of small host matrices (hot products live on the GPU)."""
    if not factors:
        raise TypeError("mdot needs at least one matrix")
    out = factors[0]
    for f in factors[1:]:
        out = np.dot(out, f)
    return out


def kdot(a, b):
    """Per-k matrix product of two (nk, ., .) stacks (utils/misc.py:49-59) through the batched HIP zgemm."""
    from libdmet_preview_amd.basis_transform.make_basis import _bgemm
    a, b = np.asarray(a), np.asarray(b)
    if a.shape[0] != b.shape[0]:
        raise ValueError("kdot: %d k-points on the left, %d on the right" % (a.shape[0], b.shape[0]))
    return _bgemm("N", "N", a, b)


def _spin_len(a, non_spin_dim):
    """Length of the leading spin axis of `a`, or None when `a` has no spin axis."""
    extra = np.ndim(a) - non_spin_dim
    if extra not in (0, 1):
        raise ValueError("expected %d or %d dimensions, got shape %s" % (non_spin_dim, non_spin_dim + 1, np.shape(a)))
    return np.shape(a)[0] if extra else None


def get_spin_dim(arrays, non_spin_dim=3):
    """Largest spin-axis length among `arrays` (1 when none of them has a spin axis)."""
    lens = [_spin_len(a, non_spin_dim) for a in arrays]
    return max([1] + [n for n in lens if n is not None])


def add_spin_dim(H, spin, non_spin_dim=3):
    """`H` with a leading spin axis of at least `spin` entries: a missing axis is added, a shorter one is filled by
    repeating block 0 (the restricted -> unrestricted promotion)."""
    H = np.asarray(H)
    if _spin_len(H, non_spin_dim) is None:
        H = H[None]
    if H.shape[0] < spin:
        H = np.broadcast_to(H[0], (spin,) + H.shape[1:]).copy()
    return H
