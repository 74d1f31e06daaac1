"""
Correlation-potential container.  Interface contract = libdmet/routine/vcor.py:19-103 (attribute names `param`,
`value`, `local`, `is_vcor_kpts`; methods `update / get / assign / evaluate / gradient / length / islocal / is_local`),
written here around two ideas of this package:

  * a potential is a LINEAR map  param -> value  whose Jacobian rows (`gradient()`) are mutually orthogonal masks, so
    `assign` is one matrix-vector product against the flattened Jacobian instead of a Python loop over parameters;
  * `value` is either one local block stack (spin | 3, nlo, nlo) -- the same matrix at every k, cell 0 only in real
    space -- or a per-k table (nkpts, spin, nlo, nlo) (`is_vcor_kpts`), which the mean-field stage adds to the Fock
    batch before the upload (routine/mfd.py `_fock_plus_vcor`).

Host bookkeeping only; the fit itself runs on the device (routine/slater.py EmbFitDevice).
"""
import numpy as np

from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.utils.misc import max_abs

SYMMETRIZE_WARN = 1e-7        # reference threshold for the "initial guess was symmetrised" warning (vcor.py:69-71)


def _abstract(name):
    def method(self, *args, **kwargs):
        log.error("Vcor.%s() must be supplied by the parametrisation (e.g. dmet.Hubbard.VcorLocal)", name)
    method.__name__ = name
    return method


class Vcor(object):
    def __init__(self):
        self.param = self.value = None          # set by update() / assign()
        self.local, self.is_vcor_kpts = True, False

    # --- parametrisation hooks (bound by the factories in dmet/Hubbard.py) ---------------------------------------
    evaluate = _abstract("evaluate")
    gradient = _abstract("gradient")
    length = _abstract("length")

    def update(self, param):
        """Adopt a parameter vector and refresh the matrix value it stands for."""
        self.param = param
        self.value = self.evaluate()

    def is_local(self):
        return self.local

    islocal = is_local

    def per_k(self):
        """True when `value` is a table over k-points rather than one local matrix."""
        return self.value is not None and np.ndim(self.value) == 4

    def get(self, i=0, kspace=True):
        """Potential seen by k-point `i` (kspace) or by cell `i` (real space: only cell 0 carries a local potential)."""
        log.eassert(self.value is not None, "correlation potential used before update() / assign()")
        if self.per_k():
            return self.value[i]
        on_site = kspace or i == 0
        return self.value if on_site else np.zeros_like(self.value)

    def assign(self, v0):
        """Least-squares projection of the matrix (or per-k table) `v0` onto the parameters (vcor.py:57-97)."""
        v0 = np.asarray(v0)
        if self.is_local():
            self._assign_local(v0)
        else:
            self._assign_kpts(v0)

    def _assign_local(self, v0):
        jac = np.asarray(self.gradient())
        log.eassert(v0.shape == jac.shape[1:], "vcor.assign: matrix of shape %s given, the parametrisation spans %s",
                    v0.shape, jac.shape[1:])
        rows = jac.reshape(jac.shape[0], -1)
        self.update(rows @ v0.ravel() / (rows * rows).sum(axis=1))
        drift = max_abs(v0 - self.get())
        if drift > SYMMETRIZE_WARN:
            log.warn("vcor.assign: guess left the parametrised space by %.5g (projected)", drift)

    def _assign_kpts(self, v0):
        """k-dependent potentials: parameter group `steps[i]` acts on the k-points `kpts_map[i]` (one k, or a +-k pair whose
        second member sees the complex conjugate); Jacobian blocks g[i] have shape (len(step), spin, nlo, nlo)."""
        jac = self.gradient()
        param = np.empty(self.length())
        for grp, (step, ks) in enumerate(zip(self.steps, self.kpts_map)):
            gc = np.conj(np.asarray(jac[grp]))
            seen = v0[ks[0]] if len(ks) == 1 else v0[ks[0]] + np.conj(v0[ks[1]])
            overlap = np.einsum('xsqp,spq->x', gc, seen)
            weight = np.einsum('xsqp,xspq->x', gc, np.asarray(jac[grp]))
            param[step] = overlap.real / weight.real
        self.update(param)
        if any(max_abs(v0[k] - self.get(k)) > SYMMETRIZE_WARN for k in range(self.nkpts)):
            log.warn("vcor.assign: per-k guess left the parametrised space (projected)")

    def __str__(self):
        return str(self.evaluate())
