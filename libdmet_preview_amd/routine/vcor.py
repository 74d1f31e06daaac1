"""
Correlation-potential container with the reference's interface (libdmet/routine/vcor.py:19-103): a parameter vector,
its matrix value (spin | 3, nlo, nlo) and the linear map between them.  Host bookkeeping only.
"""
import numpy as np

from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.utils.misc import max_abs


class Vcor(object):
    def __init__(self):
        self.param = None
        self.value = None
        self.local = True
        self.is_vcor_kpts = False

    def update(self, param):
        self.param = param
        self.value = self.evaluate()

    def islocal(self):
        return self.local

    def is_local(self):
        return self.local

    def get(self, i=0, kspace=True):
        """i is the k-point (kspace) or cell index."""
        log.eassert(self.value is not None, "Vcor not initialized yet")
        if self.value.ndim == 4:      # (nkpts, spin, nlo, nlo)
            return self.value[i]
        if kspace or i == 0:
            return self.value
        return np.zeros_like(self.value)

    def evaluate(self):
        log.error("function evaulate() is not implemented")

    def gradient(self):
        log.error("function gradient() is not implemented")

    def length(self):
        log.error("function len() is not implemented")

    def assign(self, v0):
        """Least-squares projection of a matrix onto the parameters (local potentials, vcor.py:57-71)."""
        if not self.is_local():
            raise NotImplementedError("k-dependent correlation potentials are outside the HIP path")
        g = self.gradient()
        log.eassert(v0.shape == g.shape[1:], "The correlation potential should have shape %s, rather than %s",
                    g.shape[1:], v0.shape)
        gf = g.reshape(len(g), -1)
        self.update(gf.dot(np.asarray(v0).ravel()) / np.einsum('pi,pi->p', gf, gf))
        diff = max_abs(v0 - self.get())
        if diff > 1e-7:
            log.warn("symmetrization imposed on initial guess, diff = %.5g", diff)

    def __str__(self):
        return self.evaluate().__str__()
