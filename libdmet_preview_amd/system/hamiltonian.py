"""
Model Hamiltonians in the form `Lattice.set_Ham_model` takes, mirror of libdmet/system/hamiltonian.py:18-165: the container
HamNonInt (real-space one-body stripes + a lattice two-body tensor whose layout names the `H2_format`) and the 1-band Hubbard
model on the model lattices of system/lattice.py.  Host bookkeeping: a few hopping amplitudes per cell.
"""
import numpy as np

from libdmet_preview_amd.utils import logger as log


class HamNonInt(object):
    """H1 / Fock stripes ((spin,), ncells, nao, nao) (k-space input is folded to real space), H2 with or without a spin axis in
    the layouts 'local' (nao^4 or pair^2), 'nearest' (ncells, ...), 'full' (ncells^3, ...), optional impurity JK, H0."""

    def __init__(self, lattice, H1, H2, Fock=None, ImpJK=None, kspace_input=False, spin_dim_H2=None, H0=0.0):
        ncells, nao = lattice.ncells, lattice.nao
        npair = nao * (nao + 1) // 2
        H1 = np.asarray(H1)
        log.eassert(H1.shape[-3:] == (ncells, nao, nao), "H1 shape %s not compatible with lattice", H1.shape)
        self.H1 = lattice.k2R(H1) if kspace_input else H1
        if Fock is None:
            self.Fock = self.H1
        else:
            Fock = np.asarray(Fock)
            log.eassert(Fock.shape[-3:] == self.H1.shape[-3:], "Fock shape %s not compatible with lattice", Fock.shape)
            self.Fock = lattice.k2R(Fock) if kspace_input else Fock
        H2 = np.asarray(H2)
        self.spin_dim_H2 = spin_dim_H2
        lead = () if spin_dim_H2 is None else (spin_dim_H2,)
        prefix = "" if spin_dim_H2 is None else "spin "
        layouts = [("local", ()), ("nearest", (ncells,)), ("full", (ncells,) * 3)]
        self.H2_format = None
        for name, cells in layouts:
            if H2.shape in (lead + cells + (nao,) * 4, lead + cells + (npair, npair)):
                self.H2_format = prefix + name
                break
        if self.H2_format is None:
            log.error("H2 shape %s not compatible with supercell", H2.shape)
            raise ValueError("H2 shape %s not compatible with supercell" % (H2.shape,))
        self.H2 = H2
        if ImpJK is not None:
            log.eassert(np.shape(ImpJK)[-2:] == self.H1.shape[-2:], "ImpJK shape %s not compatible with supercell", np.shape(ImpJK))
        self.ImpJK = ImpJK
        self.H0 = H0

    def getH0(self):
        return self.H0

    def getH1(self):
        return self.H1

    def getH2(self):
        return self.H2

    def getFock(self):
        return self.Fock

    def getImpJK(self):
        return self.ImpJK


def HubbardHamiltonian(lattice, U, tlist=[1.0], obc=False, compact=False, tol=1e-10, return_H1=False):
    """1-band Hubbard model H = -t sum_<ij> - t' sum_<<ij>> - ... + U sum_i n_i n_i on a model lattice (hamiltonian.py:118-165):
    hoppings between the sites of supercell 0 and their neighbours at the lattice's `neighborDist`, an on-site U as a local ERI
    (4-fold packed with `compact`).  `obc`: open boundaries (no wrap-around bonds)."""
    ncells, n = lattice.ncells, lattice.nscsites
    H1 = np.zeros((ncells, n, n))
    for order, t in enumerate(tlist):
        if abs(t) < tol:
            continue
        log.eassert(order < len(lattice.neighborDist), "%dth near neighbor distance unspecified in Lattice object", order + 1)
        for i, j in lattice.neighbor(dis=lattice.neighborDist[order], sitesA=range(n), search_range=0 if obc else 1):
            H1[j // n, j % n, i] = -t
    if return_H1:
        return H1
    if compact:
        npair = n * (n + 1) // 2
        H2 = np.zeros((npair, npair))
        diag = np.cumsum([0] + list(range(2, n + 1)))            # positions of (i, i) in the lower-triangle pair list
        H2[diag, diag] = U
    else:
        H2 = np.zeros((n,) * 4)
        np.fill_diagonal(H2, U)
    return HamNonInt(lattice, H1, H2)
