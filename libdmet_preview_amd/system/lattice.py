"""
Lattice: the k-mesh / cell bookkeeping object the hot path is written against
(reference: libdmet/system/lattice.py:31-411, 716-726).

Only the parts the embedding-construction path touches are mirrored: cell index
arithmetic (integer, via libdmetk), the k <-> R transforms (HIP), `expand`, the orbital
index sets (`val_idx`, `virt_idx`, `core_idx`, `imp_idx`) and the Hamiltonian holders that
`HF` reads.  No PySCF cell is needed: pass the number of local orbitals (or any object with
`nao_nr()`) and the k-mesh.
"""
import ctypes as C
import itertools as it

import numpy as np

from libdmet_preview_amd._lib import lib, mesh3
from libdmet_preview_amd.settings import IMAG_DISCARD_TOL
from libdmet_preview_amd.system import fourier
from libdmet_preview_amd.system.fourier import (FFTtoK, FFTtoT, k2R, R2k, make_kpts_scaled,  # noqa: F401
                                                round_to_FBZ, kpt_member, get_phase_R2k)
from libdmet_preview_amd.utils import logger as log

try:
    from collections.abc import Iterable
except ImportError:  # pragma: no cover
    from collections import Iterable


class _UnitCell(object):
    """Stand-in for a PySCF cell: unit lattice vectors, k_abs = 2 pi k_scaled."""
    def __init__(self, nao, dimension=3):
        self._nao = int(nao)
        self.dimension = dimension

    def nao_nr(self):
        return self._nao

    def lattice_vectors(self):
        return np.eye(3)

    def get_scaled_kpts(self, kpts):
        return np.asarray(kpts) / (2.0 * np.pi)

    def get_abs_kpts(self, kscaled):
        return np.asarray(kscaled) * (2.0 * np.pi)


class Lattice(object):
    def __init__(self, cell, kmesh):
        if isinstance(cell, (int, np.integer)):
            cell = _UnitCell(cell)
        self.mol = self.cell = cell
        kmesh = [int(x) for x in kmesh]
        kmesh = kmesh + [1] * (3 - len(kmesh))
        self.kmesh = kmesh
        self.nscsites = self.nao = int(cell.nao_nr())
        self.dim = getattr(cell, "dimension", 3)
        self.csize = np.asarray(kmesh)
        self.ncells = int(np.prod(self.csize))
        self.cells = fourier.make_cells(kmesh)
        self.celldict = dict(zip(map(tuple, self.cells), range(self.ncells)))
        self.kpts_scaled = make_kpts_scaled(kmesh)
        self.kpts = self.kpts_abs = cell.get_abs_kpts(self.kpts_scaled)
        self.nkpts = len(self.kpts)
        self.nsites = self.ncells * self.nscsites
        self._add = None
        self._sub = None
        _, self._neg, self._weights = fourier.kmesh_tables(kmesh)

        self.val_idx = []
        self.virt_idx = []
        self.core_idx = []

        self.hcore_lo_k = self.fock_lo_k = self.rdm1_lo_k = None
        self.hcore_lo_R = self.fock_lo_R = self.rdm1_lo_R = None
        self.ovlp_lo_k = self.ovlp_lo_R = None
        self.vhf_lo_k = None
        self.C_ao_lo = None
        self.df = None
        self.JK_imp = self.JK_core = self.Ham = None
        self.eri_symmetry = 4
        self.H2_format = None
        self._H2_local = None
        self.H0 = 0.0
        self.use_hcore_as_emb_ham = False
        self.is_model = False
        self.restricted = None

    # ---- orbital sets (lattice.py:100-163) -----------------------------------------------
    @property
    def ncore(self):
        return len(self.core_idx)

    @property
    def nval(self):
        return len(self.val_idx)

    @property
    def nvirt(self):
        return len(self.virt_idx)

    @property
    def nimp(self):
        return self.nval + self.nvirt

    limp = nimp

    @property
    def imp_idx(self):
        return list(self.val_idx) + list(self.virt_idx)

    def set_val_virt_core(self, val, virt, core):
        self.core_idx = list(core) if isinstance(core, Iterable) else list(range(0, core))
        self.val_idx = list(val) if isinstance(val, Iterable) else list(range(self.ncore, self.ncore + val))
        self.virt_idx = (list(virt) if isinstance(virt, Iterable)
                         else list(range(self.ncore + self.nval, self.ncore + self.nval + virt)))
        if self.ncore + self.nval + self.nvirt != self.nao:
            log.warn("ncore (%s) + nval (%s) + nvirt (%s) != nao (%s), \nset_val_virt_core may be incorrect.",
                     self.ncore, self.nval, self.nvirt, self.nao)

    # ---- cell arithmetic (lattice.py:194-204), integer tables from libdmetk ---------------
    def _table(self, sign):
        t = np.empty((self.ncells, self.ncells), dtype=np.int32)
        rc = lib.dmk_cell_add_table(mesh3(self.kmesh), sign, t.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise ValueError("dmk_cell_add_table failed")
        return t

    def cell_idx2pos(self, idx):
        return self.cells[idx % self.ncells]

    def cell_pos2idx(self, pos):
        return self.celldict[tuple(np.asarray(pos) % self.csize)]

    def add(self, i, j):
        if self._add is None:
            self._add = self._table(+1)
        return int(self._add[i % self.ncells, j % self.ncells])

    def subtract(self, i, j):
        if self._sub is None:
            self._sub = self._table(-1)
        return int(self._sub[i % self.ncells, j % self.ncells])

    def neg(self, i):
        return int(self._neg[i % self.ncells])

    # ---- transforms (lattice.py:209-219, 399-411) ------------------------------------------
    def FFTtoK(self, A):
        return FFTtoK(A, self.kmesh)

    def FFTtoT(self, B, tol=IMAG_DISCARD_TOL):
        return FFTtoT(B, self.kmesh, tol=tol)

    def k2R(self, A, tol=IMAG_DISCARD_TOL):
        return k2R(A, self.kmesh, tol=tol)

    def R2k(self, B):
        return R2k(B, self.kmesh)

    def k2R_basis(self, basis_k):
        return self.k2R(basis_k)

    def R2k_basis(self, basis_R):
        return self.R2k(basis_R)

    def expand(self, A, dense=False):
        """Stripe -> full (lattice.py:304-337): big[(R1),(R2)] = A[R1 - R2]; index gather on the host
        (model-size matrices only; the bath builder never materialises this)."""
        A = np.asarray(A)
        assert A.shape[-3] == self.ncells
        n = A.shape[-1]
        nc = self.ncells
        if self._sub is None:
            self._sub = self._table(-1)
        idx = self._sub.astype(np.int64)          # idx[R1, R2] = R1 - R2
        if A.ndim == 3:
            big = A[idx]                            # (R1, R2, n, n)
            return np.ascontiguousarray(big.transpose(0, 2, 1, 3)).reshape(nc * n, nc * n)
        elif A.ndim == 4:
            big = A[:, idx]
            return np.ascontiguousarray(big.transpose(0, 1, 3, 2, 4)).reshape(A.shape[0], nc * n, nc * n)
        raise ValueError("unknown shape of A, %s" % (A.shape,))

    def extract_stripe(self, A):
        nc = self.ncells
        n = A.shape[-1] // nc
        if A.ndim == 2:
            return A.reshape((nc, n, nc, n))[:, :, 0]
        elif A.ndim == 3:
            return A.reshape((A.shape[0], nc, n, nc, n))[:, :, :, 0]
        raise ValueError("unknown shape of A, %s" % (A.shape,))

    def transpose(self, A):
        A = np.asarray(A)
        neg = self._neg.astype(np.int64)
        if A.ndim == 3:
            return np.ascontiguousarray(A[neg].transpose(0, 2, 1))
        elif A.ndim == 4:
            return np.ascontiguousarray(A[:, neg].transpose(0, 1, 3, 2))
        raise ValueError("unknown shape of A, %s" % (A.shape,))

    # ---- Hamiltonian holders read by routine.mfd.HF (lattice.py:716-726) ---------------------
    def set_Ham_lo(self, fock_lo_R=None, hcore_lo_R=None, fock_lo_k=None, hcore_lo_k=None, H0=0.0,
                   use_hcore_as_emb_ham=False):
        """Install LO-basis one-body operators (stripe or k form; the other is derived by a fold)."""
        if fock_lo_R is None and fock_lo_k is not None:
            fock_lo_R = self.k2R(fock_lo_k)
        if fock_lo_k is None and fock_lo_R is not None:
            fock_lo_k = self.R2k(fock_lo_R)
        if hcore_lo_R is None and hcore_lo_k is not None:
            hcore_lo_R = self.k2R(hcore_lo_k)
        if hcore_lo_k is None and hcore_lo_R is not None:
            hcore_lo_k = self.R2k(hcore_lo_R)
        if hcore_lo_R is None:
            hcore_lo_R, hcore_lo_k = fock_lo_R, fock_lo_k
        if fock_lo_R is None:
            fock_lo_R, fock_lo_k = hcore_lo_R, hcore_lo_k
        self.fock_lo_R, self.fock_lo_k = fock_lo_R, fock_lo_k
        self.hcore_lo_R, self.hcore_lo_k = hcore_lo_R, hcore_lo_k
        self.H0 = H0
        self.use_hcore_as_emb_ham = use_hcore_as_emb_ham

    # ---- AO -> LO of the mean-field operators, the step in front of the hot path (lattice.py:416-673) --------
    def set_Ham(self, kmf, df, C_ao_lo, eri_symmetry=4, ovlp=None, hcore=None, rdm1=None, fock=None, veff=None, vhf=None,
                vj=None, vk=None, vxc=None, use_hcore_as_emb_ham=False, H0=0.0, hcore_hf_add=None):
        """hcore, fock, ovlp, rdm1, vhf in kAO -> kLO -> RLO (Hartree-Fock mean fields).

        `kmf` is duck-typed: it is only asked for what was not passed (get_ovlp, get_hcore, make_rdm1, get_jk).
        Restricted / unrestricted is read off the rank of rdm1 (the reference dispatches on the PySCF class,
        routine/pdft_helper.py:77-111): vhf = vj - vk/2 (RHF, spin-traced rdm1) or vj_a + vj_b - vk (UHF)."""
        if vxc is not None:
            raise NotImplementedError("DFT mean fields (vxc) are outside the HIP path")
        self.kmf, self.df = kmf, df
        self.C_ao_lo = np.asarray(C_ao_lo)

        def need(x, getter, what):
            if x is not None:
                return np.asarray(x)
            if kmf is None or not hasattr(kmf, getter):
                raise ValueError("set_Ham: %s was not given and kmf cannot provide it" % what)
            return np.asarray(getattr(kmf, getter)())
        ovlp, hcore, rdm1 = need(ovlp, "get_ovlp", "ovlp"), need(hcore, "get_hcore", "hcore"), need(rdm1, "make_rdm1", "rdm1")
        if (vj is None or vk is None) and (vhf is None):
            if kmf is None or not hasattr(kmf, "get_jk"):
                raise ValueError("set_Ham: neither vhf nor (vj, vk) given and kmf cannot provide them")
            vj, vk = kmf.get_jk(dm_kpts=rdm1)
        if vhf is None:
            vj, vk = np.asarray(vj), np.asarray(vk)
            vhf = vj - vk * 0.5 if rdm1.ndim == 3 else vj[0] + vj[1] - vk
        vhf = np.asarray(vhf)
        if veff is None:
            veff = vhf
        if fock is None:
            fock = hcore + veff
        fock_hf = hcore + vhf
        if hcore_hf_add is not None:
            fock_hf = fock_hf + hcore_hf_add
        self.ovlp_ao_k, self.hcore_ao_k, self.rdm1_ao_k = ovlp, hcore, rdm1
        self.fock_ao_k, self.fock_hf_ao_k, self.hcore_hf_add = np.asarray(fock), np.asarray(fock_hf), hcore_hf_add
        self.vj_ao_k = None if vj is None else np.asarray(vj)
        self.vk_ao_k = None if vk is None else np.asarray(vk)
        self.veff_ao_k, self.vhf_ao_k = np.asarray(veff), vhf
        if self.C_ao_lo.ndim == 3:
            self.spin, self.restricted = 1, True
        else:
            self.spin = self.C_ao_lo.shape[0]
            self.restricted = (self.spin == 1)
        self.eri_symmetry = eri_symmetry
        assert self.eri_symmetry in [1, 4, 8]
        if not self.restricted:
            assert self.eri_symmetry != 8
        self.transform_obj_to_lo()
        self.H0 = H0
        self.has_Ham = True
        self.use_hcore_as_emb_ham = use_hcore_as_emb_ham
        if self.use_hcore_as_emb_ham:
            log.warn("You are using hcore to construct embedding Hamiltonian...")

    setHam = set_Ham

    def set_Ham_model(self, Ham, rdm1=None, fock=None, ovlp=None, eri_symmetry=4, vj=None, vk=None, vxc=None,
                      use_hcore_as_emb_ham=True):
        """Model Hamiltonian object (getH1 / getFock / getH0 / H2_format / getH2), lattice.py:523-563."""
        if vxc is not None:
            raise NotImplementedError("DFT mean fields (vxc) are outside the HIP path")
        self.Ham = Ham
        self.hcore_lo_R = np.asarray(Ham.getH1())
        self.hcore_lo_k = self.R2k(self.hcore_lo_R)
        if ovlp is None:
            self.ovlp_lo_R = np.zeros((self.nkpts, self.nao, self.nao))
            self.ovlp_lo_R[0] = np.eye(self.nao)
        else:
            self.ovlp_lo_R = np.asarray(ovlp)
        self.ovlp_lo_k = self.R2k(self.ovlp_lo_R)
        self.fock_lo_R = np.asarray(Ham.getFock() if fock is None else fock)
        self.fock_lo_k = self.R2k(self.fock_lo_R)
        self.rdm1_lo_R = rdm1
        if rdm1 is not None:
            self.rdm1_lo_k = self.R2k(np.asarray(rdm1))
        self.check_imag()
        self.eri_symmetry = eri_symmetry
        self.use_hcore_as_emb_ham = use_hcore_as_emb_ham
        self.has_Ham = True
        self.is_model = True
        self.H2_format = Ham.H2_format
        self._H2_local = np.asarray(Ham.getH2()) if hasattr(Ham, "getH2") else None
        self.H0 = Ham.getH0()

    setHam_model = set_Ham_model

    def update_Ham(self, rdm1_lo_R, veff=None, vhf=None, **kwargs):
        """New Fock from the DMET density (lattice.py:569-592): LO stripe -> kLO -> kAO, then set_Ham again."""
        from libdmet_preview_amd.basis_transform import make_basis
        self.rdm1_lo_R = rdm1_lo_R
        self.rdm1_lo_k = self.R2k(np.asarray(rdm1_lo_R))
        self.rdm1_ao_k = make_basis.transform_rdm1_to_ao(self.rdm1_lo_k, self.C_ao_lo)
        if veff is None and vhf is None:
            vj = vk = None
        else:
            vj, vk = self.vj_ao_k, self.vk_ao_k
        self.set_Ham(self.kmf, self.df, self.C_ao_lo, self.eri_symmetry, ovlp=self.ovlp_ao_k, hcore=self.hcore_ao_k,
                     rdm1=self.rdm1_ao_k, fock=None, veff=veff, vhf=vhf, vj=vj, vk=vk, vxc=None, H0=self.H0,
                     use_hcore_as_emb_ham=self.use_hcore_as_emb_ham, hcore_hf_add=self.hcore_hf_add)

    def transform_obj_to_lo(self):
        """hcore, ovlp, fock, fock_hf, veff, vhf, rdm1: kAO -> kLO (batched device GEMMs) -> RLO (device fold)."""
        from libdmet_preview_amd.basis_transform import make_basis
        from libdmet_preview_amd.utils.misc import add_spin_dim
        if self.C_ao_lo.shape[-2] != self.hcore_ao_k.shape[-1]:
            raise NotImplementedError("transform_obj_to_lo: spin-orbital (GHF) coefficients are outside the HIP path")
        C = self.C_ao_lo
        t = make_basis.transform_h1_to_lo
        self.hcore_lo_k, self.ovlp_lo_k = t(self.hcore_ao_k, C), t(self.ovlp_ao_k, C)
        self.fock_lo_k, self.fock_hf_lo_k = t(self.fock_ao_k, C), t(self.fock_hf_ao_k, C)
        self.veff_lo_k, self.vhf_lo_k = t(self.veff_ao_k, C), t(self.vhf_ao_k, C)
        self.rdm1_lo_k = make_basis.transform_rdm1_to_lo(self.rdm1_ao_k, C, self.ovlp_ao_k)
        for name in ("hcore", "fock", "fock_hf", "veff", "vhf", "rdm1"):
            setattr(self, name + "_lo_k", add_spin_dim(getattr(self, name + "_lo_k"), self.spin))
        for name in ("hcore", "ovlp", "fock", "fock_hf", "veff", "vhf", "rdm1"):
            setattr(self, name + "_lo_R", self.k2R(getattr(self, name + "_lo_k")))
        self.check_imag()

    def check_imag(self):
        """k2R already returns real parts and warns above IMAG_DISCARD_TOL (lattice.py:675-706 is that bookkeeping)."""
        for name in ("hcore_lo_R", "fock_lo_R", "rdm1_lo_R"):
            x = getattr(self, name, None)
            if x is not None and np.iscomplexobj(x) and np.abs(np.asarray(x).imag).max() < IMAG_DISCARD_TOL:
                setattr(self, name, np.asarray(x).real)

    def getH1(self, kspace=True):
        return self.hcore_lo_k if kspace else self.hcore_lo_R

    def getFock(self, kspace=True):
        return self.fock_lo_k if kspace else self.fock_lo_R

    def getH0(self):
        return self.H0

    def get_ovlp(self, kspace=True):
        """LO overlap; identity in the (orthonormal) LO basis unless one was installed (lattice.py:728-732)."""
        if self.ovlp_lo_k is None:
            ov_R = np.zeros((self.ncells, self.nscsites, self.nscsites))
            ov_R[0] = np.eye(self.nscsites)
            self.ovlp_lo_R = ov_R
            self.ovlp_lo_k = np.asarray([np.eye(self.nscsites, dtype=np.complex128)] * self.ncells)
        return self.ovlp_lo_k if kspace else self.ovlp_lo_R

    def getImpJK(self):
        """lattice.py:754-760."""
        if self.JK_imp is not None:
            return self.JK_imp
        elif self.Ham is not None:
            return self.Ham.getImpJK()
        return None

    get_JK_imp = getImpJK

    def get_JK_core(self):
        return self.JK_core

    def getH2(self, kpts=None, compact=False, kspace=True, use_Ham=False):
        """Model lattices: the cell-local two-body tensor installed by set_H2_local (lattice.py:738-751)."""
        if kspace:
            raise NotImplementedError
        if self._H2_local is None:
            raise ValueError("no local H2 installed; call set_H2_local")
        return self._H2_local

    def set_H2_local(self, H2, H2_format="local"):
        """Install a cell-local ERI ((spin_pair,) nscsites^4) and mark the lattice as a model."""
        self._H2_local = np.asarray(H2)
        self.H2_format = H2_format
        self.eri_symmetry = 1
        self.is_model = True


# ---- model lattices (system/lattice.py:796-1109): geometry of a supercell tiling, neighbour search, the standard constructors ----

class UnitCell(object):
    """Primitive cell of a model lattice: `size` (dim, dim) lattice vectors as rows, `sites` a list of (position, name)."""

    def __init__(self, size, sites):
        self.size = np.array(size)
        log.eassert(self.size.ndim == 2 and self.size.shape[0] == self.size.shape[1], "Invalid unitcell constants")
        self.dim = self.size.shape[0]
        self.sites, self.names = [], []
        for pos, name in sites:
            log.eassert(np.shape(pos) == (self.dim,), "Invalid position for the site")
            self.sites.append(np.asarray(pos))
            self.names.append(name)
        self.nsites = len(self.sites)
        self.sitedict = dict(zip(map(tuple, self.sites), range(self.nsites)))

    def __str__(self):
        rows = ["%-10s%-10s" % (n, s) for n, s in zip(self.names, self.sites)]
        return "UnitCell Shape\n%s\nSites:\n%s\n\n" % (self.size, "\t".join(rows))


def translateSites(baseSites, usize, csize):
    """Cells of a `csize` tiling (C order) and all sites: every cell's copy of the base sites, cell by cell."""
    cells = [np.asarray(c) for c in it.product(*[range(int(n)) for n in csize])]
    sites = [np.dot(c, usize) + s for c in cells for s in baseSites]
    return cells, sites


class SuperCell(object):
    """`size` copies of a unit cell: the impurity cluster of a model."""

    def __init__(self, uc, size):
        self.unitcell, self.dim = uc, uc.dim
        self.csize = np.array(size)
        self.size = np.dot(np.diag(self.csize), uc.size)
        self.ncells = int(np.prod(self.csize))
        self.nsites = uc.nsites * self.ncells
        self.cells, self.sites = translateSites(uc.sites, uc.size, size)
        self.names = uc.names * self.ncells
        self.celldict = dict(zip(map(tuple, self.cells), range(self.ncells)))
        self.sitedict = dict(zip(map(tuple, self.sites), range(self.nsites)))

    def __str__(self):
        return "%sSuperCell Shape\n%s\nNumber of Sites:%d\n\n" % (self.unitcell, self.size, self.nsites)


class LatticeModel(Lattice):
    """A periodic tiling of `size` supercells (system/lattice.py:796-858): the Lattice of this package -- k mesh = the tiling, one
    orbital per site of the supercell -- plus the real-space geometry that model Hamiltonians are written in (site positions,
    neighbour search).  No PySCF cell is built: nothing on the device path reads one for a model."""

    def __init__(self, sc, size):
        size = [int(x) for x in np.asarray(size).ravel()]
        Lattice.__init__(self, _UnitCell(sc.nsites, dimension=sc.dim), size)
        self.supercell, self.dim = sc, sc.dim
        self.csize = np.array(size)
        self.size = np.dot(np.diag(self.csize), sc.size)
        self.cells, self.sites = translateSites(sc.sites, sc.size, size)
        self.celldict = dict(zip(map(tuple, self.cells), range(self.ncells)))
        self.sitedict = dict(zip(map(tuple, self.sites), range(self.nsites)))
        self.names = np.asarray(list(sc.names) * self.ncells)
        self.coords = np.asarray(self.sites)
        self.neighborDist = []
        self.val_idx, self.virt_idx, self.core_idx = list(range(self.nao)), [], []
        self.kmf = self.kmf_lo = None
        self.eri_symmetry = None
        self.is_model, self.has_Ham = True, False
        self.set_Ham = self.setHam = self.set_Ham_model

    def __str__(self):
        return "%sLattice Shape\n%s\nNumber of SuperCells: %4d\nNumber of Sites:      %4d\n" % (self.supercell, self.size, self.ncells, self.nsites)

    # cell arithmetic in the model's own dimension (the base class pads the mesh to three axes)
    def cell_idx2pos(self, idx):
        return np.asarray(self.cells[idx % self.ncells])

    def cell_pos2idx(self, pos):
        return self.celldict[tuple(np.asarray(pos) % self.csize)]

    def add(self, i, j):
        return self.cell_pos2idx(self.cell_idx2pos(i) + self.cell_idx2pos(j))

    def subtract(self, i, j):
        return self.cell_pos2idx(self.cell_idx2pos(i) - self.cell_idx2pos(j))

    def site_idx2pos(self, idx):
        return self.sites[idx % self.nsites]

    def site_pos2idx(self, pos):
        return self.sitedict[tuple(np.asarray(pos) % np.diag(self.size))]

    def neighbor(self, dis=1.0, max_range=1, sitesA=None, sitesB=None, search_range=1):
        """Pairs (siteA, siteB) of site INDICES at distance `dis` (system/lattice.py:894-925): siteB within `max_range` cells of
        siteA's cell, distances taken modulo up to `search_range` lattice periods (0: open boundaries).  Vectorised: one distance
        table per siteA against its candidate sites and all period shifts."""
        sitesA = range(self.nsites) if sitesA is None else sitesA
        allowed = None if sitesB is None else set(sitesB)
        nsc = self.nscsites
        cellshifts = [self.cell_pos2idx(np.asarray(s)) for s in it.product(range(-max_range, max_range + 1), repeat=self.dim)]
        shifts = np.asarray([np.dot(s, self.size) for s in it.product(range(-search_range, search_range + 1), repeat=self.dim)])
        pos = np.asarray(self.sites, dtype=float)
        out = []
        for a in sitesA:
            cells = {self.add(a // nsc, x) for x in cellshifts}
            cand = sorted(b for c in cells for b in range(c * nsc, (c + 1) * nsc) if allowed is None or b in allowed)
            d = np.linalg.norm(pos[a][None, None, :] - pos[cand][:, None, :] - shifts[None, :, :], axis=-1)
            hit = (np.abs(d - dis) < 1e-5).any(axis=1)
            out += [(a, b) for b, h in zip(cand, hit) if h]
        return out

    def update_Ham(self, rdm1_lo_R, fock_lo_k=None, ghf=False, **kwargs):
        """New mean-field Fock from the DMET density of a model (system/lattice.py:927-972): with a cell-local ERI the Coulomb and
        exchange potentials come from the cell-0 density alone and are the same at every k -- J and K on the device (dmk_jk_s4),
        fock = hcore + J - K / 2 (restricted, spin-traced density) or J_a + J_b - K_s."""
        from libdmet_preview_amd.solver import scf
        log.info("Update DMET mean-field Hamiltonian.")
        assert self.has_Ham
        if ghf:
            raise NotImplementedError("update_Ham of a GSO model lattice (spin-local ERI) is not built")
        rdm1 = np.asarray(rdm1_lo_R)
        self.rdm1_lo_R = rdm1 if rdm1.ndim == 4 else rdm1[np.newaxis]
        self.rdm1_lo_k = self.R2k(self.rdm1_lo_R)
        if fock_lo_k is None:
            if self.H2_format != "local":
                raise NotImplementedError("update_Ham with a %s lattice ERI (cell-resolved J / K) is not built" % self.H2_format)
            spin = self.rdm1_lo_R.shape[0]
            dm0 = np.ascontiguousarray(self.rdm1_lo_R[:, 0].real)                # (1/nk) sum_k rho_k = the cell-0 block
            vj, vk = scf._get_jk(dm0, np.asarray(self.getH2(compact=False, kspace=False)))
            JK = (vj - vk * 0.5) if spin == 1 else (vj[0] + vj[1] - vk)
            hcore = np.asarray(self.hcore_lo_k)
            self.fock_lo_k = hcore + (JK[:, None] if hcore.ndim == 4 else JK[0][None])
        else:
            self.fock_lo_k = fock_lo_k
        self.fock_lo_R = self.k2R(self.fock_lo_k)
        self.check_imag()

    def getH2(self, compact=False, kspace=False, use_Ham=True):
        """The lattice two-body tensor of the installed Hamiltonian (system/lattice.py:1006-1010)."""
        if self._H2_local is not None:
            return self._H2_local
        return self.Ham.getH2()


def ChainLattice(length, scsites):
    """1-D 1-band model: `length` sites in supercells of `scsites`."""
    log.eassert(length % scsites == 0, "incompatible lattice and supercell sizes")
    sc = SuperCell(UnitCell(np.eye(1), [(np.array([0]), "X")]), np.asarray([scsites]))
    lat = LatticeModel(sc, np.asarray([length // scsites]))
    lat.neighborDist = [1.0, 2.0, 3.0]
    return lat


def SquareLattice(lx, ly, scx, scy):
    """2-D 1-band model on a square lattice."""
    log.eassert(lx % scx == 0 and ly % scy == 0, "incompatible lattice and supercell sizes")
    sc = SuperCell(UnitCell(np.eye(2), [(np.array([0, 0]), "X")]), np.asarray([scx, scy]))
    lat = LatticeModel(sc, np.asarray([lx // scx, ly // scy]))
    lat.neighborDist = [1.0, np.sqrt(2.0), 2.0]
    return lat


def CubicLattice(lx, ly, lz, scx, scy, scz):
    """3-D 1-band model on a simple cubic lattice."""
    log.eassert(lx % scx == 0 and ly % scy == 0 and lz % scz == 0, "incompatible lattice and supercell sizes")
    sc = SuperCell(UnitCell(np.eye(3), [(np.array([0, 0, 0]), "X")]), np.asarray([scx, scy, scz]))
    lat = LatticeModel(sc, np.asarray([lx // scx, ly // scy, lz // scz]))
    lat.neighborDist = [1.0, np.sqrt(2.0), np.sqrt(3.0)]
    return lat


def SquareAFM(lx, ly, scx, scy):
    """2-D 1-band model in the two-site antiferromagnetic cell (lattice vectors sqrt(2) along the diagonals, sites A at the corner
    and B at the centre); system/lattice.py:1109-1127."""
    log.eassert(lx % scx == 0 and ly % scy == 0, "incompatible lattice and supercell sizes")
    uc = UnitCell(np.eye(2) * np.sqrt(2.0), [(np.zeros(2), "X1"), (np.ones(2) * (np.sqrt(2.0) * 0.5), "X2")])
    lat = LatticeModel(SuperCell(uc, np.asarray([scx, scy])), np.asarray([lx // scx, ly // scy]))
    lat.neighborDist = [1.0, np.sqrt(2.0), 2.0]
    return lat


def Square3Band(lx, ly, scx, scy):
    """2-D 3-band (CuO2) model, one Cu and two O per unit cell of side 2; system/lattice.py:1129-1148."""
    log.eassert(lx % scx == 0 and ly % scy == 0, "incompatible lattice and supercell sizes")
    uc = UnitCell(np.eye(2) * 2.0, [(np.array([0.0, 0.0]), "Cu"), (np.array([1.0, 0.0]), "O"), (np.array([0.0, 1.0]), "O")])
    lat = LatticeModel(SuperCell(uc, np.asarray([scx, scy])), np.asarray([lx // scx, ly // scy]))
    lat.neighborDist = [1.0, np.sqrt(2.0), 2.0]
    return lat
