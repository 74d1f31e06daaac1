"""
oracle/shim.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Import shim that lets the *reference* (pure-Python libDMET under /root/reference)
be imported in this container although PySCF / h5py / mpi4py are absent.  It is
used only by oracle/gen_golden.py (to capture golden vectors from the reference's
own arithmetic) and by the optional "reference present" legs of tests/.  Nothing
here travels to the GPU box in a way that matters: /root/reference does not exist
there and `available()` returns False.

How it works:
  * a sys.meta_path finder fabricates stub packages for pyscf.*, h5py, mpi4pyscf,
    mpi4py so that `from pyscf.pbc import df` and `class KUHF(pbc.scf.kuhf.KUHF)`
    evaluate; attribute access on a stub yields a stub class;
  * three real constants are planted (KPT_DIFF_TOL, BOHR, lib.einsum);
  * for the GDF ERI driver eight PySCF primitives are RESTATED in numpy
    (r_e2, _conc_mos, pack_tril, unpack_tril, hermi_sum, lib.dot, ao2mo.restore,
    prange, cartesian_prod) and sr_loop/get_naoaux are monkey-patched to serve
    in-memory DF blocks.  The reference's own control flow (k-conservation loop,
    time-reversal bookkeeping, weights, accumulation) runs unmodified.
"""
import sys
import os
import types
import importlib.abc
import importlib.machinery
import itertools
import numpy as np

REFERENCE_ROOT = os.environ.get("LIBDMET_REFERENCE_ROOT", "/root/reference")
_STUB_ROOTS = ("pyscf", "h5py", "mpi4pyscf", "mpi4py")
_installed = False


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "libdmet"))


def _stub_call(self, *a, **k):
    # `@lib.with_doc(doc)` style decorator factories: hand the function back
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    # `@mpi.parallel_call(skip_args=[1])`: return something that is again a decorator
    return _StubMeta("_stub", (), dict(_STUB_DICT))()


_STUB_DICT = {"__init__": lambda s, *a, **k: None, "__call__": _stub_call}


class _StubMeta(type):
    def __getattr__(cls, n):
        if n.startswith("__"):
            raise AttributeError(n)
        v = _StubMeta(n, (), dict(_STUB_DICT))
        setattr(cls, n, v)      # cache, so that `mpi.pool.size = n` sticks
        return v


class _StubModule(types.ModuleType):
    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        full = self.__name__ + "." + n
        if full in sys.modules:
            return sys.modules[full]
        v = _StubMeta(n, (), dict(_STUB_DICT))
        setattr(self, n, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, m):
        pass


# ----------------------------------------------------------------------------
# numpy restatements of the PySCF primitives the GDF driver touches
# (semantics: SURVEY.md Appendix D; call sites eri_transform.py:133-139, 217,
#  372-375, 432-433, 455-485, 529-543)
# ----------------------------------------------------------------------------

def cartesian_prod(arrays, out=None):
    arrays = [np.asarray(a) for a in arrays]
    dtype = np.result_type(*arrays)
    res = np.array(list(itertools.product(*arrays)), dtype=dtype)
    return res.reshape(-1, len(arrays))


def prange(start, end, step):
    for i in range(start, end, step):
        yield i, min(i + step, end)


def pack_tril(mat, axis=-1, out=None):
    mat = np.asarray(mat)
    n = mat.shape[-1]
    idx = np.tril_indices(n)
    res = mat[..., idx[0], idx[1]]
    if out is not None:
        flat = out.reshape(-1)
        flat[:res.size] = res.reshape(-1)
        return flat[:res.size].reshape(res.shape)
    return res


def unpack_tril(tril, filltriu=1, axis=-1, out=None):
    tril = np.asarray(tril)
    npair = tril.shape[-1]
    n = int((np.sqrt(8 * npair + 1) - 1) // 2)
    idx = np.tril_indices(n)
    res = np.zeros(tril.shape[:-1] + (n, n), dtype=tril.dtype)
    res[..., idx[0], idx[1]] = tril
    # HERMITIAN fill
    res[..., idx[1], idx[0]] = tril.conj()
    return res


def hermi_sum(a, axes=None, hermi=1, inplace=False, out=None):
    assert hermi == 3 and inplace
    a += a.transpose(0, 2, 1).copy()
    return a


def dot(a, b, alpha=1, c=None, beta=0):
    if c is None:
        return alpha * np.dot(a, b)
    if beta == 0:
        c[:] = alpha * np.dot(a, b)
    else:
        c *= beta
        c += alpha * np.dot(a, b)
    return c


def _conc_mos(moi, moj, compact=False):
    ni = moi.shape[1]
    nj = moj.shape[1]
    mo = np.asarray(np.hstack((moi, moj)), order="F")
    return None, None, mo, (0, ni, ni, ni + nj)


def r_e2(eri, mo_coeff, orbs_slice, tao, ao_loc, aosym="s1", out=None):
    i0, i1, j0, j1 = orbs_slice
    nL = eri.shape[0]
    n = mo_coeff.shape[0]
    e = eri.reshape(nL, n, n)
    Ci = mo_coeff[:, i0:i1]
    Cj = mo_coeff[:, j0:j1]
    res = np.einsum("pa,Lpq,qb->Lab", Ci.conj(), e, Cj, optimize=True)
    res = res.reshape(nL, -1)
    if out is not None:
        out[:] = res
        return out
    return res


def restore(symmetry, eri, norb):
    """ao2mo.restore for real input in 1-, 4- or 8-fold form."""
    eri = np.asarray(eri)
    npair = norb * (norb + 1) // 2
    symmetry = int(str(symmetry).replace("s", ""))
    if eri.size == npair * npair:
        eri4 = eri.reshape(npair, npair)
    elif eri.size == norb ** 4:
        e1 = eri.reshape(norb, norb, norb, norb)
        idx = np.tril_indices(norb)
        eri4 = e1[idx[0], idx[1]][:, idx[0], idx[1]]
    elif eri.size == npair * (npair + 1) // 2:             # 8-fold input: the packed lower triangle of the (npair, npair) matrix
        i2 = np.tril_indices(npair)
        eri4 = np.zeros((npair, npair), dtype=eri.dtype)
        eri4[i2[0], i2[1]] = eri.reshape(-1)
        eri4[i2[1], i2[0]] = eri.reshape(-1)
    else:
        raise ValueError("restore: unsupported input size")
    if symmetry == 4:
        return eri4
    idx = np.tril_indices(norb)
    if symmetry == 1:
        tmp = np.zeros((norb, norb, npair), dtype=eri4.dtype)
        tmp[idx[0], idx[1]] = eri4
        tmp[idx[1], idx[0]] = eri4
        e1 = np.zeros((norb, norb, norb, norb), dtype=eri4.dtype)
        e1[:, :, idx[0], idx[1]] = tmp
        e1[:, :, idx[1], idx[0]] = tmp
        return e1
    if symmetry == 8:
        i2 = np.tril_indices(npair)
        return eri4[i2[0], i2[1]]
    raise ValueError("restore: unsupported symmetry %s" % symmetry)


def install():
    """Install the stub finder and plant constants. Idempotent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.meta_path.insert(0, _Finder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import pyscf  # noqa: F401  (stub)
    import pyscf.lib
    import pyscf.pbc.lib.kpts_helper
    import pyscf.data.nist
    pyscf.pbc.lib.kpts_helper.KPT_DIFF_TOL = 1e-6
    pyscf.data.nist.BOHR = 0.52917721092
    L = pyscf.lib
    L.einsum = np.einsum
    L.cartesian_prod = cartesian_prod
    L.pack_tril = pack_tril
    L.unpack_tril = unpack_tril
    L.hermi_sum = hermi_sum
    L.dot = dot
    L.prange = prange
    L.SYMMETRIC = 3
    L.HERMITIAN = 1
    L.current_memory = lambda: [0.0]
    _installed = True


def patch_scf():
    """Bind the restated PySCF primitives used by libdmet.solver.scf._get_jk (solver/scf.py:255-335)."""
    install()
    from libdmet.solver import scf as rscf
    from oracle import restate_ham

    class _hf(object):
        dot_eri_dm = staticmethod(restate_ham.dot_eri_dm)

    class _ao2mo_mod(object):
        restore = staticmethod(restore)
    rscf.hf = _hf
    rscf.ao2mo = _ao2mo_mod
    return rscf


def pyscf_vec_lowdin(c, s=1):
    """pyscf.lo.orth.vec_lowdin / lowdin (PySCF >= 2.0, the pin of the reference's pyproject.toml; absent here): Loewdin
    orthonormalisation c (c^H s c)^(-1/2) with eigenvalues <= 1e-15 of the metric dropped."""
    c = np.asarray(c)
    m = c.conj().T.dot(c) if np.isscalar(s) and s == 1 else c.conj().T.dot(s).dot(c)
    e, v = np.linalg.eigh(m)
    keep = e > 1e-15
    return c.dot((v[:, keep] / np.sqrt(e[keep])).dot(v[:, keep].conj().T))


def pyscf_mo_1to1map(s):
    """pyscf.tools.mo_mapping.mo_1to1map: for every row i of |<i|j>| the column of its largest entry, columns used once."""
    s1 = abs(np.array(s, copy=True))
    out = []
    for i in range(s1.shape[0]):
        k = int(np.argmax(s1[i]))
        out.append(k)
        s1[:, k] = 0
    return out


def patch_scdm():
    """Bind the two PySCF primitives libdmet.lo.scdm.scdm_model needs (lo/scdm.py:116-150) and return libdmet.routine.localizer."""
    install()
    from libdmet.lo import scdm as rscdm

    class _lo(object):
        vec_lowdin = staticmethod(pyscf_vec_lowdin)

    class _mm(object):
        mo_1to1map = staticmethod(pyscf_mo_1to1map)
    rscdm.lo = _lo
    rscdm.mo_mapping = _mm
    from libdmet.routine import localizer
    return localizer


def quiet():
    from libdmet.utils import logger as log
    log.verbose = "RESULT"


class FakeCell(object):
    """Duck-typed pyscf Cell: identity lattice vectors, k = 2 pi * scaled."""
    def copy(self):
        import copy as _copy
        return _copy.copy(self)

    def __init__(self, nao, dimension=3):
        self._nao = nao
        self.dimension = dimension
        self.low_dim_ft_type = None

    def nao_nr(self):
        return self._nao

    def lattice_vectors(self):
        return np.eye(3)

    def get_scaled_kpts(self, kpts):
        return np.asarray(kpts) / (2.0 * np.pi)

    def get_abs_kpts(self, kscaled):
        return np.asarray(kscaled) * (2.0 * np.pi)


class FakeGDF(object):
    """Duck-typed GDF holding DF blocks in memory: blocks[(i, j)] -> (naux, nao, nao)."""
    def __init__(self, cell, kpts, blocks, naux, blockdim=240, max_memory=4000):
        self.cell = cell
        self.kpts = np.asarray(kpts)
        self.blocks = blocks
        self.naux = naux
        self._cderi = "mem"
        self.blockdim = blockdim
        self.max_memory = max_memory

    def find(self, kpt):
        d = np.abs(self.kpts - np.asarray(kpt)[None]).max(axis=1)
        idx = np.where(d < 1e-9)[0]
        assert len(idx) == 1
        return int(idx[0])


def kpts_is_zero(kpt):
    """kpts_helper.is_zero / gamma_point: the sum of |kpt| over all components below KPT_DIFF_TOL."""
    return np.abs(np.asarray(kpt)).sum() < 1e-6


def kpts_member(kpt, kpts):
    """kpts_helper.member: indices of the rows of kpts equal to kpt."""
    kpts = np.reshape(kpts, (len(kpts), -1))
    return np.where(np.abs(kpts - np.ravel(kpt)[None]).max(axis=1) < 1e-6)[0]


def kpts_unique(kpts):
    """kpts_helper.unique: (unique rows, first indices, inverse map) in order of first appearance."""
    kpts = np.asarray(kpts)
    uniq, idx, inv = [], [], np.zeros(len(kpts), dtype=int)
    for i, k in enumerate(kpts):
        for u, ku in enumerate(uniq):
            if np.abs(k - ku).max() < 1e-6:
                inv[i] = u
                break
        else:
            inv[i] = len(uniq)
            uniq.append(k)
            idx.append(i)
    return np.asarray(uniq), np.asarray(idx), inv


def patch_eri_transform():
    """Bind restated primitives into libdmet.basis_transform.eri_transform."""
    install()
    from libdmet.basis_transform import eri_transform as et

    class _AO2MO(object):
        r_e2 = staticmethod(r_e2)
    et._ao2mo = _AO2MO
    et._conc_mos = _conc_mos

    class _ao2mo_mod(object):
        restore = staticmethod(restore)
    et.ao2mo = _ao2mo_mod

    def sr_loop(gdf, kpti_kptj=None, max_memory=2000, compact=True, blksize=None):
        kpti, kptj = kpti_kptj
        i, j = gdf.find(kpti), gdf.find(kptj)
        if callable(gdf.blocks):
            blk = gdf.blocks(i, j)
        else:
            blk = gdf.blocks[(i, j)]
        blk = np.asarray(blk, dtype=np.complex128)
        naux = blk.shape[0]
        nao = blk.shape[1]
        if blksize is None:
            blksize = naux
        for b0 in range(0, naux, blksize):
            yield blk[b0:b0 + blksize].reshape(-1, nao * nao)
    et.sr_loop = sr_loop
    et.get_naoaux = lambda gdf: gdf.naux
    # pyscf.pbc.lib.kpts_helper primitives (restated): is_zero / gamma_point / member / unique
    et.is_zero = kpts_is_zero
    et.gamma_point = kpts_is_zero
    et.member = kpts_member
    et.unique = kpts_unique
    return et


def pyscf_incore_general(eri, mo_coeffs, compact=True):
    """pyscf.ao2mo.incore.general for real input: (ij|kl) -> (pq|rs) with four coefficient matrices; the pairs (p, q) and (r, s)
    come back as lower triangles when `compact` and both members of the pair use the same coefficients (PySCF's rule)."""
    c1, c2, c3, c4 = [np.asarray(c) for c in mo_coeffs]
    n = c1.shape[0]
    full = restore(1, eri, n)
    out = np.einsum('ijkl,ip,jq,kr,ls->pqrs', full, c1, c2, c3, c4, optimize=True)
    same12 = compact and c1.shape == c2.shape and np.array_equal(c1, c2)
    same34 = compact and c3.shape == c4.shape and np.array_equal(c3, c4)
    if same12:
        t = np.tril_indices(c1.shape[1])
        out = out[t[0], t[1]]
    else:
        out = out.reshape(c1.shape[1] * c2.shape[1], c3.shape[1], c4.shape[1])
    if same34:
        t = np.tril_indices(c3.shape[1])
        out = out[:, t[0], t[1]]
    else:
        out = out.reshape(out.shape[0], -1)
    return out


def patch_spinless():
    """Bind the restated PySCF primitives of routine/spinless_helper.py (ao2mo.restore, ao2mo.incore.general: :338-346) and of
    solver/scf.py; returns (spinless, spinless_helper)."""
    patch_scf()
    from libdmet.routine import spinless_helper as sh, spinless

    class _incore(object):
        general = staticmethod(pyscf_incore_general)

    class _ao2mo_mod(object):
        restore = staticmethod(restore)
        incore = _incore
    sh.ao2mo = _ao2mo_mod
    spinless.ao2mo = _ao2mo_mod
    return spinless, sh
