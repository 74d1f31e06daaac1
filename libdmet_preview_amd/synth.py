"""
Seeded synthetic inputs for the embedding-construction path.

Host-side numpy generators for the small tensors (Fock, C_ao_lo, real-space DF
kernel W0, Hubbard H1) plus the named workload table C1..C5 of BASELINE.json.
The large density-fitted AO blocks of configs C4/C5 are never materialised on
the host: they are produced on the device by the Philox kernel behind
`dmk_df_block_philox` (csrc/philox.hip) with the same counter layout.

All tensors follow the reference's conventions: cells and k-points in
`cartesian_prod` order, last mesh axis fastest (system/lattice.py:44-48,
system/fourier.py:46-53); stripe storage A[R] = <R| A |0>.
"""
import itertools
import numpy as np
from scipy import fft as scifft
import scipy.linalg as la

DEFAULT_SEED = 20241223

# name -> dict(mesh, nao, naux, nimp(val), nemb target, spin)
WORKLOADS = {
    "C1": dict(mesh=(6, 1, 1), nlo=2, naux=0, nval=2, spin=1,
               desc="1D Hubbard L=12, U/t=4, 2-site impurity"),
    "C2": dict(mesh=(6, 6, 1), nlo=4, naux=0, nval=4, spin=1,
               desc="2D Hubbard 12x12 sites, 2x2 impurity, RHF"),
    "C3": dict(mesh=(4, 1, 1), nlo=10, naux=28, nval=2, spin=1,
               desc="H-chain-like GDF cc-pVDZ, 4x1x1"),
    "C4": dict(mesh=(4, 4, 4), nlo=104, naux=416, nval=32, spin=1,
               desc="diamond-like 4x4x4, GTH-DZVP GDF"),
    "C5": dict(mesh=(6, 6, 6), nlo=200, naux=800, nval=56, spin=2,
               desc="cuprate-like 6x6x6, nao 200, naux 800, UHF"),
}


def cells_of(mesh):
    return np.array(list(itertools.product(*[range(int(n)) for n in mesh])), dtype=np.int64)


def _min_image_norm(mesh):
    c = cells_of(mesh)
    m = np.asarray(mesh)
    d = np.minimum(c, m - c)
    return np.sqrt((d.astype(float) ** 2).sum(axis=1))


def _neg_index(mesh):
    c = cells_of(mesh)
    m = np.asarray(mesh)
    n = (-c) % m
    strides = np.array([int(np.prod(m[d + 1:])) for d in range(len(m))])
    return (n * strides).sum(axis=1)


def make_fock_R(mesh, nlo, spin=1, seed=DEFAULT_SEED, decay=0.5):
    """
    Real stripe operator F[R] with F[-R] = F[R]^T (so F_k is Hermitian and
    time-reversal symmetric), entries N(0,1) exp(-decay |R|), diagonal of the
    R = 0 block shifted by 2 arange(nlo)/nlo.
    """
    rng = np.random.default_rng(seed)
    nc = int(np.prod(mesh))
    G = rng.standard_normal((spin, nc, nlo, nlo))
    neg = _neg_index(mesh)
    F = 0.5 * (G + G[:, neg].transpose(0, 1, 3, 2))
    F *= np.exp(-decay * _min_image_norm(mesh))[None, :, None, None]
    F[:, 0] += np.diag(2.0 * np.arange(nlo) / nlo)[None]
    return F


def fold_R2k(A, mesh):
    """Host fftn over the mesh axes (generator-side helper, e^{-ikR}, no factor)."""
    A = np.asarray(A)
    lead = A.shape[:-3]
    B = A.reshape(lead + tuple(mesh) + A.shape[-2:])
    ax = tuple(range(len(lead), len(lead) + len(mesh)))
    return scifft.fftn(B, axes=ax).reshape(A.shape)


def make_C_ao_lo(mesh, nao, nlo, spin=1, seed=DEFAULT_SEED + 1, decay=0.7):
    """Per-k Loewdin-orthonormal, time-reversal-symmetric AO->LO coefficients (spin,nk,nao,nlo)."""
    rng = np.random.default_rng(seed)
    nc = int(np.prod(mesh))
    CR = rng.standard_normal((spin, nc, nao, nlo))
    CR *= np.exp(-decay * _min_image_norm(mesh))[None, :, None, None]
    CR[:, 0, :nlo, :] += 2.0 * np.eye(nlo)[None]
    Ck = fold_R2k(CR, mesh)
    out = np.empty_like(Ck)
    for s in range(spin):
        for k in range(nc):
            c = Ck[s, k]
            e, v = la.eigh(c.conj().T @ c)
            out[s, k] = c @ ((v / np.sqrt(e)) @ v.conj().T)
    return out


def make_W0(mesh, naux, nao, seed=DEFAULT_SEED + 2, decay=0.6):
    """
    Real-space DF kernel W0[L, R1, p, R2, s], symmetric under (R1,p) <-> (R2,s),
    decaying with |R1| + |R2| (physical recipe of SURVEY.md section 8d).
    """
    rng = np.random.default_rng(seed)
    nc = int(np.prod(mesh))
    W = rng.standard_normal((naux, nc, nao, nc, nao))
    W = 0.5 * (W + W.transpose(0, 3, 4, 1, 2))
    d = np.exp(-decay * _min_image_norm(mesh))
    W *= d[None, :, None, None, None] * d[None, None, None, :, None]
    return W


def df_blocks_from_W0(W0, mesh):
    """L^{(ki,kj)}[L,p,s] = sum_{R1,R2} e^{-i ki R1} e^{+i kj R2} W0[L,R1,p,R2,s]  -> (nk,nk,naux,nao,nao)."""
    naux, nc, nao, _, _ = W0.shape
    W = W0.reshape((naux,) + tuple(mesh) + (nao,) + tuple(mesh) + (nao,))
    nd = len(mesh)
    ax1 = tuple(range(1, 1 + nd))
    ax2 = tuple(range(2 + nd, 2 + 2 * nd))
    A = scifft.fftn(W, axes=ax1)
    A = scifft.ifftn(A, axes=ax2) * nc
    A = A.reshape(naux, nc, nao, nc, nao)
    return np.ascontiguousarray(A.transpose(1, 3, 0, 2, 4))


def hubbard_h1_R(mesh, cell_shape, t=1.0):
    """
    1-band nearest-neighbour Hubbard hopping in stripe form (system/hamiltonian.py:118-166
    semantics): the lattice is a periodic grid of prod(mesh)*prod(cell_shape) sites, a unit
    cell holds a `cell_shape` block of sites.  Returns H1 (ncells, nsc, nsc) real.
    """
    mesh = tuple(int(x) for x in mesh)
    cs = tuple(int(x) for x in cell_shape) + (1,) * (len(mesh) - len(cell_shape))
    full = tuple(m * c for m, c in zip(mesh, cs))
    nsc = int(np.prod(cs))
    nc = int(np.prod(mesh))
    sites = list(itertools.product(*[range(c) for c in cs]))
    H = np.zeros((nc, nsc, nsc))
    cstr = [int(np.prod(mesh[d + 1:])) for d in range(len(mesh))]
    for j, sj in enumerate(sites):            # site j in cell 0
        for d in range(len(mesh)):
            if full[d] == 1:
                continue
            for step in (+1, -1):
                pos = list(sj)
                pos[d] = (pos[d] + step) % full[d]
                if full[d] == 2 and step == -1:
                    continue                 # avoid double counting on a 2-ring
                cell = [pos[a] // cs[a] for a in range(len(mesh))]
                loc = tuple(pos[a] % cs[a] for a in range(len(mesh)))
                R = sum(c * s for c, s in zip(cell, cstr))
                i = sites.index(loc)
                H[R, i, j] += -t
    return H
