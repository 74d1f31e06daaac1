"""
ERI container helpers of libdmet/routine/slater_helper.py that sit on the hot path's exit:

  init_H2             slater_helper.py:444-471
  unit2emb            slater_helper.py:494-528   zero-padded copy of the unit-cell ERI into the embedding ERI
  reorder_spin_blocks routine/slater.py:461-462  (aa, ab, bb) -> (aa, bb, ab)

and the one-body folds into the embedding space (slater_helper.py:22-156):

  transform_trans_inv[_k]   (1/nk) Re sum_k B_k^H H_k B_k, one batched complex GEMM + one long-K GEMM
  transform_local / transform_imp / transform_imp_env
  transform_4idx / transform_eri_local   (model lattices: four quarter transforms per cell, dmk_dgemm_tn_acc_rect)

The in-core 4-fold and 8-fold forms are padded on the device (dmk_pad_block_f64); `unit2emb_dev` keeps
the ERI in HBM for callers that stay on the device.
"""
import numpy as np

from libdmet_preview_amd._lib import lib, get_ctx
from libdmet_preview_amd.settings import IMAG_DISCARD_TOL
from libdmet_preview_amd.utils import devmat
from libdmet_preview_amd.utils import logger as log
from libdmet_preview_amd.utils.misc import max_abs


def init_H2(norb, eri_symmetry, dtype=np.double, spin_dim=None):
    spin_dim = () if spin_dim is None else (spin_dim,)
    if eri_symmetry == 1:
        return np.zeros(spin_dim + (norb, norb, norb, norb), dtype=dtype)
    norb_pair = norb * (norb + 1) // 2
    if eri_symmetry == 4:
        return np.zeros(spin_dim + (norb_pair, norb_pair), dtype=dtype)
    if eri_symmetry == 8:
        return np.zeros(spin_dim + (norb_pair * (norb_pair + 1) // 2,), dtype=dtype)
    raise ValueError("unknown ERI symmetry: %s" % (eri_symmetry))


def restore_eri_local(H2, norb):
    """(spin, ...) cell-local ERI in 1- or 4-fold storage -> (spin, npair, npair) 4-fold (slater_helper.py:473-492)."""
    H2 = np.asarray(H2)
    spin = H2.shape[0]
    npair = norb * (norb + 1) // 2
    if H2.size == spin * npair * npair:
        return H2.reshape(spin, npair, npair)
    ctx = get_ctx()
    out = np.empty((spin, npair, npair), dtype=np.float64)
    for s in range(spin):
        d = ctx.to_device(np.ascontiguousarray(H2[s], dtype=np.float64).reshape(-1))
        d_out = ctx.empty((npair, npair), np.float64)
        ctx.check(lib.dmk_eri_to_s4(ctx.h, int(norb), 1, d.ptr, d_out.ptr))
        out[s] = d_out.get()
    return out


def unit2emb_dev(ctx, d_unit, spin_pair, shape_in, shape_out):
    """Device form: pad (spin_pair,) + shape_in into (spin_pair,) + shape_out (1- or 2-d trailing shapes)."""
    r_in, c_in = (1, shape_in[0]) if len(shape_in) == 1 else shape_in
    r_out, c_out = (1, shape_out[0]) if len(shape_out) == 1 else shape_out
    d_out = ctx.empty((spin_pair,) + tuple(shape_out), np.float64)
    ctx.check(lib.dmk_pad_block_f64(ctx.h, int(spin_pair), int(r_in), int(c_in), d_unit.ptr, int(r_out), int(c_out),
                                    d_out.ptr))
    return d_out


def unit2emb(H2_unit, neo):
    """Allocate H2_emb and fill the impurity block with H2_unit (same storage symmetry as the input)."""
    if isinstance(H2_unit, np.ndarray):
        spin_pair = H2_unit.shape[0]
        neo_pair = neo * (neo + 1) // 2
        ctx = get_ctx()
        if H2_unit.ndim == 5:      # 1-fold: nested zero padding, innermost index pair first
            nu = H2_unit.shape[1]
            d = ctx.to_device(H2_unit, np.float64)
            d_cd = unit2emb_dev(ctx, d, spin_pair * nu * nu, (nu, nu), (neo, neo))            # (sp*nu*nu, neo, neo)
            d_b = unit2emb_dev(ctx, d_cd, spin_pair * nu, (nu, neo * neo), (neo, neo * neo))  # (sp*nu, neo, neo^2)
            d_a = unit2emb_dev(ctx, d_b, spin_pair, (nu, neo ** 3), (neo, neo ** 3))          # (sp, neo, neo^3)
            return d_a.get().reshape((spin_pair,) + (neo,) * 4)
        elif H2_unit.ndim == 3:    # 4-fold
            d = ctx.to_device(H2_unit, np.float64)
            return unit2emb_dev(ctx, d, spin_pair, H2_unit.shape[1:], (neo_pair, neo_pair)).get()
        elif H2_unit.ndim == 2:    # 8-fold
            d = ctx.to_device(H2_unit, np.float64)
            return unit2emb_dev(ctx, d, spin_pair, H2_unit.shape[1:], (neo_pair * (neo_pair + 1) // 2,)).get()
        raise ValueError
    # out-of-core container (dict-like with dataset "ccdd"; only 4-fold, slater_helper.py:519-527)
    H2_unit_old = np.asarray(H2_unit["ccdd"])
    spin_pair = H2_unit_old.shape[0]
    neo_pair = neo * (neo + 1) // 2
    ctx = get_ctx()
    d = ctx.to_device(H2_unit_old, np.float64)
    padded = unit2emb_dev(ctx, d, spin_pair, H2_unit_old.shape[1:], (neo_pair, neo_pair)).get()
    H2_emb = H2_unit
    del H2_emb["ccdd"]
    if hasattr(H2_emb, "create_dataset"):
        H2_emb.create_dataset("ccdd", padded.shape, 'f8')
        H2_emb["ccdd"][:] = padded
    else:
        H2_emb["ccdd"] = padded
    return H2_emb


def reorder_spin_blocks_dev(ctx, d_H2):
    """(aa, ab, bb) -> (aa, bb, ab) for a device ERI (3, ...) in place-equivalent new array."""
    if d_H2.shape[0] != 3:
        return d_H2
    per = d_H2.size // 3
    out = ctx.empty(d_H2.shape, np.float64)
    for dst, src in enumerate((0, 2, 1)):
        ctx.check(lib.dmk_memcpy_d2d(ctx.h, out.offset(dst * per, (per,)).ptr, d_H2.offset(src * per, (per,)).ptr, per * 8))
    return out


def reorder_spin_blocks(H2):
    H2 = np.asarray(H2)
    if H2.shape[0] != 3:
        return H2
    ctx = get_ctx()
    return reorder_spin_blocks_dev(ctx, ctx.to_device(H2, np.float64)).get()


# ---------------------------------------------------------------------------------------------
# one-body folds (slater_helper.py:22-115)
# ---------------------------------------------------------------------------------------------

def transform_trans_inv(basis, lattice, H, symmetric=True):
    """sum_ij basis[i]^T H[i-j] basis[j]   (slater_helper.py:22-35).

    symmetric=True in the reference sums the cell pairs i<j and adds the transpose, which equals the full sum
    whenever the stripe is Hermitian, H[-R] = H[R]^T (density matrices, Fock, hcore): that case runs in k space,
    (1/nk) Re sum_k B_k^H H_k B_k.  A non-Hermitian stripe with symmetric=True depends on the cell ordering and
    is evaluated literally from the expanded matrix (strict upper block triangle), model sizes only."""
    ctx = get_ctx()
    basis = np.asarray(basis, dtype=np.float64)
    H = np.asarray(H, dtype=np.float64)
    if symmetric and max_abs(H - lattice.transpose(H)) > 1e-12 * max(1.0, max_abs(H)):
        nc, n = lattice.ncells, H.shape[-1]
        big = lattice.expand(H).reshape(nc, n, nc, n)
        mask = np.triu(np.ones((nc, nc)), 1)
        U_op = (big * mask[:, None, :, None]).reshape(nc * n, nc * n)
        D_op = (big * np.eye(nc)[:, None, :, None]).reshape(nc * n, nc * n)
        B = devmat.up(ctx, basis.reshape(nc * n, -1))
        quad = lambda op: devmat.mm(ctx, "T", B, "N", devmat.mm(ctx, "N", devmat.up(ctx, op), "N", B)).get()[0].real
        U = quad(U_op)
        return np.ascontiguousarray(quad(D_op) + U + U.T)
    return devmat.quad_trans_inv(ctx, lattice, basis, H)


def transform_trans_inv_k(basis_k, H_k):
    """(1/nk) Re sum_k basis_k[k]^H H_k[k] basis_k[k]; basis_k (nkpts, nlo, nbasis)   (slater_helper.py:37-50)."""
    res = devmat.quad_k(get_ctx(), np.asarray(basis_k), np.asarray(H_k))
    if max_abs(res.imag) > IMAG_DISCARD_TOL:
        log.warn("transform_trans_inv_k: has imag part %s", max_abs(res.imag))
    return np.ascontiguousarray(res.real)


def transform_local(basis, lattice, H):
    """sum_i basis[i]^T H basis[i], H (nscsites, nscsites)."""
    return devmat.quad_local(get_ctx(), np.asarray(basis, dtype=np.float64), np.asarray(H, dtype=np.float64))


def transform_imp(basis, lattice, H):
    return devmat.quad_local(get_ctx(), np.asarray(basis, dtype=np.float64)[:1], np.asarray(H, dtype=np.float64))


def transform_imp_env(basis, lattice, H):
    """0.5 (res + res^T), res = sum_i basis[i]^T H[i] basis[0]   (slater_helper.py:105-115)."""
    ctx = get_ctx()
    basis = np.asarray(basis, dtype=np.float64)
    B = devmat.up(ctx, basis)
    s = devmat.sum_batch(ctx, devmat.mm(ctx, "T", B, "N", devmat.up(ctx, np.asarray(H, dtype=np.float64))))
    res = devmat.mm(ctx, "N", s, "N", devmat.up(ctx, basis[0])).get()[0].real
    return 0.5 * (res + res.T)


def get_rdm1_idem(rdm1, nelec, beta):
    """Project a one-particle density matrix on its idempotent part through its natural orbitals (slater_helper.py:380-421):
    rdm1 (spin, nlo, nlo) or (spin, nkpts, nlo, nlo), largest occupancy 1; `nelec` counts all k points (a pair: per spin).

    The reference diagonalises every block, reverses the order and NEGATES the natural occupations so that mfd.assignocc fills
    from the largest one down (mu0 = -0.5).  Here all blocks of -rdm1 are diagonalised in one launch (dmk_eigh_batched: levels
    ascending = minus the occupations descending, the same ordered set), the occupations come from dmk_assign_occ and the
    density is rebuilt from the eigenvectors that never left the device (dmk_occ_density)."""
    from libdmet_preview_amd.routine import mfd
    rdm1 = np.asarray(rdm1)
    shape = rdm1.shape
    if rdm1.ndim not in (3, 4):
        raise ValueError("get_rdm1_idem: rdm1 of shape %s" % (shape,))
    spin, nlo = shape[0], shape[-1]
    nblk = int(np.prod(shape[1:-2]))
    ctx = get_ctx()
    d_A = ctx.to_device(np.ascontiguousarray(-rdm1.reshape(spin * nblk, nlo, nlo)), np.complex128)
    d_w, d_Vt = mfd.eigh_dev(ctx, d_A, nlo, spin * nblk)
    ew = d_w.get().reshape((spin, nlo) if rdm1.ndim == 3 else (spin, nblk, nlo))
    ewocc, mu, nerr = mfd.assignocc(ew, nelec, beta, mu0=-0.5)
    d_occ = ctx.to_device(np.ascontiguousarray(ewocc, dtype=np.float64).reshape(spin * nblk, nlo))
    out = mfd.density_dev(ctx, d_Vt, d_occ, nlo, spin * nblk).get().reshape(shape)
    return out if np.iscomplexobj(rdm1) else np.ascontiguousarray(out.real)


# ---------------------------------------------------------------------------------------------
# two-body, model lattices (slater_helper.py:126-156)
# ---------------------------------------------------------------------------------------------

def _quarter_chain_dev(ctx, d_v, n, mats, d_out, nb):
    """d_out (nb^4) += v[ijkl] m0[ip] m1[jq] m2[kr] m3[ls]: four real GEMMs, each contracts the leading index and
    appends the new one at the end, so the index order cycles back to pqrs."""
    dims = [n, n, n, n]
    cur = d_v
    for step, d_m in enumerate(mats):
        rest = int(np.prod(dims[1:]))
        last = step == 3
        nxt = d_out if last else ctx.zeros((rest, nb), np.float64)
        ctx.check(lib.dmk_dgemm_tn_acc_rect(ctx.h, rest, nb, dims[0], 1.0, cur.ptr, rest, d_m.ptr, nb, nxt.ptr, nb))
        dims = dims[1:] + [nb]
        cur = nxt
    return d_out


def transform_4idx(vijkl, ip, jq, kr, ls):
    """Transform an ERI with 1-fold symmetry: einsum('ijkl,ip,jq,kr,ls->pqrs') (all factors with the same column count)."""
    ctx = get_ctx()
    v = np.ascontiguousarray(vijkl, dtype=np.float64)
    n, nb = v.shape[0], np.asarray(ip).shape[1]
    mats = [ctx.to_device(np.ascontiguousarray(m, dtype=np.float64)) for m in (ip, jq, kr, ls)]
    d_out = ctx.zeros((nb, nb, nb, nb), np.float64)
    return _quarter_chain_dev(ctx, ctx.to_device(v), n, mats, d_out, nb).get()


def transform_eri_local(basis, lattice, H2):
    """Cell-local H2 ((spin,) nscsites^4) into the embedding space, summed over cells (interacting bath)."""
    basis = np.asarray(basis, dtype=np.float64)
    if basis.ndim == 3:
        basis = basis[None]
    spin, ncells, nscsites, nbasis = basis.shape
    H2 = np.asarray(H2, dtype=np.float64)
    if H2.ndim == 4:
        H2 = H2[None] if spin == 1 else np.asarray([H2, H2, H2])
    ctx = get_ctx()
    d_H2 = [ctx.to_device(np.ascontiguousarray(h)) for h in H2]
    d_b = [[ctx.to_device(np.ascontiguousarray(basis[s, i])) for i in range(ncells)] for s in range(spin)]
    nblk = spin * (spin + 1) // 2
    d_res = [ctx.zeros((nbasis,) * 4, np.float64) for _ in range(nblk)]
    for i in range(ncells):
        _quarter_chain_dev(ctx, d_H2[0], nscsites, [d_b[0][i]] * 4, d_res[0], nbasis)
        if spin == 2:
            _quarter_chain_dev(ctx, d_H2[1], nscsites, [d_b[1][i]] * 4, d_res[1], nbasis)
            _quarter_chain_dev(ctx, d_H2[2], nscsites, [d_b[0][i], d_b[0][i], d_b[1][i], d_b[1][i]], d_res[2], nbasis)
    return np.asarray([d.get() for d in d_res])
