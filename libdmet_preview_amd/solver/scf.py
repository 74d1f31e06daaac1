"""
The ERI x density piece of libdmet/solver/scf.py on the MI355X:

  _get_jk     solver/scf.py:255-335   (RHF / UHF / UIHF; pyscf hf.dot_eri_dm -> dmk_jk_s4)
  _get_veff   solver/scf.py:337-352

The embedding ERI is streamed from HBM in its 4-fold packed form, once for J (both directions of the
alpha-beta block in the same pass) and once for K; 1-fold / 8-fold inputs are gathered into the 4-fold
form on the device first (dmk_eri_to_s4).  `jk_blocks_dev` is the device-resident form used by the
pipeline: it takes the three spin blocks by pointer, so the (aa, ab, bb) order of the transform and the
(aa, bb, ab) order of the Hamiltonian (routine/slater.py:461-462) need no 26 GB shuffle.
"""
import numpy as np

from libdmet_preview_amd._lib import lib, get_ctx
from libdmet_preview_amd.system import integral


def jk_dev(ctx, n, d_eri, d_dm_row=None, d_dm_col=None, d_dm_k=None, ld=None, row_ranges=None):
    """One 4-fold block: returns device (vj_row, vj_col, vk), None where the density was not given.  `row_ranges`
    ([(lo, hi)] of packed pair rows, lo a multiple of 32): the PARTIAL results of those rows of a row-sharded ERI -- the
    caller sums them over ranks."""
    npair = n * (n + 1) // 2
    out = [ctx.empty((n, n), np.float64) if d is not None else None for d in (d_dm_row, d_dm_col, d_dm_k)]
    p = lambda a: a.ptr if a is not None else None
    if row_ranges is None:
        ctx.check(lib.dmk_jk_s4(ctx.h, int(n), d_eri.ptr, int(ld or npair), p(d_dm_row), p(d_dm_col), p(d_dm_k),
                                p(out[0]), p(out[1]), p(out[2])))
    elif len(row_ranges) == 0:
        for o in out:
            if o is not None:
                o.zero_()
    else:
        rr = np.ascontiguousarray(row_ranges, dtype=np.int64).reshape(-1, 2)
        ctx.check(lib.dmk_jk_s4_rows(ctx.h, int(n), d_eri.ptr, int(ld or npair), len(rr), rr.ctypes.data, p(d_dm_row), p(d_dm_col),
                                     p(d_dm_k), p(out[0]), p(out[1]), p(out[2])))
    return tuple(out)


def jk_blocks_dev(ctx, n, d_aa, d_bb, d_ab, d_dm, with_j=True, with_k=True, row_ranges=None):
    """UIHF J/K from the three device blocks and d_dm (2, n, n): ((vj00, vj11), (vj01, vj10)), (vk00, vk11)."""
    dma, dmb = d_dm.offset(0, (n, n)), d_dm.offset(n * n, (n, n))
    vj00, _, vk00 = jk_dev(ctx, n, d_aa, dma if with_j else None, None, dma if with_k else None, row_ranges=row_ranges)
    vj11, _, vk11 = jk_dev(ctx, n, d_bb, dmb if with_j else None, None, dmb if with_k else None, row_ranges=row_ranges)
    vj01 = vj10 = None
    if with_j:
        vj01, vj10, _ = jk_dev(ctx, n, d_ab, dmb, dma, None, row_ranges=row_ranges)     # J a from b (rows), J b from a (columns)
    return ((vj00, vj11), (vj01, vj10)), (vk00, vk11)


def _block_to_dev_s4(ctx, blk, fmt, n):
    npair = n * (n + 1) // 2
    d = ctx.to_device(np.ascontiguousarray(blk, dtype=np.float64).reshape(-1))
    if fmt == 's4':
        return d
    out = ctx.empty((npair, npair), np.float64)
    ctx.check(lib.dmk_eri_to_s4(ctx.h, int(n), 1 if fmt == 's1' else 8, d.ptr, out.ptr))
    return out


def _get_jk(dm, eri, with_j=True, with_k=True):
    """
    J and K from rdm1 and ERI (RHF, UHF, UIHF).  J: ijkl,kl->ij   K: ijkl,il->jk.

    dm ((spin,) nao, nao); eri with or without spin dimension, s1 / s4 / s8.
    Returns vj (spin, nao, nao) [(2, 2, nao, nao) for UIHF] and vk (spin, nao, nao).
    """
    dm = np.asarray(dm, dtype=np.double)
    if dm.ndim == 2:
        dm = dm[np.newaxis]
    spin, nao = dm.shape[0], dm.shape[-1]
    eri = np.asarray(eri, dtype=np.double)
    eri_format, spin_dim = integral.get_eri_format(eri, nao)
    if spin_dim == 0:
        eri = eri[None]
        spin_dim = 1
    ctx = get_ctx()
    d_dm = ctx.to_device(dm)
    get = lambda a: a.get() if a is not None else None
    if spin == 1 or spin_dim == 1:
        d_E = _block_to_dev_s4(ctx, eri[0], eri_format, nao)
        vj, vk = [], []
        for s in range(spin):
            d = d_dm.offset(s * nao * nao, (nao, nao))
            j, _, k = jk_dev(ctx, nao, d_E, d if with_j else None, None, d if with_k else None)
            vj.append(get(j))
            vk.append(get(k))
        return (np.asarray(vj) if with_j else None), (np.asarray(vk) if with_k else None)
    elif spin_dim == 3:      # UIHF
        assert dm.shape[0] == 2
        blocks = [_block_to_dev_s4(ctx, eri[b], eri_format, nao) for b in range(3)]
        ((j00, j11), (j01, j10)), (k00, k11) = jk_blocks_dev(ctx, nao, blocks[0], blocks[1], blocks[2], d_dm,
                                                              with_j, with_k)
        # NOTE explicit write down vj, without broadcast (solver/scf.py:330-332)
        vj = np.asarray(((get(j00), get(j11)), (get(j01), get(j10)))) if with_j else None
        vk = np.asarray((get(k00), get(k11))) if with_k else None
        return vj, vk
    raise ValueError


def _get_veff(dm, eri):
    """HF effective potential (RHF: vj - vk/2 with a spin-traced dm; UHF: vj_a + vj_b - vk), (spin, nao, nao)."""
    dm = np.asarray(dm, dtype=np.double)
    if dm.ndim == 2:
        dm = dm[np.newaxis]
    spin = dm.shape[0]
    vj, vk = _get_jk(dm, eri)
    if spin == 1:
        veff = vj - vk * 0.5
    else:
        veff = vj[0] + vj[1] - vk
    return veff


def _get_veff_ghf(dm, eri):
    """HF effective potential of a generalised (spin-orbital) density against a spinless ERI, vj - vk (solver/scf.py:732-740)."""
    dm = np.asarray(dm, dtype=np.double)
    vj, vk = _get_jk(dm, eri)
    return vj[0] - vk[0]
