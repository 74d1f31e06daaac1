"""
ERI container helpers of libdmet/routine/slater_helper.py that sit on the hot path's exit:

  init_H2             slater_helper.py:444-471
  unit2emb            slater_helper.py:494-528   zero-padded copy of the unit-cell ERI into the embedding ERI
  reorder_spin_blocks routine/slater.py:461-462  (aa, ab, bb) -> (aa, bb, ab)

The in-core 4-fold and 8-fold forms are padded on the device (dmk_pad_block_f64); `unit2emb_dev` keeps
the ERI in HBM for callers that stay on the device.
"""
import numpy as np

from libdmet_preview_amd._lib import lib, get_ctx


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
