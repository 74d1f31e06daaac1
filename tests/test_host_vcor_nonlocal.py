"""
CPU tests of the cell-resolved correlation potential vcor.VcorNonLocal (reference routine/vcor.py:105-524): the index table of
the product class against the reference's own closures (golden G23, bit-exact: value, the non-zeros of gradient(), the
least-squares projection of assign()).  The parts that need the device (the Fourier transform inside update(), dV/dparam, the
fit) are in tests/test_gpu_vcor_nonlocal.py.
"""
import copy

import numpy as np
import pytest

from tests.test_oracle_fit import NONLOCAL_LATTICES, NONLOCAL_MODES


@pytest.mark.parametrize("lat", NONLOCAL_LATTICES, ids=[x[0] for x in NONLOCAL_LATTICES])
@pytest.mark.parametrize("mode", NONLOCAL_MODES, ids=[x[0] for x in NONLOCAL_MODES])
def test_index_table_vs_reference(golden, lat, mode):
    from libdmet_preview_amd.system.lattice import Lattice
    from libdmet_preview_amd.dmet import Hubbard
    g = golden("G23_vcor_nonlocal.npz")
    (lname, mesh, nlo, idx), (mname, res, bogo, bres) = lat, mode
    key = "tab/%s/%s" % (lname, mname)
    v = Hubbard.VcorNonLocal(res, bogo, Lattice(nlo, mesh), idx_range=idx, bogo_res=bres)
    assert not v.is_local() and not v.islocal() and not v.is_vcor_kpts
    p = g[key + "/param"]
    assert v.length() == len(p)
    v.param = p
    assert np.array_equal(v.evaluate(), g[key + "/value"])
    assert np.array_equal(np.asarray(v.cell_entries()), g[key + "/grad_nz"])
    assert (v.nparam, v.nblk, v.ncells, nlo, nlo) == tuple(g[key + "/grad_shape"])
    assert np.array_equal(v.project(g[key + "/assign_in"]), g[key + "/assign_param"])
    # every parameter owns at least one entry, every entry one parameter
    P = v.cell_entries()[0]
    assert np.array_equal(np.unique(P), np.arange(v.length()))
    flat = np.ravel_multi_index(v.cell_entries()[1:], (v.nblk, v.ncells, nlo, nlo))
    assert len(np.unique(flat)) == len(flat)


def test_copy_shares_the_lattice_and_table():
    from libdmet_preview_amd.system.lattice import Lattice
    from libdmet_preview_amd.routine import vcor
    L = Lattice(3, (2, 2, 1))
    v = vcor.VcorNonLocal(True, False, L)
    v.param = np.arange(v.length(), dtype=float)
    v.value = v.evaluate()
    w = copy.deepcopy(v)
    assert w.lattice is L and w._tab is v._tab
    w.param[0] = 99.0
    assert v.param[0] == 0.0
    with pytest.raises(Exception):
        v.project(np.zeros((2, 4, 3, 3)))
