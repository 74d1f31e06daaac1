"""
Host containers at the exit / entry of the path: routine.vcor.Vcor, system.integral.Integral / get_eri_format and the
utils.misc helpers.  Their bodies are this package's own; the CONTRACT is the reference's (routine/vcor.py:19-103,
system/integral.py:60-128, 883-927, utils/misc.py:34-86).  Stand-alone semantic checks, and -- when the reference tree is
present in this container (never on the GPU box) -- a side-by-side run against the reference's own classes under the import
shim on the same inputs.
"""
import itertools as it
import os

import numpy as np
import pytest

from libdmet_preview_amd.routine import vcor as avcor
from libdmet_preview_amd.system import integral as aint
from libdmet_preview_amd.utils import misc as amisc

HAVE_REF = os.path.isdir("/root/reference/libdmet")


class _Local(avcor.Vcor):
    """two-parameter symmetric potential on a 2 x 2 block, one spin channel"""
    def __init__(self):
        avcor.Vcor.__init__(self)
        self.grad = np.zeros((2, 1, 2, 2))
        self.grad[0, 0, 0, 0] = self.grad[0, 0, 1, 1] = 1.0
        self.grad[1, 0, 0, 1] = self.grad[1, 0, 1, 0] = 1.0

    def evaluate(self):
        return np.tensordot(self.param, self.grad, axes=(0, 0))

    def gradient(self):
        return self.grad

    def length(self):
        return 2


def test_vcor_local_contract():
    v = _Local()
    with pytest.raises(Exception):
        v.get()                                            # not initialised yet
    v.update(np.array([0.5, -0.25]))
    assert v.is_local() and v.islocal() and not v.per_k()
    assert np.array_equal(v.get(3, kspace=True), v.value) and np.array_equal(v.get(0, kspace=False), v.value)
    assert not v.get(2, kspace=False).any()                # a local potential lives in cell 0 only
    v.assign(np.array([[[1.0, 2.0], [2.0, 1.0]]]))
    assert np.allclose(v.param, [1.0, 2.0])
    from libdmet_preview_amd.utils import logger as log
    n0 = len(log.warnings_seen)
    v.assign(np.array([[[1.0, 2.0], [4.0, 3.0]]]))         # not in the parametrised space: projected, with a warning
    assert np.allclose(v.param, [2.0, 3.0]) and len(log.warnings_seen) == n0 + 1
    with pytest.raises(Exception):
        v.assign(np.zeros((1, 3, 3)))


def test_vcor_per_k_get():
    v = avcor.Vcor()
    v.value = np.arange(2 * 1 * 2 * 2, dtype=float).reshape(2, 1, 2, 2)
    assert v.per_k() and np.array_equal(v.get(1), v.value[1]) and np.array_equal(v.get(1, kspace=False), v.value[1])


def test_integral_and_eri_format():
    n = 3
    H1 = np.zeros((2, n, n))
    H2 = np.zeros((3, 6, 6))
    I = aint.Integral(n, False, False, 0.5, H1, H2)
    assert I.H1["cd"] is H1 and I.H2["ccdd"] is H2 and np.array_equal(I.ovlp, np.eye(n)) and I.H0 == 0.5
    assert I.pairNoSymm() == list(it.product(range(n), repeat=2))
    assert I.pairSymm() == list(it.combinations_with_replacement(range(n)[::-1], 2))[::-1]
    assert I.pairAntiSymm() == list(it.combinations(range(n)[::-1], 2))[::-1]
    with pytest.raises(Exception):
        aint.Integral(n, False, False, 0.0, np.zeros((2, n, n + 1)), H2)
    with pytest.raises(Exception):
        aint.Integral(n, False, False, 0.0, H1, np.zeros((6, 6, 6, 6)))
    npair, n8 = 6, 21
    cases = [((1, n, n, n, n), ("s1", 1)), ((3, n, n, n, n), ("s1", 3)), ((n, n, n, n), ("s1", 0)), ((3, npair, npair), ("s4", 3)),
             ((1, npair, npair), ("s4", 1)), ((npair, npair), ("s4", 0)), ((1, n8), ("s8", 1)), ((n8,), ("s8", 0))]
    for shape, want in cases:
        assert aint.get_eri_format(np.zeros(shape), n) == want, shape
    for bad in [(7,), (5, 5), (2, 5, 5), (2, n, n, n)]:
        with pytest.raises(Exception):
            aint.get_eri_format(np.zeros(bad), n)


def test_misc_helpers():
    assert amisc.max_abs(np.array([])) == 0.0 and amisc.max_abs(np.array([-3.0, 2.0])) == 3.0
    assert amisc.max_abs(np.array([1 + 1j, 0.5])) == pytest.approx(np.sqrt(2.0))
    a, b, c = np.arange(4.0).reshape(2, 2), np.eye(2) * 2, np.ones((2, 1))
    assert np.array_equal(amisc.mdot(a, b, c), a @ b @ c) and np.array_equal(amisc.mdot(a), a)
    assert amisc.get_spin_dim([np.zeros((4, 2, 2)), np.zeros((2, 4, 2, 2))]) == 2 and amisc.get_spin_dim([np.zeros((4, 2, 2))]) == 1
    with pytest.raises(ValueError):
        amisc.get_spin_dim([np.zeros((2, 2))])
    H = np.arange(8.0).reshape(2, 2, 2)
    out = amisc.add_spin_dim(H, 2)
    assert out.shape == (2, 2, 2, 2) and np.array_equal(out[0], H) and np.array_equal(out[1], H)
    assert amisc.add_spin_dim(out, 1) is out or np.array_equal(amisc.add_spin_dim(out, 1), out)
    assert amisc.add_spin_dim(np.zeros((1, 3, 2, 2)), 2).shape == (2, 3, 2, 2)


@pytest.mark.skipif(not HAVE_REF, reason="reference tree not present (GPU box)")
def test_against_the_reference_classes():
    from oracle import shim
    shim.install()
    shim.quiet()
    from libdmet.routine import vcor as rvcor
    from libdmet.system import integral as rint
    from libdmet.utils import misc as rmisc
    rng = np.random.default_rng(0)
    # Vcor.assign on the same parametrisation
    class RLocal(rvcor.Vcor):
        def __init__(self):
            rvcor.Vcor.__init__(self)
            self.grad = _Local().grad
        evaluate, gradient, length = _Local.evaluate, _Local.gradient, _Local.length
    v0 = rng.standard_normal((1, 2, 2))
    a, r = _Local(), RLocal()
    a.assign(v0)
    r.assign(v0)
    assert np.allclose(a.param, r.param, atol=1e-15) and np.allclose(a.get(), r.get(), atol=1e-15)
    assert np.array_equal(a.get(1, kspace=False), r.get(1, kspace=False))
    # get_eri_format / Integral pair lists / misc helpers on random shapes
    n = 4
    npair = n * (n + 1) // 2
    for shape in [(1, n, n, n, n), (3, n, n, n, n), (n, n, n, n), (3, npair, npair), (npair, npair), (1, npair * (npair + 1) // 2),
                  (npair * (npair + 1) // 2,)]:
        assert aint.get_eri_format(np.zeros(shape), n) == rint.get_eri_format(np.zeros(shape), n)
    Ia = aint.Integral(n, True, False, 0.0, np.zeros((1, n, n)), np.zeros((1, npair, npair)))
    Ir = rint.Integral(n, True, False, 0.0, np.zeros((1, n, n)), np.zeros((1, npair, npair)))
    assert Ia.pairNoSymm() == Ir.pairNoSymm() and Ia.pairSymm() == Ir.pairSymm() and Ia.pairAntiSymm() == Ir.pairAntiSymm()
    x = rng.standard_normal((3, 5)) + 1j * rng.standard_normal((3, 5))
    assert amisc.max_abs(x) == pytest.approx(rmisc.max_abs(x)) and amisc.max_abs(x.real) == pytest.approx(rmisc.max_abs(x.real))
    mats = [rng.standard_normal((3, 3)) for _ in range(4)]
    assert np.allclose(amisc.mdot(*mats), rmisc.mdot(*mats))
    H = rng.standard_normal((5, 2, 2))
    assert np.array_equal(amisc.add_spin_dim(H, 2), rmisc.add_spin_dim(H, 2))
    assert amisc.get_spin_dim([H, H[None]]) == rmisc.get_spin_dim([H, H[None]])


def test_orthonormal_completion_of_null_bath_columns():
    """completing vanishing singular directions (routine/bcs.py:46, 84 keep LAPACK's completion) must work when no
    unit vector has a residual above any fixed threshold -- nenv = nb = 4 with the kept columns spanning everything but
    (.5, .5, .5, .5) -- and for several null columns in a row, and must refuse instead of indexing past the rows."""
    import numpy as np
    import pytest
    from libdmet_preview_amd.routine.slater import orthonormal_completion
    h = np.full(4, 0.5)
    Q, _ = np.linalg.qr(np.concatenate([h[:, None], np.random.default_rng(5).standard_normal((4, 3))], axis=1))
    U = np.zeros((4, 4))
    U[:, :3] = Q[:, 1:]                                  # three kept columns, all orthogonal to h
    orthonormal_completion(U, [0, 1, 2], [3])
    assert np.abs(U.T @ U - np.eye(4)).max() < 1e-14 and abs(abs(U[:, 3] @ h) - 1.0) < 1e-14
    for nenv, nkeep, nnull in [(4, 2, 2), (7, 3, 4), (5, 0, 5), (6, 6, 0)]:
        V = np.zeros((nenv, nkeep + nnull))
        V[:, :nkeep] = np.linalg.qr(np.random.default_rng(nenv).standard_normal((nenv, max(nkeep, 1))))[0][:, :nkeep]
        orthonormal_completion(V, list(range(nkeep)), list(range(nkeep, nkeep + nnull)))
        assert np.abs(V.T @ V - np.eye(nkeep + nnull)).max() < 1e-13
    with pytest.raises(ValueError):
        orthonormal_completion(np.zeros((3, 4)), [0, 1], [2, 3])
