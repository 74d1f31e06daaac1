"""
CPU checks of the duck-type contract golden G17 (oracle/gen_golden.py gen_G17: the attribute names the REFERENCE's entry points
read from the lattice / vcor / df / cell objects, recorded while the reference's own HartreeFock / ConstructImpHam / FitVcor ran)
and of the recorder that produced it (oracle/contract.py).  The comparison with what THIS package's entry points read happens in
tests/test_gpu_chain.py (the mirror entry points need the GPU).  reference: dmet/Hubbard.py:14-37, dmet/HubPhSymm.py:74-100,
SURVEY.md section 8b.
"""
import numpy as np

from oracle import contract


def test_recorder_skips_internal_reads_and_keeps_isinstance():
    class Thing(object):
        def __init__(self):
            self.a, self.b = 1, 2

        def twice_a(self):
            return 2 * self.a                     # internal read: not part of the external contract

    t = Thing()
    log = contract.Log()
    contract.watch(t, log.stage("s1"), "thing")
    assert isinstance(t, Thing)
    assert t.twice_a() == 2 and t.b == 2
    assert log.names("thing") == {"twice_a", "b"}
    log.stage("s2")
    _ = t.a
    assert log.names("thing", "s2") == {"a"} and log.names("thing", "s1") == {"twice_a", "b"}
    assert log.stages() == ["s1", "s2"] and log.kinds() == ["thing"]
    assert "a" in contract.offered(t) and "twice_a" in contract.offered(t)


def test_G17_golden_is_self_consistent(golden):
    g = golden("G17_contract.npz")
    offered = {k.split("/", 1)[1]: set(str(x) for x in g[k]) for k in g.files if k.startswith("offered/")}
    assert set(offered) == {"lattice", "vcor", "df", "cell"}
    stages = {k.split("/")[1] for k in g.files if k.startswith("read/")}
    assert {"HartreeFock", "ConstructImpHam_ib", "ConstructImpHam_nib", "get_emb_eri_fast_gdf", "FitVcor"} <= stages
    for k in g.files:
        if k.startswith("read/"):
            kind = k.split("/")[2]
            names = {str(x) for x in g[k] if not (str(x).startswith("__") and str(x).endswith("__"))}
            assert names <= offered[kind], (k, sorted(names - offered[kind]))
    # the attributes SURVEY.md section 8b names are the ones the reference's entry points read
    lat = set()
    for k in g.files:
        if k.startswith("read/") and k.endswith("/lattice"):
            lat |= {str(x) for x in g[k]}
    for name in ("ncells", "nscsites", "imp_idx", "val_idx", "is_model", "getFock", "getH1", "get_ovlp", "R2k_basis", "FFTtoT"):
        assert name in lat, name


def test_mirror_objects_offer_what_the_reference_entry_points_read(golden):
    """The other direction of the drop-in: a REFERENCE entry point handed THIS package's Lattice / Vcor finds every attribute
    it reads (host-side objects: no GPU needed to ask hasattr)."""
    import importlib
    import os
    if not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libdmet_preview_amd", "libdmetk.so")):
        import pytest
        pytest.skip("libdmetk.so not built")
    lattice = importlib.import_module("libdmet_preview_amd.system.lattice")
    Hubbard = importlib.import_module("libdmet_preview_amd.dmet.Hubbard")
    g = golden("G17_contract.npz")
    L = lattice.Lattice(6, (2, 2, 1))
    L.val_idx, L.virt_idx, L.core_idx = [1, 2, 3], [4, 5], [0]
    for a in ("fock_lo_k", "fock_lo_R", "hcore_lo_k", "hcore_lo_R", "vhf_lo_k", "ovlp_lo_k", "JK_imp", "Ham", "JK_core", "cell", "df",
              "C_ao_lo", "eri_symmetry", "rdm1_lo_k", "rdm1_lo_R", "H0"):
        setattr(L, a, None)                       # what set_Ham leaves on an ab-initio lattice
    L.is_model, L.use_hcore_as_emb_ham = False, False
    vc = Hubbard.VcorLocal(False, False, 6, idx_range=[1, 2, 3])
    for kind, obj in (("lattice", L), ("vcor", vc)):
        need = set()
        for k in g.files:
            if k.startswith("read/") and k.endswith("/" + kind):
                need |= {str(x) for x in g[k] if not (str(x).startswith("__") and str(x).endswith("__"))}
        lacking = sorted(n for n in need if not hasattr(obj, n))
        assert not lacking, "%s: the reference's entry points read %s, which this package's object does not offer" % (kind, lacking)
