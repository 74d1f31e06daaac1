"""
CPU tests of the model lattices and the 1-band Hubbard Hamiltonian (reference system/lattice.py:796-1109 LatticeModel / UnitCell /
SuperCell / translateSites / ChainLattice / SquareLattice / CubicLattice, system/hamiltonian.py:18-165 HamNonInt /
HubbardHamiltonian) against golden G38, captured from the reference's own classes and neighbour search.
"""
import numpy as np
import pytest

MODEL_LATTICES = [("chain12_2", "chain", (12, 2)), ("chain8_4", "chain", (8, 4)), ("sq44_22", "square", (4, 4, 2, 2)), ("sq62_21", "square", (6, 2, 2, 1)),
                  ("cub442_221", "cubic", (4, 4, 2, 2, 2, 1)), ("afm42_21", "afm", (4, 2, 2, 1)), ("band3_42_21", "band3", (4, 2, 2, 1))]


def build(kind, args):
    from libdmet_preview_amd.system import lattice
    return {"chain": lattice.ChainLattice, "square": lattice.SquareLattice, "cubic": lattice.CubicLattice, "afm": lattice.SquareAFM,
            "band3": lattice.Square3Band}[kind](*args)


@pytest.mark.parametrize("name,kind,args", MODEL_LATTICES, ids=[m[0] for m in MODEL_LATTICES])
def test_geometry_neighbours_and_hubbard_hamiltonian(golden, name, kind, args):
    from libdmet_preview_amd.system import hamiltonian
    g = golden("G38_model_lattices.npz")
    L = build(kind, args)
    assert np.abs(np.asarray(L.sites) - g[name + "/sites"]).max() < 1e-15 and np.array_equal(np.asarray(L.cells), g[name + "/cells"])
    assert np.array_equal(L.size, g[name + "/size"]) and L.is_model and L.nao == L.supercell.nsites and L.val_idx == list(range(L.nao))
    assert L.ncells == len(g[name + "/cells"]) and L.nkpts == L.ncells and list(L.kmesh[:L.dim]) == list(L.csize)
    for dtag, dis in (("d1", L.neighborDist[0]), ("d2", L.neighborDist[1])):
        assert np.array_equal(np.asarray(sorted(L.neighbor(dis=dis, sitesA=range(L.nscsites)))), g["%s/nb_%s" % (name, dtag)])
        obc = np.asarray(sorted(L.neighbor(dis=dis, sitesA=range(L.nscsites), search_range=0))).reshape(-1, 2)
        assert np.array_equal(obc, g["%s/nb_%s_obc" % (name, dtag)])
    assert np.array_equal(np.asarray(sorted(L.neighbor(dis=L.neighborDist[0]))), g[name + "/nb_all"])
    for htag, kw in (("t", dict()), ("tt", dict(tlist=[1.0, -0.25])), ("ttt", dict(tlist=[1.0, 0.0, 0.1])), ("obc", dict(obc=True))):
        H = hamiltonian.HubbardHamiltonian(L, 4.0, **kw)
        assert np.array_equal(H.getH1(), g["%s/H1_%s" % (name, htag)]), htag
        assert H.getFock() is H.getH1() and H.getImpJK() is None and H.getH0() == 0.0
    H = hamiltonian.HubbardHamiltonian(L, 6.0, compact=True)
    assert np.array_equal(H.getH2(), g[name + "/H2_compact"]) and H.H2_format == str(g[name + "/H2_format"]) == "local"
    assert np.array_equal(hamiltonian.HubbardHamiltonian(L, 6.0).getH2(), g[name + "/H2_full"])
    assert np.array_equal(hamiltonian.HubbardHamiltonian(L, 4.0, return_H1=True), g[name + "/H1_t"])
    # cell arithmetic of the model's own dimension agrees with the padded mesh tables of the base class
    for i in range(L.ncells):
        assert L.add(i, L.neg(i)) == 0 and L.subtract(i, i) == 0
    # the site index round trip
    for idx in (0, L.nsites - 1):
        assert L.site_pos2idx(L.site_idx2pos(idx)) == idx


def test_ham_non_int_layouts_and_errors():
    from libdmet_preview_amd.system import lattice, hamiltonian
    L = lattice.ChainLattice(6, 2)
    n, nc = L.nao, L.ncells
    H1 = np.zeros((nc, n, n))
    for shape, spin_dim, fmt in (((n,) * 4, None, "local"), ((3, 3), None, "local"), ((nc,) + (n,) * 4, None, "nearest"),
                                 ((nc,) * 3 + (n,) * 4, None, "full"), ((3,) + (n,) * 4, 3, "spin local"), ((3, nc, 3, 3), 3, "spin nearest"),
                                 ((3,) + (nc,) * 3 + (3, 3), 3, "spin full")):
        assert hamiltonian.HamNonInt(L, H1, np.zeros(shape), spin_dim_H2=spin_dim).H2_format == fmt
    with pytest.raises(ValueError):
        hamiltonian.HamNonInt(L, H1, np.zeros((5, 5)))
    with pytest.raises(Exception):
        hamiltonian.HamNonInt(L, np.zeros((nc + 1, n, n)), np.zeros((n,) * 4))
    with pytest.raises(Exception):
        lattice.ChainLattice(7, 2)
    with pytest.raises(Exception):
        hamiltonian.HubbardHamiltonian(L, 4.0, tlist=[1.0, 0.1, 0.1, 0.1])          # more hopping ranges than neighbour distances
