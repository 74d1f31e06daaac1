"""
Pins the oracle's AO->LO transforms (oracle/restate.py transform_h1_to_lo / transform_rdm1_to_lo / transform_rdm1_to_ao,
k2R) against the reference's Lattice.set_Ham / transform_obj_to_lo / update_Ham driven under the shim (golden G11;
SURVEY.md section 8f rank 4).  CPU only.
"""
import numpy as np
import pytest

from oracle import restate as R

NAMES = ["hcore", "ovlp", "fock", "fock_hf", "veff", "vhf", "rdm1"]


def ref_lo(g, name):
    """The restated chain: returns dicts of k- and R-space LO operators."""
    mesh = tuple(int(x) for x in g[name + "/mesh"])
    C, S, hcore, vhf, rdm1 = (g["%s/in_%s" % (name, k)] for k in ("C", "S", "hcore", "vhf", "rdm1"))
    spin = 1 if C.ndim == 3 else C.shape[0]
    ao = {"hcore": hcore, "ovlp": S, "fock": hcore + vhf, "fock_hf": hcore + vhf, "veff": vhf, "vhf": vhf}
    lo_k = {k: R.transform_h1_to_lo(v, C) for k, v in ao.items()}
    lo_k["rdm1"] = R.transform_rdm1_to_lo(rdm1, C, S)
    for k in lo_k:
        if k != "ovlp":
            lo_k[k] = R.add_spin_dim(lo_k[k], spin)
    lo_R = {k: R.k2R(v, mesh) for k, v in lo_k.items()}
    return mesh, C, S, lo_k, lo_R


@pytest.mark.parametrize("name", ["rhf", "uhf"])
def test_G11_transform_obj_to_lo(golden, name):
    g = golden("G11_setham.npz")
    mesh, C, S, lo_k, lo_R = ref_lo(g, name)
    for k in NAMES:
        assert np.abs(lo_k[k] - g["%s/%s_lo_k" % (name, k)]).max() < 1e-12, k
        assert np.abs(lo_R[k] - g["%s/%s_lo_R" % (name, k)].real).max() < 1e-12, k
    new_R = g[name + "/upd_rdm1_R"]
    rk = R.R2k(new_R, mesh)
    assert np.abs(R.transform_rdm1_to_ao(rk, C) - g[name + "/upd_rdm1_ao_k"]).max() < 1e-12
