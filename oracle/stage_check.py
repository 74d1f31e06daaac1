"""
TEST INFRASTRUCTURE (never imported by the product path): full-size stage checker of the mean-field -> bath ->
C_ao_emb chain against the numpy restatement of the reference (oracle/restate.py), on the same seeded inputs.

Used by bench.py's parity leg and by tests/test_gpu_production.py::test_c5_meanfield_bath_full_size, so that the
chain is compared with an INDEPENDENT computation at the timed size (C5: 432 x eigh(200), 86 400 occupations, rho_R,
two 43144 x 56 SVDs) and not only at toy sizes -- bench.py's ERI self-check feeds the oracle the device's own
C_ao_emb, which would hide an error upstream of it.

What is compared and how (reference file:line of the stage in parentheses):
  ew           max |device - oracle| over all spin*nk*nlo eigenvalues          (routine/mfd.py:33-108)
  occupations  exact array equality at T = 0, max-abs at finite T; mu           (routine/mfd.py:887-957)
  rho_R        max-abs                                                          (routine/mfd.py:352-360)
  bath         equal number of bath orbitals per spin and the gauge-invariant projector distance
               ||B1 B1^T - B2 B2^T||_F = sqrt(||(1 - P2) B1||_F^2 + ||(1 - P1) B2||_F^2) for orthonormal B1, B2 --
               never raw vectors; the (ncells*nlo)^2 projector is not formed
                                                                                (routine/slater.py:117-220)
  C_ao_emb     max-abs against the oracle's C_ao_lo . R2k(basis) / nk^(3/4) evaluated ON THE DEVICE'S basis (the two
               bases differ by a rotation inside degenerate singular subspaces)  (eri_transform.py:118-126, 289-292)
"""
import time
import numpy as np

from oracle import restate as R

TOL = {"ew": 1e-10, "mu": 1e-10, "occ": 1e-10, "rho": 1e-10, "bath_frob": 1e-10, "c_ao_emb": 1e-12}


def projector_distance(B1, B2):
    """||B1 B1^T - B2 B2^T||_F for two (nrow, ncol) bases with orthonormal columns, without forming a projector."""
    B1 = B1.reshape(-1, B1.shape[-1])
    B2 = B2.reshape(-1, B2.shape[-1])
    r1 = B1 - B2 @ (B2.T @ B1)
    r2 = B2 - B1 @ (B1.T @ B2)
    return float(np.sqrt(np.linalg.norm(r1) ** 2 + np.linalg.norm(r2) ** 2))


def c_ao_emb_reference(mesh, C_ao_lo, basis):
    """eri_transform.py:118-126 (get_basis_k: basis_k[k] = sum_R basis[R] e^{-i k.R}) and :289-292
    (C_ao_emb = C_ao_lo . basis_k / nk^(3/4)).  Same contraction as restate.get_basis_k, written as ONE matrix product per spin
    (phase^T (nk x ncells) times basis (ncells x nlo*nemb)) instead of its einsum: 34 s -> 2 s at C5."""
    ks = R.make_kpts_scaled(mesh)
    phase = R.get_phase_R2k(mesh, ks)                       # (R, k)
    C_ao_lo = np.asarray(C_ao_lo)
    basis = np.asarray(basis)
    if C_ao_lo.ndim == 3:
        C_ao_lo = C_ao_lo[None]
    spin, nk, nlo, nemb = basis.shape
    out = np.empty((spin, nk, C_ao_lo.shape[-2], nemb), dtype=np.complex128)
    for s in range(spin):
        bk = (phase.T @ basis[s].reshape(nk, nlo * nemb)).reshape(nk, nlo, nemb)
        Cs = C_ao_lo[min(s, C_ao_lo.shape[0] - 1)]
        for k in range(nk):
            out[s, k] = Cs[k] @ bk[k]
    return out / (nk ** 0.75)


def reference_chain(mesh, Fock_R, vcor, filling, restricted, imp_idx, val_idx, beta=np.inf):
    """The oracle's own mean field and bath from the seeded real-space Fock operator."""
    nlo = Fock_R.shape[-1]
    Fk = R.R2k(Fock_R, mesh)
    rhoT, mu, E, res = R.HF(mesh, Fk, Fock_R, Fock_R, vcor, filling, restricted, beta=beta, ires=True)
    basis, info = R.get_emb_basis(mesh, nlo, rhoT, imp_idx=imp_idx, val_idx=val_idx, return_info=True)
    return {"ew": res["e"], "occ": res["mo_occ"], "mu": mu, "rho_R": rhoT, "basis": basis, "nbath_s": info["nbath_s"],
            "sigma": info["sigma"]}


def compare(mesh, Fock_R, vcor, filling, restricted, imp_idx, val_idx, C_ao_lo, got, beta=np.inf):
    """`got`: host copies of the device results -- ew (spin, nk, nlo), occ (same), mu, rho_R (spin, nk, nlo, nlo),
    basis (spin, nk, nlo, nemb), sigma (list per spin), C_ao_emb (spin, nk, nao, nemb).  Returns a flat dict of
    errors plus `ok` (every entry within TOL) and the seconds the oracle took."""
    t0 = time.perf_counter()
    ref = reference_chain(mesh, Fock_R, vcor, filling, restricted, imp_idx, val_idx, beta)
    spin = ref["ew"].shape[0]
    nlo = Fock_R.shape[-1]
    out = {}
    out["parity_ew_maxabs"] = float(np.abs(np.asarray(got["ew"]).reshape(ref["ew"].shape) - ref["ew"]).max())
    occ = np.asarray(got["occ"]).reshape(ref["occ"].shape)
    out["parity_occ_equal"] = bool(np.array_equal(occ, ref["occ"]))
    out["parity_occ_maxabs"] = float(np.abs(occ - ref["occ"]).max())
    out["parity_mu_abs"] = float(np.abs(np.asarray(got["mu"], dtype=float) - np.asarray(ref["mu"], dtype=float)).max())
    out["parity_rho_maxabs"] = float(np.abs(np.asarray(got["rho_R"]).reshape(ref["rho_R"].shape) - ref["rho_R"]).max())
    basis = np.asarray(got["basis"])
    nemb = basis.shape[-1]
    nbath_dev = [int((np.asarray(s) >= 1e-9).sum()) for s in got["sigma"]]
    out["parity_nbath"] = [nbath_dev, [int(x) for x in ref["nbath_s"]]]
    out["parity_nbath_equal"] = bool(nbath_dev == [int(x) for x in ref["nbath_s"]] and basis.shape == ref["basis"].shape)
    out["parity_sigma_maxabs"] = float(max(np.abs(np.asarray(a) - np.asarray(b)[:len(a)]).max() for a, b in zip(got["sigma"], ref["sigma"])))
    if out["parity_nbath_equal"]:
        out["parity_bath_frob"] = max(projector_distance(basis[s], ref["basis"][s]) for s in range(spin))
        B = basis.reshape(spin, -1, nemb)
        out["parity_basis_orth"] = float(max(np.abs(B[s].T @ B[s] - np.eye(nemb)).max() for s in range(spin)))
    else:
        out["parity_bath_frob"] = 1e300
        out["parity_basis_orth"] = 1e300
    Cref = c_ao_emb_reference(mesh, C_ao_lo, basis)
    out["parity_c_ao_emb_maxabs"] = float(np.abs(np.asarray(got["C_ao_emb"]).reshape(Cref.shape) - Cref).max())
    zero_t = not (beta < np.inf)
    out["parity_stages_ok"] = bool(
        out["parity_ew_maxabs"] <= TOL["ew"] and out["parity_mu_abs"] <= TOL["mu"]
        and (out["parity_occ_equal"] if zero_t else out["parity_occ_maxabs"] <= TOL["occ"])
        and out["parity_rho_maxabs"] <= TOL["rho"] and out["parity_nbath_equal"]
        and out["parity_bath_frob"] <= TOL["bath_frob"] and out["parity_basis_orth"] <= 1e-12
        and out["parity_c_ao_emb_maxabs"] <= TOL["c_ao_emb"])
    out["parity_stages_seconds"] = round(time.perf_counter() - t0, 2)
    out["parity_stages_tol"] = dict(TOL)
    return out
