"""
oracle/restate_cderi.py -- CPU restatement (numpy) of SURVEY.md section 8(f) rank 3, the on-disk DF tensor layout
(PySCF `cderi`) as the reference reads and writes it:

  get_mask_kptij_lst      basis_transform/eri_transform.py:1409-1427
  transform_gdf_to_lo     basis_transform/eri_transform.py:1312-1407   (the reference's own WRITER of the layout)
  sr_loop / _load3c       basis_transform/eri_transform.py:195-227     (reader: swap -> conjugate transpose,
                                                                        ki == kj -> Hermitian unpack of the packed triangle)

  convert_eri_to_gdf      basis_transform/eri_transform.py:1483-1535  + utils/cholesky.py:21-131 (modified Cholesky vectors of a
                                                                        molecular ERI as a Gamma-point container; golden G20)

`_load3c` is PySCF code (absent here); its behaviour is restated from the layout the reference's writer produces and
the unpack flag the reference passes (sr_loop: `unpack = is_zero(kpti - kptj) and not compact`).

TEST INFRASTRUCTURE ONLY.  Pinned against tests/golden/G13_cderi.npz (oracle/gen_golden.py gen_G13: the reference's
transform_gdf_to_lo run against a dict-backed stand-in for h5py.File, i.e. the datasets the reference itself writes).
"""
import numpy as np

from oracle.restate import KPT_DIFF_TOL, max_abs, pack_tril, round_to_FBZ, transform_ao_to_emb


def get_mask_kptij_lst(kptij_scaled, tol=KPT_DIFF_TOL):
    """kptij_scaled: (npairs, 2, 3) scaled k-points."""
    n = len(kptij_scaled)
    rnd = round_to_FBZ(np.asarray(kptij_scaled, dtype=float), tol=tol)
    mask = -np.ones(n, dtype=int)
    for i, ki in enumerate(rnd):
        if mask[i] == -1:
            for j in range(i + 1, n):
                s = ki + rnd[j]
                s = s - np.round(s)
                if max_abs(s) < tol:
                    mask[i] = j
                    mask[j] = -2
                    break
    return mask


def kptij_list(kpts):
    return np.asarray([(ki, kpts[j]) for i, ki in enumerate(kpts) for j in range(i + 1)])


def transform_gdf_to_lo(get_block, kpts_scaled, kpts_abs, naux, C_ao_lo, t_reversal_symm=True):
    """Returns the dict of datasets transform_gdf_to_lo writes (keys 'j3c-kptij', 'j3c/<k>/0')."""
    nk, nao, nlo = C_ao_lo.shape
    pairs = [(i, j) for i in range(nk) for j in range(i + 1)]
    kptij_abs = kptij_list(kpts_abs)
    kptij_scaled = kptij_list(kpts_scaled)
    mask = get_mask_kptij_lst(kptij_scaled) if t_reversal_symm else -np.ones(len(pairs), dtype=int)
    out = {"j3c-kptij": kptij_abs}
    for k, (i, j) in enumerate(pairs):
        if mask[k] == -2:
            continue
        Lpq = np.asarray(get_block(i, j), dtype=np.complex128).reshape(naux, nao * nao)
        Lij = transform_ao_to_emb(Lpq, C_ao_lo[None], i, j)[0]                       # (naux, nlo, nlo)
        gamma_pair = max_abs(kptij_scaled[k]) < KPT_DIFF_TOL
        if gamma_pair:
            data = pack_tril(Lij.real)
        elif i == j:
            data = pack_tril(Lij)
        else:
            data = Lij.reshape(naux, nlo * nlo)
        out["j3c/%d/0" % k] = data
        if mask[k] != -1:
            out["j3c/%d/0" % mask[k]] = data.conj()
    return out, mask


def load_block(feri, nk, nao, i, j):
    """What sr_loop(compact=False) yields for (kpti, kptj) from such a container."""
    pair_of = {}
    p = 0
    for a in range(nk):
        for b in range(a + 1):
            pair_of[(a, b)] = p
            p += 1
    swap = (i, j) not in pair_of
    L = np.asarray(feri["j3c/%d/0" % pair_of[(j, i) if swap else (i, j)]])
    if L.shape[1] == nao * (nao + 1) // 2 and nao > 1:
        il = np.tril_indices(nao)
        full = np.zeros((L.shape[0], nao, nao), dtype=np.complex128)
        full[:, il[1], il[0]] = L.conj()
        full[:, il[0], il[1]] = L
        L = full
    else:
        L = L.astype(np.complex128).reshape(-1, nao, nao)
    return L.conj().transpose(0, 2, 1) if swap else L


# ---- convert_eri_to_gdf (eri_transform.py:1483-1535) ------------------------------------------------------------------------------
def modified_cholesky(mat, max_error=1e-6):
    """utils/cholesky.py:21-52."""
    mat = np.asarray(mat)
    size = mat.shape[0]
    diag = np.diag(mat)
    idx = int(np.argmax(diag))
    delta_max = diag[idx]
    approx = np.zeros(size)
    vecs = [mat[idx] / delta_max ** 0.5]
    for i in range(size * 2 + 1):
        approx = approx + vecs[i] * vecs[i]
        delta = diag - approx
        idx = int(np.argmax(np.abs(delta)))
        delta_max = abs(delta[idx])
        R = np.zeros(size)
        for v in vecs:
            R = R + v[idx] * v
        vecs.append((mat[idx] - R) / delta_max ** 0.5)
        if delta_max < max_error:
            break
    return np.asarray(vecs)


def modified_cholesky_uhf(mat, max_error=1e-6):
    """utils/cholesky.py:54-105; mat = (aa, bb, ab)."""
    size = mat[0].shape[0]
    diag = np.hstack((np.diag(mat[0]), np.diag(mat[1])))
    idx = int(np.argmax(diag))
    delta_max = diag[idx]
    approx = np.zeros_like(diag)
    va, vb = [], []

    def rows(idx):
        if idx < size:
            return mat[0][idx], mat[2][idx]
        return mat[2].T[idx - size], mat[1][idx - size]

    ra, rb = rows(idx)
    va.append(ra / delta_max ** 0.5)
    vb.append(rb / delta_max ** 0.5)
    for i in range(size * 2 + 1):
        approx[:size] += va[i] * va[i]
        approx[size:] += vb[i] * vb[i]
        delta = diag - approx
        idx = int(np.argmax(np.abs(delta)))
        delta_max = abs(delta[idx])
        Ra, Rb = np.zeros(size), np.zeros(size)
        for a, b in zip(va, vb):
            c = a[idx] if idx < size else b[idx - size]
            Ra = Ra + c * a
            Rb = Rb + c * b
        ra, rb = rows(idx)
        va.append((ra - Ra) / delta_max ** 0.5)
        vb.append((rb - Rb) / delta_max ** 0.5)
        if delta_max < max_error:
            break
    return np.asarray([va, vb])


def unpack_tril_sym(tril, n):
    """pyscf.lib.unpack_tril (symmetric fill) of (..., npair) -> (..., n, n)."""
    tril = np.asarray(tril)
    out = np.zeros(tril.shape[:-1] + (n, n))
    il = np.tril_indices(n)
    out[..., il[0], il[1]] = tril
    out[..., il[1], il[0]] = tril
    return out


def convert_eri_to_gdf(eri_s4, norb, tol=1e-8):
    """eri_transform.py:1483-1535 for an ERI already restored to 4-fold symmetry: (npair, npair) or three such blocks (aa, bb, ab).
    Returns the nested dictionary of the fname=None branch."""
    eri_s4 = np.asarray(eri_s4)
    if eri_s4.ndim == 3:
        ev = modified_cholesky_uhf([eri_s4[0], eri_s4[1], eri_s4[2]], max_error=tol)
        cderi = unpack_tril_sym(ev, norb)
    else:
        cderi = unpack_tril_sym(modified_cholesky(eri_s4, max_error=tol), norb)
    return {"j3c": {"0": {"0": cderi}}, "j3c-kptij": np.zeros((1, 2, 3))}
