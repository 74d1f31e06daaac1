"""
oracle/restate_fit.py -- CPU restatement (numpy/scipy) of SURVEY.md section 8(f) rank 2: the correlation-potential
least-squares fit in the embedding space,

  Hubbard.VcorLocal            dmet/Hubbard.py:551-786     local vcor parametrisation (evaluate / gradient / diag_indices)
  slater.get_dV_dparam         routine/slater.py:851-907   (local branch, transform_local_sparseH slater_helper.py:91-100; non-local
                                                           branch :893-902 with vcor.VcorNonLocal, routine/vcor.py:105-524, golden G23)
  slater.FitVcorEmb            routine/slater.py:909-1329  errfunc, gradfunc (T = 0), gradfunc_ft (finite T)
  ftsystem.get_dw_dv           routine/ftsystem.py:151-213

The optimiser driver (routine/fit.py + fit_helper.py: Polak-Ribiere CG with a bounded scalar line search) is host
control flow; it is pinned on the product side directly against iterates captured from the reference
(tests/test_host_fit.py).

Round 6: the FitVcorEmb options idem_fit (slater_helper.py:380-421 get_rdm1_idem), C_act (slater.py:1083-1088, 1112-1124),
P_act (slater.py:878-892 with get_active_projector_full :2195-2219, transform_trans_inv_k slater_helper.py:37-50) and
return_drho_dparam (slater.py:1227-1261 over ftsystem.get_rho_grad ftsystem.py:147-221).

TEST INFRASTRUCTURE ONLY.  Pinned against the reference through tests/golden/G9_vcorfit.npz
(oracle/gen_golden.py gen_G9: the reference's own FitVcorEmb closures captured at fixed parameter vectors) and, for the
round-6 options, tests/golden/G21_fit_options.npz (gen_G21).
"""
import itertools as it
from math import sqrt

import numpy as np
import scipy.linalg as la

from oracle.restate import R2k, assignocc
from oracle.restate_ham import transform_h1

ZERO_TOL = 1e-10


# ---------------------------------------------------------------------------------------------
# VcorLocal (dmet/Hubbard.py:551-786)
# ---------------------------------------------------------------------------------------------

def triu_diag_indices(n):
    return np.cumsum([0] + list(range(n, 1, -1)))


class VcorLocal(object):
    """Local correlation potential: symmetric (spin, nscsites, nscsites) blocks on idx_range, optional pairing block."""

    def __init__(self, restricted, bogoliubov, nscsites, idx_range=None, bogo_res=False, ghf=False):
        self.restricted, self.bogoliubov, self.bogo_res, self.ghf = restricted, bogoliubov, bogo_res, ghf
        self.nscsites = nscsites
        self.idx_range = list(range(nscsites)) if idx_range is None else list(idx_range)
        nidx = len(self.idx_range)
        ntri = nidx * (nidx + 1) // 2
        self.nV = ntri if restricted else 2 * ntri
        if not bogoliubov:
            self.nD = 0
        elif restricted or bogo_res:
            self.nD = ntri
        else:
            self.nD = nidx * nidx
        self.sym_pairs = list(it.combinations_with_replacement(self.idx_range, 2))
        self.all_pairs = list(it.product(self.idx_range, repeat=2))
        self.param = np.zeros(self.nV + self.nD)
        self.value = self.evaluate()

    def length(self):
        return self.nV + self.nD

    def islocal(self):
        return True

    is_local = islocal

    def update(self, param):
        self.param = param
        self.value = self.evaluate()

    def get(self, i=0, kspace=True):
        return self.value if (kspace or i == 0) else np.zeros_like(self.value)

    def _terms(self):
        """[(param offset, spin block, sign, pair list, symmetric?)] of every branch of Hubbard.py:599-770."""
        nV = self.nV
        if not self.bogoliubov:
            if self.restricted:
                return [(0, 0, 1, self.sym_pairs, True), (0, 1, 1, self.sym_pairs, True)]
            return [(0, 0, 1, self.sym_pairs, True), (nV // 2, 1, 1, self.sym_pairs, True)]
        if self.restricted:
            return [(0, 0, 1, self.sym_pairs, True), (0, 1, -1 if self.ghf else 1, self.sym_pairs, True),
                    (nV, 2, 1, self.sym_pairs, True)]
        if self.bogo_res:
            return [(0, 0, 1, self.sym_pairs, True), (nV // 2, 1, 1, self.sym_pairs, True), (nV, 2, 1, self.sym_pairs, True)]
        return [(0, 0, 1, self.sym_pairs, True), (nV // 2, 1, 1, self.sym_pairs, True), (nV, 2, 1, self.all_pairs, False)]

    def evaluate(self):
        assert self.param.shape == (self.length(),)
        V = np.zeros((3 if self.bogoliubov else 2, self.nscsites, self.nscsites))
        for off, blk, sign, pairs, sym in self._terms():
            for idx, (i, j) in enumerate(pairs):
                V[blk, i, j] = sign * self.param[idx + off]
                if sym:
                    V[blk, j, i] = sign * self.param[idx + off]
        return V

    def gradient(self):
        g = np.zeros((self.length(), 3 if self.bogoliubov else 2, self.nscsites, self.nscsites))
        for off, blk, sign, pairs, sym in self._terms():
            for idx, (i, j) in enumerate(pairs):
                g[idx + off, blk, i, j] = sign
                if sym:
                    g[idx + off, blk, j, i] = sign
        return g

    def diag_indices(self):
        idx = triu_diag_indices(len(self.idx_range))
        if self.restricted:
            return [idx]
        return [idx, np.asarray(idx) + self.nV // 2]


class VcorNonLocal(object):
    """routine/vcor.py:105-524: a translation-invariant potential with one block per lattice vector, V(-R) = V(R)^T.  Cells are walked
    in index order: a cell that is its own inverse carries symmetric blocks, the first member of a +-R pair carries free blocks and
    fills its partner with the transposes (:128-137); per cell the parameters are [V spin 0 | V spin 1 | pairing] (:148-172).
    `kmesh` stands for the lattice (cells = the Cartesian product of the mesh extents in C order)."""

    def __init__(self, restricted, bogoliubov, kmesh, nscsites, idx_range=None, bogo_res=False):
        self.restricted, self.bogoliubov, self.bogo_res = restricted, bogoliubov, bogo_res
        self.kmesh, self.nscsites = tuple(kmesh), nscsites
        self.idx_range = list(range(nscsites)) if idx_range is None else list(idx_range)
        cells = list(it.product(*[range(n) for n in self.kmesh]))
        where = dict((c, i) for i, c in enumerate(cells))
        self.ncells = len(cells)
        self.partner = [where[tuple((-np.asarray(c)) % np.asarray(self.kmesh))] for c in cells]
        self.weight = [1 if self.partner[R] == R else (2 if self.partner[R] > R else 0) for R in range(self.ncells)]
        self.nblk = 3 if bogoliubov else (1 if restricted else 2)
        self.param = None
        self._table = self._build()

    def _build(self):
        """[(parameter, block, cell, row, col)] for every assignment of the five evaluate() bodies (:176-445)."""
        n = len(self.idx_range)
        sym = list(it.combinations_with_replacement(self.idx_range, 2))
        full = list(it.product(self.idx_range, repeat=2))
        free_pairing = self.bogoliubov and not (self.restricted or self.bogo_res)
        rows, base = [], 0
        for R in range(self.ncells):
            if self.weight[R] == 0:
                continue
            own = self.weight[R] == 1
            pairs, mate = (sym, R) if own else (full, self.partner[R])
            per_spin = len(pairs)
            nV = per_spin * (1 if self.restricted else 2)
            for t, (i, j) in enumerate(pairs):
                for b in range(1 if self.restricted else 2):
                    rows += [(base + b * per_spin + t, b, R, i, j), (base + b * per_spin + t, b, mate, j, i)]
                if self.bogoliubov and not free_pairing:
                    rows += [(base + nV + t, 2, R, i, j), (base + nV + t, 2, mate, j, i)]
            nD = 0
            if self.bogoliubov and not free_pairing:
                nD = per_spin
            elif free_pairing:
                for t, (i, j) in enumerate(full):
                    rows.append((base + nV + t, 2, R, i, j))
                    if not own:
                        rows.append((base + nV + n * n + t, 2, mate, i, j))
                nD = n * n * (1 if own else 2)
            base += nV + nD
        self.nparam = base
        return rows

    def length(self):
        return self.nparam

    def islocal(self):
        return False

    is_local = islocal

    def evaluate(self):
        V = np.zeros((self.nblk, self.ncells, self.nscsites, self.nscsites))
        for p, b, R, i, j in self._table:
            V[b, R, i, j] = self.param[p]
        return V

    def gradient(self):
        g = np.zeros((self.nparam, self.nblk, self.ncells, self.nscsites, self.nscsites))
        for p, b, R, i, j in self._table:
            g[p, b, R, i, j] = 1
        return g

    def update(self, param):
        assert len(param) == self.nparam
        self.param = param
        self.value = self.evaluate()
        self.value_k = R2k(self.value, self.kmesh)

    def get(self, i=0, kspace=True, return_all=False):
        v = self.value_k if kspace else self.value
        return v if return_all else v[:, i]

    def assign(self, v0):
        g = self.gradient()
        assert v0.shape == g.shape[1:]
        self.update(np.asarray([np.sum(g[i] * v0) / np.sum(g[i] * g[i]) for i in range(self.nparam)]))


class VcorKpoints(object):
    """routine/vcor.py:526-812 (the two modes the reference implements: restricted / unrestricted, no pairing): one Hermitian
    matrix per k point with V(-k) = V(k)^*.  k points are walked in index order; a k that is its own inverse (:526-544
    get_kpts_map) holds a real symmetric matrix (lower-triangle parameters), the first member of a +-k pair holds
    [real lower triangle | imaginary strict lower triangle] and its partner the complex conjugate.  Unrestricted: per group the
    alpha half of the parameters, then the beta half (:617-633).  `value` is (nkpts, 2, nlo, nlo) complex."""

    def __init__(self, restricted, kmesh, nscsites):
        self.restricted, self.bogoliubov, self.bogo_res = restricted, False, False
        self.kmesh, self.nscsites = tuple(kmesh), nscsites
        pts = list(it.product(*[range(n) for n in self.kmesh]))
        where = dict((c, i) for i, c in enumerate(pts))
        self.nkpts = len(pts)
        mate = [where[tuple((-np.asarray(c)) % np.asarray(self.kmesh))] for c in pts]
        self.kpts_map = [[k] if mate[k] == k else [k, mate[k]] for k in range(self.nkpts) if mate[k] >= k]
        n = nscsites
        self.n_re, self.n_im = n * (n + 1) // 2, n * (n - 1) // 2
        per_spin = [self.n_re if len(grp) == 1 else self.n_re + self.n_im for grp in self.kpts_map]
        self.nparam_kpts = [c * (1 if restricted else 2) for c in per_spin]
        self.start = np.concatenate([[0], np.cumsum(self.nparam_kpts)])
        self.is_vcor_kpts, self.param, self.value = True, None, None
        self.update(np.zeros(self.length()))

    def length(self):
        return int(self.start[-1])

    def islocal(self):
        return False

    is_local = islocal

    def spin_params(self, grp, s):
        """The parameter positions of spin block `s` in group `grp`."""
        lo, hi = self.start[grp], self.start[grp + 1]
        if self.restricted:
            return np.arange(lo, hi)
        half = (hi - lo) // 2
        return np.arange(lo + s * half, lo + (s + 1) * half)

    def evaluate(self):
        n = self.nscsites
        lo, so = np.tril_indices(n), np.tril_indices(n, -1)
        V = np.zeros((self.nkpts, 2, n, n), dtype=complex)
        for grp, ks in enumerate(self.kpts_map):
            for s in range(2):
                p = self.param[self.spin_params(grp, s)]
                m = np.zeros((n, n), dtype=complex)
                m[lo] = p[:self.n_re]
                m[(lo[1], lo[0])] = p[:self.n_re]
                if len(ks) == 2:
                    m[so] += 1j * p[self.n_re:]
                    m[(so[1], so[0])] -= 1j * p[self.n_re:]
                    V[ks[1], s] = m.conj()
                V[ks[0], s] = m
        return V

    def update(self, param):
        self.param = param
        self.value = self.evaluate()

    def get(self, i=0, kspace=True):
        return self.value[i]


# ---------------------------------------------------------------------------------------------
# dV / dparam (slater.py:851-907, local branch)
# ---------------------------------------------------------------------------------------------

def transform_local_sparseH(basis, H, thr=1e-7):
    """slater_helper.py:91-100: sum over the entries |H[j,k]| > thr of basis[:, j]^T basis[:, k] H[j,k]."""
    nb = basis.shape[-1]
    res = np.zeros((nb, nb))
    for j, k in zip(*np.nonzero(abs(H) > thr)):
        res += np.dot(basis[:, j].T, basis[:, k]) * H[j, k]
    return res


def get_active_projector_full(P_act, ovlp):
    """slater.py:2195-2219: P (P^H S P) P^H per spin and k point."""
    ovlp = np.asarray(ovlp)
    if ovlp.ndim == 3:
        ovlp = ovlp[None]
    spin, nk, nlo, _ = ovlp.shape
    assert len(P_act) == spin
    out = np.empty((spin, nk, nlo, nlo), dtype=ovlp.dtype)
    for s in range(spin):
        for k in range(nk):
            P = np.asarray(P_act[s][k])
            out[s, k] = P @ (P.conj().T @ ovlp[s][k] @ P) @ P.conj().T
    return out


def get_dV_dparam(vcor, basis, compact=True, P_full=None, kmesh=None):
    """Local branch of slater.py:851-907.  With `P_full` (spin, nk, nlo, nlo), the output of get_active_projector_full, the
    basis is projected in k space and every parameter's (k-independent) gradient matrix goes through transform_trans_inv_k
    (:878-892): Re (1 / nk) sum_k C(k)^H g C(k), C(k) = P_full(k) basis_k(k)."""
    spin, nk, nlo, nb = basis.shape
    g = vcor.gradient()
    tril = np.tril_indices(nb)
    out = np.empty((vcor.length(), spin, nb * (nb + 1) // 2)) if compact else np.empty((vcor.length(), spin, nb, nb))
    if P_full is not None:
        basis_k = np.asarray([R2k(basis[s], kmesh) for s in range(spin)])
        C = np.einsum('skij,skjb->skib', np.asarray(P_full), basis_k)
    if not vcor.islocal():
        # slater.py:893-902: the gradient of a cell-resolved potential goes to k space and through transform_trans_inv_k
        basis_k = np.asarray([R2k(basis[s], vcor.kmesh) for s in range(spin)])
        for s in range(spin):
            for ip in range(vcor.length()):
                gk = R2k(g[ip, s], vcor.kmesh)
                m = (np.einsum('kia,kij,kjb->ab', basis_k[s].conj(), gk, basis_k[s]) / nk).real
                out[ip, s] = m[tril] if compact else m
        return out
    for s in range(spin):
        for ip in range(vcor.length()):
            if P_full is None:
                m = transform_local_sparseH(basis[s], g[ip, s])
            else:
                m = (np.einsum('kia,ij,kjb->ab', C[s].conj(), g[ip, s], C[s]) / nk).real
            out[ip, s] = m[tril] if compact else m
    return out


# ---------------------------------------------------------------------------------------------
# finite-T response (ftsystem.py:151-213)
# ---------------------------------------------------------------------------------------------

def fermi_smearing_occ(mu, mo_energy, beta):
    mo_energy = np.asarray(mo_energy)
    mu = np.asarray(mu).reshape(-1, *([1] * (mo_energy.ndim - 1)))
    de = beta * (mo_energy - mu)
    occ = np.zeros_like(mo_energy)
    idx = de < 100
    occ[idx] = 1.0 / (np.exp(de[idx]) + 1.0)
    return occ


def get_dw_dv(mo_energy, mo_coeff, drho, mu, beta, fix_mu=True, compact=False, fit_idx=None):
    spin, _, norb = mo_coeff.shape
    if fit_idx is None:
        fit_idx = range(norb)
    fit_idx = list(fit_idx)
    f = fermi_smearing_occ(mu, mo_energy, beta)
    h = 1.0 - f
    dw_dv = np.zeros((spin, norb, norb), dtype=np.asarray(mo_coeff).dtype)
    for s in range(spin):
        de = mo_energy[s, :, None] - mo_energy[s]
        zero = np.abs(de) < ZERO_TOL
        inv = np.zeros_like(de)
        inv[~zero] = 1.0 / de[~zero]
        K = inv * (f[s, :, None] - f[s])
        K[zero] = (f[s, :, None] * h[s])[zero] * (-beta)
        C = mo_coeff[s]
        tmp = (C[fit_idx].T @ (2.0 * drho[s]) @ C[fit_idx].conj()) * K                  # ftsystem.py:263
        dw_dv[s] = C.conj() @ tmp @ C.T
        if not fix_mu:
            ff = f[s] * h[s]
            fsum = np.sum(ff)
            if abs(fsum) > ZERO_TOL:
                drho_dmu = (C * ff) @ C.conj().T
                dw_dmu = np.einsum('ij,ij->', drho[s], drho_dmu[fit_idx][:, fit_idx]) * 2.0 * beta
                dw_dv[s] += drho_dmu * (dw_dmu / fsum)
    if compact:
        tl = np.tril_indices(norb)
        packed = np.asarray([m[tl] for m in dw_dv]) * 2.0
        dg = np.cumsum([0] + list(range(2, norb + 1)))
        packed[:, dg] *= 0.5
        return packed
    return dw_dv


def get_rho_grad(mo_energy, mo_coeff, mu, beta, fix_mu=True):
    """ftsystem.py:147-221 with compact=True: d rho_r / d v_V for tril pairs V of the potential and r of the density,
    (npair, npair); one spin channel."""
    norb = mo_coeff.shape[-1]
    f = fermi_smearing_occ(mu, mo_energy, beta)
    h = 1.0 - f
    de = mo_energy[:, None] - mo_energy
    zero = np.abs(de) < ZERO_TOL
    inv = np.zeros_like(de)
    inv[~zero] = 1.0 / de[~zero]
    K = inv * (f - f[:, None])
    K[zero] = (f[:, None] * h)[zero] * beta
    C = mo_coeff
    scr = np.einsum('lp,mp->lmp', C.conj(), C)
    g = -np.tensordot(np.dot(scr, K), scr, axes=((-1,), (-1,))).transpose(0, 3, 1, 2)       # [l, s, m, n]
    g = g + g.transpose(1, 0, 2, 3)
    g[np.arange(norb), np.arange(norb)] *= 0.5
    tl = np.tril_indices(norb)
    g = g[tl]                                                                                   # (V, m, n)
    if not fix_mu:
        ff = f * h
        fsum = ff.sum()
        if abs(fsum) > ZERO_TOL:
            drho_dmu = np.dot(C * ff, C.conj().T) * beta
            mg = np.dot(np.einsum('ki,li->kli', C.conj(), C), ff) / fsum
            mg = mg + mg.T
            mg[np.arange(norb), np.arange(norb)] *= 0.5
            g = g + np.einsum('k,ij->kij', mg[tl], drho_dmu)
    return g.transpose(1, 2, 0)[tl].transpose(1, 0)


def get_rdm1_idem(rdm1, nelec, beta):
    """slater_helper.py:380-421: natural orbitals of every (spin[, k]) block, occupations re-assigned by assignocc on the
    NEGATED natural occupations (descending -> ascending levels, mu0 = -0.5), density rebuilt."""
    rdm1 = np.asarray(rdm1)
    shape = rdm1.shape
    spin, nlo = shape[0], shape[-1]
    blocks = rdm1.reshape(spin, -1, nlo, nlo)
    nblk = blocks.shape[1]
    ew = np.empty((spin, nblk, nlo))
    ev = np.empty((spin, nblk, nlo, nlo), dtype=rdm1.dtype)
    for s in range(spin):
        for k in range(nblk):
            ew[s, k], ev[s, k] = la.eigh(blocks[s, k])
    ew, ev = -ew[:, :, ::-1], ev[:, :, :, ::-1]
    if rdm1.ndim == 3:
        occ, _, _ = assignocc(ew[:, 0], nelec, beta, -0.5)
        occ = occ[:, None]
    else:
        occ, _, _ = assignocc(ew, nelec, beta, -0.5)
    out = np.einsum('skpm,skm,skqm->skpq', ev, occ, ev.conj())
    return out.reshape(shape)


# ---------------------------------------------------------------------------------------------
# the fit objective and its gradients (slater.py:1040-1215)
# ---------------------------------------------------------------------------------------------

class EmbFit(object):
    """errfunc / gradfunc of FitVcorEmb for given (rho target, fock_k, ovlp_k, basis, vcor)."""

    def __init__(self, rho, kmesh, basis, vcor, beta, fock_k, ovlp_k, nelec, imp_idx=None, det_idx=None,
                 mu0=None, fix_mu=False, tol_deg=1e-3, remove_diag_grad=False, idem_fit=False, C_act=None, P_full=None):
        if idem_fit:
            rho = get_rdm1_idem(rho, nelec, beta)                                   # slater.py:975-978
        self.C_act = None if C_act is None else np.asarray(C_act)
        self.spin, self.nb = basis.shape[0], basis.shape[-1]
        spin, nb = self.spin, self.nb
        self.beta, self.nelec, self.mu0, self.fix_mu, self.tol_deg = beta, nelec, mu0, fix_mu, tol_deg
        self.vcor, self.remove_diag_grad = vcor, remove_diag_grad
        if imp_idx is None and det_idx is None:
            imp_idx, det_idx = list(range(nb)), []
        imp_idx, det_idx = list(imp_idx or []), list(det_idx or [])
        self.fit_idx = imp_idx + det_idx
        nimp, nidx = len(imp_idx), len(self.fit_idx)
        self.imp_mesh, self.det_mesh = np.ix_(imp_idx, imp_idx), (det_idx, det_idx)
        self.imp_fill, self.det_fill = (slice(nimp), slice(nimp)), (range(nimp, nidx), range(nimp, nidx))
        basis_k = np.asarray([R2k(basis[s], kmesh) for s in range(spin)])
        fock_k = np.asarray(fock_k)
        if fock_k.ndim == 3:
            fock_k = fock_k[None]
        self.embH1 = transform_h1(fock_k, basis_k)
        self.ovlp = transform_h1(ovlp_k, basis_k)
        self.dV = get_dV_dparam(vcor, basis, compact=True, P_full=P_full, kmesh=kmesh)
        self.tril = np.tril_indices(nb)
        self.target = np.zeros((spin, nidx, nidx))
        for s in range(spin):
            self.target[s][self.imp_fill] = rho[s][self.imp_mesh]
            self.target[s][self.det_fill] = rho[s][self.det_mesh]

    def norm_div(self):
        """|drho| is divided by sqrt(spin) (slater.py:1094); the GSO twin sets `norm` = sqrt(2) on its single block (spinless.py:1259)."""
        return getattr(self, "norm", None) or sqrt(self.spin)

    def Vemb(self, param):
        tmp = np.tensordot(param, self.dV, axes=(0, 0))
        v = np.zeros((self.spin, self.nb, self.nb))
        for s in range(self.spin):
            v[s][self.tril] = tmp[s]
        return v

    def _solve(self, param):
        spin, nb = self.spin, self.nb
        H = self.embH1 + self.Vemb(param)
        ew, ev = np.empty((spin, nb)), np.empty((spin, nb, nb))
        for s in range(spin):
            ew[s], ev[s] = la.eigh(H[s], self.ovlp[s])
        if not self.fix_mu:
            ne = self.nelec
            mu = (0.5 * (ew[0][ne - 1] + ew[0][ne]) if spin == 1 else
                  [0.5 * (ew[s][ne[s] - 1] + ew[s][ne[s]]) for s in range(2)])
        else:
            mu = self.mu0
        occ, mu, _ = assignocc(ew, self.nelec, self.beta, mu, fix_mu=self.fix_mu, thr_deg=self.tol_deg)
        rho1 = np.zeros_like(self.target)
        for s in range(spin):
            tmp = np.dot(ev[s] * occ[s], ev[s].T)
            rho1[s][self.imp_fill] = tmp[self.imp_mesh]
            rho1[s][self.det_fill] = tmp[self.det_mesh]
        return ew, ev, occ, mu, rho1 - self.target

    def _residual(self, drho):
        """(what the norm is taken of, what enters the gradient): with C_act the residual is projected on the active orbitals,
        C^T drho C, and the gradient sees C (C^T drho C) C^T (slater.py:1083-1088, 1112-1124)."""
        if self.C_act is None:
            return drho, drho
        act = np.asarray([self.C_act[s].T @ drho[s] @ self.C_act[s] for s in range(self.spin)])
        return act, np.asarray([self.C_act[s] @ act[s] @ self.C_act[s].T for s in range(self.spin)])

    def errfunc(self, param):
        drho = self._solve(param)[4]
        return la.norm(self._residual(drho)[0]) / self.norm_div()

    def drho_dparam(self, param):
        """return_drho_dparam (slater.py:1227-1261): (spin, nparam, npair) response of the tril-packed embedding density to the
        parameters at finite temperature."""
        assert self.beta < np.inf
        ew, ev, occ, mu, _ = self._solve(param)
        mu = np.atleast_1d(mu)
        dv = np.asarray([get_rho_grad(ew[s], ev[s], mu[s], self.beta, fix_mu=self.fix_mu) for s in range(self.spin)])
        return np.einsum('psV,sVr->spr', self.dV, dv, optimize=True)

    def _finish(self, res):
        if self.remove_diag_grad:
            for s in range(self.spin):
                d = self.vcor.diag_indices()[s]
                res[d] -= np.average(res[d])
        return res

    def gradfunc(self, param):
        """T = 0 analytic gradient (slater.py:1096-1154)."""
        spin, nb = self.spin, self.nb
        ew, ev, occ, mu, drho = self._solve(param)
        act, drho = self._residual(drho)
        val = la.norm(act)
        nocc = int(np.round(np.sum(occ) / spin))
        dw = np.empty((spin, nb * (nb + 1) // 2))
        dg = (np.arange(nb), np.arange(nb))
        for s in range(spin):
            eo, evirt = ew[s, :nocc], ew[s, nocc:]
            co, cv = ev[s][:, :nocc], ev[s][:, nocc:]
            e_mn = 1.0 / (-evirt.reshape((-1, 1)) + eo)
            t = (cv[self.fit_idx].T @ drho[s] @ co[self.fit_idx]) * e_mn / (val * self.norm_div())
            full = cv @ t @ co.T
            full = (full + full.T) * 2.0
            full[dg] *= 0.5
            dw[s] = full[self.tril]
        return self._finish(np.tensordot(self.dV, dw, axes=((1, 2), (0, 1))))

    def gradfunc_ft(self, param):
        """finite-T analytic gradient (slater.py:1156-1197)."""
        ew, ev, occ, mu, drho = self._solve(param)
        act, drho = self._residual(drho)
        val = la.norm(act)
        dw_dv = get_dw_dv(ew, ev, drho, mu, self.beta, fix_mu=self.fix_mu, fit_idx=self.fit_idx, compact=True)
        res = self.dV.reshape(self.dV.shape[0], -1).dot(dw_dv.ravel()) / (2.0 * val * self.norm_div())
        return self._finish(res)


# ---------------------------------------------------------------------------------------------
# lattice-space objective (slater.py:1448-1478, FitVcorFull.errfunc)
# ---------------------------------------------------------------------------------------------

class FullFit(object):
    def __init__(self, rho, kmesh, basis, vcor, beta, Fock_k, filling, imp_idx=None, det_idx=None, fix_mu=False):
        from oracle.restate import check_nelec
        spin, nk, n, nb = basis.shape
        self.spin, self.nk, self.n, self.nb, self.beta, self.vcor, self.fix_mu = spin, nk, n, nb, beta, vcor, fix_mu
        self.imp_bath_fit = imp_idx is None and det_idx is None
        if self.imp_bath_fit:
            imp_idx, det_idx = list(range(nb)), []
        imp_idx, det_idx = list(imp_idx or []), list(det_idx or [])
        nimp, nidx = len(imp_idx), len(imp_idx) + len(det_idx)
        self.imp_mesh, self.det_mesh = np.ix_(imp_idx, imp_idx), (det_idx, det_idx)
        self.imp_fill, self.det_fill = (slice(nimp), slice(nimp)), (range(nimp, nidx), range(nimp, nidx))
        self.fit_idx = imp_idx + det_idx
        self.target = np.zeros((spin, nidx, nidx))
        for s in range(spin):
            self.target[s][self.imp_fill] = rho[s][self.imp_mesh]
            self.target[s][self.det_fill] = rho[s][self.det_mesh]
        Fock_k = np.asarray(Fock_k)
        self.Fock = Fock_k if Fock_k.ndim == 4 else Fock_k[None]
        self.basis_k = np.asarray([R2k(basis[s], kmesh) for s in range(spin)])
        self.nelec = check_nelec(spin * nk * n * filling, None)[0]

    def _solve(self, param):
        spin, nk, n = self.spin, self.nk, self.n
        self.vcor.update(param)
        ew = np.empty((spin, nk, n))
        ev = np.empty((spin, nk, n, n), dtype=np.complex128)
        for s in range(spin):
            for k in range(nk):
                ew[s, k], ev[s, k] = la.eigh(self.Fock[s, k] + self.vcor.get(k, True)[s])
        occ, mu, _ = assignocc(ew, self.nelec, self.beta, 0.0, fix_mu=self.fix_mu)
        return ew, ev, occ, mu

    def gradfunc_ft(self, param):
        """finite-T analytic lattice gradient (slater.py:1480-1640, the local-vcor branch 1631-1640): every k point
        contributes its own get_dw_dv (chemical-potential response normalised per k, as the reference does)."""
        spin, nk, n = self.spin, self.nk, self.n
        if self.imp_bath_fit:
            raise NotImplementedError                                       # slater.py:1510-1512
        ew, ev, occ, mu = self._solve(param)
        rhoT = np.einsum('skpm,skm,skqm->spq', ev, occ, ev.conj()).real / nk
        rho1 = np.zeros_like(self.target)
        for s in range(spin):
            rho1[s][self.imp_fill] = rhoT[s][self.imp_mesh]
            rho1[s][self.det_fill] = rhoT[s][self.det_mesh]
        drho = rho1 - self.target
        val = la.norm(drho)
        if getattr(self.vcor, "is_vcor_kpts", False):
            # slater.py:1519-1628: the first k of every group answers for the group; a +-k pair sees dw(k1) and its conjugate
            v = self.vcor
            lo, so = np.tril_indices(n), np.tril_indices(n, -1)
            res = np.zeros(v.length())
            for grp, ks in enumerate(v.kpts_map):
                dw = get_dw_dv(ew[:, ks[0]], ev[:, ks[0]], drho, mu, self.beta, fix_mu=self.fix_mu, fit_idx=self.fit_idx, compact=False)
                for s in range(spin):
                    if len(ks) == 1:
                        re, im = dw[s].real.copy(), None
                    elif v.restricted:
                        re, im = (dw[s].conj() + dw[s]).real, (dw[s].conj() - dw[s]).imag
                    else:
                        re, im = (dw[s] + dw[s].conj()).real, -(dw[s] - dw[s].conj()).imag
                    re[so] *= 2.0
                    pos = v.spin_params(grp, s)
                    res[pos[:v.n_re]] = re[lo]
                    if im is not None:
                        res[pos[v.n_re:]] = 2.0 * im[so]
            return res / (2.0 * val * sqrt(spin) * nk)
        g = self.vcor.gradient()                                            # (nparam, 2|spin, n, n)
        nparam = g.shape[0]
        tril = np.tril_indices(n)
        dV = np.asarray([[g[i, s][tril] for s in range(g.shape[1])] for i in range(nparam)])     # pack_tril (slater.py:1345-1348)
        if spin == 1:
            dV = dV[:, [0]]
        res = np.zeros(nparam)
        for k in range(nk):
            dw_dv = get_dw_dv(ew[:, k], ev[:, k], drho, mu, self.beta, fix_mu=self.fix_mu, fit_idx=self.fit_idx,
                              compact=True).real
            res += dV.reshape(nparam, -1).dot(dw_dv.ravel())
        return res.real / (2.0 * val * sqrt(spin) * nk)

    def errfunc(self, param):
        spin, nk, n = self.spin, self.nk, self.n
        ew, ev, occ, mu = self._solve(param)
        rho_k = np.einsum('skpm,skm,skqm->skpq', ev, occ, ev.conj())
        rho1 = np.zeros_like(self.target)
        if self.imp_bath_fit:
            rho1[:] = transform_h1(rho_k, self.basis_k)
        else:
            rhoT = rho_k.sum(axis=1).real / nk
            for s in range(spin):
                rho1[s][self.imp_fill] = rhoT[s][self.imp_mesh]
                rho1[s][self.det_fill] = rhoT[s][self.det_mesh]
        return la.norm(rho1 - self.target) / sqrt(spin)
