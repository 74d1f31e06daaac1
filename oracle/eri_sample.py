"""
ORACLE -- test infrastructure only (see oracle/eri_sample.c).  Never imported by the product path.

Sampled restatement of the reference's get_emb_eri_fast_gdf (basis_transform/eri_transform.py:235-399) for
production-size checks: the double loop over (kL, i, j) is the reference's own visiting plan (the G1 golden
`plan_tr`, recorded from the reference's driver; oracle/restate.py:tr_block_plan for meshes without a golden),
each visited AO block contributes (L|ab) for ALL auxiliary rows L but only for a, b in a sample A of embedding
orbitals (oracle/eri_sample.c:orc_half_sample), pack_tril keeps a >= b, and the contraction
eri += w (Re^T Re [+ Im^T Im]) (eri_transform.py:436-485) runs on the sampled pair columns.  Every number it
returns is an exact entry of the full Lij_s4 / ERI.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "eri_sample.c")):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_philox_rows.restype = None
        _lib.orc_philox_rows.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_half_sample.restype = None
        _lib.orc_half_sample.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_int, C.c_void_p]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_set_threads.restype = None
        _lib.orc_set_threads.argtypes = [C.c_int]
        _lib.orc_set_threads(usable_cpus())
    return _lib


def usable_cpus():
    """Affinity mask capped by the cgroup CPU quota (a box may show 256 logical CPUs and grant 16)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def num_threads():
    return int(lib().orc_num_threads())


def set_threads(n):
    """OpenMP threads of the C oracle (torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks)."""
    lib().orc_set_threads(int(n))


def philox_rows(seed, ki, kj, nao, L0, nL):
    """Rows [L0, L0 + nL) of the synthetic DF block (ki, kj): (nL, nao, nao) c128."""
    out = np.empty((nL, nao, nao), dtype=np.complex128)
    lib().orc_philox_rows(C.c_uint64(int(seed)), int(ki), int(kj), int(nao), int(L0), int(nL), out.ctypes.data_as(C.c_void_p))
    return out


def df_block_philox(seed, ki, kj, naux, nao):
    return philox_rows(seed, ki, kj, nao, 0, naux)


def plan_records(mesh, kL_list=None):
    """(kL, i, j, jm, sym) records of the time-reversal double loop, grouped by kL: the reference-recorded G1 plan
    when the mesh has one, else the restated loop."""
    mesh = [int(x) for x in mesh] + [1] * (3 - len(mesh))
    gpath = os.path.join(os.path.dirname(_HERE), "tests", "golden", "G1_ktables.npz")
    key = "%dx%dx%d/plan_tr" % tuple(mesh)
    rec = None
    if os.path.exists(gpath):
        g = np.load(gpath, allow_pickle=False)
        if key in g.files:
            # event stream recorded from the reference's own driver (oracle/gen_golden.py): [sym, i, j] per visited block,
            # [2, weight, -1] after each kL's blocks; the kL sequence is the irreducible kL (weight > 0) in order
            ev = np.asarray(g[key]).astype(int)
            weights = np.asarray(g["%dx%dx%d/weights" % tuple(mesh)]).astype(int)
            irr = iter([k for k in range(len(weights)) if weights[k] > 0])
            kL, rec = next(irr), []
            for e in ev:
                if e[0] == 2:
                    assert int(e[1]) == int(weights[kL])
                    kL = next(irr, None)
                else:
                    rec.append((kL, int(e[1]), int(e[2]), -1, int(e[0])))
    if rec is None:
        from oracle import restate as R
        weights, plan = R.tr_block_plan(R.make_kpts_scaled(mesh), True)
        rec = [(a, b, c, d, int(e)) for (a, b, c, d, e) in plan]
    by = {}
    for r in rec:
        if kL_list is None or int(r[0]) in kL_list:
            by.setdefault(int(r[0]), []).append(tuple(int(x) for x in r))
    return np.asarray(weights).astype(int), by


def half_sample_kL(seed, records, C_ao_emb, naux, A, L_list=None, max_blocks=None):
    """Sum over the visited blocks of one kL of the sampled (L|ab): (spin, nrows, nA, nA) c128 (before pack_tril)."""
    C_ao_emb = np.ascontiguousarray(C_ao_emb, dtype=np.complex128)
    spin, nk, nao, nemb = C_ao_emb.shape
    A = np.ascontiguousarray(A, dtype=np.int32)
    nA = len(A)
    Ll = None if L_list is None else np.ascontiguousarray(L_list, dtype=np.int32)
    nrows = naux if Ll is None else len(Ll)
    S = np.zeros((spin, nrows, nA, nA), dtype=np.complex128)
    stride = nk * nao * nemb
    n = 0
    for (kL, i, j, jm, sym) in records:
        lib().orc_half_sample(C.c_uint64(int(seed)), int(i), int(j), int(naux), int(nao), int(nemb), int(spin),
                              C_ao_emb[0, i].ctypes.data_as(C.c_void_p), stride, C_ao_emb[0, j].ctypes.data_as(C.c_void_p), stride,
                              nA, A.ctypes.data_as(C.c_void_p), 1 if sym else 0,
                              None if Ll is None else Ll.ctypes.data_as(C.c_void_p), 0 if Ll is None else len(Ll),
                              S.ctypes.data_as(C.c_void_p))
        n += 1
        if max_blocks is not None and n >= max_blocks:
            break
    return S


def sample_pairs(A):
    """Sampled orbital list -> (x, y, packed pair index) for A[x] >= A[y] (pack_tril order a(a+1)/2 + b)."""
    A = [int(a) for a in A]
    out = []
    for x, a in enumerate(A):
        for y, b in enumerate(A):
            if a >= b:
                out.append((x, y, a * (a + 1) // 2 + b))
    out.sort(key=lambda t: t[2])
    return out


def planes_sample(S, A):
    """pack_tril of the sampled (L|ab): (spin, nrows, nP) c128 and the packed pair indices."""
    prs = sample_pairs(A)
    xs, ys, idx = [p[0] for p in prs], [p[1] for p in prs], np.asarray([p[2] for p in prs])
    return S[:, :, xs, ys], idx


def eri_sample(mesh, seed, C_ao_emb, naux, A, kL_list, max_blocks_per_kL=None):
    """Entries eri[blk][P, Q] for P, Q in the sampled pair columns, accumulated over kL_list:
    returns (spin_pair, nP, nP) f64 in (aa, ab, bb) order, the pair indices, and the per-kL sampled planes."""
    weights, by = plan_records(mesh, set(int(k) for k in kL_list))
    spin = np.asarray(C_ao_emb).shape[0]
    nblk = spin * (spin + 1) // 2
    eri, idx, planes = None, None, {}
    for kL in kL_list:
        S = half_sample_kL(seed, by[int(kL)], C_ao_emb, naux, A, max_blocks=max_blocks_per_kL)
        P, idx = planes_sample(S, A)
        planes[int(kL)] = P
        if eri is None:
            eri = np.zeros((nblk, len(idx), len(idx)))
        w = int(weights[int(kL)])
        parts = [P.real] if w == 1 else [P.real, P.imag]
        for X in parts:
            if spin == 1:
                eri[0] += w * X[0].T @ X[0]
            else:
                eri[0] += w * X[0].T @ X[0]
                eri[1] += w * X[0].T @ X[1]
                eri[2] += w * X[1].T @ X[1]
    return eri, idx, planes
