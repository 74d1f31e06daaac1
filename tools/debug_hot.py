import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libdmet_preview_amd import _lib
from libdmet_preview_amd._lib import lib
from libdmet_preview_amd.basis_transform import eri_transform as et
from oracle import restate as R
ctx = _lib.get_ctx()
nao, naux, spin = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mesh, nemb = (2, 2, 1), 256
npair = nemb * (nemb + 1) // 2
rng = np.random.default_rng(nao)
Cemb = (rng.standard_normal((spin, 4, nao, nemb)) + 1j * rng.standard_normal((spin, 4, nao, nemb))) / np.sqrt(nao)
C_dev = ctx.to_device(Cemb)
eri_dev = ctx.zeros((spin * (spin + 1) // 2, 8, 8), np.float64)
eng = et.EriEngine(ctx, mesh, nao, naux, nemb, spin, C_dev, eri_dev, True)
print("engine ok", flush=True)
ctx.check(lib.dmk_eri_begin_kL(eng.h, 1)); ctx.sync(); print("begin ok", flush=True)
ref = np.zeros((spin, naux, npair), dtype=np.complex128)
for (i, j, sym) in [(1, 0, 1), (3, 2, 0), (0, 1, 1)]:
    blk = R.df_block_philox(5, i, j, naux, nao)
    d = ctx.to_device(blk)
    ctx.check(lib.dmk_eri_push_block(eng.h, i, j, sym, d.ptr)); ctx.sync(); print("push ok", i, j, sym, flush=True)
    Lij = R.transform_ao_to_emb(blk.reshape(naux, -1), Cemb, i, j)
    if sym:
        Lij = Lij + Lij.transpose(0, 1, 3, 2)
    ref += R.pack_tril(Lij)
    planes = eng.planes().get()
    got = planes[:, 0] + 1j * planes[:, 1]
    print("   err", np.abs(got - ref).max(), "scale", np.abs(ref).max(), flush=True)
    bad = np.argwhere(np.abs(got - ref) > 1e-9)
    if len(bad):
        ia, ib = np.tril_indices(nemb)
        print("   nbad", len(bad), "first", bad[:5], "rows/cols", [(ia[b[2]], ib[b[2]]) for b in bad[:8]], flush=True)
eng.close()
print("done")
