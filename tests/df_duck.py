"""
Test helpers: what the REFERENCE hands `get_emb_eri` -- a `pyscf.pbc.df.GDF`-shaped object -- and the reference-side names the
patch rebinds.  Shared by tests/test_host_df_object.py (CPU: resolution logic) and tests/test_gpu_df_object.py (GPU: parity).

  DuckGDF            only `cell`, `kpts`, `_cderi`, `blockdim`, `max_memory` (what golden G17 records the reference reading from
                     its DF object besides the stand-in's own block table); NO provider method (`load_block`, `get_block`,
                     `naux`).
  ao_container       the `cderi` layout PySCF writes for AO blocks: "j3c-kptij" (npairs, 2, 3) absolute k-point pairs with
                     i >= j, "j3c/<pair>/0" (naux, nao*nao) c128, lower-triangular packed (naux, nao*(nao+1)/2) when ki == kj
                     (real at Gamma) -- the reference's own writer restated by oracle/restate_cderi.py (pinned by golden G13),
                     fed identity coefficients.
  fake_h5py          a module object for sys.modules["h5py"] whose File(path, mode) returns the mapping registered for `path`
                     and records open / close: proves the path-string branch without h5py on the box.
  patched_reference  context manager: libdmet_preview_amd.patch.install() on the reference's module tree -- the real one under
                     oracle/shim.py where /root/reference exists (this container), a placeholder tree with the reference's
                     module names on the GPU box -- yielding the module `libdmet.routine.slater`.
"""
import contextlib
import importlib
import sys
import types

import numpy as np


class DuckGDF(object):
    def __init__(self, cell, kpts, cderi, blockdim=240, max_memory=4000):
        self.cell, self.kpts, self._cderi = cell, np.asarray(kpts), cderi
        self.blockdim, self.max_memory = blockdim, max_memory


def ao_container(blocks, ks, kabs, naux, nao):
    """cderi datasets of the AO blocks `blocks[(i, j)]` (naux, nao, nao): every pair i >= j (PySCF stores them all)."""
    from oracle import restate_cderi as Cd                       # checker-side restatement of the reference's writer
    eye = np.broadcast_to(np.eye(nao, dtype=np.complex128), (len(ks), nao, nao)).copy()
    out, _ = Cd.transform_gdf_to_lo(lambda i, j: blocks[(i, j)], ks, kabs, naux, eye, t_reversal_symm=False)
    return out


class _FakeFile(object):
    def __init__(self, mod, path, mapping):
        self._mod, self._path, self._map = mod, path, mapping
        self.closed = False

    def __getitem__(self, key):
        assert not self.closed, "read from a closed file"
        return self._map[key]

    def close(self):
        self.closed = True
        self._mod.closed.append(self._path)


def fake_h5py(registry):
    """Module stand-in: File(path, "r") -> a read-only view of registry[path]; `.opened` / `.closed` list the paths."""
    mod = types.ModuleType("h5py")
    mod.opened, mod.closed = [], []

    def File(path, mode="r", **kw):
        assert mode == "r", "the DF container must be opened read-only"
        if path not in registry:
            raise OSError("no such file: %s" % path)
        mod.opened.append(path)
        return _FakeFile(mod, path, registry[path])
    mod.File = File
    return mod


@contextlib.contextmanager
def patched_reference(**install_kwargs):
    """Yield `libdmet.routine.slater` with libdmet_preview_amd.patch installed (real reference tree under the import shim when
    it exists, placeholder tree otherwise); everything is undone on exit."""
    from libdmet_preview_amd import patch
    from oracle import shim
    made = []
    if shim.available():
        shim.install()
        shim.quiet()
    else:
        def mk(name):
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
                made.append(name)
            return sys.modules[name]
        mk("libdmet")
        for ours, ref, name in patch.binding_table():
            parts = ref.split(".")
            for i in range(1, len(parts) + 1):
                mk("libdmet." + ".".join(parts[:i]))
            setattr(sys.modules["libdmet." + ref], name, ("placeholder", ref, name))
        lat = mk("libdmet.system.lattice")
        lat.Lattice = type("Lattice", (), {n: ("placeholder", n) for n in patch._LATTICE_METHODS})
    handle = patch.install(**install_kwargs)
    try:
        yield importlib.import_module("libdmet.routine.slater")
    finally:
        patch.uninstall(handle)
        for name in made:
            sys.modules.pop(name, None)
