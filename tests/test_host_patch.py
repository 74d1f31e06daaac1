"""
Host logic of libdmet_preview_amd.patch (the rebinding of SURVEY.md section 8b): every entry of the binding table
exists on our side; install / uninstall rebind every holder of a name and restore it; when the reference tree is
present in this container (never on the GPU box) the table is checked against the reference's real module tree
under oracle/shim.py.  No compute is called.
"""
import importlib
import sys
import types

import pytest

from libdmet_preview_amd import patch


def test_binding_table_resolves_on_our_side():
    for ours, ref, name in patch.binding_table():
        mod = importlib.import_module("libdmet_preview_amd." + ours)
        assert callable(getattr(mod, name)), (ours, name)
    from libdmet_preview_amd.system.lattice import Lattice
    for name in patch._LATTICE_METHODS:
        assert callable(getattr(Lattice, name))


def _fake_reference(root):
    """A module tree with the reference's module names whose attributes are placeholders."""
    mods = {}

    def mk(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        mods[name] = m
        return m
    mk(root)
    for ours, ref, name in patch.binding_table():
        parts = ref.split(".")
        for i in range(1, len(parts) + 1):
            full = root + "." + ".".join(parts[:i])
            if full not in mods:
                mk(full)
        setattr(mods[root + "." + ref], name, ("placeholder", ref, name))
    lat = mods.get(root + ".system.lattice") or mk(root + ".system.lattice")

    class Lattice(object):
        pass
    for name in patch._LATTICE_METHODS:
        setattr(Lattice, name, ("placeholder", name))
    lat.Lattice = Lattice
    return mods


def test_install_rebinds_every_holder_and_uninstall_restores():
    root = "fake_libdmet_for_patch_test"
    mods = _fake_reference(root)
    try:
        handle = patch.install(reference_package=root)
        for ours, ref, name in patch.binding_table():
            mine = getattr(importlib.import_module("libdmet_preview_amd." + ours), name)
            assert getattr(mods[root + "." + ref], name) is mine, (ref, name)
        from libdmet_preview_amd.system.lattice import Lattice
        for name in patch._LATTICE_METHODS:
            assert getattr(mods[root + ".system.lattice"].Lattice, name) is getattr(Lattice, name)
        patch.uninstall(handle)
        for ours, ref, name in patch.binding_table():
            assert getattr(mods[root + "." + ref], name) == ("placeholder", ref, name)
        # a renamed entry point on the reference side is an error, not a silent skip
        delattr(mods[root + ".routine.slater"], "get_emb_eri")
        with pytest.raises(AttributeError):
            patch.install(reference_package=root)
        h = patch.install(reference_package=root, strict=False, replace_hf=False)
        assert mods[root + ".routine.mfd"].HF == ("placeholder", "routine.mfd", "HF")
        patch.uninstall(h)
    finally:
        for k in list(mods):
            sys.modules.pop(k, None)


def test_binding_table_matches_the_reference_tree():
    from oracle import shim
    if not shim.available():
        pytest.skip("reference tree not present (GPU box)")
    shim.install()
    shim.quiet()
    for ours, ref, name in patch.binding_table():
        mod = importlib.import_module("libdmet." + ref)
        assert hasattr(mod, name), (ref, name)
    lat = importlib.import_module("libdmet.system.lattice")
    for name in patch._LATTICE_METHODS:
        assert hasattr(lat.Lattice, name), name
    handle = patch.install()
    try:
        import libdmet.routine.slater as rs
        from libdmet_preview_amd.basis_transform import eri_transform as aet
        assert rs.get_emb_eri is aet.get_emb_eri
    finally:
        patch.uninstall(handle)
    import libdmet.routine.slater as rs
    assert rs.get_emb_eri is not aet.get_emb_eri
