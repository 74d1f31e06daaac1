"""
TEST INFRASTRUCTURE (never imported by the product path): attribute-access recorder for the duck-typed objects the
embedding-construction entry points are handed -- the lattice, the correlation potential, the density-fitting object, the cell
(last paragraph: `lattice.{ncells, nscsites, kmesh, imp_idx, val_idx, is_model, ...}`).

`watch(obj, log, kind)` swaps the object's class for a generated subclass whose `__getattribute__` notes every attribute name
that code OUTSIDE the object reads (reads by the object's own methods -- frames whose `self` is the object -- are its private
business and are skipped).  isinstance() keeps working (the subclass derives from the original class), which the reference's
`isinstance(mydf, df.GDF)` dispatch needs.  `Log.stage(name)` labels the entry point under which reads are filed.

Used twice with the same code:
  * oracle/gen_golden.py gen_G17: around the REFERENCE's Lattice / Vcor / FakeGDF / FakeCell while the reference's own
    HartreeFock / ConstructImpHam / FitVcor / get_emb_eri_fast_gdf run  ->  tests/golden/G17_contract.npz
    (per entry point and object kind: the names the reference reads; per object kind: every name the reference object offers);
  * tests/test_gpu_chain.py: around this package's mirror objects while the mirror entry points replay the same chain; the
    test asserts that everything OUR entry points read exists on the reference's objects (a renamed or invented attribute on
    our side fails) and that nothing outside the reference's own reads plus a short documented list is touched.
reference: dmet/Hubbard.py:14-37, dmet/HubPhSymm.py:74-100, routine/slater.py:98-220, 357-524, 909-1329.
"""
import sys


class Log(object):
    def __init__(self):
        self.reads = {}              # (stage, kind) -> set of names
        self.current = "unlabelled"

    def stage(self, name):
        self.current = name
        return self

    def note(self, kind, name):
        self.reads.setdefault((self.current, kind), set()).add(name)

    def names(self, kind, stage=None):
        out = set()
        for (st, kd), v in self.reads.items():
            if kd == kind and (stage is None or st == stage):
                out |= v
        return out

    def stages(self):
        return sorted({st for (st, _) in self.reads})

    def kinds(self):
        return sorted({kd for (_, kd) in self.reads})


_SKIP = {"__class__", "__dict__", "__getattribute__", "__setattr__", "__getstate__", "__reduce_ex__", "__reduce__", "__deepcopy__",
         "__array_struct__", "__array_interface__", "__array__", "__array_priority__", "__array_ufunc__", "__array_function__",
         "__len__", "__iter__", "__bool__", "__index__", "__float__", "__int__", "__copy__"}


def watch(obj, log, kind):
    """Record external attribute reads of `obj` under `kind`.  Returns obj (its class is replaced in place)."""
    base = type(obj)

    def __getattribute__(self, name):
        if name not in _SKIP:
            f = sys._getframe(1)
            if f.f_locals.get("self") is not self:
                log.note(kind, name)
        return base.__getattribute__(self, name)

    rec = type("Watched" + base.__name__, (base,), {"__getattribute__": __getattribute__, "__module__": base.__module__})
    object.__setattr__(obj, "__class__", rec)
    return obj


def offered(obj):
    """Every attribute name the object answers to (instance dict, class hierarchy), dunder names excluded."""
    return sorted(n for n in set(dir(obj)) if not (n.startswith("__") and n.endswith("__")))
