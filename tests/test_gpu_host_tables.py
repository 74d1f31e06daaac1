"""
The bit-exact integer bookkeeping tests of tests/test_host_abi.py (k-mesh tables, cell tables, TR weights, visiting
plans incl. the 12 264-event 6x6x6 plan, workload partitions, known answers) are CPU-marked; this module re-runs the
same test functions under the gpu marker so that they are also executed against the libdmetk.so of the GPU box's run
(round-1 verdict, weak point 9).
"""
import importlib.util
import os
import pytest

pytestmark = pytest.mark.gpu

_HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("_host_abi_for_gpu", os.path.join(_HERE, "test_host_abi.py"))
H = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(H)


def test_header_symbols_exported_on_gpu_box():
    H.test_header_symbols_exported()


@pytest.mark.parametrize("tag", H.MESHES)
def test_integer_tables_bit_exact_on_gpu_box(golden, tag):
    H.test_kmesh_tables_bit_exact(golden, tag)
    H.test_eri_plan_bit_exact(golden, tag)


def test_known_answers_and_counts_on_gpu_box():
    H.test_block_counts_match_survey()
    H.test_kpt_member_known_answers()
    H.test_lattice_expand_matches_reference_semantics()


@pytest.mark.parametrize("mesh", [(2, 2, 1), (3, 2, 1), (2, 2, 2)])
def test_general_plan_on_gpu_box(mesh):
    for tr in (True, False):
        H.test_general_plan_matches_reference_loop(mesh, tr)
