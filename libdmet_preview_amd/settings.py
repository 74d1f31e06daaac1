"""libdmet/settings.py:4-8."""
IMAG_DISCARD_TOL = 1e-7
KPT_DIFF_TOL = 1e-6     # pyscf.pbc.lib.kpts_helper.KPT_DIFF_TOL
