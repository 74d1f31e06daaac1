from libdmet_preview_amd.utils import logger  # noqa: F401
from libdmet_preview_amd.utils.misc import max_abs, add_spin_dim, get_spin_dim, mdot, kdot  # noqa: F401
