"""MI355X-native geometric multigrid pressure solver: the V-cycle / MG-PCG hot path of
rgoldade/GeometricMultigridPressureSolver as hand-written HIP kernels behind a C ABI
(include/mgps.h, csrc/).  This package is the thin host-side mirror of the reference's solver
interface; importing it loads csrc/libmgps.so and fails loudly if that library is not built."""
from . import domains  # noqa: F401  (numpy only)
from ._lib import MgpsError, build_library, lib  # noqa: F401

lib()  # fail at import time, not at first use, when the HIP library is missing

from .solver import (  # noqa: E402,F401
    GeometricMultigridPoissonSolver,
    Hierarchy,
    build_expanded_domain,
    default_options,
    expanded_layout,
    set_boundary_cell_labels,
    unit_test_boundary_cells,
    unit_test_coarsening,
    unit_test_exterior_cells,
)


def trim_host_cache():
    """Return the page-locked staging blocks the set-up keeps between solvers (mgps_trim_host_cache)."""
    lib().mgps_trim_host_cache.restype = None
    lib().mgps_trim_host_cache()


def trim_device_cache():
    """Return the device blocks released solvers left for the next one (mgps_trim_device_cache)."""
    lib().mgps_trim_device_cache.restype = None
    lib().mgps_trim_device_cache()
