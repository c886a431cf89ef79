"""Loader of the C-ABI shared library (csrc/libmgps.so, declared in include/mgps.h).

The library is the product: there is no Python or CPU fallback for any operator.  If it has not
been built, importing this module raises with the build command; if it is built but no HIP device
is visible, mgps_create reports MGPS_ERR_NO_DEVICE.
"""
import ctypes as C
import os
import subprocess

# torch first, always: its wheel bundles its own libamdhip64.so.7 and libmgps.so needs the same
# SONAME.  Loaded in this order the dynamic linker binds libmgps.so to the runtime torch already
# initialised (one HIP runtime per process); in the other order two runtimes coexist and the second
# one sees no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MGPS_LIBRARY") or os.path.join(CSRC, "libmgps.so")  # MGPS_LIBRARY: A/B runs of two builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mgps.h")


def build_library(force=False, verbose=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else []) + ["all"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libmgps.so failed (see output above)")
    return LIB_PATH


class Options(C.Structure):
    """mgps_options (include/mgps.h)."""

    _fields_ = [
        ("struct_size", C.c_int),
        ("band_width", C.c_int),
        ("band_iterations", C.c_int),
        ("jacobi_weight", C.c_float),
        ("device", C.c_int),
        ("print_stats", C.c_int),
        ("max_coarse_unknowns", C.c_int),
        ("fuse_band_passes", C.c_int),
        ("deep_band_halo", C.c_int),
        ("min_cells_per_rank", C.c_int),
        ("pcg_fp64_vectors", C.c_int),
        ("interrupt", C.c_void_p),
        ("interrupt_user", C.c_void_p),
        ("pre_sweeps", C.c_int),
        ("post_sweeps", C.c_int),
        ("stencil_path", C.c_int),
        ("precision", C.c_int),
        ("host_setup", C.c_int),
        ("borrow_device_weights", C.c_int),
    ]


class PcgStats(C.Structure):
    """mgps_pcg_stats (include/mgps.h)."""

    _fields_ = [
        ("outcome", C.c_int),
        ("iterations", C.c_int),
        ("rel_residual", C.c_double),
        ("rel_residual_recomputed", C.c_double),
        ("rhs_norm2", C.c_double),
        ("solve_ms", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libmgps.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C geometricmultigridpressuresolver_amd/csrc` (hipcc, gfx950). There is no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        L.mgps_status_string.restype = C.c_char_p
        L.mgps_last_error.restype = C.c_char_p
        L.mgps_last_error.argtypes = [C.c_void_p]
        L.mgps_hierarchy_band_count.restype = C.c_int64
        L.mgps_get_hierarchy.restype = C.c_void_p
        L.mgps_get_hierarchy.argtypes = [C.c_void_p]
        L.mgps_destroy.argtypes = [C.c_void_p]
        L.mgps_destroy.restype = None
        L.mgps_hierarchy_destroy.argtypes = [C.c_void_p]
        L.mgps_hierarchy_destroy.restype = None
        _lib = L
    return _lib


class MgpsError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"mgps status {status} ({lib().mgps_status_string(status).decode()}): {text}")
        self.status = status


def check(status, handle=None):
    if status != 0:
        raise MgpsError(status, lib().mgps_last_error(handle).decode())
