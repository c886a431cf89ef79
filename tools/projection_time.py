"""Where one pressure projection spends its time: python tools/projection_time.py BASE [gs|jacobi] [repeats].
Runs mgps_project_free_surface on the synthetic pool scene of a BASE^3 simulation grid (480 -> 512^3 solver grid) and
prints the call's own clocks (set-up / solve / whole call); MGPS_SETUP_TIMING=1 adds the set-up stages on stdout."""
import json
import sys
import time

import numpy as np

from geometricmultigridpressuresolver_amd import domains as D
from geometricmultigridpressuresolver_amd import fields as F

base = int(sys.argv[1]) if len(sys.argv) > 1 else 480
gs = (sys.argv[2] if len(sys.argv) > 2 else "gs") == "gs"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
stats = len(sys.argv) > 4 and sys.argv[4] == "stats"  # options.print_stats
tight = "tight" in sys.argv[4:]  # mgps_expanded_layout(power_of_two = 0) instead of the reference's power-of-two solver grid
pinned = "pinned" in sys.argv[4:]  # every host array in page-locked memory (mgps_host_alloc), as the Houdini shim stages them
t = time.time()
sc = D.projection_scene((base, base, base))
print("scene built in %.1f s" % (time.time() - t), flush=True)
import ctypes as C
from geometricmultigridpressuresolver_amd._lib import lib
L = lib()
L.mgps_host_alloc.restype = C.c_void_p
L.mgps_host_alloc.argtypes = [C.c_size_t]


def staged(a):
    """a copy of `a` in page-locked memory"""
    if not pinned:
        return a.copy()
    ptr = L.mgps_host_alloc(a.nbytes)
    assert ptr
    out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=a.shape)
    out[...] = a
    return out


if pinned:
    for k in ("liquid_phi", "solid_phi"):
        sc[k] = staged(sc[k])
    sc["cut_weights"] = [staged(a) for a in sc["cut_weights"]]
out = []
vel_s = [staged(a) for a in sc["velocity"]]
p_s = staged(np.zeros((base, base, base), dtype=np.float32))
for r in range(reps):
    vel = vel_s
    for d, a in zip(vel, sc["velocity"]):
        d[...] = a
    p = p_s
    p[...] = 0
    t = time.time()
    import geometricmultigridpressuresolver_amd as G
    opt = G.default_options()
    opt.print_stats = int(stats)
    _, info = F.project_free_surface(sc["liquid_phi"], sc["solid_phi"], sc["cut_weights"], vel, p, use_old_pressure=False, use_gauss_seidel=gs, power_of_two=not tight, options=opt)
    wall = (time.time() - t) * 1e3
    out.append({k: info[k] for k in ("iterations", "setup_ms", "solve_ms", "total_ms", "mg_levels", "expanded", "liquid_cells", "divergence_max")})
    out[-1]["python_wall_ms"] = wall
    print(json.dumps(out[-1]), flush=True)
