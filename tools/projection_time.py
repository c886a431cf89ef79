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
stats = len(sys.argv) > 4  # any fourth argument: options.print_stats
t = time.time()
sc = D.projection_scene((base, base, base))
print("scene built in %.1f s" % (time.time() - t), flush=True)
out = []
for r in range(reps):
    vel = [a.copy() for a in sc["velocity"]]
    p = np.zeros((base, base, base), dtype=np.float32)
    t = time.time()
    import geometricmultigridpressuresolver_amd as G
    opt = G.default_options()
    opt.print_stats = int(stats)
    _, info = F.project_free_surface(sc["liquid_phi"], sc["solid_phi"], sc["cut_weights"], vel, p, use_old_pressure=False, use_gauss_seidel=gs, options=opt)
    wall = (time.time() - t) * 1e3
    out.append({k: info[k] for k in ("iterations", "setup_ms", "solve_ms", "total_ms", "mg_levels", "expanded", "liquid_cells", "divergence_max")})
    out[-1]["python_wall_ms"] = wall
    print(json.dumps(out[-1]), flush=True)
