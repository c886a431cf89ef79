"""Set-up cost of the solver, device-side builder against the host builder: python tools/setup_time.py N LEVELS [cube|pool].
Labels and weights are on the device already (what the plugin path has after the field passes)."""
import sys
import time

import torch

import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D

n, lev = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "cube"
t = time.time()
lab, w, h = D.interior_cube(n, lev) if kind == "cube" else D.free_surface_pool(n, lev)
print("domain build", round(time.time() - t, 2), flush=True)
labd = torch.from_numpy(lab).cuda()
wd = [torch.from_numpy(a).cuda() for a in w]
only_device = len(sys.argv) > 5
for gs in (False, True):
    for host in ((0, 0, 0) if only_device else (1, 0, 0, 1, 0)):
        o = G.default_options()
        o.host_setup = host
        torch.cuda.synchronize(); t = time.time()
        s = G.GeometricMultigridPoissonSolver(labd, wd, lev, gs, options=o, do_print_stats=(2 if len(sys.argv) > 4 else 0))
        torch.cuda.synchronize(); print("create gs=%d %s builder: %.1f ms" % (gs, "host  " if host else "device", (time.time() - t) * 1e3), flush=True)
        s.close()
