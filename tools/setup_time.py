"""Set-up cost of the solver (hierarchy, lists, uploads): python tools/setup_time.py N LEVELS."""
import sys
import time

import torch

import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D

n, lev = int(sys.argv[1]), int(sys.argv[2])
t = time.time(); lab, w, h = D.interior_cube(n, lev); print("domain build", round(time.time() - t, 2))
for gs in (False, True):
    torch.cuda.synchronize(); t = time.time()
    s = G.GeometricMultigridPoissonSolver(lab, w, lev, gs, do_print_stats=True)
    torch.cuda.synchronize(); print("create gs=", gs, round(time.time() - t, 2)); s.close()
wd = [torch.from_numpy(a).cuda() for a in w]
torch.cuda.synchronize(); t = time.time()
s = G.GeometricMultigridPoissonSolver(lab, wd, lev, True)
torch.cuda.synchronize(); print("create gs= True from device weights", round(time.time() - t, 2)); s.close()
t = time.time(); H = G.Hierarchy(lab, lev); print("hierarchy only", round(time.time() - t, 2))
