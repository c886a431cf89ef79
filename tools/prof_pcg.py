"""One MG-PCG solve of the N^3 free-surface pool for rocprofv3 (--kernel-trace --stats): python tools/prof_pcg.py N precision gs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
levels = {128: 4, 256: 5, 512: 6, 1024: 7}[n]
lab, w, h = D.free_surface_pool(n, levels)
pad = 2 ** (levels - 1)
b = (D.delta_rhs(lab, n - 2 * pad, pad, h) + D.random_rhs(lab, h)).astype(np.float32)
opt = G.default_options()
opt.precision = prec
s = G.GeometricMultigridPoissonSolver(lab, w, levels, bool(gs), options=opt)
bd = s.to_device(b)
for rep in range(2):
    x = s.new_grid()
    st = s.solveGeometricConjugateGradient(x, bd, 1e-5, 500, True)
print(st)
