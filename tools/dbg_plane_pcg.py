import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D
from conftest import make_domain
from oracle.mg_oracle import Oracle
lab, w, off, lev, dx = make_domain("wide512", 40, 3, (64, 64, 512))
bp = D.random_rhs(lab, dx, seed=4).astype(np.float32)
for f32 in (False, True):
    orc = Oracle(f32=f32)
    s = orc.solver(lab.astype(np.int32), [a.astype(orc.real) for a in w], lev, False)
    x = np.zeros(lab.shape, dtype=orc.real)
    st = s.solve_pcg(x, bp.astype(orc.real), 1e-5, 500, True)
    print("oracle f32=%s" % f32, st["iterations"], st["history"][-4:])
for path in (1, 2):
    for env in ({}, {"scal": 0}):
        opt = G.default_options()
        opt.stencil_path = path
        opt.print_stats = 0
        s = G.GeometricMultigridPoissonSolver(lab, w, lev, False, options=opt)
        x = s.new_grid()
        st = s.solveGeometricConjugateGradient(x, s.to_device(bp), 1e-5, 500, True)
        print("gpu path", path, s.stencil_kernel(0), st["iterations"], st["rel_residual"], st["rel_residual_recomputed"])
        s.close()
