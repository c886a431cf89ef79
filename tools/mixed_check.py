"""Mixed precision (options.precision = 1) against fp32: V-cycle difference, PCG iteration counts and solve times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D
from conftest import make_domain

def run(name, lab, w, lev, b, tol=1e-5):
    out = {}
    for prec in (0, 1):
        opt = G.default_options()
        opt.precision = prec
        s = G.GeometricMultigridPoissonSolver(lab, w, lev, False, options=opt)
        bd = s.to_device(b)
        x = s.new_grid()
        s.applyVCycle(x, bd, False)
        v1 = x.cpu().numpy().astype(np.float64)
        s.applyVCycle(x, bd, True)
        v2 = x.cpu().numpy().astype(np.float64)
        best = None
        for rep in range(3):
            xs = s.new_grid()
            st = s.solveGeometricConjugateGradient(xs, bd, tol, 500, True)
            if best is None or st["solve_ms"] < best["solve_ms"]:
                best = st
        out[prec] = (v1, v2, best, xs.cpu().numpy().astype(np.float64))
        s.close()
    rl2 = lambda a, c: float(np.linalg.norm(a - c) / np.linalg.norm(c))
    print(f"{name}: vcycle diff {rl2(out[1][0], out[0][0]):.2e} / {rl2(out[1][1], out[0][1]):.2e}  "
          f"pcg fp32 {out[0][2]['iterations']} it {out[0][2]['solve_ms']:.2f} ms (res {out[0][2]['rel_residual']:.2e})  "
          f"mixed {out[1][2]['iterations']} it {out[1][2]['solve_ms']:.2f} ms (res {out[1][2]['rel_residual']:.2e}, outcome {out[1][2]['outcome']})  "
          f"solution diff {rl2(out[1][3], out[0][3]):.2e}", flush=True)

for kind, g in (("simple", 32), ("solid", 64), ("complex", 64)):
    lab, w, off, lev, dx = make_domain(kind, g)
    run(f"{kind}{g} random", lab, w, lev, D.random_rhs(lab, dx))
    run(f"{kind}{g} delta", lab, w, lev, D.delta_rhs(lab, g, off, dx, dtype=np.float32))
for n, levels in ((128, 4), (256, 5), (512, 6)):
    lab, w, h = D.interior_cube(n, levels)
    run(f"cube{n}", lab, w, levels, D.random_rhs(lab, h))
    lab, w, h = D.free_surface_pool(n, levels)
    pad = 2 ** (levels - 1)
    run(f"pool{n}", lab, w, levels, (D.delta_rhs(lab, n - 2 * pad, pad, h) + D.random_rhs(lab, h)).astype(np.float32))
