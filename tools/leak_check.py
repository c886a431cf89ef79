"""Create / solve / destroy in a loop: device memory (hipMemGetInfo via torch) and host RSS must stay flat.
python tools/leak_check.py [N] [cycles]"""
import resource
import sys

import torch

import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 12
levels = 5 if n <= 256 else 6
lab, w, h = D.free_surface_pool(n, levels)
b = D.random_rhs(lab, h)
base = None
for it in range(cycles):
    for fp64, prec in ((0, 0), (1, 0), (2, 0), (0, 1)):
        opt = G.default_options()
        opt.pcg_fp64_vectors, opt.precision = fp64, prec
        s = G.GeometricMultigridPoissonSolver(lab, w, levels, bool(it & 1) and not prec, options=opt)
        x = s.new_grid()
        st = s.solveGeometricConjugateGradient(x, s.to_device(b), 1e-5, 100, True)
        assert st["outcome"] == "converged"
        s.close()
        del x, s
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    G.trim_device_cache()  # the library keeps released device blocks for the next solver: hand them back before measuring
    free, total = torch.cuda.mem_get_info()
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024
    used = (total - free) >> 20
    if it == 2:
        base = (used, rss)
    print(f"cycle {it}: device used {used} MiB, host max RSS {rss} MiB", flush=True)
assert used <= base[0] + 64, ("device memory grows", base, used)
assert rss <= base[1] + 256, ("host memory grows", base, rss)
print("LEAK_CHECK_OK")
