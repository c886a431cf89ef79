#!/usr/bin/env python3
"""V-cycles per second of the interior cube with nothing else going on (no sweep timer, no stage timers): python tools/vcycle_time.py N [steps] [zero|guess] [sweeps] [levels]
(bench.py keeps the fine-level sweep timer on inside its timed region, which keeps level 0 on the launch-by-launch path)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D

n = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
guess = not (len(sys.argv) > 3 and sys.argv[3] == "zero")
levels = 1
while (n >> (levels - 1)) > 16:
    levels += 1
sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if len(sys.argv) > 5:
    levels = int(sys.argv[5])
lab, w, h = D.interior_cube(n, levels)
opt = G.default_options()
opt.pre_sweeps = opt.post_sweeps = sweeps
s = G.GeometricMultigridPoissonSolver(lab, w, levels, False, options=opt)
b = s.to_device(D.random_rhs(lab, h))
x = s.new_grid()
s.applyVCycle(x, b, False)
for _ in range(5):
    s.applyVCycle(x, b, guess)
best = 0.0
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.applyVCycle(x, b, guess)
    torch.cuda.synchronize()
    best = max(best, steps / (time.perf_counter() - t0))
print(f"{n}^3 L={levels} {'guess' if guess else 'zero'}: {best:.1f} V-cycles/s ({1e3 / best:.4f} ms)")
