#!/usr/bin/env python3
"""Compute side of the multi-GPU V-cycle on ONE GPU: rank r of a P-rank slab run with a transport that moves
nothing (exchanges, gather and scatter return at once; the values in the ghost planes are meaningless, the
kernels and launches are exactly those of the real run).  What it gives: the per-rank device time of one cycle
without any communication -- the ceiling of the strong-scaling curve -- for rank 0 (which also runs the collapsed
tail) and a middle rank.   python tools/slab_compute_bound.py [N]"""
import json
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geometricmultigridpressuresolver_amd import domains as D
from geometricmultigridpressuresolver_amd.distributed import SlabSolver, slab_partition
from nullcomm import NullComm


n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
levels = 1
while (n >> (levels - 1)) > 16:
    levels += 1
out = {"grid": n, "levels": levels, "ranks": {}}
even = "--even" in sys.argv
glob_lab = D.interior_cube_slab(n, levels, 0, 1)[0]
for P in (1, 2, 4, 8):
    cuts = [n // P * r for r in range(P + 1)] if even else slab_partition(glob_lab, levels, P, False)
    for rank in sorted({0, P // 2}):
        z0, z1 = cuts[rank], cuts[rank + 1]
        lab, w, h = D.interior_cube_slab(n, levels, z0, z1)
        comm = NullComm(rank, P, lab, levels)
        s = SlabSolver(lab, w, levels, False, comm, device=0, splits=cuts)
        b = s.to_device(D.random_rhs(lab, h, z0=z0, z1=z1))
        x = s.new_grid()
        for _ in range(3):
            s.applyVCycle(x, b, True)
        torch.cuda.synchronize()
        c0, t = comm.calls, time.perf_counter()
        reps = 10
        for _ in range(reps):
            s.applyVCycle(x, b, True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / reps * 1e3
        out["ranks"][f"P={P} rank={rank}"] = {"ms_per_cycle": round(ms, 3), "planes": [z0, z1], "exchanges_per_cycle": (comm.calls - c0) / reps,
                                               "distributed_levels": s.distributed_levels}
        print(f"P={P} rank={rank}: {ms:.3f} ms per cycle, {(comm.calls - c0) / reps:.0f} exchanges, D={s.distributed_levels}", flush=True)
        s.close()
        del s, b, x, lab, w
        torch.cuda.empty_cache()
print(json.dumps(out))
