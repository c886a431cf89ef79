"""A few V-cycles of ONE rank of a P-rank slab run with a transport that moves nothing (for rocprofv3 --kernel-trace):
python tools/slab_rank_cycle.py N P RANK [cycles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometricmultigridpressuresolver_amd import domains as D
from geometricmultigridpressuresolver_amd.distributed import SlabSolver, slab_partition
from nullcomm import NullComm


n, P, rank = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cycles = int(sys.argv[4]) if len(sys.argv) > 4 else 5
levels = 1
while (n >> (levels - 1)) > 16:
    levels += 1
cuts = slab_partition(D.interior_cube_slab(n, levels, 0, 1)[0], levels, P, False)
z0, z1 = cuts[rank], cuts[rank + 1]
lab, w, h = D.interior_cube_slab(n, levels, z0, z1)
s = SlabSolver(lab, w, levels, False, NullComm(rank, P, lab, levels), device=0, splits=cuts)
b = s.to_device(D.random_rhs(lab, h, z0=z0, z1=z1))
x = s.new_grid()
for _ in range(2):
    s.applyVCycle(x, b, True)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(cycles):
    s.applyVCycle(x, b, True)
torch.cuda.synchronize()
print("P=%d rank=%d planes [%d, %d): %.3f ms per cycle" % (P, rank, z0, z1, (time.perf_counter() - t) / cycles * 1e3))
