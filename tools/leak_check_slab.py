"""A slab rank created on the device, cycled and destroyed in a loop (null transport, one GPU): device memory must stay flat.
python tools/leak_check_slab.py [N] [P] [RANK] [cycles]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D
from geometricmultigridpressuresolver_amd.distributed import SlabSolver
from nullcomm import NullComm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cycles = int(sys.argv[4]) if len(sys.argv) > 4 else 8
levels = 1
while (n >> (levels - 1)) > 16:
    levels += 1
cuts = [n // P * r for r in range(P + 1)]
z0, z1 = cuts[rank], cuts[rank + 1]
lab, w, h = D.interior_cube_slab(n, levels, z0, z1)
comm = NullComm(rank, P, lab, levels)
comm.prepare()
base = None
for it in range(cycles):
    s = SlabSolver(lab, w, levels, bool(it & 1), comm, device=0, splits=cuts)
    x, b = s.new_grid(), s.new_grid()
    for _ in range(2):
        s.applyVCycle(x, b, False)
    s.close()
    del s, x, b
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    G.trim_device_cache()
    free, total = torch.cuda.mem_get_info()
    used = (total - free) >> 20
    if it == 2:
        base = used
    print(f"cycle {it}: device used {used} MiB", flush=True)
assert used <= base + 64, ("device memory grows", base, used)
print("SLAB_LEAK_CHECK_OK")
