"""Set-up time of one rank of a P-rank slab run (null transport, one GPU): python tools/slab_setup_time.py N P RANK [host|device]
host (default): the slab's face weights come from numpy arrays (mgps_create_slab_ranges); device: from CUDA tensors
(mgps_create_slab_device_weights: where mgps_fields_* leave them).  Since round 5 the rank is built on the device from its window
of the labels (MGPS_HOST_SETUP=1: the host builder); MGPS_SETUP_TIMING=1 prints the stages."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometricmultigridpressuresolver_amd import domains as D
from geometricmultigridpressuresolver_amd.distributed import SlabSolver
from nullcomm import NullComm


n, P, rank = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
levels = 1
while (n >> (levels - 1)) > 16:
    levels += 1
cuts = [n // P * r for r in range(P + 1)]
z0, z1 = cuts[rank], cuts[rank + 1]
lab, w, h = D.interior_cube_slab(n, levels, z0, z1)
where = sys.argv[4] if len(sys.argv) > 4 else "host"
if where == "device":
    w = [torch.from_numpy(a).cuda() for a in w]
comm = NullComm(rank, P, lab, levels)
comm.prepare()
for rep in range(3):
    torch.cuda.synchronize()
    t = time.time()
    s = SlabSolver(lab, w, levels, False, comm, device=0, splits=cuts)
    torch.cuda.synchronize()
    print("slab set-up N=%d P=%d rank=%d builder=%s weights=%s: %.1f ms" % (n, P, rank, "host" if os.environ.get("MGPS_HOST_SETUP") == "1" else "device", where, (time.time() - t) * 1e3), flush=True)
    s.close()
