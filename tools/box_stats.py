"""Class histogram of the band boxes' region lists (python tools/box_stats.py N [pool]): how many entries a launch mode reads for nothing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import geometricmultigridpressuresolver_amd as G
from geometricmultigridpressuresolver_amd import domains as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pool = len(sys.argv) > 2 and sys.argv[2] == "pool"
levels = {128: 4, 256: 5, 512: 6, 1024: 7}[n]
lab, w, h = (D.free_surface_pool if pool else D.interior_cube)(n, levels)
s = G.GeometricMultigridPoissonSolver(lab, w, levels, False)
names = {0: "skip", 1: "frozen", 2: "zero", 3: "general", 11: "closure-out", 12: "frozen-far (closure mode only)"}
for l in range(2):
    e = s.level_array(l, "box_list")
    info = s.level_array(l, "box_info").reshape(-1, 16)
    nband = len(s.level_array(l, "band"))
    cls = (e >> 16) & 15
    ring = e >> 20
    print(f"level {l}: {len(info)} groups, {len(e)} entries, {nband} band cells, {len(e) / max(nband, 1):.2f} entries per band cell")
    for c in sorted(set(cls.tolist())):
        m = cls == c
        print(f"   class {c:2d} {names.get(c, 'simple band cell, diagonal %d' % (c - 4)):34s} {m.sum():10d} {100 * m.mean():5.1f} %   rings {np.bincount(ring[m], minlength=5)[:6].tolist()}")
