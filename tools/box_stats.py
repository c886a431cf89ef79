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
    # how the listed cells sit in memory: runs of consecutive cells along x, and the 32-byte sectors / 128-byte lines a group's loads touch
    nz, ny, nx = s.level_shape(l)
    tot_runs = tot_sec = tot_line = tot_entries = 0
    for g in info:
        ent = e[g[2] : g[2] + g[7]]
        if len(ent) == 0:
            continue
        cell = np.int64(g[0]) + (ent & 31).astype(np.int64) + ((ent >> 5) & 31).astype(np.int64) * nx + ((ent >> 10) & 31).astype(np.int64) * nx * ny
        cell.sort()
        tot_runs += 1 + int((np.diff(cell) != 1).sum())
        tot_sec += len(np.unique(cell >> 3))
        tot_line += len(np.unique(cell >> 5))
        tot_entries += len(cell)
    print(f"   listed cells in memory: {tot_entries / max(tot_runs, 1):.2f} cells per x-run, {tot_sec * 32 / max(tot_entries, 1):.1f} B of 32-byte sectors and "
          f"{tot_line * 128 / max(tot_entries, 1):.1f} B of 128-byte lines per listed cell (4 B useful), {tot_entries / max(nband, 1):.2f} listed cells per band cell")
