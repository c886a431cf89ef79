#!/usr/bin/env python3
"""One V-cycle's launches in time order from a rocprofv3 --kernel-trace CSV: start, duration, gap to the previous kernel's end.
usage: cycle_timeline.py TRACE_DIR [which [marker]]   (the cycle = the span between two consecutive coarsest-level solves -- or launches of the
kernel whose name starts with `marker`: a slab rank other than 0 has no coarsest level; which: its index from the end, default 2)"""
import csv
import glob
import sys

from profsum import short


def main(d, which=2, marker=None):
    f = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
    rows = sorted(({"n": short(r["Kernel_Name"]), "g": int(r["Grid_Size_X"]), "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(f))),
                  key=lambda r: r["s"])
    marks = [i for i, r in enumerate(rows) if (r["n"].startswith(marker) if marker else r["n"].startswith("coarseSolve") or r["n"].startswith("coarseMatVec"))]
    if len(marks) < which + 1:
        print("no cycle found")
        return
    a, b = marks[-which - 1], marks[-which]
    seg = rows[a:b]
    t0 = seg[0]["s"]
    busy = sum(r["e"] - r["s"] for r in seg)
    gaps = sum(max(0, seg[i]["s"] - seg[i - 1]["e"]) for i in range(1, len(seg)))
    print(f"{d}: {len(seg)} launches per cycle, span {(rows[b]['s'] - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {gaps / 1e3:.1f} us")
    prev = None
    for r in seg:
        gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
        print(f"  {(r['s'] - t0) / 1e3:8.1f} us  {r['n']:46s} grid={r['g']:>9d}  {(r['e'] - r['s']) / 1e3:7.1f} us  gap {gap:5.1f}")
        prev = r["e"]


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2, sys.argv[3] if len(sys.argv) > 3 else None)
