#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid size) count / mean / total."""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(\w+)(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main(d, top=18):
    f = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    print(f"{d}: total kernel time {tot/1e3:.2f} ms")
    for k in sorted(agg, key=lambda k: -sum(agg[k]))[:top]:
        v = agg[k]
        print(f"  {k[0]:34s} grid={k[1]:>10d} n={len(v):4d} avg_us={sum(v)/len(v):8.1f} tot_ms={sum(v)/1e3:7.2f} {100*sum(v)/tot:5.1f}%")


if __name__ == "__main__":
    for d in sys.argv[1:]:
        main(d)
