#!/usr/bin/env python3
"""Mean of every collected counter per (kernel, grid) from rocprofv3 --pmc output directories.
usage: pmcany.py DIR [DIR ...] [--filter substring]"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(\w+)(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main(dirs, flt):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
                if flt and flt not in k[0]:
                    continue
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg, key=lambda k: (k[0], -k[1])):
        print(f"{k[0]} grid={k[1]}: " + ", ".join(f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(agg[k].items())))


if __name__ == "__main__":
    a = sys.argv[1:]
    flt = None
    if "--filter" in a:
        i = a.index("--filter")
        flt = a[i + 1]
        a = a[:i] + a[i + 2:]
    main(a, flt)
