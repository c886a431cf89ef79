#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes (one counter per pass) into profiles/<round>_pmc_hbm_traffic.json.

usage: pmcsum.py OUT.json SIZE:FETCH_DIR:WRITE_DIR [SIZE:FETCH_DIR:WRITE_DIR ...]

FETCH_SIZE / WRITE_SIZE are KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section;
calibrated with tools/kbench copyK / stream3K): FETCH_SIZE reports half of the bytes read, so
traffic = 2 * FETCH_SIZE + WRITE_SIZE.  Besides every (kernel, grid) pair the output names, per size, the
fine-level damped-Jacobi sweep (the stencil kernel with OP_JACOBI = <0> and the largest launch), which is
what bench.py's roofline.traffic quotes.
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(\w+)(<[^>]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def read(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main(out, specs):
    res = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (KB per dispatch, mean over "
        "dispatches). gfx950 correction: FETCH_SIZE reports 1/2 of the bytes read (calibrated on tools/kbench "
        "copyK / stream3K), so traffic = 2*FETCH_SIZE + WRITE_SIZE. L2<->fabric bytes: Infinity-Cache hits included.",
        "kernels": {},
        "fine_jacobi_sweep": {},
    }
    for spec in specs:
        size, fd, wd = spec.split(":")
        fetch, cnt = read(fd, "FETCH_SIZE")
        write, _ = read(wd, "WRITE_SIZE")
        best = None
        for k in sorted(fetch, key=lambda k: -fetch[k]):
            if k not in write or k[0].startswith("void at::") or k[0].startswith("__amd"):
                continue
            e = {
                "FETCH_SIZE_KB": fetch[k],
                "WRITE_SIZE_KB": write[k],
                "traffic_bytes": (2 * fetch[k] + write[k]) * 1024,
                "dispatches": cnt[k],
            }
            res["kernels"][f"{k[0]}@{size}^3 grid={k[1]}"] = e
            if re.match(r"stencil(Quad|Plane)Kernel<0(, *false)?(, *float)?(, *false)?>", k[0]) and (best is None or k[1] > best[0][1] or
                                                                     (k[1] == best[0][1] and False)):
                if best is None or e["traffic_bytes"] > best[1]["traffic_bytes"]:
                    best = (k, e)
        if best:
            res["fine_jacobi_sweep"][size] = dict(best[1], kernel=best[0][0], grid=best[0][1])
    json.dump(res, open(out, "w"), indent=1)
    for s, e in res["fine_jacobi_sweep"].items():
        print(s, e)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
