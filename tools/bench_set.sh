#!/bin/bash
# Runs on the GPU box: the bench lines this repo tracks -> gpurun_out/bench*.json + a one-line digest each.
export TMPDIR=/tmp
mkdir -p gpurun_out
for sz in 256 512 1024; do timeout -k 10 300 python bench.py --size $sz --steps 20 --warmup 5 --no-cpu > gpurun_out/bench$sz.json 2> gpurun_out/bench$sz.err || exit 1; done
timeout -k 10 300 python bench.py --size 512 --steps 20 --warmup 5 --no-cpu --smoother gs > gpurun_out/bench512gs.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload free_surface_pcg --size 512 > gpurun_out/pcg512.json 2>/dev/null || exit 1
python - <<PY
import json
for sz in ("256","512","1024","512gs"):
    d=json.load(open(f"gpurun_out/bench{sz}.json")); r=d["roofline"]; print(sz, round(d["value"],1), "V/s", round(r["achieved"]), "GB/s", round(r["frac"],3), round(r["ms_per_launch"],4))
d=json.load(open("gpurun_out/pcg512.json")); print(d["tiled_gs"], d["jacobi"])
PY
