#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/gs_fetch --output-format csv -- python3 bench.py --size 512 --steps 3 --warmup 1 --no-cpu --smoother gs > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/gs_write --output-format csv -- python3 bench.py --size 512 --steps 3 --warmup 1 --no-cpu --smoother gs > /dev/null 2>&1
python3 tools/pmcsum.py gpurun_out/gs_pmc.json 512:gpurun_out/gs_fetch:gpurun_out/gs_write > /dev/null
python3 - <<PY
import json
d=json.load(open("gpurun_out/gs_pmc.json"))["kernels"]
for k,v in d.items():
    if "tiledGS" in k or "bandFused" in k or "prolong" in k or "restrict" in k: print(k, round(v["traffic_bytes"]/1e6,1),"MB", v["dispatches"])
PY
