#!/bin/bash
# A/B of environment settings on one box, interleaved: tools/ab_env.sh TAG "A=1 B=2" "A=0" ...  (each argument after TAG: one
# setting, a space-separated VAR=value list or "-" for none).  Zero-guess V-cycles at 512 / 1024 (cube) and MG-PCG on the
# 512^3 free-surface pool
tag=$1; shift
out=gpurun_out/${tag}_ab.txt; : > $out
pj() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
if 'value' in d: print('$1', round(d['value'],2), d['unit'], round(d['ms_per_step'],4))
else: print('$1', 'jacobi', d['jacobi']['iterations'], round(d['jacobi']['solve_ms'],2), 'gs', d['tiled_gs']['iterations'], round(d['tiled_gs']['solve_ms'],2))"; }
for rnd in 1 2; do
  for setting in "$@"; do
    vars=$setting; [ "$setting" = "-" ] && vars=""
    for sz in ${AB_SIZES:-256 512 1024}; do
      env $vars python3 bench.py --size $sz --steps 40 --warmup 5 --no-cpu --no-frac512 --zero-guess 2>/dev/null | pj "[$setting] zero $sz" >> $out
    done
    env $vars python3 bench.py --size 512 --workload free_surface_pcg 2>/dev/null | pj "[$setting] pcg 512" >> $out
  done
done
sort $out
