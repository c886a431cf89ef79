#!/bin/bash
# A/B of two builds on one box, interleaved (usage: tools/ab_lib.sh TAG LIB_B): the in-tree library (A) against LIB_B
tag=$1; B=$2
export TMPDIR=/tmp
out=gpurun_out/${tag}_ab.txt; : > $out
for rnd in 1 2 3; do
  for v in A B; do
    for sz in 256 512 1024; do
      if [ $v = A ]; then r=$(python3 bench.py --size $sz --steps 30 --warmup 5 --no-cpu --no-frac512 2>/dev/null); else r=$(MGPS_LIBRARY=$PWD/$B python3 bench.py --size $sz --steps 30 --warmup 5 --no-cpu --no-frac512 2>/dev/null); fi
      echo "$r" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $sz', round(d['value'],1), 'band', round(d['stages_ms_per_cycle']['boundary_smoother'],3), 'stage', round(d['band_stage']['ms_per_stage'],4))" >> $out
    done
    if [ $v = A ]; then r=$(python3 bench.py --size 512 --workload free_surface_pcg 2>/dev/null); else r=$(MGPS_LIBRARY=$PWD/$B python3 bench.py --size 512 --workload free_surface_pcg 2>/dev/null); fi
    echo "$r" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v pcg512 jacobi', round(d['jacobi']['solve_ms'],2), 'gs', round(d['tiled_gs']['solve_ms'],2))" >> $out
  done
done
sort $out
