#!/bin/bash
# A/B of an environment switch on one box (usage: tools/r4_ab.sh TAG VAR): bench at 256/512/1024 + 512 gs + pcg, VAR unset vs VAR=0, two rounds
tag=$1; var=$2
export TMPDIR=/tmp
out=gpurun_out/${tag}_ab.txt; : > $out
for rnd in 1 2; do
  for val in default 0; do
    for sz in 256 512 1024; do
      if [ $val = default ]; then r=$(python3 bench.py --size $sz --steps 30 --warmup 5 --no-cpu --no-frac512 2>/dev/null); else r=$(env $var=0 python3 bench.py --size $sz --steps 30 --warmup 5 --no-cpu --no-frac512 2>/dev/null); fi
      echo "$r" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$val $sz', round(d['value'],1), 'band', round(d['stages_ms_per_cycle']['boundary_smoother'],3), 'stage', round(d['band_stage']['ms_per_stage'],4))" >> $out
    done
    if [ $val = default ]; then r=$(python3 bench.py --size 512 --workload free_surface_pcg 2>/dev/null); else r=$(env $var=0 python3 bench.py --size 512 --workload free_surface_pcg 2>/dev/null); fi
    echo "$r" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$val pcg512 jacobi', round(d['jacobi']['solve_ms'],2), 'gs', round(d['tiled_gs']['solve_ms'],2))" >> $out
  done
done
sort $out
