#!/bin/bash
# A/B of two builds on one box, the 512^3 pool MG-PCG only (usage: tools/ab_pcg.sh TAG LIB_B): the in-tree library (A) against LIB_B, interleaved
tag=$1; B=$2
out=gpurun_out/${tag}_abpcg.txt; : > $out
for rnd in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then r=$(python3 bench.py --size 512 --workload free_surface_pcg 2>/dev/null); else r=$(MGPS_LIBRARY=$PWD/$B python3 bench.py --size 512 --workload free_surface_pcg 2>/dev/null); fi
    echo "$r" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', ' '.join('%s %.2f' % (k, d[k]['solve_ms']) for k in ('jacobi','tiled_gs','jacobi_fp64_iterate','tiled_gs_fp64_iterate')))" >> $out
  done
done
sort $out
