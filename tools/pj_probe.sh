#!/bin/bash
export TMPDIR=/tmp
run() { label=$1; sz=$2; shift 2
  r=$(env "$@" python3 bench.py --size $sz --steps 30 --warmup 5 --no-cpu --no-frac512 2>/dev/null)
  echo "$r" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d['stages_ms_per_cycle_fine_level']
print('$label $sz', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'fine:', {k: round(v,3) for k,v in f.items()})"
}
for rnd in 1 2; do
run default 1024 X=1
run plane 1024 MGPS_STENCIL=plane
run plane_fuse 1024 MGPS_STENCIL=plane MGPS_FUSE_PROLONG=1
done
