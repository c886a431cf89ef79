#!/bin/bash
# A/B of two builds on one box, interleaved: tools/ab_bench.sh TAG LIB_A LIB_B  (bench at 256/512/1024, three rounds each)
tag=$1; A=$2; B=$3
for rnd in 1 2 3; do
  for sz in ${AB_SIZES:-256 512 1024}; do
    for v in A B; do
      lib=$A; [ $v = B ] && lib=$B
      MGPS_LIBRARY=$PWD/$lib python3 bench.py --size $sz --steps 40 --warmup 5 --no-cpu --no-frac512 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $sz', round(d['value'],1), round(d['ms_per_step'],4))" >> gpurun_out/${tag}_ab.txt
    done
  done
done
sort gpurun_out/${tag}_ab.txt
