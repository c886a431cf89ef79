#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats and the two PMC passes of bench.py at each size.
# usage: tools/collect_profiles.sh ROUND SIZE [SIZE ...]   -> gpurun_out/prof_<size>_{trace,fetch,write}
set -e
round=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
for sz in "$@"; do
  steps=10; [ "$sz" -ge 1024 ] && steps=4
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${sz}_trace --output-format csv -- python3 bench.py --size $sz --steps $steps --warmup 2 --no-cpu --no-frac512 > gpurun_out/prof_${sz}_trace.json 2> gpurun_out/prof_${sz}_trace.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_${sz}_fetch --output-format csv -- python3 bench.py --size $sz --steps 3 --warmup 1 --no-cpu --no-frac512 > /dev/null 2> gpurun_out/prof_${sz}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_${sz}_write --output-format csv -- python3 bench.py --size $sz --steps 3 --warmup 1 --no-cpu --no-frac512 > /dev/null 2> gpurun_out/prof_${sz}_write.err
  echo "size $sz done"
done
