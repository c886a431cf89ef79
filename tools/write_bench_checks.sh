#!/bin/bash
# on the GPU box: store the check values of the bench configurations in bench_check.json (copied to gpurun_out/ for the way back)
set -e
for args in "--size 128 --levels 4" "--size 128 --levels 4 --sweeps 2" "--size 256" "--size 512" "" "--size 256 --smoother gs" "--size 512 --smoother gs"; do
  python bench.py $args --steps 2 --warmup 1 --no-cpu --no-frac512 --write-check > /dev/null 2> gpurun_out/write_check.err || { tail -5 gpurun_out/write_check.err; exit 1; }
done
cp bench_check.json gpurun_out/bench_check.json
cat bench_check.json
