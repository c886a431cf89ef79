#!/bin/bash
# usage (GPU box): bash tools/dist_fuzz.sh "5 6 7 8" -> the 2- and 4-rank slab tests of tests/dist_worker.py (mode gpu: V-cycle, PCG, A.x, every set-up array
# against the host builder) with other seeds of the random-label domain
for seed in $1; do
  for np in 2 4; do
    MGPS_DIST_SEED=$seed OMP_NUM_THREADS=2 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$np --master-addr 127.0.0.1 --master-port $((29500 + seed * 10 + np)) tests/dist_worker.py gpu > gpurun_out/dist_fuzz_${seed}_${np}.log 2>&1
    echo "seed $seed ranks $np rc $? ok $(grep -c WORKER_OK gpurun_out/dist_fuzz_${seed}_${np}.log)"
  done
done
