#!/bin/bash
# quick GPU iteration: set-up + parity subset, bench at the three sizes, kernel traces (usage: tools/r3_quick.sh TAG)
set -e
tag=$1
python -m pytest tests/test_device_setup.py -x -q -m gpu > gpurun_out/${tag}_t1.log 2>&1 || { tail -30 gpurun_out/${tag}_t1.log; exit 1; }
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not 1024 and not full_size and not 512" > gpurun_out/${tag}_t2.log 2>&1 || { tail -30 gpurun_out/${tag}_t2.log; exit 1; }
for sz in 256 512 1024; do python3 bench.py --size $sz --steps 30 --warmup 5 --no-cpu --no-frac512 > gpurun_out/${tag}_bench$sz.json 2> gpurun_out/${tag}_bench$sz.err; done
export TMPDIR=/tmp
for sz in 256 1024; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof_${sz}_trace --output-format csv -- python3 bench.py --size $sz --steps 6 --warmup 2 --no-cpu --no-frac512 > /dev/null 2> gpurun_out/${tag}_prof_${sz}.err
  python3 tools/profsum.py gpurun_out/${tag}_prof_${sz}_trace > gpurun_out/${tag}_profsum_$sz.txt
  rm -rf gpurun_out/${tag}_prof_${sz}_trace
done
tail -n 1 gpurun_out/${tag}_t1.log; tail -n 1 gpurun_out/${tag}_t2.log
