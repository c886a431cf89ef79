#!/bin/bash
# quick GPU iteration (usage: tools/r4_quick.sh TAG [pytest -k expression]): parity subset, then the bench set
tag=$1; kexpr=${2:-"not 1024 and not full_size and not 512"}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetry.py -x -q -m gpu -k "$kexpr" > gpurun_out/${tag}_tests.log 2>&1 || { tail -40 gpurun_out/${tag}_tests.log; exit 1; }
tail -n 2 gpurun_out/${tag}_tests.log
bash tools/bench_set.sh > gpurun_out/${tag}_set.txt 2>&1; tail -6 gpurun_out/${tag}_set.txt
