#!/bin/bash
round=${1:-r05}
# on the GPU box: BASELINE config 3 (512^3 free-surface pool, MG-PCG) per smoother: kernel trace summary, one iteration's launches in
# time order, and the HBM traffic of the band-stage launches (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes)
export TMPDIR=/tmp
mkdir -p gpurun_out
for gs in 1 0; do
  tag=$([ $gs = 1 ] && echo gs || echo jacobi)
  rocprofv3 --kernel-trace -d gpurun_out/pool_${tag}_trace --output-format csv -- python3 tools/prof_pcg.py 512 0 $gs > gpurun_out/pool_${tag}.log 2>&1
  python3 tools/profsum.py gpurun_out/pool_${tag}_trace > gpurun_out/${round}_pcg512_${tag}_kernel_summary.txt
  python3 tools/cycle_timeline.py gpurun_out/pool_${tag}_trace 3 > gpurun_out/${round}_pcg512_${tag}_iteration_timeline.txt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pool_${tag}_fetch --output-format csv -- python3 tools/prof_pcg.py 512 0 $gs > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pool_${tag}_write --output-format csv -- python3 tools/prof_pcg.py 512 0 $gs > /dev/null 2>&1
  python3 tools/pmcsum.py gpurun_out/${round}_pmc_pcg512_${tag}.json 512:gpurun_out/pool_${tag}_fetch:gpurun_out/pool_${tag}_write > /dev/null
  rm -rf gpurun_out/pool_${tag}_trace gpurun_out/pool_${tag}_fetch gpurun_out/pool_${tag}_write
done
python3 tools/box_stats.py 512 pool > gpurun_out/${round}_box_stats_512.txt 2>&1; python3 tools/box_stats.py 512 >> gpurun_out/${round}_box_stats_512.txt 2>&1
echo pool-profile-done
