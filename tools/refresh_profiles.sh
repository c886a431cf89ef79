set -e
# usage (on the GPU box): bash tools/refresh_profiles.sh r02   -> gpurun_out/<round>_*: kernel traces, PMC traffic, summaries
round=${1:-r02}
export TMPDIR=/tmp PYTHONPATH=$PWD
bash tools/collect_profiles.sh $round 256 512 1024
python3 tools/pmcsum.py gpurun_out/${round}_pmc_hbm_traffic.json 256:gpurun_out/prof_256_fetch:gpurun_out/prof_256_write 512:gpurun_out/prof_512_fetch:gpurun_out/prof_512_write 1024:gpurun_out/prof_1024_fetch:gpurun_out/prof_1024_write > gpurun_out/pmcsum.log 2>&1
for sz in 256 512 1024; do python3 tools/profsum.py gpurun_out/prof_${sz}_trace; cp gpurun_out/prof_${sz}_trace/*/*_kernel_stats.csv gpurun_out/${round}_vcycle${sz}_jacobi_kernel_stats.csv; done > gpurun_out/${round}_vcycle_kernel_summary.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pcg --output-format csv -- python3 bench.py --workload free_surface_pcg --size 512 > gpurun_out/${round}_pcg512_free_surface.json 2> gpurun_out/prof_pcg.err
python3 tools/profsum.py gpurun_out/prof_pcg > gpurun_out/${round}_pcg512_kernel_summary.txt
cp gpurun_out/prof_pcg/*/*_kernel_stats.csv gpurun_out/${round}_pcg512_kernel_stats.csv
# tiled Gauss-Seidel: trace + PMC at 512^3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gs_trace --output-format csv -- python3 bench.py --size 512 --steps 10 --warmup 2 --no-cpu --no-frac512 --smoother gs > gpurun_out/${round}_bench512_gs_traced.json 2> gpurun_out/prof_gs.err
python3 tools/profsum.py gpurun_out/prof_gs_trace > gpurun_out/${round}_vcycle512_gs_kernel_summary.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/gs_fetch --output-format csv -- python3 bench.py --size 512 --steps 3 --warmup 1 --no-cpu --no-frac512 --smoother gs > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/gs_write --output-format csv -- python3 bench.py --size 512 --steps 3 --warmup 1 --no-cpu --no-frac512 --smoother gs > /dev/null 2>&1
python3 tools/pmcsum.py gpurun_out/${round}_pmc_gs512.json 512:gpurun_out/gs_fetch:gpurun_out/gs_write > /dev/null
echo refresh-profiles-done
