set -e
export TMPDIR=/tmp PYTHONPATH=$PWD
bash tools/collect_profiles.sh r01 256 512 1024
python3 tools/pmcsum.py gpurun_out/r01_pmc_hbm_traffic.json 256:gpurun_out/prof_256_fetch:gpurun_out/prof_256_write 512:gpurun_out/prof_512_fetch:gpurun_out/prof_512_write 1024:gpurun_out/prof_1024_fetch:gpurun_out/prof_1024_write > gpurun_out/pmcsum.log 2>&1
for sz in 256 512 1024; do python3 tools/profsum.py gpurun_out/prof_${sz}_trace; done > gpurun_out/r01_vcycle_kernel_summary.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pcg --output-format csv -- python3 bench.py --workload free_surface_pcg --size 512 > gpurun_out/prof_pcg.json 2> gpurun_out/prof_pcg.err
python3 tools/profsum.py gpurun_out/prof_pcg > gpurun_out/r01_pcg512_kernel_summary.txt
echo refresh-profiles-done
