#!/bin/bash
# usage (on the GPU box): bash tools/final_round.sh r02 -> gpurun_out/<round>_*: everything kept under profiles/ for a round
round=${1:-r05}
export TMPDIR=/tmp PYTHONPATH=$PWD
mkdir -p gpurun_out
# (SKIP_REFRESH=1: the kernel traces / PMC passes were collected by a call of their own -- the two halves together pass gpurun's 20 minutes)
if [ -z "$SKIP_REFRESH" ]; then
  bash tools/refresh_profiles.sh $round > gpurun_out/refresh.log 2>&1 || { tail -5 gpurun_out/refresh.log; exit 1; }
  echo "profiles refreshed"
fi
python bench.py > gpurun_out/${round}_bench1024_jacobi_1gpu.json 2> gpurun_out/bench1024.err || exit 1
python bench.py --size 512 > gpurun_out/${round}_bench512_jacobi.json 2>/dev/null || exit 1
python bench.py --size 256 > gpurun_out/${round}_bench256_jacobi.json 2>/dev/null || exit 1
python bench.py --size 512 --smoother gs --no-frac512 > gpurun_out/${round}_bench512_gs.json 2>/dev/null || exit 1
python bench.py --size 256 --smoother gs --no-frac512 > gpurun_out/${round}_bench256_gs.json 2>/dev/null || exit 1
python bench.py --size 128 --levels 4 --sweeps 2 --no-frac512 > gpurun_out/${round}_bench128_L4_2plus2.json 2>/dev/null || exit 1
# BASELINE config 5: the cycle in the preconditioner's form (zero initial guess), fp32 beside mixed precision
for sz in 512 1024; do
  python bench.py --size $sz --zero-guess --no-cpu --no-frac512 --steps 20 --warmup 5 > gpurun_out/${round}_bench${sz}_zeroguess_fp32.json 2>/dev/null || exit 1
  python bench.py --size $sz --zero-guess --precision mixed --no-cpu --no-frac512 --steps 20 --warmup 5 > gpurun_out/${round}_bench${sz}_zeroguess_mixed.json 2>/dev/null || exit 1
done
# config 5 with the plugin's smoother: 512^3 pool MG-PCG, tiled Gauss-Seidel, fp32 against mixed precision
python tools/prof_pcg.py 512 0 1 2>/dev/null | tail -n 1 > gpurun_out/${round}_pcg512_gs_fp32_vs_mixed.txt
python tools/prof_pcg.py 512 1 1 2>/dev/null | tail -n 1 >> gpurun_out/${round}_pcg512_gs_fp32_vs_mixed.txt
# multi-GPU compute ceiling (null transport on one GPU), slab set-up time, one cycle's launches of a middle rank and of rank 0
python tools/slab_compute_bound.py 1024 > gpurun_out/${round}_slab_compute_bound_1024.json 2> gpurun_out/slab_cb.err || echo "slab_compute_bound failed"
python tools/slab_setup_time.py 1024 8 3 host > gpurun_out/${round}_slab_setup_time_1024.txt 2> gpurun_out/slab_st.err || echo "slab_setup_time failed"
python tools/slab_setup_time.py 1024 8 3 device >> gpurun_out/${round}_slab_setup_time_1024.txt 2>> gpurun_out/slab_st.err || echo "slab_setup_time (device) failed"
python tools/slab_setup_time.py 1024 8 0 device >> gpurun_out/${round}_slab_setup_time_1024.txt 2>> gpurun_out/slab_st.err || echo "slab_setup_time (rank 0) failed"
MGPS_SETUP_TIMING=1 python tools/slab_setup_time.py 1024 8 3 device 2>&1 | grep "mgps set-up: slab" | tail -8 >> gpurun_out/${round}_slab_setup_time_1024.txt
MGPS_SETUP_TIMING=1 python tools/slab_setup_time.py 1024 8 0 device 2>&1 | grep "mgps set-up: slab" | tail -8 >> gpurun_out/${round}_slab_setup_time_1024.txt
for r in 4 0; do
  rocprofv3 --kernel-trace -d gpurun_out/slab_rank$r --output-format csv -- python3 tools/slab_rank_cycle.py 1024 8 $r 8 > /dev/null 2>&1
  python3 tools/profsum.py gpurun_out/slab_rank$r > gpurun_out/${round}_slab_rank${r}_of_8_kernel_summary.txt
  python3 tools/cycle_timeline.py gpurun_out/slab_rank$r 3 residualZKernel > gpurun_out/${round}_slab_rank${r}_of_8_cycle_timeline.txt
  rm -rf gpurun_out/slab_rank$r
done
# the --gpus N line with its self-checks: the RCCL transport with one rank, and rehearsals of 2 and 4 ranks over the host-staged transport
python bench.py --gpus 1 --force-slab --no-cpu --no-frac512 --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/${round}_bench1024_force_slab.json
python bench.py --gpus 2 --rehearse-gloo --size 512 --no-cpu --no-frac512 --steps 5 --warmup 2 2>/dev/null | tail -n 1 > gpurun_out/${round}_bench512_rehearse_gloo2.json
python bench.py --gpus 4 --rehearse-gloo --size 512 --no-cpu --no-frac512 --steps 5 --warmup 2 2>/dev/null | tail -n 1 > gpurun_out/${round}_bench512_rehearse_gloo4.json
# BASELINE config 3 where the band stage hurts: kernel summaries, one iteration's launches, PMC traffic of the band-stage launches
bash tools/r5_pool_profile.sh $round > /dev/null 2>&1
tools/kbench 1024 64 2>&1 | head -4 > gpurun_out/${round}_stream_ceilings_1024.txt
# the plugin's own configuration untraced (512^3 pool MG-PCG, every CG vector mode), and the Gauss-Seidel band stage A/B
python bench.py --workload free_surface_pcg --size 512 2>/dev/null | tail -n 1 > gpurun_out/${round}_pcg512_free_surface_untraced.json
python bench.py --workload free_surface_pcg --size 1024 2>/dev/null | tail -n 1 > gpurun_out/${round}_pcg1024_free_surface_untraced.json
for v in 1 0; do MGPS_GS_SNAPSHOT=$v python bench.py --size 512 --smoother gs --no-frac512 --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/${round}_bench512_gs_snapshot$v.json; done
# residual + restriction without the residual grid (1024^3 fine level by size): the pair against the two separate passes
MGPS_FUSE_RR=0 python bench.py --no-cpu --no-frac512 --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/${round}_bench1024_fuse_rr0.json
python bench.py --no-cpu --no-frac512 --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/${round}_bench1024_fuse_rr1.json
# SQ / TCP / TCC counters of the 1024^3 and 512^3 kernels (the kernel the bench line names included)
bash tools/sq_pmc.sh ${round} 1024 > /dev/null 2>&1; bash tools/sq_pmc.sh ${round} 512 > /dev/null 2>&1
tools/facebench 1024 > gpurun_out/${round}_facebench.txt 2>&1; tools/facebench 512 >> gpurun_out/${round}_facebench.txt 2>&1
for sz in 256 512; do
  rocprofv3 --kernel-trace -d gpurun_out/tl_$sz --output-format csv -- python3 bench.py --size $sz --steps 6 --warmup 2 --no-cpu --no-frac512 > /dev/null 2>&1
  python3 tools/cycle_timeline.py gpurun_out/tl_$sz 7 > gpurun_out/${round}_cycle_timeline_$sz.txt  # (a cycle of the timed region: the last five carry the stage timers)
  rm -rf gpurun_out/tl_$sz
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${round}_bench*.json")):
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], round(d["value"], 1), "V/s", round(r["achieved"]), "GB/s frac", round(r["frac"], 3), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
echo final-round-done
