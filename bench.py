#!/usr/bin/env python3
"""Headline benchmark: V-cycles/sec of the geometric multigrid hot path + achieved HBM GB/s of the
fine-level smoother (BASELINE.json metric), on synthetic interior-liquid cubes.

    python bench.py --gpus N --steps K --warmup W [--size 256|512|1024] [--levels L]
                    [--smoother jacobi|gs] [--no-cpu]

Default workload: BASELINE config 4, the 1024^3 interior-liquid cube (7 levels), on every N, so the
per-N values form the strong-scaling curve the metric asks for; N > 1 splits the grid into Z-slabs
(one rank per GPU, RCCL ghost-plane exchange, coarse tail collapsed to rank 0).

A "step" is one applyVCycle(useInitialGuess=true) on grids that are already
resident in HBM.  Rank 0 prints ONE JSON line.  `roofline` is the fine-level full-domain smoother
(the dominant kernel): algorithmic bytes (13 B per allocated fine cell) over
its mean launch duration measured with HIP events inside the timed region, against the 8 TB/s
HBM3E peak.  `cpu_baseline` is the fp64 CPU oracle (a port of the reference's CPU path, OpenMP)
timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SMOOTHER_BYTES_PER_CELL = 13.0  # x r 4 + b r 4 + label 1 + x' w 4
VCYCLE_BYTES_PER_FINE_CELL = 60.7


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=0, help="solver grid edge N (default 1024 = BASELINE config 4; 256 and 512 are the other metric sizes)")
    ap.add_argument("--levels", type=int, default=0, help="multigrid levels (default: coarsest level 16^3)")
    ap.add_argument("--smoother", choices=["jacobi", "gs"], default="jacobi")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-same-grid", action="store_true",
                    help="also run the OPTIMISED CPU cycle (fp32) on the bench grid itself when the bounded sample is smaller (1024^3: about 60 GB of "
                         "host memory and a few minutes; needs MemAvailable >= 96 GB)")
    ap.add_argument("--no-frac512", action="store_true", help="skip the extra 512^3 smoother measurement (roofline.frac_512)")
    ap.add_argument("--force-slab", action="store_true", help="use the multi-GPU code path (RCCL transport, slab solver) even with one rank")
    ap.add_argument("--even-slabs", action="store_true", help="multi-GPU: nz / N planes per rank instead of cuts balanced by active cells")
    ap.add_argument("--test-fail-rccl", action="store_true", help="test hook: the RCCL transport counts as failed on rank 0; every rank must fall back to the host-staged one")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N-rank path on a box with fewer GPUs: ranks share the GPUs there are, gloo process group, "
                    "host-staged transport (RCCL refuses two ranks on one device).  The line it prints is marked and is not a measurement")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample")
    ap.add_argument("--check-oracle", action="store_true",
                    help="free_surface_pcg only: also solve with the fp64 CPU oracle and report the pressure-field difference")
    ap.add_argument("--sweeps", type=int, default=1, help="full-domain smoother sweeps per stroke (options.pre_sweeps = post_sweeps); "
                    "1 = the reference's schedule, 2 = BASELINE config 1's '2+2'")
    ap.add_argument("--precision", choices=["fp32", "mixed"], default="fp32",
                    help="mixed = BASELINE config 5's storage: the fine level's iterate and residual in binary16 (options.precision = 1)")
    ap.add_argument("--zero-guess", action="store_true",
                    help="time applyVCycle(useInitialGuess=false): the cycle as the CG preconditioner runs it (CG.h:86, 180). The mixed-precision "
                         "cycle is built for this form; from an initial guess it adds an fp32 residual and a correction pass")
    ap.add_argument("--write-check", action="store_true",
                    help="--gpus 1 only: store this configuration's check value (relative residual after three V-cycles) in bench_check.json, "
                         "the table every run -- a first N > 1 run above all -- compares its own value with")
    ap.add_argument("--test-drop-exchange", type=int, default=0,
                    help="test hook (--rehearse-gloo): the host-staged transport drops every N-th exchange during the check cycles; the check must fail")
    ap.add_argument("--workload", choices=["vcycle", "free_surface_pcg"], default="vcycle",
                    help="free_surface_pcg = BASELINE config 3 (single GPU): MG-PCG to 1e-5 on the free-surface pool")
    return ap.parse_args()


def default_levels(n):
    lev = 1
    while (n >> (lev - 1)) > 16:
        lev += 1
    return lev  # 256 -> 5 (BASELINE config 2), 512 -> 6, 1024 -> 7


def cpu_baseline(n, levels, use_gs, budget_s, sweeps=1, same_grid=False):
    """fp64 oracle V-cycle on the host cores.  Bounded: shrink the grid until one V-cycle fits the
    budget; the unit stays V-cycles/sec of *that* grid and the sample string says which."""
    import numpy as np

    from geometricmultigridpressuresolver_amd import domains as D
    from oracle.mg_oracle import Oracle

    orc = Oracle()
    cores = orc.get_threads()
    sample_n, sample_levels = n, levels
    # ~1.3e-8 s per cell per V-cycle per core-ish guess; stay safely inside the budget
    while sample_n > 64 and (sample_n**3) * 6e-8 / max(cores, 1) * 8 > budget_s:
        sample_n //= 2
        sample_levels = max(2, sample_levels - 1)
    lab, w, h = D.interior_cube(sample_n, sample_levels, dtype=np.float64)
    s = orc.solver(lab.astype(np.int32), w, sample_levels, use_gs, pre_sweeps=sweeps, post_sweeps=sweeps)
    b = D.random_rhs(lab, h, dtype=np.float64)
    x = np.zeros_like(b)
    s.apply_vcycle(x, b, False)  # warm-up (page faults, first touch)
    times = []
    t_end = time.time() + budget_s
    while len(times) < 5 and (time.time() < t_end or not times):
        t0 = time.time()
        s.apply_vcycle(x, b, True)
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    out = {
        "value": 1.0 / med,
        "unit": "V-cycles/sec",
        "cores": cores,
        "kind": "port",
        "sample": f"{len(times)} V-cycles of the fp64 OpenMP oracle on a {sample_n}^3 interior cube, {sample_levels} levels, "
        f"{'tiled GS' if use_gs else 'Jacobi'} smoother (median {med*1e3:.1f} ms)",
        "grid": sample_n,
    }
    if sample_n != n:  # not like for like: say what the same port would do on the bench grid (work scales with the cell count)
        out["same_grid"] = {"grid": n, "status": f"not run: the fp64 port needs about {13 * 8 * n**3 / 1e9:.0f} GB at {n}^3",
                            "extrapolated_value": (1.0 / med) * (sample_n / n) ** 3}
    # the optimised CPU variant BASELINE.md section 2 / SURVEY 8(d) ask for next to the faithful port, so that the GPU / CPU ratio is
    # not read off the reference's structural costs: mgo_solver_apply_vcycle_fast (oracle/mg_oracle.c) -- fp32 storage, Jacobi sweeps
    # that ping-pong instead of copying the grid (Ops.h:289), the residual in one pass (Ops.h:728-731), one-byte labels, band passes
    # over precomputed rows, OpenMP over x-rows; same V-cycle, results equal to the faithful cycle's
    try:
        if use_gs or sweeps != 1:
            raise ValueError("the optimised cycle covers the Jacobi smoother with one sweep per stroke")
        o32 = Oracle(f32=True)
        s32 = o32.solver(lab.astype(np.int32), [a.astype(np.float32) for a in w], sample_levels, use_gs)
        b32, x32 = b.astype(np.float32), np.zeros(b.shape, dtype=np.float32)
        s32.apply_vcycle_fast(x32, b32, False)
        t32 = []
        for _ in range(3):
            t0 = time.time()
            s32.apply_vcycle_fast(x32, b32, True)
            t32.append(time.time() - t0)
        out["optimised_variant"] = {"value": 1.0 / sorted(t32)[1], "unit": "V-cycles/sec", "cores": o32.get_threads(), "grid": sample_n,
                                    "what": "fp32 storage, ping-pong Jacobi (no whole-grid copy), one-pass residual, uint8 labels, band rows precomputed, OpenMP"}
        if sample_n != n:
            out["optimised_variant"]["extrapolated_to_bench_grid"] = out["optimised_variant"]["value"] * (sample_n / n) ** 3
    except Exception as e:  # the baseline is a reported extra, never a reason to lose the bench line
        out["optimised_variant"] = {"error": str(e)}
    if same_grid and sample_n != n and "error" not in out["optimised_variant"]:
        # like for like instead of an extrapolation: the optimised cycle on the bench grid itself (outside the default run's budget)
        try:
            avail_gb = next(int(line.split()[1]) for line in open("/proc/meminfo") if line.startswith("MemAvailable")) / 1e6
            need_gb = 60.0 * (n / 1024) ** 3
            if avail_gb < 1.6 * need_gb:
                out["same_grid"]["status"] += f"; optimised cycle not run either: {avail_gb:.0f} GB available, about {need_gb:.0f} GB needed"
            else:
                del lab, w, b, x, s, s32, b32, x32
                labn, wn, hn = D.interior_cube(n, levels, dtype=np.float32)
                sn = o32.solver(labn.astype(np.int32), wn, levels, False)
                bn = D.random_rhs(labn, hn, dtype=np.float32)
                xn = np.zeros_like(bn)
                sn.apply_vcycle_fast(xn, bn, False)
                tn = []
                for _ in range(2):
                    t0 = time.time()
                    sn.apply_vcycle_fast(xn, bn, True)
                    tn.append(time.time() - t0)
                out["same_grid"] = {"grid": n, "status": "run (optimised cycle, fp32; the fp64 port does not fit)", "value": 1.0 / min(tn), "unit": "V-cycles/sec",
                                    "cores": o32.get_threads()}
        except Exception as e:
            out["same_grid"]["status"] += f"; optimised cycle failed: {e}"
    return out


def free_surface_pcg(args):
    """BASELINE config 3: N^3 free-surface pool (sine liquid surface, ghost-fluid weights up to 1/0.01,
    solid box with cut-cell weights), MG-preconditioned CG (tiled GS smoother, as the plugin) to 1e-5 on
    the delta + random rhs.  Not the headline metric; prints its own JSON line."""
    import numpy as np
    import torch

    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    n = args.size or 512
    levels = args.levels or default_levels(n)
    torch.cuda.set_device(0)
    lab, w, h = D.free_surface_pool(n, levels)
    pad = 2 ** (levels - 1)
    b = D.delta_rhs(lab, n - 2 * pad, pad, h) + D.random_rhs(lab, h)
    out = {"metric": "MG-PCG solve", "grid": n, "levels": levels, "tolerance": 1e-5,
           "active_cells": int(D.active_mask(lab).sum()), "general_boundary_note": "ghost-fluid + cut-cell rows"}
    # the two smoothers with fp32 CG vectors, then the same with the CG vectors in fp64 (options.pcg_fp64_vectors)
    # (... and with the iterate alone in fp64, options.pcg_fp64_vectors = 2)
    for use_gs, fp64 in ((True, 0), (False, 0), (True, 2), (False, 2), (True, 1), (False, 1)):
        opt = G.default_options()
        opt.pcg_fp64_vectors = int(fp64)
        solver = G.GeometricMultigridPoissonSolver(lab, w, levels, use_gs, device=0, options=opt)
        bd = solver.to_device(b)
        best = None
        for rep in range(3):
            x = solver.new_grid()
            st = solver.solveGeometricConjugateGradient(x, bd, 1e-5, 2500, True)
            if best is None or st["solve_ms"] < best["solve_ms"]:
                best = st
        key = ("tiled_gs" if use_gs else "jacobi") + {0: "", 1: "_fp64_vectors", 2: "_fp64_iterate"}[fp64]
        out[key] = {k: best[k] for k in ("outcome", "iterations", "rel_residual", "rel_residual_recomputed", "solve_ms")}
        if not fp64:
            # roofline of the whole iteration, Jacobi form: 132 algorithmic B per ACTIVE cell and iteration (DESIGN.md section 11:
            # V-cycle 60.7 B per fine cell x 8/7 for the coarser levels + band stages, A.p + dot 9, update 25, xpay 13) against
            # 8 TB/s; the cells a launch visits beyond the active ones (run ends, padding) are overhead, not work
            gbps = 132.0 * out["active_cells"] * best["iterations"] / (best["solve_ms"] * 1e-3) / 1e9
            out[key]["algorithmic_GBps"] = gbps
            out[key]["frac_of_hbm_peak"] = gbps / HBM_PEAK_GBS
            out[key]["visited_cells_fine_sweep"] = solver.swept_cells(0)[1 if use_gs else 0]
        if use_gs and not fp64 and args.check_oracle:
            x_gpu = x.cpu().numpy().astype(np.float64)
        solver.close()
    if args.check_oracle:  # the same solve in fp64 on the host (checker only, never timed as the product)
        from oracle.mg_oracle import Oracle

        orc = Oracle()
        t0 = time.time()
        ref = orc.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], levels, True)
        x_ref = np.zeros(lab.shape)
        st = ref.solve_pcg(x_ref, b.astype(np.float64), 1e-5, 2500, True)
        out["oracle_fp64"] = {
            "iterations": st["iterations"], "rel_residual_recomputed": st["rel_residual_recomputed"],
            "seconds": time.time() - t0, "threads": orc.get_threads(),
            "pressure_rel_l2_diff_gpu_vs_oracle": float(np.linalg.norm(x_gpu - x_ref) / np.linalg.norm(x_ref)),
            "pressure_rel_max_diff": float(np.abs(x_gpu - x_ref).max() / np.abs(x_ref).max()),
        }
    print(json.dumps(out))


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (one rank per
    GPU, 127.0.0.1 rendezvous) and relay rank 0's JSON line.  This parent never touches the GPU (no torch import, no
    HIP call) and nothing re-execs after GPU initialisation."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in res.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    raise SystemExit(res.returncode if res.returncode else (0 if line else 1))


def main():
    args = parse()
    # multi-process GPU work on this pool's driver needs dmabuf IPC (RCCL fails with "hipIpcGetMemHandle: invalid argument" otherwise);
    # the image exports it already -- kept here for a shell that does not (read when the HIP runtime initialises, children inherit it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.workload == "free_surface_pcg":
        return free_surface_pcg(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (args.gpus > 1 or args.force_slab) and world == 1 and "RANK" not in os.environ:
        return self_launch(args)  # (--force-slab with one rank needs the rendezvous too: the RCCL transport is initialised through it)
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    if args.rehearse_gloo:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    slab_run = world > 1 or args.force_slab  # --force-slab: rehearse the multi-GPU code path with one rank
    # RCCL prints a version banner on stdout when a communicator comes up; stdout must carry the JSON line only
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if slab_run:
        import torch.distributed as dist

        # torch.distributed carries the rendezvous, the barriers and the max over ranks of the timings -- host-side, over gloo.  The
        # data path is the library's own RCCL communicator (RcclComm: ncclSend / ncclRecv over xGMI on the solver's stream), the
        # ONLY RCCL communicator of the process (rounds 2-4 opened torch's "nccl" group beside it: two communicators on one device,
        # a combination that had never run on more than one GPU)
        dist.init_process_group("gloo")

    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    n = args.size or 1024
    levels = args.levels or default_levels(n)
    use_gs = args.smoother == "gs"
    assert n % world == 0 and (n // world) % 16 == 0, "grid planes per rank must be a multiple of 16"
    opt = G.default_options()
    opt.precision = 1 if args.precision == "mixed" else 0
    opt.pre_sweeps = opt.post_sweeps = args.sweeps
    # the cuts: even, or (Jacobi) balanced by active cells -- the EXTERIOR padding leaves the end ranks half empty otherwise
    cuts = [n // world * r for r in range(world + 1)]
    if slab_run and world > 1 and not args.even_slabs:
        from geometricmultigridpressuresolver_amd.distributed import slab_partition

        cuts = slab_partition(D.interior_cube_slab(n, levels, 0, 1)[0], levels, world, use_gs, opt)
    z0, z1 = cuts[rank], cuts[rank + 1]

    # every rank: labels of the whole grid (1 byte per cell), weights and rhs of its own Z-slab only
    lab, w, h = D.interior_cube_slab(n, levels, z0, z1)
    if slab_run:
        from geometricmultigridpressuresolver_amd.distributed import RcclComm, SlabSolver, TorchDistComm

        # before anything is timed: rank-stamped data through every entry of the transport, verified on arrival (mgps_comm_preflight)
        from geometricmultigridpressuresolver_amd.distributed import comm_preflight

        transport, transport_error = ("gloo, host staged (rehearsal)" if args.rehearse_gloo else "rccl"), None
        if args.rehearse_gloo:
            comm = TorchDistComm()
            ranks_seen = comm_preflight(comm, 1 << 16)
        else:
            # RCCL has never carried more than one rank of this code (the pool's boxes have one GPU).  If the communicator does not
            # come up or the preflight sees wrong data on ANY rank, all ranks fall back together to the host-staged transport: the
            # line then still checks the slab code's answer, says so in slab.transport and carries the error -- it is not a
            # measurement of the xGMI path.  (A hang inside librccl cannot be caught this way.)
            comm = None
            try:
                comm = RcclComm(device=local_rank)
                ranks_seen = comm_preflight(comm, 1 << 16)
                if args.test_fail_rccl and rank == 0:
                    raise RuntimeError("--test-fail-rccl")
                if ranks_seen != world:
                    raise RuntimeError(f"preflight all-reduce counted {ranks_seen} ranks of {world}")
            except Exception as e:  # noqa: BLE001 -- whatever went wrong, every rank must hear of it
                transport_error = f"rank {rank}: {e!r}"
            errors = [None] * world
            dist.all_gather_object(errors, transport_error)
            errors = [e for e in errors if e]
            if errors:
                transport_error = "; ".join(errors)[:600]
                print(f"[bench] RCCL transport failed its preflight, falling back to the host-staged transport: {transport_error}", file=sys.stderr)
                transport = "gloo, host staged -- FALLBACK, not a measurement of the RCCL path"
                comm = TorchDistComm()
                ranks_seen = comm_preflight(comm, 1 << 16)
        torch.cuda.synchronize()
        t_setup = time.perf_counter()
        solver = SlabSolver(lab, w, levels, use_gs, comm, device=local_rank, options=opt, splits=cuts)
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t_setup) * 1e3
    else:
        solver = G.GeometricMultigridPoissonSolver(lab, w, levels, use_gs, device=local_rank, options=opt)
    del w
    b = solver.to_device(D.random_rhs(lab, h, z0=z0, z1=z1))
    x = solver.new_grid()

    def barrier():
        torch.cuda.synchronize()
        if slab_run:
            dist.barrier()
            torch.cuda.synchronize()

    solver.applyVCycle(x, b, False)
    guess = not args.zero_guess
    for _ in range(args.warmup):
        solver.applyVCycle(x, b, guess)
    solver.profile_enable(True)
    barrier()
    exchanges0 = solver.exchange_count if slab_run else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.applyVCycle(x, b, guess)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0  # this rank's own clock: its queue drained, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    exchanges_per_cycle = ((solver.exchange_count - exchanges0) / args.steps) if slab_run else 0
    smooth_ms, smooth_groups = solver.profile_read()
    solver.profile_enable(2)  # stage breakdown: a few extra cycles outside the timed region (its event records cost a few %)
    for _ in range(min(args.steps, 5)):
        solver.applyVCycle(x, b, guess)
    stages = solver.stage_times()
    solver.profile_enable(False)
    # ---- does the run compute the right thing?  Relative residual |b - A x| / |b| after three V-cycles from zero (outside the timed
    # region), against the value the single-GPU solver stored for this configuration in bench_check.json (--write-check)
    check_key = f"{n}^3 L{levels} {args.smoother} sweeps{args.sweeps} {args.precision}"
    if args.test_drop_exchange and slab_run and args.rehearse_gloo:
        comm.drop_every = args.test_drop_exchange
    xc, rc = solver.new_grid(), solver.new_grid()
    for it in range(3):
        solver.applyVCycle(xc, b, it > 0)
    solver.computePoissonResidual(rc, xc, b)
    check_value = solver.l2Norm(rc) / solver.l2Norm(b)
    del xc, rc
    table_path = os.path.join(ROOT, "bench_check.json")
    try:
        table = json.load(open(table_path))
    except Exception:
        table = {}
    if args.write_check and not slab_run and rank == 0:
        table[check_key] = check_value
        json.dump(table, open(table_path, "w"), indent=1, sort_keys=True)
    ref = table.get(check_key)
    check = {"what": "|b - A x| / |b| after three V-cycles from zero", "value": check_value, "reference": ref, "reference_from": "bench_check.json (--gpus 1 --write-check)",
             "tolerance_rel": 1e-4,
             "status": "no reference for this configuration" if ref is None else ("ok" if abs(check_value - ref) <= 1e-4 * abs(ref) else "FAILED")}
    slab_diag = None
    if slab_run:  # the job is as slow as its slowest rank
        t = torch.tensor([elapsed, smooth_ms, own_elapsed, setup_ms, float(exchanges_per_cycle)], dtype=torch.float64)
        tmin = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        elapsed, smooth_ms = float(t[0]), float(t[1])
        # what a first hardware run needs to be read: how uneven the ranks are (a rank's own cycle time: its queue drained, before
        # the barrier), what the set-up cost and how many exchanges a cycle issues (middle ranks: two neighbours each)
        slab_diag = {"rank_cycle_ms_max": float(t[2]) / args.steps * 1e3, "rank_cycle_ms_min": float(tmin[2]) / args.steps * 1e3,
                     "setup_ms_max": float(t[3]), "setup_ms_min": float(tmin[3]), "exchanges_per_cycle_max": float(t[4]),
                     "exchanges_per_cycle_min": float(tmin[4]),
                     "distributed_levels": solver.distributed_levels, "ghost_planes": solver.ghost_planes,
                     "band_stage": [solver.band_stage_form(l) for l in range(solver.distributed_levels)],
                     "preflight": "ok", "rccl_ranks_seen": ranks_seen, "transport": transport, "transport_error": transport_error, "check": check}

    cells = float(n) ** 3  # whole job; a rank holds cells / world of them
    active_cells = float(((lab[z0:z1] == 0) | (lab[z0:z1] == 3)).sum())  # this rank's INTERIOR + BOUNDARY cells
    # the sweep skips chunks / blocks / tiles without active cells (as the reference skips constant tiles):
    # algorithmic bytes count the cells a launch actually visits on this rank, not the allocation
    swept = solver.swept_cells(0)[1 if use_gs else 0]
    sweeps_per_group = 1  # Jacobi: one sweep; GS: two half sweeps touch every tile once = one sweep
    t_sweep = smooth_ms * 1e-3 / max(smooth_groups, 1) / sweeps_per_group  # (profile_read counts sweeps: --sweeps 2 doubles the count)
    # mixed precision: the binary16 sweep moves 9 B per cell (iterate 2 + rhs 4 + code 1 + new iterate 2) instead of 13
    sweep_bytes = 9.0 if args.precision == "mixed" else SMOOTHER_BYTES_PER_CELL
    if t_sweep <= 0:
        raise SystemExit("bench.py: the fine-level sweep was not timed (no event pairs came back from mgps_profile_read)")
    achieved = sweep_bytes * swept / t_sweep / 1e9  # per GPU
    vps = args.steps / elapsed
    out = {
        "metric": "V-cycles/sec",
        "value": vps,
        "unit": "V-cycles/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f32 arithmetic, f16 storage of the fine-level iterate and residual",
        "data": ("synthetic; REHEARSAL over gloo with ranks sharing a GPU -- not a measurement" if args.rehearse_gloo else
                 "synthetic; RCCL FAILED ITS PREFLIGHT, host-staged fallback -- not a measurement of the multi-GPU path" if (slab_run and transport_error) else "synthetic"),
        "config": {
            "workload": f"{n}^3 interior-liquid cube, {levels}-level V-cycle, reference schedule "
            f"(3 band Jacobi + {args.sweeps} x {'2 tiled-GS half sweeps' if use_gs else 'damped-Jacobi sweep'} + 3 band Jacobi per stroke), "
            "useInitialGuess=%s, " % ("true" if guess else "false (the preconditioner's form)") + ("fp32 storage" if args.precision == "fp32" else "mixed precision (options.precision = 1)"),
            "grid": n,
            "levels": levels,
            "smoother": "tiled_gs" if use_gs else "jacobi",
            "parallelism": f"zslab{world}",
            "slab_cuts": cuts,
            "distributed_levels": solver.distributed_levels if slab_run else 0,
        },
        **({"slab": slab_diag} if slab_diag else {}),
        "check": check,
        "vcycle_algorithmic_GBps": VCYCLE_BYTES_PER_FINE_CELL * cells * vps / 1e9,
        # the whole cycle against the HBM peak: SURVEY 8(d)'s 60.7 B per fine cell (band passes excluded) x the cells a sweep
        # visits (active runs), per GPU
        "vcycle_frac_visited": VCYCLE_BYTES_PER_FINE_CELL * swept * vps / 1e9 / HBM_PEAK_GBS,
        # device time per cycle by stage, all levels (the reference's stopwatch scopes, MG.cpp:436-878; rank 0's figures)
        "stages_ms_per_cycle": {k: v / max(stages["cycles"], 1) for k, v in stages.items() if k not in ("cycles", "fine")},
        "stages_ms_per_cycle_fine_level": {k: v / max(stages["cycles"], 1) for k, v in stages.get("fine", {}).items()},
        "roofline": {
            "kernel": "fine-level tiled Gauss-Seidel sweep (tiledGSPureKernel + tiledGSMixedKernel, two colours)" if use_gs
            else "fine-level damped-Jacobi sweep (%s<OP_JACOBI>)" % ("stencilPlaneKernel" if solver.stencil_kernel(0) == "plane" else "stencilQuadKernel"),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "ms_per_launch": t_sweep * 1e3,
            "launches": smooth_groups,
            "cells_per_launch": swept,
            "cells_allocated": float(n) * n * (z1 - z0),
            "achieved_over_allocated_cells": sweep_bytes * (float(n) * n * (z1 - z0)) / t_sweep / 1e9,
            # the same launch priced three ways: the cells it visits (frac), the active cells alone (the runs carry row-end
            # padding), every allocated cell (SURVEY 8(d)'s "13 N^3")
            "frac_active": sweep_bytes * active_cells / t_sweep / 1e9 / HBM_PEAK_GBS,
            "frac_allocated": sweep_bytes * (float(n) * n * (z1 - z0)) / t_sweep / 1e9 / HBM_PEAK_GBS,
            "cells_active": active_cells,
            "note": "per GPU; achieved = %g B x cells the launch visits (listed runs / blocks, inside the level's active x range: since round 3 the sweeps "
                    "leave the EXTERIOR padding at the row ends alone, an eighth of every row on the BASELINE cubes, so the cell count is 12.5 %% below "
                    "round 2's for the same grid) / mean launch time (HIP events on the solver's stream)" % sweep_bytes,
        },
    }
    # the band stage of the fine level (launchBandBox: closure + plain launch per stroke, two strokes per cycle): 16 B per band
    # cell algorithmic (iterate in, rhs, iterate out + the closure's Jacobi value), device time from the stage marks
    if not slab_run and not use_gs and args.sweeps == 1:
        try:
            nband = len(solver.level_array(0, "band"))
            band_ms = stages["fine"]["boundary_smoother"] / max(stages["cycles"], 1) / 4.0  # four stages per cycle on level 0
            out["band_stage"] = {
                "kernel": "bandBoxKernel (closure / plain, mean of the four stage launches of the fine level)",
                "bound": "hbm",
                "cells": nband,
                "ms_per_stage": band_ms,
                "achieved": 16.0 * nband / (band_ms * 1e-3) / 1e9 if band_ms > 0 else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": 16.0 * nband / (band_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if band_ms > 0 else None,
                "note": "O(N^2) cells at a line-granular cost: an x-face band row is three cells of one 128-B line per array "
                        "(traffic: profiles/r05_pmc_hbm_traffic.json, bandBoxKernel entries of the fine level, fetched + written per band cell)",
            }
            try:  # measured bytes per band cell of the closure / plain launch at this size, from the committed PMC passes
                kern = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_hbm_traffic.json")))["kernels"]
                fine = {k: v for k, v in kern.items() if k.startswith("bandBoxKernel<float, ") and f"@{n}^3" in k}
                top = max(int(k.split("grid=")[1]) for k in fine) if fine else 0
                for k, v in fine.items():
                    if int(k.split("grid=")[1]) == top and ", false, false, false>" in k:
                        which = "closure" if k.startswith("bandBoxKernel<float, true") else "plain"
                        out["band_stage"][f"traffic_bytes_per_band_cell_{which}"] = v["traffic_bytes"] / nband
            except Exception:
                pass
        except Exception as e:
            out["band_stage"] = {"error": str(e)}
    # HBM traffic of the same kernel at the same size from the committed rocprofv3 PMC passes
    # (profiles/r01_pmc_hbm_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 fetch correction)
    for pmc_file in ("r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["fine_jacobi_sweep"]
            if str(n) in pmc and not use_gs and world == 1 and args.precision == "fp32":
                out["roofline"]["traffic"] = pmc[str(n)]["traffic_bytes"]
                out["roofline"]["traffic_source"] = (
                    f"profiles/{pmc_file}, {pmc[str(n)]['kernel']} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                    "per launch; L2<->fabric bytes incl. Infinity-Cache hits)")
                out["roofline"]["algorithmic_bytes"] = sweep_bytes * swept
                break
        except Exception:
            pass
    # the north-star target is quoted on the 512^3 fine smoother: measure it beside the bench grid (N = 1 default runs)
    if n != 512 and world == 1 and not args.force_slab and not args.no_frac512:
        try:
            solver.close()
            del solver, x, b
            torch.cuda.empty_cache()
            lab5, w5, h5 = D.interior_cube(512, default_levels(512))
            s5 = G.GeometricMultigridPoissonSolver(lab5, w5, default_levels(512), use_gs, device=local_rank, options=opt)
            b5, x5 = s5.to_device(D.random_rhs(lab5, h5)), s5.new_grid()
            s5.applyVCycle(x5, b5, False)
            for _ in range(5):
                s5.applyVCycle(x5, b5, guess)
            s5.profile_enable(True)
            for _ in range(20):
                s5.applyVCycle(x5, b5, guess)
            ms5, groups5 = s5.profile_read()
            swept5 = s5.swept_cells(0)[1 if use_gs else 0]
            t5 = ms5 * 1e-3 / max(groups5, 1)
            out["roofline"]["frac_512"] = sweep_bytes * swept5 / t5 / 1e9 / HBM_PEAK_GBS
            out["roofline"]["ms_per_launch_512"] = t5 * 1e3
            s5.close()
        except Exception as e:
            out["roofline"]["frac_512_error"] = str(e)
    if not args.no_cpu and rank == 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(n, levels, use_gs, args.cpu_seconds, args.sweeps, args.cpu_same_grid)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if slab_run:
        dist.destroy_process_group()
    if check["status"] == "FAILED":
        print(f"bench.py: the check FAILED: {check}", file=sys.stderr, flush=True)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
