"""Multi-rank tests of the Z-slab path (one process per rank, torch.distributed.run, 127.0.0.1).

* cpu (gloo, world size 2 and 4): the exchange / collapse schedule restated over the oracle's
  operators (tests/slab_emulation.py) reproduces the whole-grid oracle to round-off, and stops doing so
  when exchanges are dropped.

* gpu: two (and four) ranks share the one GPU of the test box; the slab ranks are set up on the device from their window of
  the labels (round 5; every list against the host builder's), the slab orchestration of libmgps.so (the box form of the
  band stage on cut levels with two list messages per stroke -- or an exchange per band pass with deep_band_halo = 0 -- a
  ghost plane per whole-grid operator that reads across a cut, collapse of the coarse tail to rank 0) runs over
  TorchDistComm/gloo and must reproduce the single-GPU solver on four domains (box, cut-cell solid, free-surface scene,
  random labels).  Only the transport differs from production (RCCL refuses two ranks on one device).
"""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_workers(mode, nproc, timeout, extra_env=None):
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(HERE, "dist_worker.py"), mode,
    ]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.update(extra_env or {})
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, env=env)
    ok = [f"WORKER_OK {r}" in res.stdout for r in range(nproc)]
    assert res.returncode == 0 and all(ok), res.stdout[-4000:]
    return res.stdout


@pytest.mark.gpu
def test_two_slabs_match_single_gpu():
    out = run_workers("gpu", 2, 420)
    print(out[-1500:])


@pytest.mark.gpu
def test_four_slabs_match_single_gpu():
    """Four ranks on the one GPU: the two middle ranks exchange with both neighbours (the 2-rank run has edge
    ranks only), 32 / 16 planes per rank."""
    out = run_workers("gpu", 4, 600)
    print(out[-1500:])


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 4])
def test_slabs_plane_marching_sweep(nproc):
    """stencilPlaneKernel -- the sweep a 1024^3 slab run executes -- on cut slabs (ghost planes below / above): forced
    onto a small free-surface + cut-cell grid with options.stencil_path = 2; edge ranks (2) and middle ranks (4)."""
    out = run_workers("plane", nproc, 420)
    print(out[-800:])


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 4])
def test_slabs_balanced_by_active_cells(nproc):
    """mgps_slab_partition + mgps_create_slab_ranges: slabs of different sizes (equal active cells instead of equal planes),
    the collapse through gatherv / scatterv; Jacobi with the library's cuts, Gauss-Seidel with the caller's own."""
    out = run_workers("balanced", nproc, 420)
    print(out[-800:])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,nproc", [("gpu", 2), ("plane", 2), ("plane", 4)])
def test_slabs_residual_restriction_pair_on_cut_levels(mode, nproc):
    """MGPS_FUSE_RR=1: residual + restriction of a down-stroke as the z-folded pair (residualZKernel + restrictXYKernel) on every
    level that fits -- by size only 4 MiB planes take it -- including CUT levels (round 5): the ranks exchange r on their boundary
    planes and the marches fold the neighbours' planes in as their edge terms.  Must reproduce the whole-grid solver, which takes
    the pair too; the worker asserts that the cut fine level really took it."""
    out = run_workers(mode, nproc, 420, {"MGPS_FUSE_RR": "1", "MGPS_EXPECT_FUSED_RR": "1"})
    print(out[-800:])


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 4])
def test_slab_setup_failure_on_one_rank_reaches_every_rank(nproc):
    """labels that break the BOUNDARY-cell rule on the last rank's planes only: every rank's constructor must return an error --
    the set-up folds each rank's status into an all-reduce before the next collective -- and nobody hangs"""
    out = run_workers("violation", nproc, 300)
    print(out[-600:])


@pytest.mark.gpu
def test_rccl_transport_single_rank():
    """The production transport (librccl through dlopen) with a world of one."""
    run_workers("rccl1", 1, 300)


def run_bench(extra, timeout=600):
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--size", "256", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-frac512"] + extra
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS="2"))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--gpus", "1", "--force-slab"], ["--gpus", "2", "--rehearse-gloo"], ["--gpus", "4", "--rehearse-gloo"]])
def test_bench_slab_line_checks_itself(extra):
    """bench.py --gpus N (round 5): before the timed region every rank pushes rank-stamped data through every entry of the
    transport (slab.preflight, slab.rccl_ranks_seen); after it the relative residual of three V-cycles is compared with the
    value the single-GPU solver stored in bench_check.json (slab.check).  --force-slab: the RCCL transport with one rank;
    --rehearse-gloo: 2 and 4 ranks sharing the GPU over the host-staged transport."""
    import json

    res = run_bench(extra)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    n = int(extra[1])
    assert line["n_gpus"] == n and line["slab"]["preflight"] == "ok" and line["slab"]["rccl_ranks_seen"] == n
    assert line["slab"]["check"]["status"] == "ok" and line["check"]["reference"] is not None, line["slab"]["check"]
    assert line["slab"]["ghost_planes"] == 5 and all(f == "boxes" for f in line["slab"]["band_stage"])


@pytest.mark.gpu
def test_bench_check_catches_a_dropped_exchange():
    """the same line with the transport's test hook dropping every third exchange during the check cycles: the check must
    fail and the process must exit non-zero"""
    import json

    res = run_bench(["--gpus", "2", "--rehearse-gloo", "--test-drop-exchange", "3"])
    assert res.returncode != 0, res.stdout[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["slab"]["check"]["status"] == "FAILED", line["slab"]["check"]


@pytest.mark.gpu
def test_bench_falls_back_when_rccl_fails_its_preflight():
    """RCCL has never carried two ranks of this code.  When its communicator or the preflight fails on any rank, every rank
    switches to the host-staged transport together: the line still checks the slab code's answer and says what happened."""
    import json

    res = run_bench(["--gpus", "1", "--force-slab", "--test-fail-rccl"])
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert "FALLBACK" in line["slab"]["transport"] and "--test-fail-rccl" in line["slab"]["transport_error"], line["slab"]
    assert "not a measurement" in line["data"] and line["slab"]["check"]["status"] == "ok"


@pytest.mark.parametrize("nproc", [2, 4])
def test_slab_schedule_emulation_cpu(nproc):
    out = run_workers("cpu", nproc, 600)
    assert "cpu emulation" in out
