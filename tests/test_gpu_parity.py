"""GPU parity tests: every HIP operator, the V-cycle and MG-PCG, called through the C ABI
(libmgps.so), against the fp64 CPU oracle on the same seeded inputs.

Tolerances (fp32 storage + fp32 arithmetic on the GPU vs the reference's fp64):
  * single operator pass:      max-abs error <= 5e-6 * max|oracle|
  * one V-cycle:               relative L2 error <= 1e-5
  * reductions (fp64 accum.):  relative error <= 1e-6
  * PCG to 1e-5:               iteration count within +2 of the oracle's, recomputed residual <= 2e-5
  * symmetry <Ma,b> = <Mb,a>:  relative difference <= 1e-4 (the reference's fp64 bound is 1e-10)
"""
import numpy as np
import pytest

from conftest import rel_err, rel_l2

pytestmark = pytest.mark.gpu

OP_TOL = 5e-6
VCYCLE_TOL = 1e-5


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _setup(kind, g, use_gs, domain_factory, oracle, levels=None, solver_shape=None, options=None):
    import geometricmultigridpressuresolver_amd as G

    if kind in ("wide", "odd", "wide512", "widesolid"):
        levels, solver_shape = _wide_args(kind)
    lab, w, off, lev, dx = domain_factory(kind, g, levels, solver_shape)
    gpu = G.GeometricMultigridPoissonSolver(lab, w, lev, use_gs, options=options)
    lab32 = lab.astype(np.int32)
    w64 = [a.astype(np.float64) for a in w]
    orc = oracle.solver(lab32, w64, lev, use_gs)
    return gpu, orc, lab, lab32, w64, off, lev, dx


def _rand_active(lab, seed, scale=1.0):
    from geometricmultigridpressuresolver_amd import domains as D

    rng = np.random.Generator(np.random.PCG64(seed))
    v = rng.random(lab.shape) * scale
    v[~D.active_mask(lab)] = 0
    return v


DOMAINS = [("simple", 32), ("complex", 32), ("solid", 48), ("wide", 24), ("odd", 36), ("random", 4)]


def _wide_args(kind):
    """"wide": a 248 x 24 x 24 box in a 256 x 32 x 32 solver grid (3 levels), a row = one wavefront of the quad sweep
    (by size these grids take the quad kernel; test_plane_sweep_* force the plane-marching one onto them);
    "odd": a 36^3 complex domain in a 44^3 grid (3 levels: 44, 22, 11) -- level 1 has nx % 4 != 0, which
    takes the scalar sweep and the per-cell band list instead of the quad forms;
    "widesolid": free surface + cut-cell solid in a 264 x 40 x 32 grid: ragged block edges in x and y"""
    return {"wide": (3, (32, 32, 256)), "odd": (3, (44, 44, 44)), "wide512": (3, (64, 64, 512)),
            "widesolid": (3, (32, 40, 264))}.get(kind, (None, None))


@pytest.mark.parametrize("kind,g", DOMAINS)
def test_hierarchy_matches_oracle(kind, g, domain_factory, oracle, torch_cuda):
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle)
    assert gpu.getMGLevels() == orc.levels
    H = gpu.hierarchy()
    for l in range(orc.levels):
        assert (H.level_labels(l) == orc.level_labels(l)).all()
        assert (H.band_cells(l) == orc.band(l)).all()


@pytest.mark.parametrize("kind,g", DOMAINS)
def test_fine_level_operators(kind, g, domain_factory, oracle, torch_cuda):
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle)
    x0 = _rand_active(lab, 1)
    b0 = _rand_active(lab, 2, dx * dx)
    band = orc.band(0)

    # applyPoissonMatrix
    y = np.zeros_like(x0)
    oracle.apply_poisson(y, x0, lab32, w64)
    xd, yd = gpu.to_device(x0), gpu.new_grid()
    gpu.applyPoissonMatrix(yd, xd)
    assert rel_err(yd.cpu().numpy(), y) < OP_TOL

    # computePoissonResidual
    r = np.zeros_like(x0)
    oracle.residual(r, x0, b0, lab32, w64)
    bd, rd = gpu.to_device(b0), gpu.new_grid()
    gpu.computePoissonResidual(rd, xd, bd)
    assert rel_err(rd.cpu().numpy(), r) < OP_TOL

    # jacobiPoissonSmoother
    xj = x0.copy()
    oracle.jacobi(xj, b0, lab32, w64)
    xjd = gpu.to_device(x0)
    gpu.jacobiPoissonSmoother(xjd, bd)
    assert rel_err(xjd.cpu().numpy(), xj) < OP_TOL

    # boundaryJacobiPoissonSmoother, three passes
    xb = x0.copy()
    xbd = gpu.to_device(x0)
    for _ in range(3):
        oracle.boundary_jacobi(xb, b0, lab32, band, w64)
        gpu.boundaryJacobiPoissonSmoother(xbd, bd)
    assert rel_err(xbd.cpu().numpy(), xb) < OP_TOL

    # tiledGaussSeidelPoissonSmoother, all four colour/direction combinations in V-cycle order
    xg = x0.copy()
    xgd = gpu.to_device(x0)
    for odd, fwd in ((True, True), (False, True), (False, False), (True, False)):
        oracle.tiled_gs(xg, b0, lab32, odd, fwd, w64)
        gpu.tiledGaussSeidelPoissonSmoother(xgd, bd, odd, fwd)
    assert rel_err(xgd.cpu().numpy(), xg) < 4 * OP_TOL


@pytest.mark.parametrize("kind,g", DOMAINS)
def test_coarse_level_operators(kind, g, domain_factory, oracle, torch_cuda):
    """Levels >= 1 use unit weights (MG.cpp:572-575) and smaller, differently aligned grids."""
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle)
    for l in range(1, orc.levels):
        ll = orc.level_labels(l)
        x0 = _rand_active(ll, 10 + l)
        b0 = _rand_active(ll, 20 + l)
        bd = gpu.to_device(b0, l)
        xj = x0.copy()
        oracle.jacobi(xj, b0, ll)
        xjd = gpu.to_device(x0, l)
        gpu.jacobiPoissonSmoother(xjd, bd, level=l)
        assert rel_err(xjd.cpu().numpy(), xj) < OP_TOL, l
        r = np.zeros_like(x0)
        oracle.residual(r, x0, b0, ll)
        rd = gpu.new_grid(l)
        gpu.computePoissonResidual(rd, gpu.to_device(x0, l), bd, level=l)
        assert rel_err(rd.cpu().numpy(), r) < OP_TOL, l
        xb = x0.copy()
        oracle.boundary_jacobi(xb, b0, ll, orc.band(l))
        xbd = gpu.to_device(x0, l)
        gpu.boundaryJacobiPoissonSmoother(xbd, bd, level=l)
        assert rel_err(xbd.cpu().numpy(), xb) < OP_TOL, l
        xg = x0.copy()
        xgd = gpu.to_device(x0, l)
        for odd, fwd in ((True, True), (False, True), (False, False), (True, False)):
            oracle.tiled_gs(xg, b0, ll, odd, fwd)
            gpu.tiledGaussSeidelPoissonSmoother(xgd, bd, odd, fwd, level=l)
        assert rel_err(xgd.cpu().numpy(), xg) < 4 * OP_TOL, l


@pytest.mark.parametrize("kind,g", DOMAINS)
def test_fused_band_stage(kind, g, domain_factory, oracle, torch_cuda):
    """The band stage of a stroke (three passes fused into one launch on whole-grid levels) against three
    oracle passes and against three single-pass launches of the same library, on every level."""
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle)
    for l in range(orc.levels):
        ll = orc.level_labels(l)
        x0 = _rand_active(ll, 30 + l)
        b0 = _rand_active(ll, 40 + l)
        bd = gpu.to_device(b0, l)
        xb = x0.copy()
        for _ in range(3):
            oracle.boundary_jacobi(xb, b0, ll, orc.band(l), w64 if l == 0 else None)
        fused, single = gpu.to_device(x0, l), gpu.to_device(x0, l)
        gpu.boundaryJacobiStage(fused, bd, level=l)
        for _ in range(3):
            gpu.boundaryJacobiPoissonSmoother(single, bd, level=l)
        assert rel_err(fused.cpu().numpy(), xb) < OP_TOL, l
        # same arithmetic per cell; allow for a different fused-multiply-add contraction only
        assert rel_err(fused.cpu().numpy(), single.cpu().numpy().astype(np.float64)) < 1e-6, l


@pytest.mark.parametrize("kind,g", DOMAINS + [("wide512", 40)])
def test_transfer_operators(kind, g, domain_factory, oracle, torch_cuda):
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle)
    for l in range(orc.levels - 1):
        fl, cl = orc.level_labels(l), orc.level_labels(l + 1)
        fine = _rand_active(fl, 30 + l)
        coarse = np.zeros(cl.shape)
        oracle.downsample(coarse, fine, cl)
        cd = gpu.new_grid(l + 1)
        gpu.downsample(cd, gpu.to_device(fine, l), fine_level=l)
        assert rel_err(cd.cpu().numpy(), coarse) < OP_TOL, l
        assert (cd.cpu().numpy()[~np.isin(cl, (0, 3))] == 0).all()

        csrc = _rand_active(cl, 40 + l)
        fdst = _rand_active(fl, 50 + l)
        ref = fdst.copy()
        oracle.upsample_add(ref, csrc, fl)
        fd = gpu.to_device(fdst, l)
        gpu.upsampleAndAdd(fd, gpu.to_device(csrc, l + 1), fine_level=l)
        assert rel_err(fd.cpu().numpy(), ref) < OP_TOL, l


@pytest.mark.parametrize("kind,g", DOMAINS)
def test_vector_ops_and_reductions(kind, g, domain_factory, oracle, torch_cuda):
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle)
    a = _rand_active(lab, 3) - 0.4 * (lab == 0)
    b = _rand_active(lab, 4)
    ad, bd = gpu.to_device(a), gpu.to_device(b)
    a32, b32 = ad.cpu().numpy().astype(np.float64), bd.cpu().numpy().astype(np.float64)
    assert abs(gpu.dotProduct(ad, bd) - oracle.dot(a32, b32, lab32)) <= 1e-6 * abs(oracle.dot(a32, b32, lab32))
    assert abs(gpu.squaredL2Norm(ad) - oracle.squared_l2(a32, lab32)) <= 1e-6 * oracle.squared_l2(a32, lab32)
    assert abs(gpu.l2Norm(ad) - oracle.l2(a32, lab32)) <= 1e-6 * oracle.l2(a32, lab32)
    assert gpu.infNorm(ad) == pytest.approx(oracle.inf_norm(a32, lab32), rel=1e-7)
    neg = -np.abs(a32) - 1.0 * (np.isin(lab, (0, 3)))
    negd = gpu.to_device(neg)
    assert gpu.infNorm(negd) == 0.0  # the reference's signed max: max(0, max v), Ops.h:1303-1312
    assert gpu.infNorm(negd, reference_signed_max=False) == pytest.approx(np.abs(negd.cpu().numpy()).max(), rel=1e-7)

    ref = a32.copy()
    oracle.add_to_vector(ref, b32, -0.37, lab32)
    gpu.addToVector(ad, bd, -0.37)
    assert rel_err(ad.cpu().numpy(), ref) < 1e-6
    ref2 = np.zeros_like(a32)
    oracle.add_vectors(ref2, a32, b32, 1.7, lab32)
    dd = gpu.new_grid()
    ad = gpu.to_device(a)
    gpu.addVectors(dd, ad, bd, 1.7)
    assert rel_err(dd.cpu().numpy(), ref2) < 1e-6
    gpu.addVectors(bd, ad, bd, 1.7)  # destination aliases the scaled source (CG.h:191)
    assert rel_err(bd.cpu().numpy(), ref2) < 1e-6
    ref3 = a32.copy()
    oracle.scale_vector(ref3, 0.25, lab32)
    gpu.scaleVector(ad, 0.25)
    assert rel_err(ad.cpu().numpy(), ref3) < 1e-6


@pytest.mark.parametrize("kind,g", DOMAINS)
def test_coarse_direct_solve(kind, g, domain_factory, oracle, torch_cuda):
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle)
    L = orc.levels - 1
    cl = orc.level_labels(L)
    b = _rand_active(cl, 60)
    xd = gpu.new_grid(L)
    gpu.coarseDirectSolve(xd, gpu.to_device(b, L))
    x = xd.cpu().numpy().astype(np.float64)
    y = np.zeros_like(x)
    oracle.apply_poisson(y, x, cl)
    assert rel_err(y, b) < 5e-5  # fp32 dense inverse; the elongated "wide" coarse grid is the worst case


@pytest.mark.parametrize("use_gs", [False, True])
def test_tight_expansion_matches_oracle(use_gs, oracle, torch_cuda):
    """the tight (non power-of-two) expansion `mgps_expanded_layout(power_of_two=0)` -- extents
    padded to multiples of 2^levels only (64 x 48 x 80 instead of 64 x 64 x 128 here, 2.1x fewer cells) -- gives
    the same (offset, levels) contract and the same pressure field as the reference-sized grid."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    bl, bw, dx = D.build_complex_domain((40, 24, 56))
    lab_t, w_t, off_t, lev_t = G.build_expanded_domain(bl, bw, 0, power_of_two=False)
    lab_p, w_p, off_p, lev_p = G.build_expanded_domain(bl, bw, 0, power_of_two=True)
    assert (off_t, lev_t) == (off_p, lev_p) and lab_t.size < 0.5 * lab_p.size
    rng = np.random.Generator(np.random.PCG64(3))
    base_b = np.where(bl == D.INTERIOR, rng.random(bl.shape) * dx * dx, 0.0)

    def embed(lab, off):
        b = np.zeros(lab.shape)
        b[off : off + bl.shape[0], off : off + bl.shape[1], off : off + bl.shape[2]] = base_b
        b[~D.active_mask(lab)] = 0
        return b

    fields = []
    for lab, w, off, lev in ((lab_t, w_t, off_t, lev_t), (lab_p, w_p, off_p, lev_p)):
        gpu = G.GeometricMultigridPoissonSolver(lab, w, lev, use_gs)
        b = embed(lab, off)
        xd, bd = gpu.new_grid(), gpu.to_device(b)
        st = gpu.solveGeometricConjugateGradient(xd, bd, 1e-6, 200, True)
        assert st["outcome"] == "converged"
        x = xd.cpu().numpy()
        fields.append(x[off : off + bl.shape[0], off : off + bl.shape[1], off : off + bl.shape[2]].astype(np.float64))
        if lab is lab_t:  # and the tight grid against the fp64 oracle on the same grid
            orc = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], lev, use_gs)
            x_ref = np.zeros(lab.shape)
            so = orc.solve_pcg(x_ref, bd.cpu().numpy().astype(np.float64), 1e-6, 200, True)
            assert abs(so["iterations"] - st["iterations"]) <= 2
            assert rel_l2(x, x_ref) < 2e-5
        gpu.close()
    assert rel_l2(fields[0], fields[1]) < 2e-5  # same solution whatever the padding


@pytest.mark.parametrize("use_gs", [False, True])
@pytest.mark.parametrize("kind,g", DOMAINS + [("wide512", 40)])
def test_vcycle_matches_oracle(kind, g, use_gs, domain_factory, oracle, torch_cuda):
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, use_gs, domain_factory, oracle)
    b = _rand_active(lab, 5, dx * dx)
    x_ref = np.zeros_like(b)
    xd, bd = gpu.new_grid(), gpu.to_device(b)
    b_as_f32 = bd.cpu().numpy().astype(np.float64)
    for it in range(3):
        orc.apply_vcycle(x_ref, b_as_f32, it > 0)
        gpu.applyVCycle(xd, bd, it > 0)
        assert rel_l2(xd.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1), it
    x = xd.cpu().numpy()
    assert (x[~np.isin(lab, (0, 3))] == 0).all()  # zero outside active cells (Ops.h:821-823)


@pytest.mark.parametrize("use_gs", [False, True])
def test_vcycle_symmetry_fp32(use_gs, domain_factory, oracle, torch_cuda):
    """<M a, b> = <M b, a> for M = 4 chained V-cycles (Test.cpp:1808-1875), fp32 bound 1e-4."""
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup("solid", 48, use_gs, domain_factory, oracle)
    a, b = gpu.to_device(_rand_active(lab, 6, dx * dx)), gpu.to_device(_rand_active(lab, 7, dx * dx))
    xa, xb = gpu.new_grid(), gpu.new_grid()
    for it in range(4):
        gpu.applyVCycle(xa, a, it > 0)
        gpu.applyVCycle(xb, b, it > 0)
    da, db = gpu.dotProduct(xa, b), gpu.dotProduct(xb, a)
    assert abs(da - db) / max(abs(da), abs(db)) < 1e-4


def test_single_level_vcycle(domain_factory, oracle, torch_cuda):
    """mgLevels == 1 returns after the fine smoothing stroke (MG.cpp:516-517)."""
    import geometricmultigridpressuresolver_amd as G

    lab, w, off, lev, dx = domain_factory("simple", 32)
    for use_gs in (False, True):
        gpu = G.GeometricMultigridPoissonSolver(lab, w, 1, use_gs)
        orc = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], 1, use_gs)
        b = _rand_active(lab, 8, dx * dx)
        bd = gpu.to_device(b)
        x_ref = np.zeros_like(b)
        orc.apply_vcycle(x_ref, bd.cpu().numpy().astype(np.float64), False)
        xd = gpu.new_grid()
        gpu.applyVCycle(xd, bd, False)
        assert rel_l2(xd.cpu().numpy(), x_ref) < VCYCLE_TOL


@pytest.mark.parametrize("kind,g", [("simple", 64), ("solid", 64)])
@pytest.mark.parametrize("use_gs", [True, False])
def test_mg_pcg_matches_oracle(kind, g, use_gs, domain_factory, oracle, torch_cuda):
    """testConjugateGradient (Test.cpp:675-1009): delta rhs, MG-preconditioned CG to 1e-5."""
    from geometricmultigridpressuresolver_amd import domains as D

    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, use_gs, domain_factory, oracle)
    b = D.delta_rhs(lab, g, off, dx, dtype=np.float32)
    bd = gpu.to_device(b)
    x_ref = np.zeros(lab.shape)
    ref = orc.solve_pcg(x_ref, b.astype(np.float64), 1e-5, 2500, True)
    xd = gpu.new_grid()
    st = gpu.solveGeometricConjugateGradient(xd, bd, 1e-5, 2500, True)
    assert st["outcome"] == "converged"
    assert abs(st["iterations"] - ref["iterations"]) <= 2
    assert st["rel_residual_recomputed"] < 2e-5
    # same pressure field: both solve A x = b to 1e-5; compare in the A-independent relative L2 sense
    assert rel_l2(xd.cpu().numpy(), x_ref) < 2e-4


@pytest.mark.parametrize("kind,g,use_mg", [("solid", 64, True), ("complex", 32, True), ("complex", 32, False)])
@pytest.mark.parametrize("use_gs", [True, False])
def test_pcg_fp64_vectors(kind, g, use_mg, use_gs, domain_factory, oracle, torch_cuda):
    """options.pcg_fp64_vectors: CG vectors in fp64 around the fp32 preconditioner.  Same iteration count as the fp64
    oracle (+-2) and as the fp32-vector solve (+-1); the recomputed residual is then a true fp64 residual of the
    iterate and agrees with the recurrence's; the returned pressure matches the oracle's."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    lab, w, off, lev, dx = domain_factory(kind, g)
    b = (D.delta_rhs(lab, g, off, dx) + D.random_rhs(lab, dx)).astype(np.float32)
    orc = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], lev, use_gs)
    x_ref = np.zeros(lab.shape)
    tol, cap = (1e-6, 2500) if use_mg else (1e-4, 4000)
    ref = orc.solve_pcg(x_ref, b.astype(np.float64), tol, cap, use_mg)
    stats = {}
    for fp64 in (0, 1):
        opt = G.default_options()
        opt.pcg_fp64_vectors = fp64
        s = G.GeometricMultigridPoissonSolver(lab, w, lev, use_gs, options=opt)
        x = s.new_grid()
        stats[fp64] = s.solveGeometricConjugateGradient(x, s.to_device(b), tol, cap, use_mg)
        if fp64:
            x64 = x.cpu().numpy()
        s.close()
    st = stats[1]
    assert st["outcome"] == "converged" and abs(st["iterations"] - ref["iterations"]) <= 2
    assert abs(st["iterations"] - stats[0]["iterations"]) <= (1 if use_mg else 3)
    assert st["rel_residual_recomputed"] < tol and abs(st["rel_residual_recomputed"] - st["rel_residual"]) < 1e-3 * st["rel_residual"]
    assert rel_l2(x64, x_ref) < (2e-5 if use_mg else 2e-3)


@pytest.mark.parametrize("fp64", [0, 1])
def test_pcg_interrupt_callback(fp64, domain_factory, torch_cuda):
    """options.interrupt (the reference polls UT_Interrupt::opInterrupt in every loop, e.g. Ops.h:319): polled before
    every CG iteration and before every level of both strokes of the preconditioning V-cycle; a non-zero answer stops
    the solve with MGPS_ERR_INTERRUPTED and leaves the iterate reached so far in x.  A callback that never fires changes
    nothing."""
    import ctypes as C

    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    lab, w, off, lev, dx = domain_factory("solid", 64)
    b = D.random_rhs(lab, dx)
    polls = []
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p)

    def make(stop_after):
        def cb(user):
            polls.append(1)
            return int(stop_after is not None and len(polls) > stop_after)

        return CB(cb)

    results = {}
    for stop_after in (None, 25):  # 25 polls: about three iterations in
        del polls[:]
        cb = make(stop_after)
        opt = G.default_options()
        opt.pcg_fp64_vectors = fp64
        opt.interrupt = C.cast(cb, C.c_void_p)
        s = G.GeometricMultigridPoissonSolver(lab, w, lev, False, options=opt)
        x = s.new_grid()
        if stop_after is None:
            st = s.solveGeometricConjugateGradient(x, s.to_device(b), 1e-6, 200, True)
            assert st["outcome"] == "converged" and len(polls) > 3 * (st["iterations"] + 1)  # per iteration + per level and stroke
            results["full"] = st["iterations"]
        else:
            with pytest.raises(G.MgpsError) as err:
                s.solveGeometricConjugateGradient(x, s.to_device(b), 1e-6, 200, True)
            assert err.value.status == 9 and len(polls) == stop_after + 1  # MGPS_ERR_INTERRUPTED
            assert float(x.abs().max()) > 0  # the iterations' worth of solution is there
        s.close()
    assert results["full"] > 3


def test_plain_c_caller_flow(domain_factory, torch_cuda):
    """What a C caller without any tensor library does: mgps_device_count, mgps_grid_alloc, mgps_copy_to_device, a
    V-cycle, mgps_copy_to_host, mgps_grid_free -- same numbers as the torch-tensor route."""
    import ctypes as C

    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D
    from geometricmultigridpressuresolver_amd._lib import check, lib

    L = lib()
    count = C.c_int()
    check(L.mgps_device_count(C.byref(count)))
    assert count.value >= 1
    lab, w, off, lev, dx = domain_factory("solid", 64)
    b = np.ascontiguousarray(D.random_rhs(lab, dx), dtype=np.float32)
    s = G.GeometricMultigridPoissonSolver(lab, w, lev, True)
    xd, bd = C.c_void_p(), C.c_void_p()
    check(L.mgps_grid_alloc(s.h, 0, C.byref(xd)), s.h)
    check(L.mgps_grid_alloc(s.h, 0, C.byref(bd)), s.h)
    L.mgps_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mgps_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mgps_apply_vcycle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.mgps_grid_free.argtypes = [C.c_void_p, C.c_void_p]
    check(L.mgps_copy_to_device(s.h, bd, b.ctypes.data_as(C.c_void_p), b.nbytes), s.h)
    for it in range(2):
        check(L.mgps_apply_vcycle(s.h, xd, bd, int(it > 0)), s.h)
    out = np.empty_like(b)
    check(L.mgps_copy_to_host(s.h, out.ctypes.data_as(C.c_void_p), xd, out.nbytes), s.h)
    check(L.mgps_grid_free(s.h, xd), s.h)
    check(L.mgps_grid_free(s.h, bd), s.h)
    xt, bt = s.new_grid(), s.to_device(b)
    for it in range(2):
        s.applyVCycle(xt, bt, it > 0)
    assert np.array_equal(out, xt.cpu().numpy())
    s.close()


@pytest.mark.parametrize("kind,g", [("simple", 64), ("complex", 32)])
@pytest.mark.parametrize("use_gs", [False, True])
def test_fp64_iterate_converges_like_fp64_vectors(kind, g, use_gs, torch_cuda):
    """options.pcg_fp64_vectors = 2 (default): the fp64 iterate with group-wise fp32 updates and residual replacement must converge
    like the all-fp64 loop at tolerances below what fp32 storage of x resolves (1e-6, 1e-7) -- same iteration count to within one,
    recomputed residual under the tolerance.  Round 5 regression: a solve from the zero guess puts the whole solution into its
    first group of updates; without a flush tied to the residual's drop (and a restart of the direction when a replacement jumps)
    the 64^3 box with Gauss-Seidel crawled at 1e-5 for 200 iterations."""
    import geometricmultigridpressuresolver_amd as G
    from conftest import make_domain
    from geometricmultigridpressuresolver_amd import domains as D

    lab, w, off, lev, dx = make_domain(kind, g)
    rhs = D.random_rhs(lab, dx)
    for tol in (1e-6, 1e-7):
        its = {}
        for mode in (1, 2):
            opt = G.default_options()
            opt.pcg_fp64_vectors = mode
            s = G.GeometricMultigridPoissonSolver(lab, w, lev, use_gs, options=opt)
            x = s.new_grid()
            st = s.solveGeometricConjugateGradient(x, s.to_device(rhs), tol, 100, True)
            s.close()
            assert st["outcome"] == "converged" and st["rel_residual_recomputed"] <= tol, (mode, tol, st)
            its[mode] = st["iterations"]
        assert its[2] <= its[1] + 1, (tol, its)


def test_diagonal_pcg(domain_factory, oracle, torch_cuda):
    """useMGPreconditioner off: Jacobi-preconditioned CG (Plug.cpp:485-618)."""
    from geometricmultigridpressuresolver_amd import domains as D

    gpu, orc, lab, lab32, w64, off, lev, dx = _setup("solid", 48, True, domain_factory, oracle)
    b = D.delta_rhs(lab, 48, off, dx, dtype=np.float32)
    x_ref = np.zeros(lab.shape)
    ref = orc.solve_pcg(x_ref, b.astype(np.float64), 1e-5, 2500, False)
    xd = gpu.new_grid()
    st = gpu.solveGeometricConjugateGradient(xd, gpu.to_device(b), 1e-5, 2500, False)
    assert st["outcome"] == "converged"
    assert abs(st["iterations"] - ref["iterations"]) <= max(3, ref["iterations"] // 20)
    assert st["rel_residual_recomputed"] < 2e-5


def test_pcg_early_outs(domain_factory, oracle, torch_cuda):
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup("simple", 32, True, domain_factory, oracle)
    xd, zd = gpu.new_grid(), gpu.new_grid()
    st = gpu.solveGeometricConjugateGradient(xd, zd, 1e-5, 10, True)
    assert st["outcome"] == "rhs_zero"  # CG.h:36-40
    b = _rand_active(lab, 9, dx * dx)
    bd = gpu.to_device(b)
    st = gpu.solveGeometricConjugateGradient(xd, bd, 1e-5, 100, True)
    assert st["outcome"] == "converged"
    st2 = gpu.solveGeometricConjugateGradient(xd, bd, 1e-4, 100, True)
    assert st2["outcome"] == "already_converged" and st2["iterations"] == 0  # CG.h:60-64


def test_zero_inactive_restores_the_invariant(domain_factory, oracle, torch_cuda):
    """An initial guess that carries values in air / solid cells changes the operator of the cells next to them (the
    sweeps sum neighbours unmasked, relying on the zero invariant of Ops.h:821-823); mgps_zero_inactive restores the
    invariant and the solve then equals the one from the clean guess."""
    from geometricmultigridpressuresolver_amd import domains as D

    gpu, orc, lab, lab32, w64, off, lev, dx = _setup("solid", 48, True, domain_factory, oracle)
    b = gpu.to_device(D.random_rhs(lab, dx))
    guess = _rand_active(lab, 12)
    dirty = guess + 5.0 * (~D.active_mask(lab))
    xc, xd = gpu.to_device(guess), gpu.to_device(dirty)
    gpu.zeroInactive(xd)
    assert np.array_equal(xd.cpu().numpy(), xc.cpu().numpy())
    gpu.applyVCycle(xc, b, True)
    gpu.applyVCycle(xd, b, True)
    assert np.array_equal(xd.cpu().numpy(), xc.cpu().numpy())


def test_host_buffer_forms(domain_factory, oracle, torch_cuda):
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup("complex", 32, True, domain_factory, oracle)
    b = _rand_active(lab, 11, dx * dx).astype(np.float32)
    x = gpu.applyVCycleHost(np.zeros_like(b), b, False)
    x_ref = np.zeros(lab.shape)
    orc.apply_vcycle(x_ref, b.astype(np.float64), False)
    assert rel_l2(x, x_ref) < VCYCLE_TOL
    xs, st = gpu.solvePcgHost(np.zeros_like(b), b, 1e-5, 100, True)
    assert st["outcome"] == "converged"
    # the double-precision host forms (the reference's StoreReal): same numbers as the float forms of the same values
    x64 = gpu.applyVCycleHost(np.zeros(b.shape, dtype=np.float64), b.astype(np.float64), False)
    assert x64.dtype == np.float64 and np.array_equal(x64.astype(np.float32), x)
    xs64, st64 = gpu.solvePcgHost(np.zeros(b.shape, dtype=np.float64), b.astype(np.float64), 1e-5, 100, True)
    assert st64["iterations"] == st["iterations"] and np.array_equal(xs64.astype(np.float32), xs)


def test_distinct_handles_are_independent(domain_factory, torch_cuda):
    """The boundary's threading rule: a handle is not thread-safe, distinct handles are
    independent.  Two host threads create, use and destroy their own solvers at the same time, each on its own
    stream; both reproduce what they compute alone (set-up runs its own helper threads: they must not meet)."""
    import threading

    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    torch = torch_cuda
    jobs = [("simple", 64, False), ("solid", 64, True)]
    inputs = [domain_factory(kind, g) for kind, g, _ in jobs]

    def run(idx, out):
        lab, w, off, lev, dx = inputs[idx]
        with torch.cuda.stream(torch.cuda.Stream()):
            s = G.GeometricMultigridPoissonSolver(lab, w, lev, jobs[idx][2])
            b = s.to_device(D.random_rhs(lab, dx))
            x = s.new_grid()
            for it in range(3):
                s.applyVCycle(x, b, it > 0)
            xp = s.new_grid()
            st = s.solveGeometricConjugateGradient(xp, b, 1e-6, 100, True)
            s.synchronize()
            out[idx] = (x.cpu().numpy(), xp.cpu().numpy(), st["iterations"])
            s.close()

    alone = [None, None]
    for i in range(2):
        run(i, alone)
    for _ in range(3):
        together = [None, None]
        threads = [threading.Thread(target=run, args=(i, together)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for i in range(2):
            assert together[i] is not None
            assert np.array_equal(together[i][0], alone[i][0]) and np.array_equal(together[i][1], alone[i][1])
            assert together[i][2] == alone[i][2]


def test_gathered_dot_matches_separate_reduction(torch_cuda):
    """MG-PCG takes <z, r> from the last stroke of the preconditioning V-cycle (sweep partial sums + band-scatter
    corrections).  With MGPS_CHECK_FUSED_DOT=1 (read once per process, hence the child process) the library compares
    every such value with a separate reduction and fails the solve on a relative difference above 1e-9: both
    smoothers, unit-weight and cut-cell / ghost-fluid domains."""
    import subprocess
    import sys

    from conftest import ROOT

    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np\n"
        "import geometricmultigridpressuresolver_amd as G\n"
        "from geometricmultigridpressuresolver_amd import domains as D\n"
        "from conftest import make_domain\n"
        "for kind, g in (('simple', 64), ('solid', 64), ('complex', 32)):\n"
        "    lab, w, off, lev, dx = make_domain(kind, g)\n"
        "    for gs in (False, True):\n"
        "        s = G.GeometricMultigridPoissonSolver(lab, w, lev, gs)\n"
        "        x = s.new_grid(); b = s.to_device(D.random_rhs(lab, dx))\n"
        "        st = s.solveGeometricConjugateGradient(x, b, 1e-6, 200, True)\n"
        "        assert st['outcome'] == 'converged', st\n"
        "        s.close()\n"
        "print('GATHER_OK')\n"
    ) % (ROOT, ROOT)
    env = dict(**__import__("os").environ, MGPS_CHECK_FUSED_DOT="1")
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "GATHER_OK" in res.stdout, res.stdout[-3000:]


def test_convergence_trace(domain_factory, oracle, torch_cuda):
    """testOneLevelVCycle (Test.cpp:1877-1960): b = 0, sine error, Jacobi V-cycles with
    useInitialGuess; the error norm must contract monotonically and track the oracle's trace."""
    from geometricmultigridpressuresolver_amd import domains as D

    gpu, orc, lab, lab32, w64, off, lev, dx = _setup("simple", 64, False, domain_factory, oracle)
    x0 = D.sine_initial_guess(lab, dx)
    xd, zd = gpu.to_device(x0), gpu.new_grid()
    x_ref, z = x0.astype(np.float64), np.zeros(lab.shape)
    prev = gpu.l2Norm(xd)
    for it in range(12):
        gpu.applyVCycle(xd, zd, True)
        orc.apply_vcycle(x_ref, z, True)
        cur = gpu.l2Norm(xd)
        assert cur < prev
        assert cur == pytest.approx(oracle.l2(x_ref, lab32), rel=1e-3)
        prev = cur


def test_vcycle_512_matches_oracle(oracle, torch_cuda):
    """The 512^3 roofline size against the fp64 oracle itself (about 40 s of host time): two chained V-cycles on
    the interior cube, solution and residual-norm trace."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    n, levels = 512, 6
    lab, w, h = D.interior_cube(n, levels)
    gpu = G.GeometricMultigridPoissonSolver(lab, w, levels, False)
    b32 = D.random_rhs(lab, h, seed=1)
    bd, xd, rd = gpu.to_device(b32), gpu.new_grid(), gpu.new_grid()
    lab32 = lab.astype(np.int32)
    orc = oracle.solver(lab32, [a.astype(np.float64) for a in w], levels, False)
    b64 = b32.astype(np.float64)
    x_ref, r_ref = np.zeros(lab.shape), np.zeros(lab.shape)
    for it in range(2):
        gpu.applyVCycle(xd, bd, it > 0)
        orc.apply_vcycle(x_ref, b64, it > 0)
        assert rel_l2(xd.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1)
        gpu.computePoissonResidual(rd, xd, bd)
        oracle.residual(r_ref, x_ref, b64, lab32, [a.astype(np.float64) for a in w])
        assert gpu.l2Norm(rd) == pytest.approx(float(np.sqrt((r_ref * r_ref).sum())), rel=1e-4)
    gpu.close()


@pytest.mark.parametrize("n,levels", [(256, 5), (512, 6), (1024, 7)])
def test_full_size_properties(n, levels, torch_cuda):
    """BASELINE configs 2 and 4 and the 512^3 roofline size (interior cubes, coarsest level 16^3) through
    size-independent properties: the oracle is too slow for routine full-size comparison, so check linearity
    of the V-cycle, zero-outside-active, symmetry and contraction at the benchmark sizes themselves."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    lab, w, h = D.interior_cube(n, levels)
    for use_gs in (False, True) if n < 1024 else (False,):
        gpu = G.GeometricMultigridPoissonSolver(lab, w, levels, use_gs)
        a = gpu.to_device(D.random_rhs(lab, h, seed=1))
        b = gpu.to_device(D.random_rhs(lab, h, seed=2))
        xa, xb, xab = gpu.new_grid(), gpu.new_grid(), gpu.new_grid()
        gpu.applyVCycle(xa, a)
        gpu.applyVCycle(xb, b)
        ab = a + 2.0 * b
        gpu.applyVCycle(xab, ab)
        lin = (xab - (xa + 2.0 * xb)).norm().item() / xab.norm().item()
        assert lin < 5e-6  # M(a + 2b) = M a + 2 M b
        da, db = gpu.dotProduct(xa, b), gpu.dotProduct(xb, a)
        assert abs(da - db) / max(abs(da), abs(db)) < 1e-4
        inactive = gpu.to_device((~D.active_mask(lab)).astype(np.float32))
        assert (xa * inactive).abs().max().item() == 0.0
        r = gpu.new_grid()
        norms = [gpu.l2Norm(a)]
        # chained V-cycles contract the residual (asymptotic factor ~0.7 with one Jacobi sweep per stroke); on
        # white-noise rhs the first cycles barely move its L2 norm (1.12 |b| after one cycle at 512^3, flat
        # over the first two at 1024^3 -- the fp64 oracle shows the same trace, test_vcycle_512_matches_oracle),
        # so monotonicity is required from the third cycle on
        for it in range(5):
            if it:
                gpu.applyVCycle(xa, a, True)
            gpu.computePoissonResidual(r, xa, a)
            norms.append(gpu.l2Norm(r))
            assert it < 2 or norms[-1] < norms[-2]
        assert norms[-1] < 0.6 * norms[0]
        gpu.close()
        del gpu, a, b, xa, xb, xab, ab, r, inactive
        torch_cuda.cuda.empty_cache()


# ---- the plane-marching sweep (stencilPlaneKernel): by size it only runs where an x-y plane exceeds 2 MiB (1024^2),
# which no oracle-sized domain reaches; options.stencil_path = 2 puts it on every level whose shape allows it
PLANE_DOMAINS = [("wide", 24), ("wide512", 40), ("widesolid", 24)]


def _plane_options():
    import geometricmultigridpressuresolver_amd as G

    opt = G.default_options()
    opt.stencil_path = 2
    return opt


@pytest.mark.parametrize("kind,g", PLANE_DOMAINS)
def test_plane_sweep_operators_match_oracle(kind, g, domain_factory, oracle, torch_cuda):
    """stencilPlaneKernel<JACOBI | RESIDUAL | APPLY> (Ops.h:262-367, 621-732) against the oracle, 5e-6, and bit for bit
    against the quad kernel of the same library (same arithmetic per cell)."""
    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle, options=_plane_options())
    quad, *_ = _setup(kind, g, False, domain_factory, oracle)
    assert gpu.stencil_kernel(0) == "plane" and quad.stencil_kernel(0) == "quad"
    if kind == "widesolid":
        assert int((gpu.hierarchy().level_labels(0) == 3).sum()) > 1000 and any(((a != 0) & (a != 1)).any() for a in w64)
    x0 = _rand_active(lab, 1)
    b0 = _rand_active(lab, 2, dx * dx)
    xd, bd = gpu.to_device(x0), gpu.to_device(b0)
    for name, ref_fn, run in (
        ("apply", lambda out: oracle.apply_poisson(out, x0, lab32, w64), lambda s, out: s.applyPoissonMatrix(out, xd)),
        ("residual", lambda out: oracle.residual(out, x0, b0, lab32, w64), lambda s, out: s.computePoissonResidual(out, xd, bd)),
    ):
        ref = np.zeros_like(x0)
        ref_fn(ref)
        out_p, out_q = gpu.new_grid(), quad.new_grid()
        out_p.fill_(7.0)  # the public forms define every cell of the destination
        run(gpu, out_p)
        run(quad, out_q)
        assert rel_err(out_p.cpu().numpy(), ref) < OP_TOL, name
        assert np.array_equal(out_p.cpu().numpy(), out_q.cpu().numpy()), name
    xj = x0.copy()
    oracle.jacobi(xj, b0, lab32, w64)
    xp, xq = gpu.to_device(x0), quad.to_device(x0)
    gpu.jacobiPoissonSmoother(xp, bd)
    quad.jacobiPoissonSmoother(xq, bd)
    assert rel_err(xp.cpu().numpy(), xj) < OP_TOL
    assert np.array_equal(xp.cpu().numpy(), xq.cpu().numpy())


@pytest.mark.parametrize("kind,g", PLANE_DOMAINS)
def test_plane_sweep_vcycle_and_pcg_match_oracle(kind, g, domain_factory, oracle, torch_cuda):
    """V-cycles (activity-skipping plane blocks, solver-owned grids) and MG-PCG through the plane-marching sweep:
    the PCG takes <p, A p> from stencilPlaneKernel<APPLY, DOT> and <z, r> from stencilPlaneKernel<JACOBI, DOT>."""
    from geometricmultigridpressuresolver_amd import domains as D

    gpu, orc, lab, lab32, w64, off, lev, dx = _setup(kind, g, False, domain_factory, oracle, options=_plane_options())
    assert gpu.stencil_kernel(0) == "plane"
    b = _rand_active(lab, 5, dx * dx)
    x_ref = np.zeros_like(b)
    xd, bd = gpu.new_grid(), gpu.to_device(b)
    b_as_f32 = bd.cpu().numpy().astype(np.float64)
    for it in range(3):
        orc.apply_vcycle(x_ref, b_as_f32, it > 0)
        gpu.applyVCycle(xd, bd, it > 0)
        assert rel_l2(xd.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1), it
    assert (xd.cpu().numpy()[~np.isin(lab, (0, 3))] == 0).all()
    # tolerance 1e-4: on the elongated 512 x 64 x 64 box with only 3 levels fp32 CG vectors reach their rounding floor
    # near 1e-5 (the fp32 build of the oracle needs 19 iterations there against 16 in fp64, like both GPU kernels)
    bp = (D.random_rhs(lab, dx, seed=4)).astype(np.float32)
    x_ref = np.zeros(lab.shape)
    ref = orc.solve_pcg(x_ref, bp.astype(np.float64), 1e-4, 500, True)
    xs = gpu.new_grid()
    st = gpu.solveGeometricConjugateGradient(xs, gpu.to_device(bp), 1e-4, 500, True)
    assert st["outcome"] == "converged" and abs(st["iterations"] - ref["iterations"]) <= 2
    assert rel_l2(xs.cpu().numpy(), x_ref) < 1e-3
    quad, *_ = _setup(kind, g, False, domain_factory, oracle)  # same arithmetic per cell and the same summation order per block? no:
    xq = quad.new_grid()                                      # the dot partials are grouped differently, so compare to round-off
    sq = quad.solveGeometricConjugateGradient(xq, quad.to_device(bp), 1e-4, 500, True)
    assert sq["iterations"] == st["iterations"] and rel_l2(xs.cpu().numpy(), xq.cpu().numpy().astype(np.float64)) < 1e-5


def test_plane_sweep_gathered_dots(torch_cuda):
    """MGPS_CHECK_FUSED_DOT=1 (read once per process, hence the child) with the plane-marching sweep forced: every
    <z, r> gathered from stencilPlaneKernel<JACOBI, DOT> + band scatters must equal a separate reduction to 1e-9."""
    import subprocess
    import sys

    from conftest import ROOT

    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import numpy as np\n"
        "import geometricmultigridpressuresolver_amd as G\n"
        "from geometricmultigridpressuresolver_amd import domains as D\n"
        "from conftest import make_domain\n"
        "for kind, g, shape in (('wide', 24, (32, 32, 256)), ('widesolid', 24, (32, 40, 264))):\n"
        "    lab, w, off, lev, dx = make_domain(kind, g, 3, shape)\n"
        "    s = G.GeometricMultigridPoissonSolver(lab, w, lev, False)\n"
        "    assert s.stencil_kernel(0) == 'plane'\n"
        "    x = s.new_grid(); b = s.to_device(D.random_rhs(lab, dx))\n"
        "    st = s.solveGeometricConjugateGradient(x, b, 1e-6, 200, True)\n"
        "    assert st['outcome'] == 'converged', st\n"
        "    s.close()\n"
        "print('GATHER_OK')\n"
    ) % (ROOT, ROOT)
    env = dict(**__import__("os").environ, MGPS_CHECK_FUSED_DOT="1", MGPS_STENCIL="plane")
    res = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "GATHER_OK" in res.stdout, res.stdout[-3000:]


def test_plane_sweep_natural_dispatch_matches_oracle(oracle, torch_cuda):
    """The default dispatch (no option, no environment) on BASELINE's own plane size: a 1024 x 1024 x 48 solver grid, 5 levels
    (coarsest 64 x 64 x 3), 4 MiB x-y planes as at 1024^3 -- against the oracle on the same 50 M cells: one Jacobi sweep, the
    residual, and two V-cycles (free surface: general BOUNDARY cells, skipped runs).  Rounds 1-2 sent such planes to the
    plane-marching sweep; since round 3 the quad kernel is the faster one up to 4 MiB planes (launchStencil: 1.82 against
    1.85 ms at 1024^3) and takes them -- asserted here; the plane-marching kernel keeps larger planes (no oracle-sized grid has
    them together with a coarsest level the direct solver takes) and is tested through options.stencil_path = 2 /
    MGPS_STENCIL=plane: the tests above, tests/test_device_setup.py's random domains, tests/test_gpu_symmetry.py."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    shape = (16, 992, 992)
    z, y, x = np.meshgrid(np.arange(16) / 16, np.arange(992) / 992, np.arange(992) / 992, indexing="ij")
    phi = y - 0.5 + 0.02 * np.sin(4.0 * np.pi * z) + 0.0 * x
    del z, y, x
    bl, bw, dx = D._complex_from_phi(phi, shape, True, (0.4, 0.6), np.float32, 1.0 / 992)
    del phi
    lab, w, off, lev = D.expand_domain(bl, bw, levels=5, solver_shape=(48, 1024, 1024))
    assert int((lab == 3).sum()) > 100000
    gpu = G.GeometricMultigridPoissonSolver(lab, w, lev, False)
    assert gpu.stencil_kernel(0) == "quad" and gpu.stencil_kernel(1) == "quad"
    # by size (4 MiB planes) the fine level's down-stroke takes residual + restriction as the z-folded pair (residualZKernel with its
    # edge and general-cell launches + restrictXYKernel, round 4) -- the V-cycles below run through it; level 1 (1 MiB planes) does not
    assert gpu.residual_restrict_fused(0) and not gpu.residual_restrict_fused(1)
    lab32 = lab.astype(np.int32)
    w64 = [a.astype(np.float64) for a in w]
    x0 = _rand_active(lab, 1)
    b0 = _rand_active(lab, 2, dx * dx)
    xd, bd = gpu.to_device(x0), gpu.to_device(b0)
    ref = np.zeros_like(x0)
    oracle.residual(ref, x0, b0, lab32, w64)
    rd = gpu.new_grid()
    gpu.computePoissonResidual(rd, xd, bd)
    assert rel_err(rd.cpu().numpy(), ref) < OP_TOL
    # the pair as an operator against the oracle's downsample(residual): free surface (blocks the liquid leaves: edge launches) and
    # a cut-cell solid (general BOUNDARY cells: the four patch launches)
    lab1 = gpu.hierarchy().level_labels(1).astype(np.int32)
    coarse_ref = np.zeros(lab1.shape)
    oracle.downsample(coarse_ref, ref, lab1)
    cd = gpu.new_grid(1)
    gpu.residualDownsample(cd, xd, bd, 0)
    assert rel_err(cd.cpu().numpy(), coarse_ref) < OP_TOL
    del coarse_ref, cd
    xj = x0.copy()
    oracle.jacobi(xj, b0, lab32, w64)
    gpu.jacobiPoissonSmoother(xd, bd)
    assert rel_err(xd.cpu().numpy(), xj) < OP_TOL
    del ref, xj, rd
    orc = oracle.solver(lab32, w64, lev, False)
    x_ref = np.zeros(lab.shape)
    b_as_f32 = bd.cpu().numpy().astype(np.float64)
    xs = gpu.new_grid()
    for it in range(2):
        orc.apply_vcycle(x_ref, b_as_f32, it > 0)
        gpu.applyVCycle(xs, bd, it > 0)
        assert rel_l2(xs.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1), it
    gpu.close()


@pytest.mark.parametrize("case", ["stair", "wsolid", "rag264"])
def test_residual_restriction_pair_matches_oracle(case, torch_cuda):
    """residualZKernel / residualZEdgeKernel / residualZGeneralKernel + restrictXYKernel (round 4) against the ORACLE's
    downsample(residual(x)) as one operator (mgps_residual_downsample), forced onto small grids with MGPS_FUSE_RR=1 (by size only
    4 MiB planes take the pair): stair -- a step in the liquid on block boundaries, so that blocks without active cells owe rz a
    plane of terms (the edge launch); wsolid -- free surface + cut-cell solid: general BOUNDARY cells (the four patch launches);
    rag264 -- ragged tiles at the end of every axis.  Also against the two separate passes of the same solver (MGPS_FUSE_RR=0)."""
    import subprocess
    import sys

    from conftest import ROOT

    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
        "import geometricmultigridpressuresolver_amd as G\n"
        "from geometricmultigridpressuresolver_amd import domains as D\n"
        "from oracle.mg_oracle import Oracle\n"
        "case, fused = sys.argv[1], sys.argv[2] == '1'\n"
        "if case == 'wsolid':\n"
        "    bl, bw, dx = D.build_complex_domain((24, 32, 256), use_solid=True)\n"
        "    lab, w, off, lev = D.expand_domain(bl, bw, levels=3, solver_shape=(32, 40, 264))\n"
        "else:\n"
        "    shape = (20, 44, 248) if case == 'rag264' else (24, 32, 248)\n"
        "    bl = np.full(shape, D.DIRICHLET, dtype=np.uint8)\n"
        "    if case == 'stair':\n"
        "        bl[1:8, 1:12, 1:-1] = D.INTERIOR\n"
        "        bl[1:16, 12:31, 1:-1] = D.INTERIOR\n"
        "    else:\n"
        "        bl[1:-1, 1:-1, 1:-1] = D.INTERIOR\n"
        "    bw = []\n"
        "    for axis in range(3):\n"
        "        wa = np.zeros(D.face_shape(*shape, axis), dtype=np.float32)\n"
        "        back, fwd = D._shift_pair(bl, axis)\n"
        "        wa[D._inner_faces(wa, axis)] = np.where((back == D.INTERIOR) | (fwd == D.INTERIOR), 1.0, 0.0)\n"
        "        bw.append(wa)\n"
        "    dx = 1.0 / 248\n"
        "    lab, w, off, lev = D.expand_domain(bl, bw, levels=3, solver_shape=(28, 52, 264) if case == 'rag264' else (32, 40, 264))\n"
        "s = G.GeometricMultigridPoissonSolver(lab, w, lev, False)\n"
        "assert s.stencil_kernel(0) == 'plane' and s.residual_restrict_fused(0) == fused, (s.stencil_kernel(0), s.residual_restrict_fused(0))\n"
        "rng = np.random.default_rng(5)\n"
        "act = D.active_mask(lab)\n"
        "x0 = np.where(act, rng.standard_normal(lab.shape), 0.0).astype(np.float32)\n"
        "b0 = np.where(act, rng.standard_normal(lab.shape) * dx * dx, 0.0).astype(np.float32)\n"
        "cd = s.new_grid(1)\n"
        "s.residualDownsample(cd, s.to_device(x0), s.to_device(b0), 0)\n"
        "orc = Oracle()\n"
        "r = np.zeros(lab.shape); orc.residual(r, x0.astype(np.float64), b0.astype(np.float64), lab.astype(np.int32), [a.astype(np.float64) for a in w])\n"
        "lab1 = s.hierarchy().level_labels(1).astype(np.int32)\n"
        "ref = np.zeros(lab1.shape); orc.downsample(ref, r, lab1)\n"
        "err = np.abs(cd.cpu().numpy() - ref).max() / np.abs(ref).max()\n"
        "assert np.abs(ref).max() > 0 and err < 5e-6, err\n"
        "np.save(sys.argv[3], cd.cpu().numpy())\n"
        "print('PAIR_OK', err)\n"
    ) % (ROOT, ROOT)
    import tempfile

    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for fused in ("1", "0"):
            path = __import__("os").path.join(tmp, f"c{fused}.npy")
            env = dict(**__import__("os").environ, MGPS_FUSE_RR=fused, MGPS_STENCIL="plane")
            res = subprocess.run([sys.executable, "-c", code, case, fused, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
            assert res.returncode == 0 and "PAIR_OK" in res.stdout, res.stdout[-3000:]
            outs.append(np.load(path))
    assert np.abs(outs[0] - outs[1]).max() <= 2e-6 * np.abs(outs[1]).max()


def test_plane_sweep_by_size_matches_oracle(oracle, torch_cuda):
    """The plane-marching sweep reached by the SIZE rule (no option, no environment): x-y planes of 1280 x 1024 cells
    = 5 MiB > kPlaneSweepMinPlaneBytes -- the production path of every level with planes larger than 1024^2, with its two
    planes of look-ahead, the zero-start variant and the active x range (1248 active cells of a 1280-cell row).  Against the
    oracle on the same 63 M cells: residual, one Jacobi sweep and two V-cycles, the second from the first one's result."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    shape = (16, 992, 1248)
    z, y, x = np.meshgrid(np.arange(16) / 16, np.arange(992) / 992, np.arange(1248) / 1248, indexing="ij")
    phi = y - 0.5 + 0.02 * np.sin(4.0 * np.pi * z) + 0.0 * x
    del z, y, x
    bl, bw, dx = D._complex_from_phi(phi, shape, True, (0.4, 0.6), np.float32, 1.0 / 992)
    del phi
    lab, w, off, lev = D.expand_domain(bl, bw, levels=5, solver_shape=(48, 1024, 1280))
    gpu = G.GeometricMultigridPoissonSolver(lab, w, lev, False)
    assert gpu.stencil_kernel(0) == "plane" and gpu.stencil_kernel(1) == "quad"
    lab32 = lab.astype(np.int32)
    w64 = [a.astype(np.float64) for a in w]
    x0 = _rand_active(lab, 1)
    b0 = _rand_active(lab, 2, dx * dx)
    xd, bd = gpu.to_device(x0), gpu.to_device(b0)
    ref = np.zeros_like(x0)
    oracle.residual(ref, x0, b0, lab32, w64)
    rd = gpu.new_grid()
    gpu.computePoissonResidual(rd, xd, bd)
    assert rel_err(rd.cpu().numpy(), ref) < OP_TOL
    xj = x0.copy()
    oracle.jacobi(xj, b0, lab32, w64)
    gpu.jacobiPoissonSmoother(xd, bd)
    assert rel_err(xd.cpu().numpy(), xj) < OP_TOL
    del ref, xj, rd
    orc = oracle.solver(lab32, w64, lev, False)
    x_ref = np.zeros(lab.shape)
    b_as_f32 = bd.cpu().numpy().astype(np.float64)
    xs = gpu.new_grid()
    for it in range(2):
        orc.apply_vcycle(x_ref, b_as_f32, it > 0)
        gpu.applyVCycle(xs, bd, it > 0)
        assert rel_l2(xs.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1), it
    gpu.close()


# ---- BASELINE configs 1 and 3 as stated -----------------------------------------------------------------------------
def test_config1_128_L4_2plus2(oracle, torch_cuda):
    """BASELINE config 1: 128^3 interior-liquid cube, 4-level V-cycle, 2+2 damped-Jacobi sweeps (options.pre_sweeps =
    post_sweeps = 2; the reference's own count is 1+1, MG.cpp:466-486) -- the reference CPU path's leg is the fp64
    oracle with the same counts.  V-cycle by V-cycle parity, the testMultigrid convergence check (b = 0, sine error,
    Test.cpp:1877-1960) and the symmetry check (Test.cpp:1808-1841) on this configuration."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    n, levels = 128, 4
    lab, w, h = D.interior_cube(n, levels)
    opt = G.default_options()
    opt.pre_sweeps = opt.post_sweeps = 2
    gpu = G.GeometricMultigridPoissonSolver(lab, w, levels, False, options=opt)
    lab32 = lab.astype(np.int32)
    orc = oracle.solver(lab32, [a.astype(np.float64) for a in w], levels, False, pre_sweeps=2, post_sweeps=2)
    one = oracle.solver(lab32, [a.astype(np.float64) for a in w], levels, False)
    b32 = D.random_rhs(lab, h, seed=1)
    bd, xd = gpu.to_device(b32), gpu.new_grid()
    b64 = b32.astype(np.float64)
    x_ref, x_one = np.zeros(lab.shape), np.zeros(lab.shape)
    for it in range(3):
        gpu.applyVCycle(xd, bd, it > 0)
        orc.apply_vcycle(x_ref, b64, it > 0)
        one.apply_vcycle(x_one, b64, it > 0)
        assert rel_l2(xd.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1), it
    assert rel_l2(x_ref, x_one) > 1e-3  # the sweep counts do change the cycle: the comparison above is not vacuous
    # convergence check: 2+2 contracts the sine error monotonically and at least as fast as 1+1
    x0 = D.sine_initial_guess(lab, h)
    xs, zd = gpu.to_device(x0), gpu.new_grid()
    x_ref, x_one, z = x0.astype(np.float64), x0.astype(np.float64), np.zeros(lab.shape)
    prev = gpu.l2Norm(xs)
    for it in range(8):
        gpu.applyVCycle(xs, zd, True)
        orc.apply_vcycle(x_ref, z, True)
        one.apply_vcycle(x_one, z, True)
        cur = gpu.l2Norm(xs)
        assert cur < prev and cur == pytest.approx(oracle.l2(x_ref, lab32), rel=1e-3)
        prev = cur
    assert oracle.l2(x_ref, lab32) < oracle.l2(x_one, lab32)
    # symmetry of the 2+2 cycle (fp32 bound 1e-4)
    a, b = gpu.to_device(D.random_rhs(lab, h, seed=6)), gpu.to_device(D.random_rhs(lab, h, seed=7))
    xa, xb = gpu.new_grid(), gpu.new_grid()
    gpu.applyVCycle(xa, a)
    gpu.applyVCycle(xb, b)
    da, db = gpu.dotProduct(xa, b), gpu.dotProduct(xb, a)
    assert abs(da - db) / max(abs(da), abs(db)) < 1e-4
    gpu.close()


@pytest.mark.parametrize("use_gs,pre,post", [(True, 2, 1), (False, 1, 3), (True, 2, 2)])
def test_sweep_counts_match_oracle(use_gs, pre, post, domain_factory, oracle, torch_cuda):
    """options.pre_sweeps / post_sweeps on a cut-cell + free-surface domain, both smoothers, V-cycle and PCG (the
    gathered <z, r> must come from the last repetition only)."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    lab, w, off, lev, dx = domain_factory("solid", 48)
    opt = G.default_options()
    opt.pre_sweeps, opt.post_sweeps = pre, post
    gpu = G.GeometricMultigridPoissonSolver(lab, w, lev, use_gs, options=opt)
    orc = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], lev, use_gs, pre_sweeps=pre, post_sweeps=post)
    b = _rand_active(lab, 5, dx * dx)
    xd, bd = gpu.new_grid(), gpu.to_device(b)
    b64 = bd.cpu().numpy().astype(np.float64)
    x_ref = np.zeros_like(b)
    for it in range(2):
        orc.apply_vcycle(x_ref, b64, it > 0)
        gpu.applyVCycle(xd, bd, it > 0)
        assert rel_l2(xd.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1), it
    if pre == post:  # a symmetric cycle is a valid CG preconditioner
        x_ref = np.zeros(lab.shape)
        ref = orc.solve_pcg(x_ref, b64, 1e-5, 200, True)
        xs = gpu.new_grid()
        st = gpu.solveGeometricConjugateGradient(xs, bd, 1e-5, 200, True)
        assert st["outcome"] == "converged" and abs(st["iterations"] - ref["iterations"]) <= 2
        assert rel_l2(xs.cpu().numpy(), x_ref) < 2e-4
    gpu.close()


def test_large_coarsest_level_is_factorised_on_the_device(oracle, torch_cuda):
    """A coarsest level past 8192 unknowns (here 22^3 = 10 648 on a 26^3 grid: a 104^3 cube with 3 levels): the reference
    factorises whatever the coarsest level holds (MG.cpp:405-411, solve :680); this library builds the dense matrix of
    MG.cpp:359-382 on the device, factorises and inverts it there (hipSOLVER potrf / potri in fp64) and solves by one dense
    mat-vec.  Direct solve: A x = b to the fp32 inverse's accuracy; two V-cycles against the oracle; a second solver on the
    same labels takes the kept inverse and gives the same bits."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    n, levels = 104, 3
    lab, w, h = D.interior_cube(n, levels)
    gpu = G.GeometricMultigridPoissonSolver(lab, w, levels, False)
    orc = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], levels, False)
    assert orc.coarse_unknowns > 8192
    cl = orc.level_labels(levels - 1)
    b = _rand_active(cl, 61)
    xd = gpu.new_grid(levels - 1)
    gpu.coarseDirectSolve(xd, gpu.to_device(b, levels - 1))
    x = xd.cpu().numpy().astype(np.float64)
    y = np.zeros_like(x)
    oracle.apply_poisson(y, x, cl)
    assert rel_err(y, b) < 5e-5
    rhs = D.random_rhs(lab, h)
    x_ref = np.zeros(lab.shape)
    xg, bg = gpu.new_grid(), gpu.to_device(rhs)
    for it in range(2):
        orc.apply_vcycle(x_ref, rhs.astype(np.float64), it > 0)
        gpu.applyVCycle(xg, bg, it > 0)
        assert rel_l2(xg.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1)
    again = G.GeometricMultigridPoissonSolver(lab, w, levels, False)
    x2 = again.new_grid()
    for it in range(2):
        again.applyVCycle(x2, again.to_device(rhs), it > 0)
    assert np.array_equal(x2.cpu().numpy(), xg.cpu().numpy())
    again.close()
    gpu.close()


@pytest.mark.parametrize("levels,use_gs", [(5, True), (6, False)])
def test_config3_512_free_surface_pcg(levels, use_gs, oracle, torch_cuda):
    """BASELINE config 3: 512^3 free-surface pool (sine liquid surface, ghost-fluid weights up to 1/0.01, cut-cell solid
    box), MG-preconditioned CG with the plugin's smoother (tiled Gauss-Seidel, Plug.cpp:466) -- and with damped Jacobi, what
    MGPS_DOP_SMOOTHER=jacobi selects in the shipped DOP node -- to 1e-5 on the delta +
    random rhs.  Stated criteria, against the fp64 oracle solving the same system with the same smoother (about 15 s on 16 host cores):
      * iteration count within +-2 of the oracle's;
      * pressure field relative L2 difference < 1e-5 (the "same pressure field" criterion of the north star);
      * fp32 CG vectors: the recurrence residual CG tests is < 1e-5; the residual *recomputed* in fp32 from the fp32
        iterate floors at eps * cond (ghost-fluid weights of 100 amplify the rounding of x) and is only required < 1e-2;
      * options.pcg_fp64_vectors = 1 (all CG vectors fp64) and = 2 (the default: the iterate in fp64 with group-wise fp32 updates
        and residual replacement): the recomputed residual is a true fp64 residual and must itself be < 1e-5."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    # levels = 5 is SURVEY 8(d)'s definition of the config (coarsest level 32^3: ~13 000 unknowns on this pool, factorised on the
    # device): the plugin's smoother with every CG vector mode; levels = 6 (coarsest 16^3, factorised on the host; what rounds 1 and 2
    # ran): the Jacobi switch of the DOP shim with the default mode
    n = 512
    lab, w, h = D.free_surface_pool(n, levels)
    pad = 2 ** (levels - 1)
    b = (D.delta_rhs(lab, n - 2 * pad, pad, h) + D.random_rhs(lab, h)).astype(np.float32)
    results = {}
    variants = (0, 1, 2) if use_gs else (2,)
    for fp64 in variants:
        opt = G.default_options()
        opt.pcg_fp64_vectors = fp64
        gpu = G.GeometricMultigridPoissonSolver(lab, w, levels, use_gs, options=opt)
        xd = gpu.new_grid()
        st = gpu.solveGeometricConjugateGradient(xd, gpu.to_device(b), 1e-5, 2500, True)
        results[fp64] = (st, xd.cpu().numpy().astype(np.float64))
        gpu.close()
        del gpu, xd
        torch_cuda.cuda.empty_cache()
    orc = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], levels, use_gs)
    x_ref = np.zeros(lab.shape)
    ref = orc.solve_pcg(x_ref, b.astype(np.float64), 1e-5, 2500, True)
    assert ref["rel_residual_recomputed"] < 1e-5
    for fp64 in variants:
        st, x = results[fp64]
        assert st["outcome"] == "converged" and abs(st["iterations"] - ref["iterations"]) <= 2, (fp64, st, ref["iterations"])
        assert st["rel_residual"] < 1e-5
        assert rel_l2(x, x_ref) < 1e-5, (fp64, rel_l2(x, x_ref))
        assert st["rel_residual_recomputed"] < (1e-5 if fp64 else 1e-2), (fp64, st)


@pytest.mark.parametrize("gs", [False, True])
def test_extents_not_multiples_of_four(gs, oracle, torch_cuda):
    """nx = 66: no 16-byte quads, so the sweeps take the scalar kernel while the vector ops walk the activity runs of the
    flat array (runs straddle rows); two V-cycles and an MG-PCG against the oracle."""
    import geometricmultigridpressuresolver_amd as G
    from geometricmultigridpressuresolver_amd import domains as D

    bl, bw, dx = D.build_complex_domain((32, 36, 58), use_solid=True, dtype=np.float32)
    lab, w, off, lev = D.expand_domain(bl, bw, levels=2, solver_shape=(40, 44, 66))
    s = G.GeometricMultigridPoissonSolver(lab, w, lev, gs)
    assert s.stencil_kernel(0) == "scalar"
    ref = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], lev, gs)
    b = _rand_active(lab, 3, dx * dx)
    x_ref = np.zeros(lab.shape)
    xd, bd = s.new_grid(), s.to_device(b)
    b64 = bd.cpu().numpy().astype(np.float64)
    for it in range(2):
        ref.apply_vcycle(x_ref, b64, it > 0)
        s.applyVCycle(xd, bd, it > 0)
        assert rel_l2(xd.cpu().numpy(), x_ref) < VCYCLE_TOL * (it + 1)
    st = s.solveGeometricConjugateGradient(s.new_grid(), bd, 1e-6, 200, True)
    assert st["outcome"] == "converged"
    s.close()
