"""Plugin-side field pre/post-processing (include/mgps_fields.h).

CPU: the oracle pipeline (material labels -> valid faces -> multigrid labels / weights -> right-hand side ->
MG-PCG -> pressure -> pressure gradient) leaves the liquid divergence-free -- the reference's own end-to-end
check (Plug.cpp:704-706) -- and its labels / weights satisfy the reference's structural unit tests.
GPU: every device pass against the oracle on the same inputs, and the same end-to-end property through the
HIP solver.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import geometricmultigridpressuresolver_amd as G  # noqa: E402
from geometricmultigridpressuresolver_amd import domains as D  # noqa: E402
from oracle.mg_oracle import FieldsOracle  # noqa: E402

SHAPE = (40, 32, 48)  # gz, gy, gx


@pytest.fixture(scope="module")
def fo():
    return FieldsOracle()


def _oracle_pipeline(fo, oracle, sc, with_solid):
    cw, phi = sc["cut_weights"], sc["liquid_phi"]
    sv = sc["solid_velocity"] if with_solid else None
    material = fo.material_labels(phi, sc["solid_phi"], cw)
    valid = fo.valid_faces(material, cw)
    eshape, offset, levels = G.expanded_layout(SHAPE, 0, power_of_two=False)
    lab = fo.domain_labels(material, eshape, offset)
    w = fo.boundary_weights(cw, phi, valid, material, eshape, offset)
    oracle.set_boundary_labels(lab, w)
    rhs = fo.rhs(material, sc["velocity"], cw, eshape, offset, sv)
    return material, valid, eshape, offset, levels, lab, w, rhs


@pytest.mark.parametrize("with_solid", [False, True])
def test_oracle_projection_is_divergence_free(fo, oracle, with_solid):
    sc = D.projection_scene(SHAPE, with_solid_velocity=with_solid, dtype=np.float64)
    material, valid, eshape, offset, levels, lab, w, rhs = _oracle_pipeline(fo, oracle, sc, with_solid)
    sv = sc["solid_velocity"] if with_solid else None
    assert {0, 1, 2} == set(np.unique(material)) and (material == 1).sum() > 5000
    assert oracle.unit_test_exterior(lab) and oracle.unit_test_boundary(lab, w)  # Ops.cpp:602, Ops.h:1771
    # every valid face carries a positive weight; liquid/air faces are scaled by 1/theta >= 1 (Plug.cpp:827-853)
    for a in range(3):
        assert (sc["cut_weights"][a][valid[a] == 1] > 0).all()
    assert max(float(x.max()) for x in w) > 1.0
    s = oracle.solver(lab, w, levels, True)
    x = np.zeros(eshape)
    st = s.solve_pcg(x, rhs, 1e-10, 500, True)
    assert st["rel_residual_recomputed"] < 1e-9
    pressure = np.zeros(SHAPE)
    fo.solution_to_pressure(pressure, x, material, offset)
    vel = [v.copy() for v in sc["velocity"]]
    fo.pressure_gradient(vel, sc["cut_weights"], sc["liquid_phi"], pressure, valid, material)
    before = np.abs(fo.rhs(material, sc["velocity"], sc["cut_weights"], eshape, offset, sv)).max()
    after = np.abs(fo.rhs(material, vel, sc["cut_weights"], eshape, offset, sv)).max()
    assert before > 0.1 and after < 1e-8 * before
    total, mx, count = fo.divergence(material, vel, sc["cut_weights"], sv)
    assert count == (material == 1).sum() and abs(total) < 1e-8 * count and 0 <= mx < 1e-8
    # the warm-start copy is the inverse of the pressure copy on liquid cells
    assert (fo.pressure_to_solution(pressure, material, eshape, offset) == np.where(np.isin(lab, (0, 3)), x, 0)).all()


def _dev(a, torch):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("with_solid", [False, True])
def test_device_passes_match_oracle(fo, oracle, with_solid):
    import torch

    from geometricmultigridpressuresolver_amd import fields as F

    sc = D.projection_scene(SHAPE, with_solid_velocity=with_solid)
    material, valid, eshape, offset, levels, lab, w, rhs = _oracle_pipeline(fo, oracle, sc, with_solid)
    cw = [_dev(a, torch) for a in sc["cut_weights"]]
    phi, sphi = _dev(sc["liquid_phi"], torch), _dev(sc["solid_phi"], torch)
    vel = [_dev(a, torch) for a in sc["velocity"]]
    sv = [_dev(a, torch) for a in sc["solid_velocity"]] if with_solid else None
    mat_d = F.buildMaterialCellLabels(phi, sphi, cw)
    assert (mat_d.cpu().numpy() == material).all()
    valid_d = F.buildValidFaces(mat_d, cw)
    for a in range(3):
        assert (valid_d[a].cpu().numpy() == valid[a]).all()
    lab_d, w_d = F.buildMGDomain(mat_d, cw, phi, valid_d, eshape, offset)
    assert (lab_d.cpu().numpy() == lab).all()
    for a in range(3):
        assert np.abs(w_d[a].cpu().numpy() - w[a]).max() <= 2e-6 * np.abs(w[a]).max()
    rhs_d = F.buildRHS(mat_d, vel, cw, eshape, offset, sv)
    assert np.abs(rhs_d.cpu().numpy() - rhs).max() < 1e-5
    # pressure copy in / out and the gradient update on a seeded pressure field
    rng = np.random.default_rng(5)
    p_host = np.where(material == 1, rng.random(SHAPE), 0.0).astype(np.float32)
    x_d = F.applyOldPressure(_dev(p_host, torch), mat_d, eshape, offset)
    assert (x_d.cpu().numpy() == fo.pressure_to_solution(p_host, material, eshape, offset).astype(np.float32)).all()
    p_back = torch.full(SHAPE, 9.0, dtype=torch.float32, device="cuda")
    F.applySolutionToPressure(p_back, x_d, mat_d, offset)
    assert (p_back.cpu().numpy() == np.where(material == 1, p_host, 9.0)).all()
    vel_ref = [v.astype(np.float64) for v in sc["velocity"]]
    fo.pressure_gradient(vel_ref, sc["cut_weights"], sc["liquid_phi"], p_host, valid, material)
    vel_d = [v.clone() for v in vel]
    F.applyPressureGradient(vel_d, phi, _dev(p_host, torch), valid_d, mat_d)
    for a in range(3):
        assert np.abs(vel_d[a].cpu().numpy() - vel_ref[a]).max() < 2e-5 * np.abs(vel_ref[a]).max()
    got = F.computeResultingDivergence(mat_d, vel, cw, sv)
    ref = fo.divergence(material, sc["velocity"], sc["cut_weights"], sc["solid_velocity"] if with_solid else None)
    assert got[2] == ref[2] and abs(got[0] - ref[0]) < 1e-4 * ref[2] and got[1] == pytest.approx(ref[1], rel=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("MGPS_FUZZ_SEEDS", "10"))))
def test_random_scenes_match_oracle(seed, fo, oracle):
    """Randomised geometry (grid shape, fill level, wave, solid box) through the whole chain: device field passes
    and the HIP V-cycle / MG-PCG against the oracle on the same scene -- ragged tiles, thin slabs, bands that
    wrap the solid, general BOUNDARY cells on cut faces and at the free surface."""
    import torch

    from geometricmultigridpressuresolver_amd import fields as F

    rng = np.random.default_rng(100 + seed)
    shape = tuple(int(v) for v in rng.integers(14, 60, 3))
    sc = D.projection_scene(shape, seed=seed, with_solid_velocity=bool(seed & 1), randomize=True)
    cw, phi = sc["cut_weights"], sc["liquid_phi"]
    sv = sc["solid_velocity"]
    material = fo.material_labels(phi, sc["solid_phi"], cw)
    if (material == 1).sum() < 200:
        pytest.skip("scene has almost no liquid")
    valid = fo.valid_faces(material, cw)
    eshape, offset, levels = G.expanded_layout(shape, 0, power_of_two=bool(seed % 3 == 0))
    lab = fo.domain_labels(material, eshape, offset)
    w = fo.boundary_weights(cw, phi, valid, material, eshape, offset)
    oracle.set_boundary_labels(lab, w)
    rhs = fo.rhs(material, sc["velocity"], cw, eshape, offset, sv)
    # device passes
    cw_d = [_dev(a, torch) for a in cw]
    mat_d = F.buildMaterialCellLabels(_dev(phi, torch), _dev(sc["solid_phi"], torch), cw_d)
    assert (mat_d.cpu().numpy() == material).all()
    valid_d = F.buildValidFaces(mat_d, cw_d)
    lab_d, w_d = F.buildMGDomain(mat_d, cw_d, _dev(phi, torch), valid_d, eshape, offset)
    assert (lab_d.cpu().numpy() == lab).all()
    rhs_d = F.buildRHS(mat_d, [_dev(a, torch) for a in sc["velocity"]], cw_d, eshape, offset,
                       [_dev(a, torch) for a in sv] if sv is not None else None)
    assert np.abs(rhs_d.cpu().numpy() - rhs).max() < 1e-5
    # solver on the device-built domain against the oracle on the oracle-built one
    w32 = [a.cpu().numpy() for a in w_d]
    for use_gs in (False, True):
        gpu = G.GeometricMultigridPoissonSolver(lab_d.cpu().numpy(), w32, levels, use_gs)
        orc = oracle.solver(lab, [a.astype(np.float64) for a in w32], levels, use_gs)
        assert gpu.getMGLevels() == orc.levels
        b64 = rhs_d.cpu().numpy().astype(np.float64)
        x_ref = np.zeros(eshape)
        xd = gpu.new_grid()
        for it in range(2):
            orc.apply_vcycle(x_ref, b64, it > 0)
            gpu.applyVCycle(xd, rhs_d, it > 0)
            err = np.linalg.norm(xd.cpu().numpy() - x_ref) / max(np.linalg.norm(x_ref), 1e-300)
            assert err < 2e-5 * (it + 1), (shape, use_gs, it, err)
        xo, xg = np.zeros(eshape), gpu.new_grid()
        so = orc.solve_pcg(xo, b64, 1e-5, 300, True)
        sg = gpu.solveGeometricConjugateGradient(xg, rhs_d, 1e-5, 300, True)
        assert sg["outcome"] == "converged" and abs(sg["iterations"] - so["iterations"]) <= 2, (shape, so, sg)
        gpu.close()
        if not use_gs:  # the mixed-precision cycle (binary16 fine-level iterate / residual) on the same scene
            opt = G.default_options()
            opt.precision = 1
            mix = G.GeometricMultigridPoissonSolver(lab_d.cpu().numpy(), w32, levels, False, options=opt)
            xm = mix.new_grid()
            mix.applyVCycle(xm, rhs_d, False)
            x1 = np.zeros(eshape)
            orc.apply_vcycle(x1, b64, False)
            assert np.linalg.norm(xm.cpu().numpy() - x1) <= 2e-3 * max(np.linalg.norm(x1), 1e-300), (shape, "mixed cycle")
            xm = mix.new_grid()
            sm = mix.solveGeometricConjugateGradient(xm, rhs_d, 1e-5, 300, True)
            assert sm["outcome"] == "converged" and sm["iterations"] <= 1.5 * sg["iterations"] + 1, (shape, sm, sg)
            assert np.linalg.norm(xm.cpu().numpy() - xo) <= 5e-5 * max(np.linalg.norm(xo), 1e-300), (shape, "mixed pcg")
            mix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("with_solid", [False, True])
def test_solver_from_device_weights_equals_host_weights(with_solid, fo, oracle):
    """mgps_create_device_weights (rows of the BOUNDARY cells evaluated by a kernel, weights never on the host)
    builds the same solver as mgps_create: identical V-cycle and PCG results; and it rejects inconsistent input."""
    import torch

    sc = D.projection_scene(SHAPE, with_solid_velocity=with_solid)
    material, valid, eshape, offset, levels, lab, w, rhs = _oracle_pipeline(fo, oracle, sc, with_solid)
    lab8, w32 = lab.astype(np.uint8), [a.astype(np.float32) for a in w]
    b = rhs.astype(np.float32)
    results = []
    # host labels + host weights (mgps_create), host labels + device weights (mgps_create_device_weights), both on
    # the device (mgps_create_device)
    for labels, weights in ((lab8, w32), (lab8, [_dev(a, torch) for a in w32]), (_dev(lab8, torch), [_dev(a, torch) for a in w32])):
        for use_gs in (False, True):
            s = G.GeometricMultigridPoissonSolver(labels, weights, levels, use_gs)
            x, bd = s.new_grid(), s.to_device(b)
            s.applyVCycle(x, bd, False)
            xp = s.new_grid()
            st = s.solveGeometricConjugateGradient(xp, bd, 1e-6, 200, True)
            results.append((x.cpu().numpy(), xp.cpu().numpy(), st["iterations"]))
            s.close()
    for k in range(2):
        for other in (k + 2, k + 4):
            assert np.array_equal(results[k][0], results[other][0]) and np.array_equal(results[k][1], results[other][1])
            assert results[k][2] == results[other][2]
    # every face weight 1: cells that are BOUNDARY only through a cut face between two liquid cells lose their reason
    bad = [torch.ones_like(_dev(a, torch)) for a in w32]
    with pytest.raises(G.MgpsError):
        G.GeometricMultigridPoissonSolver(lab8, bad, levels, False)


@pytest.mark.gpu
def test_device_projection_is_divergence_free():
    """The whole projection on the device: fields -> multigrid domain -> MG-PCG -> pressure -> velocity; the
    reference's own end-to-end check is the resulting divergence (Plug.cpp:704-706)."""
    import torch

    from geometricmultigridpressuresolver_amd import fields as F

    sc = D.projection_scene(SHAPE, with_solid_velocity=True)
    cw = [_dev(a, torch) for a in sc["cut_weights"]]
    phi, sphi = _dev(sc["liquid_phi"], torch), _dev(sc["solid_phi"], torch)
    vel = [_dev(a, torch) for a in sc["velocity"]]
    sv = [_dev(a, torch) for a in sc["solid_velocity"]]
    eshape, offset, levels = G.expanded_layout(SHAPE, 0, power_of_two=False)
    material = F.buildMaterialCellLabels(phi, sphi, cw)
    valid = F.buildValidFaces(material, cw)
    labels, weights = F.buildMGDomain(material, cw, phi, valid, eshape, offset)
    rhs = F.buildRHS(material, vel, cw, eshape, offset, sv)
    # labels and weights stay where the field passes wrote them (mgps_create_device fetches its own host copy of the labels)
    solver = G.GeometricMultigridPoissonSolver(labels, weights, levels, True)
    x = solver.new_grid()
    st = solver.solveGeometricConjugateGradient(x, rhs, 1e-6, 200, True)
    assert st["outcome"] == "converged"
    pressure = torch.zeros(SHAPE, dtype=torch.float32, device="cuda")
    F.applySolutionToPressure(pressure, x, material, offset)
    before = F.buildRHS(material, vel, cw, eshape, offset, sv).abs().max().item()
    F.applyPressureGradient(vel, phi, pressure, valid, material)
    after = F.buildRHS(material, vel, cw, eshape, offset, sv).abs().max().item()
    total, mx, count = F.computeResultingDivergence(material, vel, cw, sv)
    assert after < 2e-4 * before and mx <= after * 1.0001 and count == (material == 1).sum().item()
    solver.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_one_call_projection_matches_the_pass_by_pass_pipeline(dtype):
    """mgps_project_free_surface -- what the Houdini shim calls with flattened SIM fields (float, or the reference's
    double: converted on the device) -- gives the same pressure, velocity and valid faces as the device passes driven
    one by one, and leaves the liquid divergence-free (Plug.cpp:704-706)."""
    import torch

    from geometricmultigridpressuresolver_amd import fields as F

    sc = D.projection_scene(SHAPE, with_solid_velocity=True)
    # pass by pass (float32 on the device)
    cw = [_dev(a, torch) for a in sc["cut_weights"]]
    phi, sphi = _dev(sc["liquid_phi"], torch), _dev(sc["solid_phi"], torch)
    vel = [_dev(a, torch) for a in sc["velocity"]]
    sv = [_dev(a, torch) for a in sc["solid_velocity"]]
    eshape, offset, levels = G.expanded_layout(SHAPE, 0, power_of_two=True)
    material = F.buildMaterialCellLabels(phi, sphi, cw)
    valid = F.buildValidFaces(material, cw)
    labels, weights = F.buildMGDomain(material, cw, phi, valid, eshape, offset)
    rhs = F.buildRHS(material, vel, cw, eshape, offset, sv)
    solver = G.GeometricMultigridPoissonSolver(labels, weights, levels, True)
    x = solver.new_grid()
    st = solver.solveGeometricConjugateGradient(x, rhs, 1e-6, 200, True)
    pressure = torch.zeros(SHAPE, dtype=torch.float32, device="cuda")
    F.applySolutionToPressure(pressure, x, material, offset)
    F.applyPressureGradient(vel, phi, pressure, valid, material)
    solver.close()
    # one call on host arrays
    h = lambda a: np.array(a, dtype=dtype, order="C", copy=True)  # noqa: E731  (copies: the call updates velocity and pressure in place)
    vel_h = [h(a) for a in sc["velocity"]]
    p_h = np.zeros(SHAPE, dtype=dtype)
    valid_h, info = F.project_free_surface(h(sc["liquid_phi"]), h(sc["solid_phi"]), [h(a) for a in sc["cut_weights"]], vel_h, p_h,
                                           [h(a) for a in sc["solid_velocity"]], use_old_pressure=False, tolerance=1e-6, max_iterations=200)
    assert info["iterations"] == st["iterations"] and info["mg_levels"] == levels and info["offset"] == offset and info["expanded"] == tuple(eshape)
    assert info["liquid_cells"] == (material == 1).sum().item()
    for a in range(3):
        assert (valid_h[a] == valid[a].cpu().numpy()).all()
        assert np.abs(vel_h[a] - vel[a].cpu().numpy()).max() < 1e-6 * max(1.0, np.abs(vel_h[a]).max())
    assert np.abs(p_h - pressure.cpu().numpy()).max() < 1e-6 * np.abs(p_h).max()
    before = np.abs(rhs.cpu().numpy()).max()
    assert info["divergence_max"] < 2e-4 * before and info["residual_l2"] > 0
    # staging buffers from mgps_host_alloc (page-locked): same answer, and released blocks are handed out again
    if dtype == np.float32:
        import ctypes as C

        from geometricmultigridpressuresolver_amd._lib import lib

        L = lib()
        L.mgps_host_alloc.restype = C.c_void_p
        L.mgps_host_alloc.argtypes = [C.c_size_t]
        L.mgps_host_free.argtypes = [C.c_void_p]
        nbytes = int(np.prod(SHAPE)) * 4
        ptr = L.mgps_host_alloc(nbytes)
        assert ptr
        pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=SHAPE)
        pinned[...] = 0
        vel3 = [h(a) for a in sc["velocity"]]
        F.project_free_surface(h(sc["liquid_phi"]), h(sc["solid_phi"]), [h(a) for a in sc["cut_weights"]], vel3, pinned,
                               [h(a) for a in sc["solid_velocity"]], use_old_pressure=False, tolerance=1e-6, max_iterations=200)
        assert np.array_equal(pinned, p_h)
        del pinned
        L.mgps_host_free(ptr)
        again = L.mgps_host_alloc(nbytes)
        assert again == ptr
        L.mgps_host_free(again)
    # warm start from the answer: CG leaves at its first test (CG.h:60-64)
    vel2 = [h(a) for a in sc["velocity"]]
    _, info2 = F.project_free_surface(h(sc["liquid_phi"]), h(sc["solid_phi"]), [h(a) for a in sc["cut_weights"]], vel2, p_h.copy(),
                                      [h(a) for a in sc["solid_velocity"]], use_old_pressure=True, tolerance=1e-5, max_iterations=200)
    assert info2["iterations"] == 0 and info2["outcome"] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("use_old", [False, True])
def test_projection_clears_stale_pressure_outside_the_liquid(use_old):
    """The reference clears the pressure field before it writes the solution into the liquid cells (`makeConstant(0)`,
    Plug.cpp:641): a cell that was liquid in the last sub-step and is air or solid now must not keep its old value, which the
    gradient pass would read across ghost-fluid faces.  Garbage outside the liquid in the incoming pressure changes nothing."""
    import torch

    from geometricmultigridpressuresolver_amd import fields as F

    sc = D.projection_scene(SHAPE, with_solid_velocity=True)
    h = lambda a: np.array(a, dtype=np.float32, order="C", copy=True)  # noqa: E731
    material = F.buildMaterialCellLabels(_dev(sc["liquid_phi"], torch), _dev(sc["solid_phi"], torch), [_dev(a, torch) for a in sc["cut_weights"]]).cpu().numpy()
    liquid = material == 1
    assert liquid.any() and (~liquid).any()

    def run(p0):
        vel = [h(a) for a in sc["velocity"]]
        p = p0.copy()
        valid, info = F.project_free_surface(h(sc["liquid_phi"]), h(sc["solid_phi"]), [h(a) for a in sc["cut_weights"]], vel, p,
                                             [h(a) for a in sc["solid_velocity"]], use_old_pressure=use_old, tolerance=1e-6, max_iterations=200)
        return vel, p, info

    clean = np.zeros(SHAPE, dtype=np.float32)
    if use_old:  # a warm start that means something inside the liquid
        clean[liquid] = np.random.default_rng(5).standard_normal(int(liquid.sum())).astype(np.float32) * 1e-3
    dirty = clean.copy()
    dirty[~liquid] = np.random.default_rng(6).standard_normal(int((~liquid).sum())).astype(np.float32) * 1e3
    vel_c, p_c, info_c = run(clean)
    vel_d, p_d, info_d = run(dirty)
    assert (p_d[~liquid] == 0).all() and (p_c[~liquid] == 0).all()
    assert info_c["iterations"] == info_d["iterations"]
    assert np.array_equal(p_c, p_d)
    for a in range(3):
        assert np.array_equal(vel_c[a], vel_d[a])


@pytest.mark.gpu
def test_projection_without_liquid_publishes_valid_faces_and_zero_pressure():
    """No liquid cell: nothing to solve, but what the reference would publish is still published -- the valid faces it built
    (Plug.cpp:286) and an all-zero pressure (Plug.cpp:641); velocities stay as they came."""
    from geometricmultigridpressuresolver_amd import fields as F

    sc = D.projection_scene(SHAPE, with_solid_velocity=False)
    h = lambda a: np.array(a, dtype=np.float32, order="C", copy=True)  # noqa: E731
    phi = np.full(SHAPE, 1.0, dtype=np.float32)  # air everywhere
    vel = [h(a) for a in sc["velocity"]]
    vel0 = [a.copy() for a in vel]
    p = np.full(SHAPE, 7.0, dtype=np.float32)
    valid, info = F.project_free_surface(phi, h(sc["solid_phi"]), [h(a) for a in sc["cut_weights"]], vel, p, None, use_old_pressure=True,
                                         tolerance=1e-6, max_iterations=200)
    assert info["liquid_cells"] == 0 and info["iterations"] == 0
    assert (p == 0).all()
    for a in range(3):
        assert np.array_equal(vel[a], vel0[a]) and (valid[a] == 0).all()
