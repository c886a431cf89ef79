"""CPU tests of the C-ABI library (libmgps.so): it loads without a GPU, exports every symbol that
include/mgps.h declares, and its host-side logic (domain expansion, structural checks, multigrid
hierarchy, coarsest-level direct solve, error conventions) agrees with the oracle.  No device
compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.sparse.linalg as spla

import geometricmultigridpressuresolver_amd as G
from conftest import ROOT, make_domain
from geometricmultigridpressuresolver_amd import domains as D
from geometricmultigridpressuresolver_amd._lib import Options, lib
from test_oracle_properties import assemble

KINDS = [("simple", 16), ("complex", 24), ("solid", 24)]


def test_library_exports_every_declared_symbol():
    import glob

    header = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))))
    names = sorted(set(re.findall(r"\b(mgps_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 55 and "mgps_fields_rhs" in names
    missing = [n for n in names if not hasattr(lib(), n)]
    assert not missing, missing


def test_options_struct_matches_header():
    o = G.default_options()
    assert o.struct_size == C.sizeof(Options)
    assert (o.band_width, o.band_iterations) == (3, 3)  # MG.cpp:141-142
    assert o.jacobi_weight == pytest.approx(2.0 / 3.0)  # Ops.h:291
    assert o.device == -1 and o.max_coarse_unknowns == 32768
    assert (o.pre_sweeps, o.post_sweeps, o.stencil_path) == (1, 1, 0)  # MG.cpp:466-486, 740-757: one sweep per stroke
    assert lib().mgps_status_string(0) == b"ok" and lib().mgps_status_string(2) == b"no HIP device"


def test_expanded_layout_rule(oracle):
    for shape in [(128, 128, 128), (64, 64, 64), (400, 600, 600), (30, 100, 50), (17, 33, 65)]:
        bz, by, bx = shape
        (ez, ey, ex), off, lev = G.expanded_layout(shape)
        dims, off2, lev2 = oracle.expanded_layout(bx, by, bz)
        assert (ex, ey, ez) == dims and off == off2 and lev == lev2
    # caller-chosen level count and tight (non power-of-two) extents
    (ez, ey, ex), off, lev = G.expanded_layout((480, 480, 480), levels=5, power_of_two=False)
    assert (ez, ey, ex, off, lev) == (512, 512, 512, 16, 5)
    (ez, ey, ex), off, lev = G.expanded_layout((400, 600, 600), levels=0, power_of_two=False)
    assert lev == 8 and off == 128 and (ez, ey, ex) == (768, 1024, 1024)


@pytest.mark.parametrize("kind,g", KINDS)
def test_expanded_domain_matches_oracle(kind, g, oracle):
    if kind == "simple":
        bl, bw, dx = D.build_simple_domain(g, 1, dtype=np.float32)
    else:
        bl, bw, dx = D.build_complex_domain(g, use_solid=(kind == "solid"), dtype=np.float32)
    lab, w, off, lev = G.build_expanded_domain(bl, bw)
    olab, ow, ooff, olev = oracle.build_expanded_domain(bl.astype(np.int32), [a.astype(np.float64) for a in bw])
    assert (off, lev) == (ooff, olev)
    assert (lab == olab).all()
    for a in range(3):
        assert (w[a] == ow[a].astype(np.float32)).all()
    plab, pw, poff, plev = D.expand_domain(bl, bw)  # the numpy restatement used by the harness
    assert (plab == lab).all() and poff == off and plev == lev
    assert G.unit_test_boundary_cells(lab, w) and G.unit_test_exterior_cells(lab)


@pytest.mark.parametrize("kind,g", KINDS)
def test_structural_checks_reject_corruption(kind, g, oracle):
    lab, w, off, lev, dx = make_domain(kind, g)
    bad = lab.copy()
    bad[0, 0, 0] = D.DIRICHLET
    assert not G.unit_test_exterior_cells(bad) and not oracle.unit_test_exterior(bad.astype(np.int32))
    k, j, i = [a[0] for a in np.nonzero(lab == D.BOUNDARY)]
    bad = lab.copy()
    bad[k, j, i] = D.INTERIOR  # an INTERIOR cell next to a non-active one
    assert not G.unit_test_boundary_cells(bad, w)
    assert not oracle.unit_test_boundary(bad.astype(np.int32), [a.astype(np.float64) for a in w])
    H = G.Hierarchy(lab, lev)
    c1 = H.level_labels(1)
    assert G.unit_test_coarsening(c1, lab)
    k, j, i = [a[0] for a in np.nonzero(c1 == D.DIRICHLET)]
    c1[k, j, i] = D.EXTERIOR
    assert not G.unit_test_coarsening(c1, lab)


@pytest.mark.parametrize("kind,g", KINDS + [("solid", 40)])
def test_hierarchy_matches_oracle(kind, g, oracle):
    lab, w, off, lev, dx = make_domain(kind, g)
    H = G.Hierarchy(lab, lev)
    s = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], lev, False)
    assert H.levels == s.levels and H.coarse_unknowns == s.coarse_unknowns
    for l in range(H.levels):
        nx, ny, nz = s.level_dims(l)
        assert H.level_shape(l) == (nz, ny, nx)
        assert (H.level_labels(l) == s.level_labels(l)).all()
        assert (H.band_cells(l) == s.band(l)).all()  # same cells in the reference's (tile,k,j,i) order


@pytest.mark.parametrize("kind,g", KINDS + [("solid", 40), ("solid", 72), ("simple", 100)])
@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_band_boxes_replay(kind, g, depth):
    """Host side of the box form of the fused band stage (what single-device solvers run since round 3): every closure cell
    owned by exactly one box, regions within the workgroup budget, and the group-by-group replay of all three uses of the
    kernel -- plain stage, closure stage (band passes + the sweep's values on the band closure), plain stage fed from the
    closure stage's snapshot -- bit-identical to pass-by-pass band smoothing and a full Jacobi sweep (Ops.h:524-619, 262-367).
    Level 0 runs with the domain's face weights, so the general BOUNDARY cells (operator rows) take part."""
    lab, w, off, lev, dx = make_domain(kind, g)
    H = G.Hierarchy(lab, lev)
    any_general = False
    for l in range(H.levels):
        nband = len(H.band_cells(l))
        groups, cells, general = H.check_band_boxes(l, depth, w if l == 0 else None)
        assert (groups > 0) == (nband > 0) and cells >= nband
        any_general = any_general or general > 0
        if l == 0:  # and with unit weights on the same labels
            H.check_band_boxes(l, depth)
    assert any_general == (kind != "simple")


def test_hierarchy_non_cubic(oracle):
    bl = np.full((20, 12, 28), D.INTERIOR, dtype=np.uint8)
    bl[-4:] = D.DIRICHLET
    bw = [np.ones(D.face_shape(20, 12, 28, a), dtype=np.float32) for a in range(3)]
    for a in range(3):
        sl = [slice(None)] * 3
        sl[2 - a] = 0
        bw[a][tuple(sl)] = 0
        sl[2 - a] = -1
        bw[a][tuple(sl)] = 0
    bw[2][-4:] = 0  # air/air faces closed (z axis faces inside the Dirichlet slab)
    lab, w, off, lev = G.build_expanded_domain(bl, bw)
    assert lab.shape == (32, 32, 64) and lev == 3
    H = G.Hierarchy(lab, lev)
    s = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], lev, True)
    for l in range(H.levels):
        assert (H.level_labels(l) == s.level_labels(l)).all() and (H.band_cells(l) == s.band(l)).all()


@pytest.mark.parametrize("shape", [(24, 40, 56), (36, 20, 44)])
@pytest.mark.parametrize("width", [1, 2, 3, 4])
def test_band_list_random_labels_any_width(shape, width, oracle):
    """The tile-by-tile band builder (a 16^3 tile + a halo of width-1 cells) against the oracle's grid-wide
    breadth-first version: random blobs of every label, extents that are not multiples of the tile edge."""
    rng = np.random.default_rng(7 + width)
    nz, ny, nx = shape
    noise = rng.random(shape)
    for _ in range(2):  # smooth a little so that INTERIOR regions are several cells thick
        noise = (noise + np.roll(noise, 1, 0) + np.roll(noise, 1, 1) + np.roll(noise, 1, 2) + np.roll(noise, -1, 0)
                 + np.roll(noise, -1, 1) + np.roll(noise, -1, 2)) / 7
    lab = np.full(shape, D.INTERIOR, dtype=np.uint8)
    lab[noise < np.quantile(noise, 0.15)] = D.EXTERIOR
    lab[noise > np.quantile(noise, 0.85)] = D.DIRICHLET
    lab[0], lab[-1], lab[:, 0], lab[:, -1], lab[:, :, 0], lab[:, :, -1] = (D.EXTERIOR,) * 6
    ones = [np.ones(D.face_shape(nz, ny, nx, a), dtype=np.float32) for a in range(3)]
    D.set_boundary_labels(lab, ones)
    opt = G.default_options()
    opt.band_width = width
    H = G.Hierarchy(lab, 1, options=opt)
    assert (H.band_cells(0) == oracle.build_boundary_cells(lab.astype(np.int32), width)).all()


@pytest.mark.parametrize("block,expect", [(2, 1), (4, 2)])
def test_level_cap_quirk(block, expect, oracle):
    """MG.cpp:243-248: the first level without a solvable cell caps the hierarchy at level - 1 (one more than needed is
    dropped).  A small liquid block in air: coarsening turns it into DIRICHLET once a coarse cell has an air child."""
    n = 32
    lab = np.full((n, n, n), D.EXTERIOR, dtype=np.uint8)
    lab[8:24, 8:24, 8:24] = D.DIRICHLET
    lab[12 : 12 + block, 12 : 12 + block, 12 : 12 + block] = D.INTERIOR
    w = [np.ones(D.face_shape(n, n, n, a), dtype=np.float32) for a in range(3)]
    D.set_boundary_labels(lab, w)
    H = G.Hierarchy(lab, 4)
    s = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], 4, False)
    assert H.levels == s.levels == expect


@pytest.mark.parametrize("kind,g", KINDS)
def test_coarse_solve_matches_scipy_and_oracle(kind, g, oracle):
    lab, w, off, lev, dx = make_domain(kind, g)
    H = G.Hierarchy(lab, lev)
    cl = H.level_labels(H.levels - 1)
    A, idx, act = assemble(cl.astype(np.int32), None)
    rng = np.random.default_rng(11)
    b = np.zeros(cl.shape, dtype=np.float32)
    b[act] = rng.random(A.shape[0])
    x = H.coarse_solve(b)
    ref = spla.spsolve(A.tocsc(), b[act].astype(np.float64))
    assert np.abs(x[act] - ref).max() < 2e-6 * np.abs(ref).max()  # fp32 in / out
    s = oracle.solver(lab.astype(np.int32), [a.astype(np.float64) for a in w], lev, False)
    assert np.abs(x - s.coarse_solve(b.astype(np.float64))).max() < 2e-6 * np.abs(ref).max()


def test_error_conventions():
    lab, w, off, lev, dx = make_domain("simple", 16)
    L = lib()
    h = C.c_void_p()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    nz, ny, nx = lab.shape
    # odd extent
    assert L.mgps_hierarchy_create(C.byref(h), nx - 1, ny, nz, p(lab), 2, None) == 1
    assert b"even" in L.mgps_last_error(None)
    # extents not divisible by 2^(levels-1)
    assert L.mgps_hierarchy_create(C.byref(h), nx, ny, nz, p(lab), 7, None) == 1
    # options struct of the wrong size
    o = G.default_options()
    o.struct_size = 4
    assert L.mgps_hierarchy_create(C.byref(h), nx, ny, nz, p(lab), 2, C.byref(o)) == 1
    # no EXTERIOR shell
    bad = lab.copy()
    bad[0] = D.DIRICHLET
    assert L.mgps_hierarchy_create(C.byref(h), nx, ny, nz, p(bad), 2, None) == 5
    # nothing to solve
    empty = np.full_like(lab, D.EXTERIOR)
    assert L.mgps_hierarchy_create(C.byref(h), nx, ny, nz, p(empty), 2, None) == 5
    # coarsest level too large for the direct solver
    o = G.default_options()
    o.max_coarse_unknowns = 4
    assert L.mgps_hierarchy_create(C.byref(h), nx, ny, nz, p(lab), 2, C.byref(o)) == 6
    assert not h.value
    # NULL handles never crash
    assert L.mgps_apply_vcycle(None, None, None, 0) == 1
    assert L.mgps_levels(None) == 0
    L.mgps_destroy(None)


@pytest.mark.parametrize("n,levels", [(32, 2), (32, 3), (64, 4)])
def test_thin_exterior_shell_is_refused_on_every_level(n, levels):
    """The reference asserts unitTestExteriorCells on EVERY level (MG.cpp:235, 252).  A liquid cube behind a 1-cell shell
    passes the fine-level check, but its coarse levels put active cells on the grid border, which the band builder,
    the row builder and the transfer kernels would index past: the hierarchy must refuse it (status 5, the message
    names the padding), never abort or throw across the C ABI; the same labels with 2^(levels-1) cells of padding pass."""
    L = lib()
    h = C.c_void_p()
    lab = np.full((n, n, n), D.EXTERIOR, dtype=np.uint8)
    lab[1:-1, 1:-1, 1:-1] = D.INTERIOR
    ones = [np.ones(D.face_shape(n, n, n, a), dtype=np.float32) for a in range(3)]
    D.set_boundary_labels(lab, ones)
    assert G.unit_test_exterior_cells(lab)
    rc = L.mgps_hierarchy_create(C.byref(h), n, n, n, lab.ctypes.data_as(C.c_void_p), levels, None)
    assert rc == 5 and not h.value
    msg = L.mgps_last_error(None)
    assert b"EXTERIOR shell" in msg and str(2 ** (levels - 1)).encode() in msg
    pad = 2 ** (levels - 1)
    good = np.full((n, n, n), D.EXTERIOR, dtype=np.uint8)
    good[pad:-pad, pad:-pad, pad:-pad] = D.INTERIOR
    D.set_boundary_labels(good, ones)
    H = G.Hierarchy(good, levels)
    assert H.levels == levels
    for l in range(levels):
        assert G.unit_test_exterior_cells(H.level_labels(l))


def test_status_strings_cover_every_code():
    header = open(os.path.join(ROOT, "include", "mgps.h")).read()
    codes = dict((name, int(v)) for name, v in re.findall(r"(MGPS_(?:OK|ERR_[A-Z_]+)) = (\d+)", header))
    assert codes["MGPS_ERR_INTERNAL"] == 10 and len(codes) == 11
    for name, v in codes.items():
        assert lib().mgps_status_string(v) != b"unknown status", name


def test_host_alloc_without_device_returns_null():
    import torch

    L = lib()
    L.mgps_host_alloc.restype = C.c_void_p
    L.mgps_host_alloc.argtypes = [C.c_size_t]
    L.mgps_host_free.argtypes = [C.c_void_p]
    p = L.mgps_host_alloc(1 << 20)
    if torch.cuda.is_available():
        assert p
        L.mgps_host_free(p)
    else:
        assert not p  # callers fall back to ordinary memory (the Houdini shim's Staging does)
    L.mgps_host_free(None)


def test_create_without_device_fails_loudly():
    """The product has no CPU path: on a box without a HIP device the constructor reports it."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    lab, w, off, lev, dx = make_domain("simple", 16)
    with pytest.raises(G.MgpsError) as e:
        G.GeometricMultigridPoissonSolver(lab, w, lev, True)
    assert e.value.status == 2 and "no CPU path" in str(e.value)
