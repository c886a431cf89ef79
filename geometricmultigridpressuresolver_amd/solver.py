"""Host-side mirror of the reference's solver interface over the C ABI.

Names follow the reference so that a test written against it reads the same here:

    HDK::GeometricMultigridPoissonSolver(labels, weights, mgLevels, useGaussSeidel)   MG.h:20-24
        .applyVCycle(solution, rhs, useInitialGuess)                                  MG.h:26-29
        .getMGLevels()                                                                MG.h:31
    HDK::GeometricMultigridOperators::{jacobiPoissonSmoother, tiledGaussSeidelPoissonSmoother,
        boundaryJacobiPoissonSmoother, applyPoissonMatrix, computePoissonResidual, downsample,
        upsampleAndAdd, dotProduct, squaredL2Norm, l2Norm, infNorm, addToVector, addVectors,
        scaleVector}                                                                  Ops.h:19-174
    HDK::solveGeometricConjugateGradient(...)                                         CG.h:18-27

Grids are torch CUDA float32 tensors of shape (nz, ny, nx) -- torch is used for device memory and
streams only; every operation is a call into libmgps.so on the tensor's data pointer.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Options, PcgStats, check, lib

PCG_OUTCOMES = {0: "converged", 1: "rhs_zero", 2: "already_converged", 3: "max_iterations"}


def default_options():
    o = Options()
    lib().mgps_default_options(C.byref(o))
    return o


def _np_u8(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.uint8)


def _np_f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- domain expansion (host arrays), Ops.h:1328-1644 ---------------------------------------------
def expanded_layout(base_shape, levels=0, power_of_two=True):
    bz, by, bx = base_shape
    dims = (C.c_int * 3)()
    off = C.c_int()
    lev = C.c_int()
    check(lib().mgps_expanded_layout(bx, by, bz, int(levels), int(bool(power_of_two)), dims, C.byref(off), C.byref(lev)))
    return (dims[2], dims[1], dims[0]), off.value, lev.value


def build_expanded_domain(base_labels, base_weights, levels=0, power_of_two=True):
    """buildExpandedCellLabels + buildExpandedBoundaryWeights x3 + setBoundaryCellLabels
    (Plug.cpp:344-362).  Returns (labels uint8, [wx, wy, wz] float32, offset, levels)."""
    base_labels = _np_u8(base_labels)
    bz, by, bx = base_labels.shape
    (ez, ey, ex), off, lev = expanded_layout(base_labels.shape, levels, power_of_two)
    labels = np.empty((ez, ey, ex), dtype=np.uint8)
    check(lib().mgps_expand_labels(_p(labels), _p(base_labels), bx, by, bz, ex, ey, ez, off))
    weights = []
    for axis in range(3):
        bw = _np_f32(base_weights[axis])
        shape = [ez, ey, ex]
        shape[2 - axis] += 1
        w = np.empty(shape, dtype=np.float32)
        check(lib().mgps_expand_weights(_p(w), _p(bw), axis, bx, by, bz, ex, ey, ez, off))
        weights.append(w)
    check(lib().mgps_set_boundary_labels(_p(labels), _p(weights[0]), _p(weights[1]), _p(weights[2]), ex, ey, ez))
    return labels, weights, off, lev


def set_boundary_cell_labels(labels, weights):
    labels = _np_u8(labels)
    nz, ny, nx = labels.shape
    w = [_np_f32(a) for a in weights]
    check(lib().mgps_set_boundary_labels(_p(labels), _p(w[0]), _p(w[1]), _p(w[2]), nx, ny, nz))
    return labels


def unit_test_boundary_cells(labels, weights=None):
    labels = _np_u8(labels)
    nz, ny, nx = labels.shape
    ok = C.c_int()
    w = [None] * 3 if weights is None else [_p(_np_f32(a)) for a in weights]
    keep = weights
    check(lib().mgps_check_boundary_cells(_p(labels), w[0], w[1], w[2], nx, ny, nz, C.byref(ok)))
    return bool(ok.value)


def unit_test_exterior_cells(labels):
    labels = _np_u8(labels)
    nz, ny, nx = labels.shape
    ok = C.c_int()
    check(lib().mgps_check_exterior_cells(_p(labels), nx, ny, nz, C.byref(ok)))
    return bool(ok.value)


def unit_test_coarsening(coarse, fine):
    coarse, fine = _np_u8(coarse), _np_u8(fine)
    nz, ny, nx = fine.shape
    ok = C.c_int()
    check(lib().mgps_check_coarsening(_p(coarse), _p(fine), nx, ny, nz, C.byref(ok)))
    return bool(ok.value)


class Hierarchy:
    """Host-only multigrid hierarchy (no GPU needed): coarse labels, band lists, coarse solve."""

    def __init__(self, labels=None, mg_levels=None, options=None, _borrowed=None):
        self._own = _borrowed is None
        if _borrowed is not None:
            self.h = C.c_void_p(_borrowed)
            return
        labels = _np_u8(labels)
        nz, ny, nx = labels.shape
        h = C.c_void_p()
        opt = options if options is not None else default_options()
        check(lib().mgps_hierarchy_create(C.byref(h), nx, ny, nz, _p(labels), int(mg_levels), C.byref(opt)))
        self.h = h

    def close(self):
        if self._own and self.h:
            lib().mgps_hierarchy_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def levels(self):
        return lib().mgps_hierarchy_levels(self.h)

    def level_shape(self, level):
        d = (C.c_int * 3)()
        check(lib().mgps_hierarchy_level_dims(self.h, level, d))
        return (d[2], d[1], d[0])

    def level_labels(self, level):
        out = np.empty(self.level_shape(level), dtype=np.uint8)
        check(lib().mgps_hierarchy_level_labels(self.h, level, _p(out)))
        return out

    def band_cells(self, level):
        n = lib().mgps_hierarchy_band_count(self.h, level)
        out = np.empty((max(n, 0), 3), dtype=np.int32)
        if n > 0:
            check(lib().mgps_hierarchy_band_cells(self.h, level, _p(out)))
        return out

    def check_band_boxes(self, level, depth=3, weights=None):
        """host self-check of the box form of the fused band stage (level 0 with `weights` = [wx, wy, wz]: general BOUNDARY
        cells take part); returns (groups, region cells, general entries)"""
        g, n, q = C.c_int64(), C.c_int64(), C.c_int64()
        w = [None, None, None]
        if weights is not None:
            w = [_np_f32(a) for a in weights]
        check(lib().mgps_hierarchy_check_band_boxes(self.h, int(level), int(depth), *[(_p(a) if a is not None else None) for a in w],
                                                    C.byref(g), C.byref(n), C.byref(q)))
        return g.value, n.value, q.value

    @property
    def coarse_unknowns(self):
        return lib().mgps_hierarchy_coarse_unknowns(self.h)

    def coarse_solve(self, b):
        b = _np_f32(b)
        x = np.zeros_like(b)
        check(lib().mgps_hierarchy_coarse_solve(self.h, _p(x), _p(b)))
        return x


class GeometricMultigridPoissonSolver:
    """MG.h:10-53 on the GPU.  labels: (nz, ny, nx) uint8 host array; weights: three float32 host
    arrays with one extra entry along their axis."""

    def __init__(self, labels, weights, mg_levels, use_gauss_seidel, do_print_stats=False, device=None, options=None):
        on_device = all(torch.is_tensor(a) and a.is_cuda for a in weights)
        labels_on_device = on_device and torch.is_tensor(labels) and labels.is_cuda and labels.dtype == torch.uint8
        if labels_on_device:  # mgps_create_device: the library fetches its own host copy (1 byte per cell)
            labels = labels.contiguous()
        else:
            if torch.is_tensor(labels):
                labels = labels.cpu().numpy()
            labels = _np_u8(labels)
        nz, ny, nx = labels.shape
        if on_device:  # weights written on the device (fields.buildMGDomain) never cross to the host
            w = [a.contiguous() for a in weights]
            assert all(a.dtype == torch.float32 for a in w)
        else:
            w = [_np_f32(a) for a in weights]
        assert tuple(w[0].shape) == (nz, ny, nx + 1) and tuple(w[1].shape) == (nz, ny + 1, nx) and tuple(w[2].shape) == (nz + 1, ny, nx)
        opt = options if options is not None else default_options()
        opt.print_stats = int(do_print_stats) if not isinstance(do_print_stats, bool) else int(do_print_stats)
        if device is not None:
            opt.device = torch.device(device).index if not isinstance(device, int) else device
        elif on_device:
            opt.device = w[0].device.index
        self.h = C.c_void_p()
        wp = [C.c_void_p(a.data_ptr()) for a in w] if on_device else [_p(a) for a in w]
        create = lib().mgps_create_device if labels_on_device else lib().mgps_create_device_weights if on_device else lib().mgps_create
        lp = C.c_void_p(labels.data_ptr()) if labels_on_device else _p(labels)
        check(create(C.byref(self.h), nx, ny, nz, lp, wp[0], wp[1], wp[2], int(mg_levels), int(bool(use_gauss_seidel)), C.byref(opt)))
        self.shape = (nz, ny, nx)
        self.use_gauss_seidel = bool(use_gauss_seidel)
        dev_index = opt.device if opt.device >= 0 else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        self.use_torch_stream()

    # -- plumbing --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            lib().mgps_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_torch_stream(self):
        """Enqueue on torch's current stream of the solver's device."""
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(lib().mgps_set_stream(self.h, C.c_void_p(s)), self.h)

    def synchronize(self):
        check(lib().mgps_synchronize(self.h), self.h)

    def hierarchy(self):
        return Hierarchy(_borrowed=lib().mgps_get_hierarchy(self.h))

    LEVEL_ARRAYS = {"codes": (0, np.uint8), "band": (1, np.int32), "band_diag": (2, np.uint8), "rows": (3, np.float32), "chunks": (4, np.int32),
                    "plane_blocks": (5, np.int32), "pure_even": (6, np.int32), "pure_odd": (7, np.int32), "mixed_even": (8, np.int32),
                    "mixed_odd": (9, np.int32), "tile_bnd_start": (10, np.int32), "box_info": (11, np.int32), "box_list": (12, np.uint32),
                    "box_general": (13, np.int32)}

    def level_array(self, level, name):
        """mgps_level_array: one of the set-up arrays of a level as it sits on the device (tests / tools)."""
        which, dt = self.LEVEL_ARRAYS[name]
        n = C.c_int64()
        check(lib().mgps_level_array(self.h, int(level), which, None, C.byref(n)), self.h)
        out = np.empty(n.value, dtype=dt)
        if n.value:
            check(lib().mgps_level_array(self.h, int(level), which, _p(out), C.byref(n)), self.h)
        return out

    def level_shape(self, level):
        d = (C.c_int * 3)()
        check(lib().mgps_level_dims(self.h, level, d), self.h)
        return (d[2], d[1], d[0])

    def new_grid(self, level=0):
        return torch.zeros(self.level_shape(level), dtype=torch.float32, device=self.device)

    def to_device(self, a, level=0):
        t = torch.from_numpy(_np_f32(a)).to(self.device)
        assert tuple(t.shape) == self.level_shape(level)
        return t

    def _g(self, t, level=0):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
        assert tuple(t.shape) == self.level_shape(level), (tuple(t.shape), self.level_shape(level))
        return C.c_void_p(t.data_ptr())

    # -- MG.h ------------------------------------------------------------------------------------
    def getMGLevels(self):
        return lib().mgps_levels(self.h)

    def applyVCycle(self, solution, rhs, use_initial_guess=False):
        check(lib().mgps_apply_vcycle(self.h, self._g(solution), self._g(rhs), int(bool(use_initial_guess))), self.h)

    # -- Ops.h -----------------------------------------------------------------------------------
    def jacobiPoissonSmoother(self, solution, rhs, level=0):
        check(lib().mgps_jacobi_smooth(self.h, level, self._g(solution, level), self._g(rhs, level)), self.h)

    def tiledGaussSeidelPoissonSmoother(self, solution, rhs, do_smooth_odd_tiles, do_smooth_forward, level=0):
        check(
            lib().mgps_tiled_gs_smooth(
                self.h, level, self._g(solution, level), self._g(rhs, level), int(bool(do_smooth_odd_tiles)), int(bool(do_smooth_forward))
            ),
            self.h,
        )

    def boundaryJacobiPoissonSmoother(self, solution, rhs, level=0):
        check(lib().mgps_boundary_jacobi_smooth(self.h, level, self._g(solution, level), self._g(rhs, level)), self.h)

    def boundaryJacobiStage(self, solution, rhs, level=0):
        """options.band_iterations band passes, as a smoothing stroke runs them (fused where possible)"""
        check(lib().mgps_boundary_jacobi_stage(self.h, level, self._g(solution, level), self._g(rhs, level)), self.h)

    def applyPoissonMatrix(self, destination, source, level=0):
        check(lib().mgps_apply_poisson(self.h, level, self._g(destination, level), self._g(source, level)), self.h)

    def computePoissonResidual(self, residual, solution, rhs, level=0):
        check(lib().mgps_residual(self.h, level, self._g(residual, level), self._g(solution, level), self._g(rhs, level)), self.h)

    def downsample(self, destination, source, fine_level=0):
        check(lib().mgps_downsample(self.h, fine_level, self._g(destination, fine_level + 1), self._g(source, fine_level)), self.h)

    def residualDownsample(self, destination, solution, rhs, fine_level=0):
        """mgps_residual_downsample: destination (level fine_level + 1) = downsample(rhs - A solution) as a down-stroke forms it"""
        check(lib().mgps_residual_downsample(self.h, fine_level, self._g(destination, fine_level + 1), self._g(solution, fine_level), self._g(rhs, fine_level)), self.h)

    def upsampleAndAdd(self, destination, source, fine_level=0):
        check(lib().mgps_upsample_add(self.h, fine_level, self._g(destination, fine_level), self._g(source, fine_level + 1)), self.h)

    def coarseDirectSolve(self, solution, rhs):
        L = self.getMGLevels() - 1
        check(lib().mgps_coarse_solve(self.h, self._g(solution, L), self._g(rhs, L)), self.h)

    def dotProduct(self, a, b, level=0):
        out = C.c_double()
        check(lib().mgps_dot(self.h, level, self._g(a, level), self._g(b, level), C.byref(out)), self.h)
        return out.value

    def squaredL2Norm(self, a, level=0):
        out = C.c_double()
        check(lib().mgps_squared_l2_norm(self.h, level, self._g(a, level), C.byref(out)), self.h)
        return out.value

    def l2Norm(self, a, level=0):
        out = C.c_double()
        check(lib().mgps_l2_norm(self.h, level, self._g(a, level), C.byref(out)), self.h)
        return out.value

    def infNorm(self, a, level=0, reference_signed_max=True):
        out = C.c_double()
        check(lib().mgps_inf_norm(self.h, level, self._g(a, level), int(bool(reference_signed_max)), C.byref(out)), self.h)
        return out.value

    def addToVector(self, destination, source, scale, level=0):
        check(lib().mgps_add_to_vector(self.h, level, self._g(destination, level), self._g(source, level), C.c_double(scale)), self.h)

    def addVectors(self, destination, source, scaled_source, scale, level=0):
        check(
            lib().mgps_add_vectors(
                self.h, level, self._g(destination, level), self._g(source, level), self._g(scaled_source, level), C.c_double(scale)
            ),
            self.h,
        )

    def scaleVector(self, vector, scale, level=0):
        check(lib().mgps_scale_vector(self.h, level, self._g(vector, level), C.c_double(scale)), self.h)

    def zeroInactive(self, grid, level=0):
        """grid = 0 outside active cells (the invariant every operator relies on; see mgps_zero_inactive)"""
        check(lib().mgps_zero_inactive(self.h, level, self._g(grid, level)), self.h)

    # -- CG.h ------------------------------------------------------------------------------------
    def solveGeometricConjugateGradient(self, solution, rhs, tolerance=1e-5, max_iterations=2500, use_mg_preconditioner=True):
        st = PcgStats()
        check(
            lib().mgps_solve_pcg(
                self.h, self._g(solution), self._g(rhs), C.c_double(tolerance), int(max_iterations), int(bool(use_mg_preconditioner)), C.byref(st)
            ),
            self.h,
        )
        return {
            "outcome": PCG_OUTCOMES.get(st.outcome, st.outcome),
            "iterations": st.iterations,
            "rel_residual": st.rel_residual,
            "rel_residual_recomputed": st.rel_residual_recomputed,
            "rhs_norm2": st.rhs_norm2,
            "solve_ms": st.solve_ms,
        }

    # -- measurement hooks ---------------------------------------------------------------------------
    def profile_enable(self, on=True):
        """on: False / True (fine-smoother events) / 2 (also per-stage events, see stage_times)"""
        check(lib().mgps_profile_enable(self.h, int(on)), self.h)

    def profile_read(self):
        ms, n = C.c_double(), C.c_int()
        check(lib().mgps_profile_read(self.h, C.byref(ms), C.byref(n)), self.h)
        return ms.value, n.value

    def stage_times(self):
        """{stage: ms} summed over levels and cycles since the last call (mgps_stage_times), plus 'cycles'"""
        ms = (C.c_double * 6)()
        n = C.c_int()
        check(lib().mgps_stage_times(self.h, ms, C.byref(n)), self.h)
        names = ("boundary_smoother", "smoother", "residual", "downsample", "direct_solve", "upsample_add")
        out = {k: ms[i] for i, k in enumerate(names)}
        out["cycles"] = n.value
        fine = (C.c_double * 6)()
        check(lib().mgps_stage_times_fine(self.h, fine), self.h)
        out["fine"] = {k: fine[i] for i, k in enumerate(names)}
        return out

    def swept_cells(self, level=0):
        """(stencil sweep cells, tiled-GS sweep cells) one full-domain pass of `level` visits"""
        a, b = C.c_longlong(), C.c_longlong()
        check(lib().mgps_swept_cells(self.h, int(level), C.byref(a), C.byref(b)), self.h)
        return a.value, b.value

    def residual_restrict_fused(self, level=0):
        """mgps_residual_restrict_fused: does a down-stroke of this level run residual + restriction as the z-folded pair?"""
        f = C.c_int()
        check(lib().mgps_residual_restrict_fused(self.h, int(level), C.byref(f)), self.h)
        return bool(f.value)

    def stencil_kernel(self, level=0):
        """'quad' | 'plane' | 'scalar': the kernel the Jacobi / residual / A.x sweeps of `level` launch"""
        k = C.c_int()
        check(lib().mgps_stencil_kernel(self.h, int(level), C.byref(k)), self.h)
        return {1: "quad", 2: "plane", 3: "scalar"}[k.value]

    # -- host-buffer forms (what the Houdini shim calls) --------------------------------------------
    def applyVCycleHost(self, solution, rhs, use_initial_guess=False):
        if np.asarray(solution).dtype == np.float64:
            x = np.ascontiguousarray(solution, dtype=np.float64)
            b = np.ascontiguousarray(rhs, dtype=np.float64)
            check(lib().mgps_apply_vcycle_host_f64(self.h, _p(x), _p(b), int(bool(use_initial_guess))), self.h)
            return x
        x = _np_f32(solution)
        b = _np_f32(rhs)
        check(lib().mgps_apply_vcycle_host(self.h, _p(x), _p(b), int(bool(use_initial_guess))), self.h)
        return x

    def solvePcgHost(self, solution, rhs, tolerance=1e-5, max_iterations=2500, use_mg_preconditioner=True):
        if np.asarray(solution).dtype == np.float64:  # the reference's StoreReal: narrowed / widened on the device
            x = np.ascontiguousarray(solution, dtype=np.float64)
            b = np.ascontiguousarray(rhs, dtype=np.float64)
            st = PcgStats()
            check(lib().mgps_solve_pcg_host_f64(self.h, _p(x), _p(b), C.c_double(tolerance), int(max_iterations), int(bool(use_mg_preconditioner)),
                                                C.byref(st)), self.h)
            return x, {"outcome": PCG_OUTCOMES.get(st.outcome, st.outcome), "iterations": st.iterations, "rel_residual": st.rel_residual}
        x = _np_f32(solution)
        b = _np_f32(rhs)
        st = PcgStats()
        check(
            lib().mgps_solve_pcg_host(
                self.h, _p(x), _p(b), C.c_double(tolerance), int(max_iterations), int(bool(use_mg_preconditioner)), C.byref(st)
            ),
            self.h,
        )
        return x, {"outcome": PCG_OUTCOMES.get(st.outcome, st.outcome), "iterations": st.iterations, "rel_residual": st.rel_residual}
