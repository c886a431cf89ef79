"""Python mirror of include/mgps_fields.h: the plugin-side field pre/post-processing of
HDK_GeometricFreeSurfacePressureSolver::solveGasSubclass (Plug.cpp:297-426, 631-714) on the device, with
torch CUDA tensors as device memory.  Names follow the reference's functions.

Grids are (nz, ny, nx) tensors (x fastest); face grids of axis a have one more entry along a (axis 0 = x =
last tensor dimension).  Material labels int32: 0 SOLID, 1 LIQUID, 2 AIR.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib

SOLID_CELL, LIQUID_CELL, AIR_CELL = 0, 1, 2


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _g(shape):
    gz, gy, gx = shape
    return gx, gy, gz


def _face_shape(shape, axis):
    s = list(shape)
    s[2 - axis] += 1
    return tuple(s)


def _chk(t, shape, dtype):
    assert t.is_cuda and t.is_contiguous() and t.dtype == dtype and tuple(t.shape) == tuple(shape), (t.shape, shape, t.dtype)
    return t


def buildMaterialCellLabels(liquid_surface, solid_surface, cut_cell_weights):
    """Util.cpp:87-148.  SDFs sampled at cell centres, cut-cell weights on the three face grids."""
    shape = tuple(liquid_surface.shape)
    cw = [_chk(cut_cell_weights[a], _face_shape(shape, a), torch.float32) for a in range(3)]
    out = torch.empty(shape, dtype=torch.int32, device=liquid_surface.device)
    check(lib().mgps_fields_material_labels(_p(out), _p(_chk(liquid_surface, shape, torch.float32)), _p(_chk(solid_surface, shape, torch.float32)),
                                            _p(cw[0]), _p(cw[1]), _p(cw[2]), *_g(shape), _stream()))
    return out


def buildValidFaces(material, cut_cell_weights):
    """Plug.cpp:716-744: one uint8 face grid per axis."""
    shape = tuple(material.shape)
    out = []
    for a in range(3):
        v = torch.empty(_face_shape(shape, a), dtype=torch.uint8, device=material.device)
        check(lib().mgps_fields_valid_faces(a, _p(v), _p(_chk(material, shape, torch.int32)),
                                            _p(_chk(cut_cell_weights[a], _face_shape(shape, a), torch.float32)), *_g(shape), _stream()))
        out.append(v)
    return out


def buildMGDomain(material, cut_cell_weights, liquid_surface, valid_faces, expanded_shape, offset):
    """buildMGDomainLabels + buildMGBoundaryWeights x3 + setBoundaryCellLabels, written straight into the
    expanded solver grid (Plug.cpp:344-362, 746-865): returns (labels uint8, [wx, wy, wz] float32)."""
    shape, eshape = tuple(material.shape), tuple(expanded_shape)
    dev = material.device
    labels = torch.empty(eshape, dtype=torch.uint8, device=dev)
    check(lib().mgps_fields_domain_labels(_p(labels), _p(material), *_g(shape), *_g(eshape), int(offset), _stream()))
    weights = []
    for a in range(3):
        w = torch.empty(_face_shape(eshape, a), dtype=torch.float32, device=dev)
        check(lib().mgps_fields_boundary_weights(a, _p(w), _p(cut_cell_weights[a]), _p(_chk(liquid_surface, shape, torch.float32)),
                                                 _p(_chk(valid_faces[a], _face_shape(shape, a), torch.uint8)), _p(material), *_g(shape),
                                                 *_g(eshape), int(offset), _stream()))
        weights.append(w)
    check(lib().mgps_fields_set_boundary_labels(_p(labels), _p(weights[0]), _p(weights[1]), _p(weights[2]), *_g(eshape), _stream()))
    return labels, weights


def buildRHS(material, velocity, cut_cell_weights, expanded_shape, offset, solid_velocity=None):
    """Plug.cpp:867-943."""
    shape, eshape = tuple(material.shape), tuple(expanded_shape)
    rhs = torch.empty(eshape, dtype=torch.float32, device=material.device)
    sv = solid_velocity if solid_velocity is not None else [None, None, None]
    check(lib().mgps_fields_rhs(_p(rhs), _p(material), *[_p(_chk(velocity[a], _face_shape(shape, a), torch.float32)) for a in range(3)],
                                *[_p(sv[a]) for a in range(3)], *[_p(cut_cell_weights[a]) for a in range(3)], *_g(shape), *_g(eshape),
                                int(offset), _stream()))
    return rhs


def applyOldPressure(pressure, material, expanded_shape, offset):
    """Plug.cpp:945-997: the warm-start solution grid."""
    shape, eshape = tuple(material.shape), tuple(expanded_shape)
    x = torch.empty(eshape, dtype=torch.float32, device=material.device)
    check(lib().mgps_fields_pressure_to_solution(_p(x), _p(_chk(pressure, shape, torch.float32)), _p(material), *_g(shape), *_g(eshape),
                                                 int(offset), _stream()))
    return x


def applySolutionToPressure(pressure, solution, material, offset):
    """Plug.cpp:999-1047 (in place on `pressure`)."""
    shape, eshape = tuple(material.shape), tuple(solution.shape)
    check(lib().mgps_fields_solution_to_pressure(_p(_chk(pressure, shape, torch.float32)), _p(solution), _p(material), *_g(shape),
                                                 *_g(eshape), int(offset), _stream()))
    return pressure


def applyPressureGradient(velocity, liquid_surface, pressure, valid_faces, material):
    """Plug.cpp:1049-1131 (in place on the three velocity face grids)."""
    shape = tuple(material.shape)
    for a in range(3):
        check(lib().mgps_fields_pressure_gradient(a, _p(_chk(velocity[a], _face_shape(shape, a), torch.float32)), _p(liquid_surface),
                                                  _p(pressure), _p(valid_faces[a]), _p(material), *_g(shape), _stream()))
    return velocity


def computeResultingDivergence(material, velocity, cut_cell_weights, solid_velocity=None):
    """Plug.cpp:1133-1207: (accumulated divergence, max divergence, liquid cell count)."""
    shape = tuple(material.shape)
    out = (C.c_double * 3)()
    sv = solid_velocity if solid_velocity is not None else [None, None, None]
    check(lib().mgps_fields_divergence(out, _p(material), *[_p(velocity[a]) for a in range(3)], *[_p(sv[a]) for a in range(3)],
                                       *[_p(cut_cell_weights[a]) for a in range(3)], *_g(shape), _stream()))
    return out[0], out[1], out[2]


# ---- the whole projection in one call on host arrays (what the Houdini shim calls) ---------------------------------
class Projection(C.Structure):
    """mgps_projection (include/mgps_fields.h)."""

    _fields_ = [
        ("struct_size", C.c_int), ("gx", C.c_int), ("gy", C.c_int), ("gz", C.c_int), ("real_bytes", C.c_int),
        ("liquid_phi", C.c_void_p), ("solid_phi", C.c_void_p), ("cut_weights", C.c_void_p * 3), ("velocity", C.c_void_p * 3),
        ("solid_velocity", C.c_void_p * 3), ("pressure", C.c_void_p), ("valid_faces", C.c_void_p * 3),
        ("use_old_pressure", C.c_int), ("use_mg_preconditioner", C.c_int), ("use_gauss_seidel", C.c_int),
        ("tolerance", C.c_double), ("max_iterations", C.c_int), ("power_of_two", C.c_int),
        ("stats", _lib.PcgStats), ("mg_levels", C.c_int), ("offset", C.c_int), ("expanded", C.c_int * 3),
        ("liquid_cells", C.c_double), ("residual_inf", C.c_double), ("residual_l2", C.c_double),
        ("divergence_sum", C.c_double), ("divergence_max", C.c_double),
        ("setup_ms", C.c_double), ("solve_ms", C.c_double), ("total_ms", C.c_double),
    ]


def project_free_surface(liquid_phi, solid_phi, cut_weights, velocity, pressure, solid_velocity=None, use_old_pressure=True,
                         use_mg_preconditioner=True, use_gauss_seidel=True, tolerance=1e-5, max_iterations=2500, power_of_two=True,
                         options=None):
    """solveGasSubclass (Plug.cpp:252-707) on numpy host arrays of one dtype (float32 or float64): `velocity` and `pressure`
    are updated in place; returns (valid_faces[3] uint8, info dict)."""
    import numpy as np

    dt = np.dtype(pressure.dtype)
    assert dt in (np.dtype(np.float32), np.dtype(np.float64))
    shape = tuple(liquid_phi.shape)

    def chk(a, sh):
        assert a.dtype == dt and a.flags.c_contiguous and tuple(a.shape) == tuple(sh), (a.dtype, a.shape, sh)
        return a.ctypes.data_as(C.c_void_p)

    pr = Projection()
    pr.struct_size = C.sizeof(Projection)
    pr.gz, pr.gy, pr.gx = shape
    pr.real_bytes = dt.itemsize
    pr.liquid_phi, pr.solid_phi, pr.pressure = chk(liquid_phi, shape), chk(solid_phi, shape), chk(pressure, shape)
    valid = []
    for a in range(3):
        fs = _face_shape(shape, a)
        pr.cut_weights[a] = chk(cut_weights[a], fs)
        pr.velocity[a] = chk(velocity[a], fs)
        pr.solid_velocity[a] = chk(solid_velocity[a], fs) if solid_velocity is not None else None
        valid.append(np.zeros(fs, dtype=np.uint8))
        pr.valid_faces[a] = valid[a].ctypes.data_as(C.c_void_p)
    pr.use_old_pressure, pr.use_mg_preconditioner, pr.use_gauss_seidel = int(use_old_pressure), int(use_mg_preconditioner), int(use_gauss_seidel)
    pr.tolerance, pr.max_iterations, pr.power_of_two = float(tolerance), int(max_iterations), int(power_of_two)
    check(lib().mgps_project_free_surface(C.byref(pr), C.byref(options) if options is not None else None))
    info = {
        "iterations": pr.stats.iterations, "outcome": pr.stats.outcome, "rel_residual": pr.stats.rel_residual,
        "rel_residual_recomputed": pr.stats.rel_residual_recomputed, "mg_levels": pr.mg_levels, "offset": pr.offset,
        "expanded": (pr.expanded[2], pr.expanded[1], pr.expanded[0]), "liquid_cells": pr.liquid_cells,
        "residual_inf": pr.residual_inf, "residual_l2": pr.residual_l2, "divergence_sum": pr.divergence_sum,
        "divergence_max": pr.divergence_max, "setup_ms": pr.setup_ms, "solve_ms": pr.solve_ms, "total_ms": pr.total_ms,
    }
    return valid, info
