"""Python mirror of include/mgps_fields.h: the plugin-side field pre/post-processing of
HDK_GeometricFreeSurfacePressureSolver::solveGasSubclass (Plug.cpp:297-426, 631-714) on the device, with
torch CUDA tensors as device memory.  Names follow the reference's functions.

Grids are (nz, ny, nx) tensors (x fastest); face grids of axis a have one more entry along a (axis 0 = x =
last tensor dimension).  Material labels int32: 0 SOLID, 1 LIQUID, 2 AIR.
"""
import ctypes as C

import torch

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
