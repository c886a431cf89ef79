// C++ host-side mirror of the reference's solver interface over the C ABI (include/mgps.h).
//
// The reference's hot path is C++ against Houdini's UT_VoxelArray; this header gives a maintainer the
// same names and argument meaning on flat, x-fastest arrays, so the call sequence of
// HDK_GeometricFreeSurfacePressureSolver::solveGasSubclass (Plug.cpp:344-362, 426-629) ports line by
// line (see INTEGRATION.md for the UT_VoxelArray <-> flat conversion it needs around it):
//
//   HDK::GeometricMultigridPoissonSolver(labels, weights[3], mgLevels, useGaussSeidel)   MG.h:20-24
//       .applyVCycle(solution, rhs, useInitialGuess)                                      MG.h:26-29
//       .getMGLevels()                                                                    MG.h:31
//   HDK::GeometricMultigridOperators::{applyPoissonMatrix, computePoissonResidual, dotProduct,
//       squaredL2Norm, l2Norm, infNorm, addToVector, addVectors, scaleVector,
//       buildExpandedCellLabels, buildExpandedBoundaryWeights, setBoundaryCellLabels}     Ops.h:19-174
//   HDK::solveGeometricConjugateGradient(...)                                             CG.h:18-27
//
// Differences that are deliberate and visible in the signatures:
//   * grids are mgps::DeviceGrid (a float array in HBM owned by the solver's device) or
//     std::vector<float> for the host forms -- not UT_VoxelArray<double>;
//   * the free functions take the solver (it owns the labels / weights on the device) instead of
//     label and weight grids;
//   * errors are exceptions of type mgps::Error carrying the C status, thrown on THIS side of the ABI
//     (the reference asserts or carries on; nothing throws across the C boundary).
// Header only; link with libmgps.so.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "mgps.h"

namespace HDK {
class GeometricMultigridPoissonSolver;
}

namespace mgps {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

inline void check(int status, const mgps_solver *h = nullptr)
{
    if (status != MGPS_OK) throw Error(status, std::string(mgps_status_string(status)) + ": " + mgps_last_error(h));
}

// Host-side voxel grid, x fastest (the logical order of UT_VoxelArray).
template <class T>
struct Grid {
    int nx = 0, ny = 0, nz = 0;
    std::vector<T> v;
    Grid() = default;
    Grid(int x, int y, int z, T fill = T()) : nx(x), ny(y), nz(z), v(size_t(x) * y * z, fill) {}
    T &operator()(int i, int j, int k) { return v[(size_t(k) * ny + j) * nx + i]; }
    const T &operator()(int i, int j, int k) const { return v[(size_t(k) * ny + j) * nx + i]; }
};

// A grid in device memory (level 0 unless stated), owned by its solver.
class DeviceGrid {
public:
    DeviceGrid(HDK::GeometricMultigridPoissonSolver &s, int level = 0);
    ~DeviceGrid();
    DeviceGrid(const DeviceGrid &) = delete;
    DeviceGrid &operator=(const DeviceGrid &) = delete;
    void upload(const std::vector<float> &host);
    void download(std::vector<float> &host) const;
    float *data() { return p_; }
    const float *data() const { return p_; }

private:
    HDK::GeometricMultigridPoissonSolver &s_;
    int level_;
    float *p_ = nullptr;
};

}  // namespace mgps

namespace HDK {

namespace GeometricMultigridOperators {

enum CellLabels { INTERIOR_CELL = MGPS_INTERIOR_CELL, EXTERIOR_CELL = MGPS_EXTERIOR_CELL,
                  DIRICHLET_CELL = MGPS_DIRICHLET_CELL, BOUNDARY_CELL = MGPS_BOUNDARY_CELL };  // Ops.h:11

// buildExpandedCellLabels (Ops.h:1328-1456): returns (exteriorOffset, mgLevels)
inline std::pair<int, int> buildExpandedCellLabels(mgps::Grid<uint8_t> &expanded, const mgps::Grid<uint8_t> &base)
{
    int dims[3], offset = 0, levels = 0;
    mgps::check(mgps_expanded_layout(base.nx, base.ny, base.nz, 0, 1, dims, &offset, &levels));
    expanded = mgps::Grid<uint8_t>(dims[0], dims[1], dims[2]);
    mgps::check(mgps_expand_labels(expanded.v.data(), base.v.data(), base.nx, base.ny, base.nz, dims[0], dims[1], dims[2], offset));
    return {offset, levels};
}

// buildExpandedBoundaryWeights (Ops.h:1458-1572)
inline void buildExpandedBoundaryWeights(mgps::Grid<float> &expanded, const mgps::Grid<float> &base,
                                         const mgps::Grid<uint8_t> &expandedLabels, int exteriorOffset, int axis)
{
    const int ex = expandedLabels.nx, ey = expandedLabels.ny, ez = expandedLabels.nz;
    expanded = mgps::Grid<float>(ex + (axis == 0), ey + (axis == 1), ez + (axis == 2));
    mgps::check(mgps_expand_weights(expanded.v.data(), base.v.data(), axis, base.nx - (axis == 0), base.ny - (axis == 1),
                                    base.nz - (axis == 2), ex, ey, ez, exteriorOffset));
}

// setBoundaryCellLabels (Ops.h:1574-1644)
inline void setBoundaryCellLabels(mgps::Grid<uint8_t> &labels, const std::array<mgps::Grid<float>, 3> &weights)
{
    mgps::check(mgps_set_boundary_labels(labels.v.data(), weights[0].v.data(), weights[1].v.data(), weights[2].v.data(),
                                         labels.nx, labels.ny, labels.nz));
}

}  // namespace GeometricMultigridOperators

// MG.h:10-53
class GeometricMultigridPoissonSolver {
public:
    GeometricMultigridPoissonSolver(const mgps::Grid<uint8_t> &initialCellLabels,
                                    const std::array<mgps::Grid<float>, 3> &boundaryWeights, int mgLevels,
                                    bool useGaussSeidel, bool doPrintStats = false, int device = -1)
    {
        mgps_options opt;
        mgps_default_options(&opt);
        opt.print_stats = doPrintStats;
        opt.device = device;
        mgps::check(mgps_create(&h_, initialCellLabels.nx, initialCellLabels.ny, initialCellLabels.nz,
                                initialCellLabels.v.data(), boundaryWeights[0].v.data(), boundaryWeights[1].v.data(),
                                boundaryWeights[2].v.data(), mgLevels, useGaussSeidel ? 1 : 0, &opt));
    }
    ~GeometricMultigridPoissonSolver() { mgps_destroy(h_); }
    GeometricMultigridPoissonSolver(const GeometricMultigridPoissonSolver &) = delete;
    GeometricMultigridPoissonSolver &operator=(const GeometricMultigridPoissonSolver &) = delete;

    void applyVCycle(mgps::DeviceGrid &solution, const mgps::DeviceGrid &rhs, bool useInitialGuess = false)
    {
        mgps::check(mgps_apply_vcycle(h_, solution.data(), rhs.data(), useInitialGuess), h_);
    }
    // host-array form: upload, one V-cycle, download
    void applyVCycle(std::vector<float> &solution, const std::vector<float> &rhs, bool useInitialGuess = false)
    {
        mgps::check(mgps_apply_vcycle_host(h_, solution.data(), rhs.data(), useInitialGuess), h_);
    }
    int getMGLevels() const { return mgps_levels(h_); }
    mgps_solver *handle() { return h_; }

private:
    mgps_solver *h_ = nullptr;
};

namespace GeometricMultigridOperators {

using Solver = GeometricMultigridPoissonSolver;
using mgps::DeviceGrid;

inline void applyPoissonMatrix(Solver &s, DeviceGrid &destination, const DeviceGrid &source)  // Ops.h:621-714
{
    mgps::check(mgps_apply_poisson(s.handle(), 0, destination.data(), source.data()), s.handle());
}
inline void computePoissonResidual(Solver &s, DeviceGrid &residual, const DeviceGrid &solution, const DeviceGrid &rhs)  // Ops.h:716-732
{
    mgps::check(mgps_residual(s.handle(), 0, residual.data(), solution.data(), rhs.data()), s.handle());
}
inline double dotProduct(Solver &s, const DeviceGrid &a, const DeviceGrid &b)  // Ops.h:1020-1085
{
    double v = 0;
    mgps::check(mgps_dot(s.handle(), 0, a.data(), b.data(), &v), s.handle());
    return v;
}
inline double squaredL2Norm(Solver &s, const DeviceGrid &a)  // Ops.h:1205-1265
{
    double v = 0;
    mgps::check(mgps_squared_l2_norm(s.handle(), 0, a.data(), &v), s.handle());
    return v;
}
inline double l2Norm(Solver &s, const DeviceGrid &a)  // Ops.h:1197-1203
{
    double v = 0;
    mgps::check(mgps_l2_norm(s.handle(), 0, a.data(), &v), s.handle());
    return v;
}
inline double infNorm(Solver &s, const DeviceGrid &a)  // Ops.h:1267-1326: max(0, max v), as the reference
{
    double v = 0;
    mgps::check(mgps_inf_norm(s.handle(), 0, a.data(), 1, &v), s.handle());
    return v;
}
inline void addToVector(Solver &s, DeviceGrid &destination, const DeviceGrid &source, double scale)  // Ops.h:1087-1137
{
    mgps::check(mgps_add_to_vector(s.handle(), 0, destination.data(), source.data(), scale), s.handle());
}
inline void addVectors(Solver &s, DeviceGrid &destination, const DeviceGrid &source, const DeviceGrid &scaledSource,
                       double scale)  // Ops.h:1139-1195
{
    mgps::check(mgps_add_vectors(s.handle(), 0, destination.data(), source.data(), scaledSource.data(), scale), s.handle());
}
inline void scaleVector(Solver &s, DeviceGrid &vector, double scale)  // Ops.h:974-1018
{
    mgps::check(mgps_scale_vector(s.handle(), 0, vector.data(), scale), s.handle());
}

}  // namespace GeometricMultigridOperators

// solveGeometricConjugateGradient (CG.h:18-27) with the functors the plugin binds (Plug.cpp:430-483):
// A = applyPoissonMatrix, M^-1 = applyVCycle of `s` (useMGPreconditioner) or the diagonal.
inline mgps_pcg_stats solveGeometricConjugateGradient(GeometricMultigridPoissonSolver &s, mgps::DeviceGrid &solution,
                                                      const mgps::DeviceGrid &rhs, double tolerance, int maxIterations,
                                                      bool useMGPreconditioner = true)
{
    mgps_pcg_stats st{};
    mgps::check(mgps_solve_pcg(s.handle(), solution.data(), rhs.data(), tolerance, maxIterations, useMGPreconditioner, &st),
                s.handle());
    return st;
}
inline mgps_pcg_stats solveGeometricConjugateGradient(GeometricMultigridPoissonSolver &s, std::vector<float> &solution,
                                                      const std::vector<float> &rhs, double tolerance, int maxIterations,
                                                      bool useMGPreconditioner = true)
{
    mgps_pcg_stats st{};
    mgps::check(mgps_solve_pcg_host(s.handle(), solution.data(), rhs.data(), tolerance, maxIterations, useMGPreconditioner, &st),
                s.handle());
    return st;
}

}  // namespace HDK

namespace mgps {

inline DeviceGrid::DeviceGrid(HDK::GeometricMultigridPoissonSolver &s, int level) : s_(s), level_(level)
{
    check(mgps_grid_alloc(s.handle(), level, &p_), s.handle());
}
inline DeviceGrid::~DeviceGrid() { mgps_grid_free(s_.handle(), p_); }
inline void DeviceGrid::upload(const std::vector<float> &host) { check(mgps_grid_upload(s_.handle(), level_, p_, host.data()), s_.handle()); }
inline void DeviceGrid::download(std::vector<float> &host) const
{
    check(mgps_grid_download(s_.handle(), level_, host.data(), p_), s_.handle());
}

}  // namespace mgps
