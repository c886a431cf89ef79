// Internal declarations shared by the host-side hierarchy builder (mgps_host.cpp), the HIP kernels
// (mgps_kernels.hip) and the solver / C-ABI layer (mgps_solver.hip).  Not part of the public ABI.
#pragma once

#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>

#include "mgps.h"

namespace mgps {

constexpr int kTile = 16;  // UT_VoxelArray tile edge: decides the Gauss-Seidel colouring (Ops.h:436-448)

inline bool isActive(uint8_t l) { return l == MGPS_INTERIOR_CELL || l == MGPS_BOUNDARY_CELL; }

struct Dims {
    int nx = 0, ny = 0, nz = 0;
    size_t cells() const { return size_t(nx) * ny * nz; }
    size_t idx(int i, int j, int k) const { return (size_t(k) * ny + j) * nx + i; }
};

// One level of the hierarchy, host side.
struct HostLevel {
    Dims d;
    std::vector<uint8_t> labels;
    std::vector<int32_t> band;       // linear indices of the band cells, reference order (tile,k,j,i)
    std::vector<int32_t> tilesOdd;   // tiles (linear tile id) holding active cells, (tx+ty+tz) odd
    std::vector<int32_t> tilesEven;
    int64_t activeCells = 0;
};

void setLastGlobalError(const std::string &msg);
const char *lastGlobalError();

// ---- device side ------------------------------------------------------------------------------
// Read-only description of one level handed to the kernels.
struct GridP {
    int nx, ny, nz;
    const uint8_t *lab;
    const float *wx, *wy, *wz;  // fine level only; nullptr = unit weights (MG.cpp:572-575)
};

enum StencilOp { OP_JACOBI = 0, OP_RESIDUAL = 1, OP_APPLY = 2 };

// All launchers enqueue on `stream` (a hipStream_t passed as void*) and return a hipError_t as int.
int launchStencil(void *stream, StencilOp op, const GridP &g, float *out, const float *x, const float *b, float omega);
int launchBandJacobi(void *stream, const GridP &g, float *x, const float *b, const int32_t *band, int nband,
                     float *bandTmp, float omega);
int launchTiledGS(void *stream, const GridP &g, float *x, const float *b, const int32_t *tiles, int ntiles,
                  int forward);
int launchRestrict(void *stream, const GridP &coarse, float *coarseOut, const float *fine);
int launchProlongAdd(void *stream, const GridP &fine, float *fineInOut, const float *coarse);
int launchCoarseSolve(void *stream, int n, const float *inverse, const int32_t *cells, float *x, const float *b,
                      float *gathered);
int launchAxpy(void *stream, const GridP &g, float *dst, const float *src, const float *scaleDev, float scaleHost,
               float sign);
int launchXpay(void *stream, const GridP &g, float *dst, const float *a, const float *s, const float *scaleDev,
               float scaleHost);
int launchScale(void *stream, const GridP &g, float *v, float scale);
int launchDiagInverse(void *stream, const GridP &g, float *dinv);
int launchMulMasked(void *stream, const GridP &g, float *dst, const float *a, const float *b);
// reductions: kind 0 = dot(a,b), 1 = sum a^2, 2 = max(0, max a) (reference infNorm), 3 = max |a|.
// `partials` holds kReducePartials doubles; the result lands in *resultDev.
constexpr int kReducePartials = 2048;
int launchReduce(void *stream, int kind, const GridP &g, const float *a, const float *b, double *partials,
                 double *resultDev);

}  // namespace mgps

// Host-only hierarchy (C-ABI opaque type).
struct mgps_hierarchy {
    int levels = 0;
    int bandWidth = 3;
    std::vector<mgps::HostLevel> lv;
    // coarsest-level direct solver
    int coarseN = 0;
    int coarseBW = 0;
    std::vector<int32_t> coarseCell;    // unknown id -> linear cell of the coarsest grid
    std::vector<int32_t> coarseIndex;   // linear cell -> unknown id or -1
    std::vector<double> coarseL;        // banded Cholesky factor, coarseN x (coarseBW+1)
    std::vector<float> coarseInverse;   // dense coarseN x coarseN inverse (built on demand for the GPU)
    void bandedSolve(double *v) const;
    void buildDenseInverse();
};
