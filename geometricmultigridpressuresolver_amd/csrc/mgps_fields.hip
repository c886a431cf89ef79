// Plugin-side field pre/post-processing on the device (include/mgps_fields.h).
// Every pass is one thread per cell or face over a dense x-fastest grid: consecutive lanes read consecutive
// addresses of every input, HBM-bound streaming kernels with a handful of bytes per cell.  No reference
// counterpart of the layout: the reference walks 16^3 tiles of UT_VoxelArray on the host (Plug.cpp:716-1207).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mgps_fields.h"
#include "mgps_internal.h"

using namespace mgps;

namespace {

enum : int { kSolid = 0, kLiquid = 1, kAir = 2 };  // Util.h:17

struct Box {
    int gx, gy, gz;
    __host__ __device__ size_t cells() const { return size_t(gx) * gy * gz; }
};

__device__ __forceinline__ size_t cellAt(const Box &g, int i, int j, int k) { return (size_t(k) * g.gy + j) * g.gx + i; }
__device__ __forceinline__ size_t faceAt(const Box &g, int axis, int i, int j, int k)
{
    return (size_t(k) * (g.gy + (axis == 1)) + j) * (g.gx + (axis == 0)) + i;
}
// cellToFaceMap(cell, axis, dir)
__device__ __forceinline__ size_t cellFace(const Box &g, int axis, int dir, int i, int j, int k)
{
    return faceAt(g, axis, i + (axis == 0 && dir), j + (axis == 1 && dir), k + (axis == 2 && dir));
}
__device__ __forceinline__ float ghostFluidTheta(float phi0, float phi1)  // Util.h:25-42 + the clamp of Plug.cpp:850-851
{
    float theta = 0.f;
    if (phi0 < 0.f) theta = phi1 < 0.f ? 1.f : phi0 / (phi0 - phi1);
    else if (phi1 < 0.f) theta = phi1 / (phi1 - phi0);
    return fminf(fmaxf(theta, 0.01f), 1.f);
}
// thread -> (i, j, k) of a grid of extents (nx, ny, nz); false past the end
__device__ __forceinline__ bool unflatten(int nx, int ny, int nz, int &i, int &j, int &k)
{
    const size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= size_t(nx) * ny * nz) return false;
    i = int(t % nx);
    j = int((t / nx) % ny);
    k = int(t / (size_t(nx) * ny));
    return true;
}

__global__ void materialLabelsKernel(Box g, int32_t *__restrict__ material, const float *__restrict__ phi,
                                     const float *__restrict__ solidPhi, const float *__restrict__ cwx,
                                     const float *__restrict__ cwy, const float *__restrict__ cwz)
{
    int i, j, k;
    if (!unflatten(g.gx, g.gy, g.gz, i, j, k)) return;
    const float *cw[3] = {cwx, cwy, cwz};
    const int ext[3] = {g.gx, g.gy, g.gz};
    const size_t c = cellAt(g, i, j, k);
    bool open[3][2], inFluid = false;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            open[a][d] = cw[a][cellFace(g, a, d, i, j, k)] > 0.f;
            inFluid = inFluid || open[a][d];
        }
    int label = kSolid;
    if (inFluid) {
        bool liquid = phi[c] <= 0.f;
        if (!liquid && solidPhi[c] >= 0.f) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    int n[3] = {i, j, k};
                    n[a] += d ? 1 : -1;
                    if (open[a][d] && n[a] >= 0 && n[a] < ext[a] && phi[cellAt(g, n[0], n[1], n[2])] <= 0.f) liquid = true;
                }
        }
        label = liquid ? kLiquid : kAir;
    }
    material[c] = label;
}

__global__ void validFacesKernel(Box g, int axis, uint8_t *__restrict__ valid, const int32_t *__restrict__ material,
                                 const float *__restrict__ cw)
{
    int i, j, k;
    if (!unflatten(g.gx + (axis == 0), g.gy + (axis == 1), g.gz + (axis == 2), i, j, k)) return;
    const size_t f = faceAt(g, axis, i, j, k);
    const int ext[3] = {g.gx, g.gy, g.gz};
    int b[3] = {i, j, k}, fw[3] = {i, j, k};
    b[axis] -= 1;
    uint8_t v = 0;
    if (cw[f] > 0.f && b[axis] >= 0 && fw[axis] < ext[axis])
        v = material[cellAt(g, b[0], b[1], b[2])] == kLiquid || material[cellAt(g, fw[0], fw[1], fw[2])] == kLiquid;
    valid[f] = v;
}

__global__ void domainLabelsKernel(Box g, Box e, int offset, uint8_t *__restrict__ expanded, const int32_t *__restrict__ material)
{
    int bi, bj, bk;  // (the base box only: the launcher has filled the expanded grid with EXTERIOR)
    if (!unflatten(g.gx, g.gy, g.gz, bi, bj, bk)) return;
    const int m = material[cellAt(g, bi, bj, bk)];
    expanded[cellAt(e, bi + offset, bj + offset, bk + offset)] =
        m == kLiquid ? MGPS_INTERIOR_CELL : m == kAir ? MGPS_DIRICHLET_CELL : MGPS_EXTERIOR_CELL;
}

__global__ void boundaryWeightsKernel(Box g, Box e, int offset, int axis, float *__restrict__ expanded, const float *__restrict__ cw,
                                      const float *__restrict__ phi, const uint8_t *__restrict__ valid,
                                      const int32_t *__restrict__ material)
{
    int bi, bj, bk;  // (the faces of the base box only: the launcher has zeroed the expanded face grid)
    if (!unflatten(g.gx + (axis == 0), g.gy + (axis == 1), g.gz + (axis == 2), bi, bj, bk)) return;
    const int i = bi + offset, j = bj + offset, k = bk + offset;
    float w = 0.f;
    {
        const size_t f = faceAt(g, axis, bi, bj, bk);
        if (valid[f]) {  // a valid face has both cells inside the grid
            int b[3] = {bi, bj, bk};
            b[axis] -= 1;
            const size_t cb = cellAt(g, b[0], b[1], b[2]), cf = cellAt(g, bi, bj, bk);
            const int mb = material[cb], mf = material[cf];
            w = cw[f];
            if ((mb == kLiquid && mf == kAir) || (mb == kAir && mf == kLiquid)) w /= ghostFluidTheta(phi[cb], phi[cf]);
        }
    }
    expanded[faceAt(e, axis, i, j, k)] = w;
}

__global__ void setBoundaryLabelsKernel(Box e, uint8_t *__restrict__ lab, const float *__restrict__ wx,
                                        const float *__restrict__ wy, const float *__restrict__ wz)
{
    int i, j, k;
    if (!unflatten(e.gx, e.gy, e.gz, i, j, k)) return;
    const size_t c = cellAt(e, i, j, k);
    // INTERIOR cells have all six neighbours inside the grid (the EXTERIOR shell); BOUNDARY written by another
    // thread reads as "not DIRICHLET / EXTERIOR" just like INTERIOR, so the in-place update is race-free
    if (lab[c] != MGPS_INTERIOR_CELL) return;
    const float *w[3] = {wx, wy, wz};
    const ptrdiff_t stride[3] = {1, e.gx, ptrdiff_t(e.gx) * e.gy};
    bool bnd = false;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const uint8_t nl = lab[ptrdiff_t(c) + (d ? stride[a] : -stride[a])];
            bnd = bnd || nl == MGPS_DIRICHLET_CELL || nl == MGPS_EXTERIOR_CELL || w[a][cellFace(e, a, d, i, j, k)] != 1.f;
        }
    if (bnd) lab[c] = MGPS_BOUNDARY_CELL;
}

// weighted divergence of a LIQUID cell; signBackward = +1 gives the right-hand side (Plug.cpp:912), -1 the
// divergence report (Plug.cpp:1180)
__device__ __forceinline__ float cellDivergence(const Box &g, int i, int j, int k, float signBackward, const float *const v[3],
                                                const float *const sv[3], const float *const cw[3])
{
    float div = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const size_t f = cellFace(g, a, d, i, j, k);
            const float sign = d == 0 ? signBackward : -signBackward, w = cw[a][f];
            if (w > 0.f) div += sign * w * v[a][f];
            if (sv[a] && w < 1.f) div += sign * (1.f - w) * sv[a][f];
        }
    return div;
}

__global__ void rhsKernel(Box g, Box e, int offset, float *__restrict__ rhs, const int32_t *__restrict__ material, const float *vx,
                          const float *vy, const float *vz, const float *svx, const float *svy, const float *svz, const float *cwx,
                          const float *cwy, const float *cwz)
{
    int bi, bj, bk;  // (the base box only: the launcher has zeroed the expanded grid)
    if (!unflatten(g.gx, g.gy, g.gz, bi, bj, bk)) return;
    if (material[cellAt(g, bi, bj, bk)] != kLiquid) return;
    const float *v[3] = {vx, vy, vz}, *sv[3] = {svx, svy, svz}, *cw[3] = {cwx, cwy, cwz};
    rhs[cellAt(e, bi + offset, bj + offset, bk + offset)] = cellDivergence(g, bi, bj, bk, 1.f, v, sv, cw);
}

__global__ void pressureToSolutionKernel(Box g, Box e, int offset, float *__restrict__ x, const float *__restrict__ pressure,
                                         const int32_t *__restrict__ material)
{
    int bi, bj, bk;  // (the base box only: the launcher has zeroed the expanded grid)
    if (!unflatten(g.gx, g.gy, g.gz, bi, bj, bk)) return;
    const size_t c = cellAt(g, bi, bj, bk);
    if (material[c] == kLiquid) x[cellAt(e, bi + offset, bj + offset, bk + offset)] = pressure[c];
}

__global__ void solutionToPressureKernel(Box g, Box e, int offset, float *__restrict__ pressure, const float *__restrict__ x,
                                         const int32_t *__restrict__ material)
{
    int i, j, k;
    if (!unflatten(g.gx, g.gy, g.gz, i, j, k)) return;
    const size_t c = cellAt(g, i, j, k);
    if (material[c] == kLiquid) pressure[c] = x[cellAt(e, i + offset, j + offset, k + offset)];
}

__global__ void pressureGradientKernel(Box g, int axis, float *__restrict__ velocity, const float *__restrict__ phi,
                                       const float *__restrict__ pressure, const uint8_t *__restrict__ valid,
                                       const int32_t *__restrict__ material)
{
    int i, j, k;
    if (!unflatten(g.gx + (axis == 0), g.gy + (axis == 1), g.gz + (axis == 2), i, j, k)) return;
    const size_t f = faceAt(g, axis, i, j, k);
    if (!valid[f]) return;  // valid faces have both cells inside the grid (Plug.cpp:1086-1087 never skips one)
    int b[3] = {i, j, k};
    b[axis] -= 1;
    const size_t cb = cellAt(g, b[0], b[1], b[2]), cf = cellAt(g, i, j, k);
    float grad = pressure[cf] - pressure[cb];
    if (material[cb] != kLiquid || material[cf] != kLiquid) grad /= ghostFluidTheta(phi[cb], phi[cf]);
    velocity[f] -= grad;
}

constexpr int kDivBlocks = 1024;
__global__ __launch_bounds__(256) void divergenceKernel(Box g, double *__restrict__ partials, const int32_t *__restrict__ material,
                                                        const float *vx, const float *vy, const float *vz, const float *svx,
                                                        const float *svy, const float *svz, const float *cwx, const float *cwy,
                                                        const float *cwz)
{
    const float *v[3] = {vx, vy, vz}, *sv[3] = {svx, svy, svz}, *cw[3] = {cwx, cwy, cwz};
    double sum = 0.0, mx = 0.0, count = 0.0;
    const size_t n = g.cells();
    for (size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n; t += size_t(gridDim.x) * blockDim.x) {
        if (material[t] != kLiquid) continue;
        const int i = int(t % g.gx), j = int((t / g.gx) % g.gy), k = int(t / (size_t(g.gx) * g.gy));
        const double d = double(cellDivergence(g, i, j, k, -1.f, v, sv, cw));
        sum += d;
        mx = d > mx ? d : mx;
        count += 1.0;
    }
    __shared__ double s[3][256];
    s[0][threadIdx.x] = sum;
    s[1][threadIdx.x] = mx;
    s[2][threadIdx.x] = count;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (int(threadIdx.x) < off) {
            s[0][threadIdx.x] += s[0][threadIdx.x + off];
            s[1][threadIdx.x] = s[1][threadIdx.x] > s[1][threadIdx.x + off] ? s[1][threadIdx.x] : s[1][threadIdx.x + off];
            s[2][threadIdx.x] += s[2][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = s[0][0];
        partials[kDivBlocks + blockIdx.x] = s[1][0];
        partials[2 * kDivBlocks + blockIdx.x] = s[2][0];
    }
}

int bad(const char *what)
{
    setLastGlobalError(std::string(what) + ": NULL pointer, axis outside 0..2 or non-positive extent");
    return MGPS_ERR_INVALID_ARGUMENT;
}
int done(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MGPS_OK;
    setLastGlobalError(std::string(what) + ": " + hipGetErrorString(e));
    return MGPS_ERR_HIP;
}
inline unsigned blocks(size_t n) { return unsigned((n + 255) / 256); }
inline bool okBox(int x, int y, int z) { return x > 0 && y > 0 && z > 0; }
inline bool okExpanded(int gx, int gy, int gz, int ex, int ey, int ez, int off)
{
    return okBox(ex, ey, ez) && off >= 0 && gx + off <= ex && gy + off <= ey && gz + off <= ez;
}

}  // namespace

extern "C" {

int mgps_fields_material_labels(int32_t *material, const float *liquid_phi, const float *solid_phi, const float *cwx,
                                const float *cwy, const float *cwz, int gx, int gy, int gz, void *stream)
try {
    if (!material || !liquid_phi || !solid_phi || !cwx || !cwy || !cwz || !okBox(gx, gy, gz)) return bad("mgps_fields_material_labels");
    const Box g{gx, gy, gz};
    materialLabelsKernel<<<blocks(g.cells()), 256, 0, static_cast<hipStream_t>(stream)>>>(g, material, liquid_phi, solid_phi, cwx, cwy, cwz);
    return done("mgps_fields_material_labels");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_valid_faces(int axis, uint8_t *valid, const int32_t *material, const float *cut_weights, int gx, int gy,
                            int gz, void *stream)
try {
    if (axis < 0 || axis > 2 || !valid || !material || !cut_weights || !okBox(gx, gy, gz)) return bad("mgps_fields_valid_faces");
    const Box g{gx, gy, gz};
    const size_t n = size_t(gx + (axis == 0)) * (gy + (axis == 1)) * (gz + (axis == 2));
    validFacesKernel<<<blocks(n), 256, 0, static_cast<hipStream_t>(stream)>>>(g, axis, valid, material, cut_weights);
    return done("mgps_fields_valid_faces");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_domain_labels(uint8_t *expanded_labels, const int32_t *material, int gx, int gy, int gz, int ex, int ey,
                              int ez, int offset, void *stream)
try {
    if (!expanded_labels || !material || !okBox(gx, gy, gz) || !okExpanded(gx, gy, gz, ex, ey, ez, offset))
        return bad("mgps_fields_domain_labels");
    const Box g{gx, gy, gz}, e{ex, ey, ez};
    // fill + a kernel over the base box: the reference's power-of-two expansion makes the solver grid up to ten times the
    // simulation grid (480^3 -> 1024^3), and one thread per expanded cell spent 2.5-4 ms per pass there
    if (hipMemsetAsync(expanded_labels, MGPS_EXTERIOR_CELL, e.cells(), static_cast<hipStream_t>(stream)) != hipSuccess) return bad("mgps_fields_domain_labels");
    domainLabelsKernel<<<blocks(g.cells()), 256, 0, static_cast<hipStream_t>(stream)>>>(g, e, offset, expanded_labels, material);
    return done("mgps_fields_domain_labels");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_boundary_weights(int axis, float *expanded_weights, const float *cut_weights, const float *liquid_phi,
                                 const uint8_t *valid, const int32_t *material, int gx, int gy, int gz, int ex, int ey,
                                 int ez, int offset, void *stream)
try {
    if (axis < 0 || axis > 2 || !expanded_weights || !cut_weights || !liquid_phi || !valid || !material || !okBox(gx, gy, gz) ||
        !okExpanded(gx, gy, gz, ex, ey, ez, offset))
        return bad("mgps_fields_boundary_weights");
    const Box g{gx, gy, gz}, e{ex, ey, ez};
    const size_t n = size_t(ex + (axis == 0)) * (ey + (axis == 1)) * (ez + (axis == 2));
    const size_t nb = size_t(gx + (axis == 0)) * (gy + (axis == 1)) * (gz + (axis == 2));
    if (hipMemsetAsync(expanded_weights, 0, n * sizeof(float), static_cast<hipStream_t>(stream)) != hipSuccess) return bad("mgps_fields_boundary_weights");
    boundaryWeightsKernel<<<blocks(nb), 256, 0, static_cast<hipStream_t>(stream)>>>(g, e, offset, axis, expanded_weights, cut_weights,
                                                                                    liquid_phi, valid, material);
    return done("mgps_fields_boundary_weights");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_set_boundary_labels(uint8_t *expanded_labels, const float *wx, const float *wy, const float *wz, int ex,
                                    int ey, int ez, void *stream)
try {
    if (!expanded_labels || !wx || !wy || !wz || !okBox(ex, ey, ez)) return bad("mgps_fields_set_boundary_labels");
    const Box e{ex, ey, ez};
    setBoundaryLabelsKernel<<<blocks(e.cells()), 256, 0, static_cast<hipStream_t>(stream)>>>(e, expanded_labels, wx, wy, wz);
    return done("mgps_fields_set_boundary_labels");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_rhs(float *expanded_rhs, const int32_t *material, const float *vx, const float *vy, const float *vz,
                    const float *svx, const float *svy, const float *svz, const float *cwx, const float *cwy,
                    const float *cwz, int gx, int gy, int gz, int ex, int ey, int ez, int offset, void *stream)
try {
    if (!expanded_rhs || !material || !vx || !vy || !vz || !cwx || !cwy || !cwz || !okBox(gx, gy, gz) ||
        !okExpanded(gx, gy, gz, ex, ey, ez, offset) || ((svx || svy || svz) && !(svx && svy && svz)))
        return bad("mgps_fields_rhs");
    const Box g{gx, gy, gz}, e{ex, ey, ez};
    if (hipMemsetAsync(expanded_rhs, 0, e.cells() * sizeof(float), static_cast<hipStream_t>(stream)) != hipSuccess) return bad("mgps_fields_rhs");
    rhsKernel<<<blocks(g.cells()), 256, 0, static_cast<hipStream_t>(stream)>>>(g, e, offset, expanded_rhs, material, vx, vy, vz, svx, svy,
                                                                             svz, cwx, cwy, cwz);
    return done("mgps_fields_rhs");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_pressure_to_solution(float *expanded_x, const float *pressure, const int32_t *material, int gx, int gy,
                                     int gz, int ex, int ey, int ez, int offset, void *stream)
try {
    if (!expanded_x || !pressure || !material || !okBox(gx, gy, gz) || !okExpanded(gx, gy, gz, ex, ey, ez, offset))
        return bad("mgps_fields_pressure_to_solution");
    const Box g{gx, gy, gz}, e{ex, ey, ez};
    if (hipMemsetAsync(expanded_x, 0, e.cells() * sizeof(float), static_cast<hipStream_t>(stream)) != hipSuccess) return bad("mgps_fields_pressure_to_solution");
    pressureToSolutionKernel<<<blocks(g.cells()), 256, 0, static_cast<hipStream_t>(stream)>>>(g, e, offset, expanded_x, pressure, material);
    return done("mgps_fields_pressure_to_solution");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_solution_to_pressure(float *pressure, const float *expanded_x, const int32_t *material, int gx, int gy,
                                     int gz, int ex, int ey, int ez, int offset, void *stream)
try {
    if (!pressure || !expanded_x || !material || !okBox(gx, gy, gz) || !okExpanded(gx, gy, gz, ex, ey, ez, offset))
        return bad("mgps_fields_solution_to_pressure");
    const Box g{gx, gy, gz}, e{ex, ey, ez};
    solutionToPressureKernel<<<blocks(g.cells()), 256, 0, static_cast<hipStream_t>(stream)>>>(g, e, offset, pressure, expanded_x, material);
    return done("mgps_fields_solution_to_pressure");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_pressure_gradient(int axis, float *velocity, const float *liquid_phi, const float *pressure,
                                  const uint8_t *valid, const int32_t *material, int gx, int gy, int gz, void *stream)
try {
    if (axis < 0 || axis > 2 || !velocity || !liquid_phi || !pressure || !valid || !material || !okBox(gx, gy, gz))
        return bad("mgps_fields_pressure_gradient");
    const Box g{gx, gy, gz};
    const size_t n = size_t(gx + (axis == 0)) * (gy + (axis == 1)) * (gz + (axis == 2));
    pressureGradientKernel<<<blocks(n), 256, 0, static_cast<hipStream_t>(stream)>>>(g, axis, velocity, liquid_phi, pressure, valid, material);
    return done("mgps_fields_pressure_gradient");
}
MGPS_API_CATCH(nullptr)

int mgps_fields_divergence(double out_host[3], const int32_t *material, const float *vx, const float *vy, const float *vz,
                           const float *svx, const float *svy, const float *svz, const float *cwx, const float *cwy,
                           const float *cwz, int gx, int gy, int gz, void *stream)
try {
    if (!out_host || !material || !vx || !vy || !vz || !cwx || !cwy || !cwz || !okBox(gx, gy, gz) ||
        ((svx || svy || svz) && !(svx && svy && svz)))
        return bad("mgps_fields_divergence");
    const Box g{gx, gy, gz};
    hipStream_t s = static_cast<hipStream_t>(stream);
    double *partials = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&partials), 3 * kDivBlocks * sizeof(double)) != hipSuccess) {
        setLastGlobalError("mgps_fields_divergence: hipMalloc failed");
        return MGPS_ERR_ALLOC;
    }
    divergenceKernel<<<kDivBlocks, 256, 0, s>>>(g, partials, material, vx, vy, vz, svx, svy, svz, cwx, cwy, cwz);
    std::vector<double> host(3 * kDivBlocks);
    hipError_t e = hipMemcpyAsync(host.data(), partials, host.size() * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(partials);
    if (e != hipSuccess) {
        setLastGlobalError(std::string("mgps_fields_divergence: ") + hipGetErrorString(e));
        return MGPS_ERR_HIP;
    }
    double sum = 0.0, mx = 0.0, count = 0.0;  // fixed order: reproducible
    for (int b = 0; b < kDivBlocks; ++b) {
        sum += host[size_t(b)];
        mx = host[size_t(kDivBlocks + b)] > mx ? host[size_t(kDivBlocks + b)] : mx;
        count += host[size_t(2 * kDivBlocks + b)];
    }
    out_host[0] = sum;
    out_host[1] = mx;
    out_host[2] = count;
    return MGPS_OK;
}
MGPS_API_CATCH(nullptr)

}  // extern "C"

// ---- one-call projection (mgps_project_free_surface) --------------------------------------------------------------
namespace {
__global__ void narrowRealKernel(float *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    const size_t c = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c < n) dst[c] = float(src[c]);
}
__global__ void widenRealKernel(double *__restrict__ dst, const float *__restrict__ src, size_t n)
{
    const size_t c = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c < n) dst[c] = double(src[c]);
}

// number of LIQUID cells (material label 1, Util.h:17)
__global__ void countLiquidKernel(const int32_t *__restrict__ material, size_t n, unsigned long long *__restrict__ count)
{
    unsigned long long mine = 0;
    for (size_t c = size_t(blockIdx.x) * blockDim.x + threadIdx.x; c < n; c += size_t(gridDim.x) * blockDim.x) mine += material[c] == 1;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

// device buffers of one call; everything is released when the object goes out of scope
struct DevPool {
    std::vector<void *> blocks;
    ~DevPool()
    {
        (void)hipDeviceSynchronize();  // (deviceFree keeps the blocks for the next call and does not wait for queued kernels)
        for (void *b : blocks) (void)mgps::deviceFree(b);
    }
    template <class T>
    T *get(size_t count)
    {
        void *p = nullptr;
        if (mgps::deviceAlloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != 0) throw std::bad_alloc();
        blocks.push_back(p);
        return static_cast<T *>(p);
    }
};
size_t faceCount(int gx, int gy, int gz, int axis) { return size_t(gx + (axis == 0)) * (gy + (axis == 1)) * (gz + (axis == 2)); }
}  // namespace

extern "C" {

int mgps_project_free_surface(mgps_projection *p, const mgps_options *opt)
try {
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    if (!p || p->struct_size != int(sizeof(mgps_projection))) {
        setLastGlobalError("mgps_project_free_surface: NULL or struct_size mismatch");
        return MGPS_ERR_INVALID_ARGUMENT;
    }
    const int gx = p->gx, gy = p->gy, gz = p->gz;
    bool ok = gx > 0 && gy > 0 && gz > 0 && (p->real_bytes == 4 || p->real_bytes == 8) && p->liquid_phi && p->solid_phi && p->pressure;
    for (int a = 0; a < 3; ++a) ok = ok && p->cut_weights[a] && p->velocity[a];
    const bool haveSolidVel = p->solid_velocity[0] && p->solid_velocity[1] && p->solid_velocity[2];
    if (!ok || (!haveSolidVel && (p->solid_velocity[0] || p->solid_velocity[1] || p->solid_velocity[2]))) {
        setLastGlobalError("mgps_project_free_surface: missing field or bad extents (solid velocities: all three or none)");
        return MGPS_ERR_INVALID_ARGUMENT;
    }
    mgps_options o;
    mgps_default_options(&o);
    if (opt) {
        if (opt->struct_size != int(sizeof(mgps_options))) {
            setLastGlobalError("mgps_options.struct_size mismatch: call mgps_default_options first");
            return MGPS_ERR_INVALID_ARGUMENT;
        }
        o = *opt;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        setLastGlobalError("no HIP device is visible (this library has no CPU path)");
        return MGPS_ERR_NO_DEVICE;
    }
    if (o.device >= 0 && hipSetDevice(o.device) != hipSuccess) {
        setLastGlobalError("hipSetDevice failed");
        return MGPS_ERR_NO_DEVICE;
    }
    hipStream_t s = nullptr;
    DevPool pool;
    const size_t cells = size_t(gx) * gy * gz;
    const bool dbl = p->real_bytes == 8;
    double *stage = nullptr;  // device staging for double host arrays
    if (dbl) stage = pool.get<double>(faceCount(gx + 1, gy + 1, gz + 1, 3));
    auto failHip = [&](const char *what, hipError_t e) {
        setLastGlobalError(std::string("mgps_project_free_surface: ") + what + ": " + hipGetErrorString(e));
        return int(MGPS_ERR_HIP);
    };
    hipError_t he = hipSuccess;
    auto upload = [&](const void *host, size_t n) -> float * {  // float copy of a host real array on the device
        float *d = pool.get<float>(n);
        if (he != hipSuccess) return d;
        if (!dbl) he = hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice);
        else {
            he = hipMemcpy(stage, host, n * sizeof(double), hipMemcpyHostToDevice);
            if (he == hipSuccess) {
                narrowRealKernel<<<unsigned((n + 255) / 256), 256, 0, s>>>(d, stage, n);
                he = hipStreamSynchronize(s);
            }
        }
        return d;
    };
    auto download = [&](void *host, const float *d, size_t n) {
        if (he != hipSuccess) return;
        if (!dbl) he = hipMemcpy(host, d, n * sizeof(float), hipMemcpyDeviceToHost);
        else {
            widenRealKernel<<<unsigned((n + 255) / 256), 256, 0, s>>>(stage, d, n);
            he = hipMemcpy(host, stage, n * sizeof(double), hipMemcpyDeviceToHost);
        }
    };
    // What the labels and weights are made from goes up first; the velocities, the solid velocities and the old pressure --
    // needed from the right-hand side on -- follow on a copy stream while the passes below and the solver's set-up run
    // (float arrays in page-locked memory, mgps_host_alloc: from pageable memory the call degrades to a blocking copy).
    struct CopyStream {
        hipStream_t s = nullptr;
        ~CopyStream()
        {
            if (s) {
                (void)hipStreamSynchronize(s);
                (void)hipStreamDestroy(s);
            }
        }
    } copy;
    if (!dbl && hipStreamCreateWithFlags(&copy.s, hipStreamNonBlocking) != hipSuccess) copy.s = nullptr;
    auto uploadLater = [&](const void *host, size_t n) -> float * {
        if (!copy.s) return upload(host, n);
        float *d = pool.get<float>(n);
        if (he == hipSuccess) he = hipMemcpyAsync(d, host, n * sizeof(float), hipMemcpyHostToDevice, copy.s);
        return d;
    };
    float *phi = upload(p->liquid_phi, cells), *solidPhi = upload(p->solid_phi, cells);
    float *cw[3], *vel[3], *svel[3] = {nullptr, nullptr, nullptr};
    for (int a = 0; a < 3; ++a) cw[a] = upload(p->cut_weights[a], faceCount(gx, gy, gz, a));
    float *pressure = uploadLater(p->pressure, cells);
    for (int a = 0; a < 3; ++a) {
        vel[a] = uploadLater(p->velocity[a], faceCount(gx, gy, gz, a));
        if (haveSolidVel) svel[a] = uploadLater(p->solid_velocity[a], faceCount(gx, gy, gz, a));
    }
    if (he != hipSuccess) return failHip("upload", he);
#define PROJ_TRY(call)               \
    do {                             \
        const int rc_ = (call);      \
        if (rc_ != MGPS_OK) return rc_; \
    } while (0)
    // Plug.cpp:270, 286
    int32_t *material = pool.get<int32_t>(cells);
    PROJ_TRY(mgps_fields_material_labels(material, phi, solidPhi, cw[0], cw[1], cw[2], gx, gy, gz, s));
    uint8_t *valid[3];
    for (int a = 0; a < 3; ++a) {
        valid[a] = pool.get<uint8_t>(faceCount(gx, gy, gz, a));
        PROJ_TRY(mgps_fields_valid_faces(a, valid[a], material, cw[a], gx, gy, gz, s));
    }
    // Plug.cpp:316-362: MG labels and weights at +offset of the expanded grid, then the BOUNDARY labels
    int dims[3], offset = 0, levels = 0;
    PROJ_TRY(mgps_expanded_layout(gx, gy, gz, 0, p->power_of_two, dims, &offset, &levels));
    const int ex = dims[0], ey = dims[1], ez = dims[2];
    const size_t ecells = size_t(ex) * ey * ez;
    uint8_t *labels = pool.get<uint8_t>(ecells);
    PROJ_TRY(mgps_fields_domain_labels(labels, material, gx, gy, gz, ex, ey, ez, offset, s));
    float *w[3];
    for (int a = 0; a < 3; ++a) {
        w[a] = pool.get<float>(faceCount(ex, ey, ez, a));
        PROJ_TRY(mgps_fields_boundary_weights(a, w[a], cw[a], phi, valid[a], material, gx, gy, gz, ex, ey, ez, offset, s));
    }
    PROJ_TRY(mgps_fields_set_boundary_labels(labels, w[0], w[1], w[2], ex, ey, ez, s));
    p->mg_levels = levels;
    p->offset = offset;
    p->expanded[0] = ex;
    p->expanded[1] = ey;
    p->expanded[2] = ez;
    // liquid cell count first: a domain without liquid has nothing to solve.  (The reference has no such exit: it would run
    // its solver on an empty system.  What it would publish -- the valid faces it computed and an all-zero pressure,
    // Plug.cpp:286, 641 -- is published here as well; velocities stay as they came.)
    double div[3] = {0, 0, 0};
    {
        unsigned long long *count = pool.get<unsigned long long>(1), liquid = 0;
        he = hipMemsetAsync(count, 0, sizeof(unsigned long long), s);
        if (he == hipSuccess) {
            countLiquidKernel<<<unsigned(std::min<size_t>((cells + 255) / 256, 4096)), 256, 0, s>>>(material, cells, count);
            he = hipMemcpy(&liquid, count, sizeof(liquid), hipMemcpyDeviceToHost);
        }
        if (he != hipSuccess) return failHip("liquid cell count", he);
        div[2] = double(liquid);
    }
    p->liquid_cells = div[2];
    std::memset(&p->stats, 0, sizeof(p->stats));
    p->residual_inf = p->residual_l2 = p->divergence_sum = p->divergence_max = 0;
    if (div[2] == 0) {
        if (copy.s && (he = hipStreamSynchronize(copy.s)) != hipSuccess) return failHip("upload", he);
        if ((he = hipMemsetAsync(pressure, 0, cells * sizeof(float), s)) != hipSuccess) return failHip("pressure clear", he);
        download(p->pressure, pressure, cells);
        for (int a = 0; a < 3; ++a)
            if (p->valid_faces[a] && he == hipSuccess) he = hipMemcpy(p->valid_faces[a], valid[a], faceCount(gx, gy, gz, a), hipMemcpyDeviceToHost);
        if (he != hipSuccess) return failHip("download", he);
        p->setup_ms = p->total_ms = std::chrono::duration<double, std::milli>(clock::now() - t0).count();
        p->solve_ms = 0;
        p->stats.outcome = MGPS_PCG_RHS_ZERO;
        return MGPS_OK;
    }
    mgps_solver *mg = nullptr;  // Plug.cpp:463-466 (before the right-hand side here: the velocities may still be on their way)
    o.borrow_device_weights = 1;  // (the pool outlives the solver: `guard` below is destroyed first)
    int rc = mgps_create_device(&mg, ex, ey, ez, labels, w[0], w[1], w[2], levels, p->use_gauss_seidel, &o);
    if (rc != MGPS_OK) return rc;
    struct Guard {
        mgps_solver *h;
        ~Guard() { mgps_destroy(h); }
    } guard{mg};
    if (copy.s && (he = hipStreamSynchronize(copy.s)) != hipSuccess) return failHip("upload", he);
    // Plug.cpp:386, 413
    float *rhs = pool.get<float>(ecells), *x = nullptr;
    PROJ_TRY(mgps_fields_rhs(rhs, material, vel[0], vel[1], vel[2], svel[0], svel[1], svel[2], cw[0], cw[1], cw[2], gx, gy, gz, ex, ey, ez, offset, s));
    rc = mgps_grid_alloc(mg, 0, &x);  // zero-filled
    if (rc == MGPS_OK && p->use_old_pressure) rc = mgps_fields_pressure_to_solution(x, pressure, material, gx, gy, gz, ex, ey, ez, offset, s);
    if (rc != MGPS_OK) {
        setLastGlobalError(mgps_last_error(mg));
        return rc;
    }
    (void)hipDeviceSynchronize();
    const auto t1 = clock::now();
    rc = mgps_solve_pcg(mg, x, rhs, p->tolerance, p->max_iterations, p->use_mg_preconditioner, &p->stats);  // Plug.cpp:474-483 / 609-618
    if (rc == MGPS_OK) {  // Plug.cpp:625-628
        float *r = nullptr;
        rc = mgps_grid_alloc(mg, 0, &r);
        if (rc == MGPS_OK) rc = mgps_residual(mg, 0, r, x, rhs);
        if (rc == MGPS_OK) rc = mgps_inf_norm(mg, 0, r, 1, &p->residual_inf);
        if (rc == MGPS_OK) rc = mgps_l2_norm(mg, 0, r, &p->residual_l2);
    }
    if (rc != MGPS_OK && rc != MGPS_ERR_INTERRUPTED) {
        setLastGlobalError(mgps_last_error(mg));
        return rc;
    }
    const int solveRc = rc;
    (void)hipDeviceSynchronize();
    const auto t2 = clock::now();
    // Plug.cpp:641-707: the pressure field is cleared before the solution is written into the liquid cells (`makeConstant(0)`,
    // Plug.cpp:641) -- the warm start above has consumed the caller's old values; a cell that was liquid in the last sub-step and
    // is air or solid now must not keep its old pressure (the gradient pass reads it across ghost-fluid faces)
    if ((he = hipMemsetAsync(pressure, 0, cells * sizeof(float), s)) != hipSuccess) return failHip("pressure clear", he);
    PROJ_TRY(mgps_fields_solution_to_pressure(pressure, x, material, gx, gy, gz, ex, ey, ez, offset, s));
    for (int a = 0; a < 3; ++a) PROJ_TRY(mgps_fields_pressure_gradient(a, vel[a], phi, pressure, valid[a], material, gx, gy, gz, s));
    PROJ_TRY(mgps_fields_divergence(div, material, vel[0], vel[1], vel[2], svel[0], svel[1], svel[2], cw[0], cw[1], cw[2], gx, gy, gz, s));
    p->divergence_sum = div[0];
    p->divergence_max = div[1];
#undef PROJ_TRY
    download(p->pressure, pressure, cells);
    for (int a = 0; a < 3; ++a) {
        download(p->velocity[a], vel[a], faceCount(gx, gy, gz, a));
        if (p->valid_faces[a] && he == hipSuccess) he = hipMemcpy(p->valid_faces[a], valid[a], faceCount(gx, gy, gz, a), hipMemcpyDeviceToHost);
    }
    if (he != hipSuccess) return failHip("download", he);
    const auto t3 = clock::now();
    p->setup_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    p->solve_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    p->total_ms = std::chrono::duration<double, std::milli>(t3 - t0).count();
    if (solveRc != MGPS_OK) setLastGlobalError("mgps_project_free_surface: interrupted");
    return solveRc;
}
MGPS_API_CATCH(nullptr)

}  // extern "C"
