// Tuning tool (not part of the product): where does a tiled Gauss-Seidel half sweep spend its time?
// Includes the product's kernel file so that the shipped kernels themselves are timed next to stripped variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igeometricmultigridpressuresolver_amd/csrc tools/gsbench.hip -o tools/gsbench
//   tools/gsbench 512
#include "../geometricmultigridpressuresolver_amd/csrc/mgps_kernels.hip"

#include <cstdio>
#include <functional>
#include <vector>

using namespace mgps;

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

namespace mgps {
namespace {
// load + store only: the memory side of a pure tile
template <bool REMAP>
__global__ __launch_bounds__(256) void gsMemOnly(GridP g, float *__restrict__ x, const float *__restrict__ b, const int32_t *__restrict__ tiles)
{
    __shared__ float sx[kHalo3];
    __shared__ float sb[kTile3];
    const int tilesX = (g.nx + kTile - 1) / kTile, tilesY = (g.ny + kTile - 1) / kTile;
    const int tile = tiles[REMAP ? remapBlock(blockIdx.x, gridDim.x) : blockIdx.x];
    const int i0 = (tile % tilesX) * kTile, j0 = ((tile / tilesX) % tilesY) * kTile, k0 = (tile / (tilesX * tilesY)) * kTile;
    gsLoadTile<false>(g, x, b, i0, j0, k0, sx, sb, nullptr);
    __syncthreads();
    for (int r = threadIdx.x; r < kTile * kTile * 4; r += blockDim.x) {
        const int q = r & 3, cj = (r >> 2) % kTile, ck = (r >> 2) / kTile;
        const float *src = sx + haloIdx(4 * q, cj, ck);
        const float *bq = sb + (ck * kTile + cj) * kTile + 4 * q;
        *reinterpret_cast<float4 *>(x + (size_t(k0 + ck) * g.ny + j0 + cj) * g.nx + i0 + 4 * q) =
            make_float4(src[0] + 1e-9f * bq[0], src[1], src[2], src[3]);
    }
}
// compute only: the 46 steps on whatever LDS holds, one store per thread so that nothing is optimised away
__global__ __launch_bounds__(256) void gsComputeOnly(GridP g, float *__restrict__ x, const int32_t *__restrict__ tiles, int forward)
{
    __shared__ float sx[kHalo3];
    __shared__ float sb[kTile3];
    for (int h = threadIdx.x; h < kHalo3; h += blockDim.x) sx[h] = float(h & 7);
    for (int h = threadIdx.x; h < kTile3; h += blockDim.x) sb[h] = float(h & 3);
    __syncthreads();
    const int li = threadIdx.x % kTile, lj = threadIdx.x / kTile;
    for (int step = 0; step < kPlanes; ++step) {
        const int s = forward ? step : kPlanes - 1 - step;
        const int lk = s - li - lj;
        if (lk >= 0 && lk < kTile) {
            const int h = haloIdx(li, lj, lk);
            const float xc = sx[h];
            const float lap = 6.f * xc - (sx[h - 1] + sx[h + 1] + sx[h - kHalo] + sx[h + kHalo] + sx[h - kHalo * kHalo] + sx[h + kHalo * kHalo]);
            sx[h] = xc + (sb[(lk * kTile + lj) * kTile + li] - lap) * (1.f / 6.f);
        }
        __syncthreads();
    }
    x[size_t(tiles[blockIdx.x]) * 256 + threadIdx.x] = sx[haloIdx(li, lj, 3)];
}
// the shipped pure kernel with the XCD-aware block remap
__global__ __launch_bounds__(256) void gsPureRemap(GridP g, float *__restrict__ x, const float *__restrict__ b, const int32_t *__restrict__ tiles, int forward)
{
    __shared__ float sx[kHalo3];
    __shared__ float sb[kTile3];
    const int tilesX = (g.nx + kTile - 1) / kTile, tilesY = (g.ny + kTile - 1) / kTile;
    const int tile = tiles[remapBlock(blockIdx.x, gridDim.x)];
    const int i0 = (tile % tilesX) * kTile, j0 = ((tile / tilesX) % tilesY) * kTile, k0 = (tile / (tilesX * tilesY)) * kTile;
    gsLoadTile<false>(g, x, b, i0, j0, k0, sx, sb, nullptr);
    __syncthreads();
    const int li = threadIdx.x % kTile, lj = threadIdx.x / kTile;
    for (int step = 0; step < kPlanes; ++step) {
        const int s = forward ? step : kPlanes - 1 - step;
        const int lk = s - li - lj;
        if (lk >= 0 && lk < kTile) {
            const int h = haloIdx(li, lj, lk);
            const float xc = sx[h];
            const float lap = 6.f * xc - (sx[h - 1] + sx[h + 1] + sx[h - kHalo] + sx[h + kHalo] + sx[h - kHalo * kHalo] + sx[h + kHalo * kHalo]);
            sx[h] = xc + (sb[(lk * kTile + lj) * kTile + li] - lap) * (1.f / 6.f);
        }
        __syncthreads();
    }
    for (int r = threadIdx.x; r < kTile * kTile * 4; r += blockDim.x) {
        const int q = r & 3, cj = (r >> 2) % kTile, ck = (r >> 2) / kTile;
        const float *src = sx + haloIdx(4 * q, cj, ck);
        *reinterpret_cast<float4 *>(x + (size_t(k0 + ck) * g.ny + j0 + cj) * g.nx + i0 + 4 * q) = make_float4(src[0], src[1], src[2], src[3]);
    }
}
// memory-side breakdown: which streams of a tile cost what.  PARTS bit 0: the two x-halo cells per row, bit 1: the y/z halo rows,
// bit 2: rhs, bit 3: store of the tile
template <int PARTS, bool REMAP>
__global__ __launch_bounds__(256) void gsParts(GridP g, float *__restrict__ x, const float *__restrict__ b, const int32_t *__restrict__ tiles)
{
    __shared__ float sx[kHalo3];
    __shared__ __attribute__((aligned(16))) float sb[kTile3];
    const int tilesX = (g.nx + kTile - 1) / kTile, tilesY = (g.ny + kTile - 1) / kTile;
    const int tile = tiles[REMAP ? remapBlock(blockIdx.x, gridDim.x) : blockIdx.x];
    const int i0 = (tile % tilesX) * kTile, j0 = ((tile / tilesX) % tilesY) * kTile, k0 = (tile / (tilesX * tilesY)) * kTile;
    const int tid = threadIdx.x;
    float4 xv[6];
    float hv[3] = {0.f, 0.f, 0.f};
    float4 bv[4];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int r = tid + m * 256;
        const int q = r & 3, row = r >> 2, lj = row % kHalo, lk = row / kHalo;
        const int gj = j0 + lj - 1, gk = k0 + lk - 1;
        const bool interior = lj >= 1 && lj <= kTile && lk >= 1 && lk <= kTile;
        const bool in = r < kHalo * kHalo * 4 && (interior || (PARTS & 2));
        xv[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) xv[m] = *reinterpret_cast<const float4 *>(x + (ptrdiff_t(gk) * g.ny + gj) * g.nx + i0 + 4 * q);
    }
    if (PARTS & 1) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int r = tid + m * 256;
            const int side = r & 1, row = r >> 1, lj = row % kHalo, lk = row / kHalo;
            const int gi = side ? i0 + kTile : i0 - 1, gj = j0 + lj - 1, gk = k0 + lk - 1;
            const bool interior = lj >= 1 && lj <= kTile && lk >= 1 && lk <= kTile;
            if (r < kHalo * kHalo * 2 && interior) hv[m] = x[(ptrdiff_t(gk) * g.ny + gj) * g.nx + gi];
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int r = tid + m * 256;
        const int q = r & 3, lj = (r >> 2) % kTile, lk = (r >> 2) / kTile;
        bv[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((PARTS & 4) && (PARTS & 16)) bv[m] = streamLoad4(b + (size_t(k0 + lk) * g.ny + j0 + lj) * g.nx + i0 + 4 * q);
        else if (PARTS & 4) bv[m] = *reinterpret_cast<const float4 *>(b + (size_t(k0 + lk) * g.ny + j0 + lj) * g.nx + i0 + 4 * q);
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int r = tid + m * 256;
        if (r < kHalo * kHalo * 4) {
            const int q = r & 3, row = r >> 2;
            float *dst = sx + row * kHalo + 1 + 4 * q;
            dst[0] = xv[m].x;
            dst[1] = xv[m].y;
            dst[2] = xv[m].z;
            dst[3] = xv[m].w;
        }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int r = tid + m * 256;
        if (r < kHalo * kHalo * 2) sx[(r >> 1) * kHalo + ((r & 1) ? kHalo - 1 : 0)] = hv[m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) *reinterpret_cast<float4 *>(sb + (tid + m * 256) * 4) = bv[m];
    __syncthreads();
    float acc = 0.f;
    for (int r = threadIdx.x; r < kTile * kTile * 4; r += blockDim.x) {
        const int q = r & 3, cj = (r >> 2) % kTile, ck = (r >> 2) / kTile;
        const float *src = sx + haloIdx(4 * q, cj, ck);
        const float *bq = sb + (ck * kTile + cj) * kTile + 4 * q;
        const float4 v = make_float4(src[0] + 1e-9f * bq[0] + src[-1], src[1] + src[kHalo], src[2] + src[kHalo * kHalo], src[3] + src[4]);
        if ((PARTS & 8) && (PARTS & 16)) __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(x + (size_t(k0 + ck) * g.ny + j0 + cj) * g.nx + i0 + 4 * q));
        else if (PARTS & 8) *reinterpret_cast<float4 *>(x + (size_t(k0 + ck) * g.ny + j0 + cj) * g.nx + i0 + 4 * q) = v;
        else acc += v.x + v.y + v.z + v.w;
    }
    if (!(PARTS & 8) && acc == 123.f) x[0] = acc;
}

// all global loads of a full tile issued before the first LDS write: 13 loads in flight per thread instead of one
__device__ __forceinline__ void gsLoadTileMLP(const GridP &g, const float *__restrict__ x, const float *__restrict__ b, int i0, int j0, int k0,
                                              float *sx, float *sb)
{
    const int tid = threadIdx.x;
    float4 xv[6];
    float hv[3];
    float4 bv[4];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int r = tid + m * 256;
        const int q = r & 3, row = r >> 2, lj = row % kHalo, lk = row / kHalo;
        const int gj = j0 + lj - 1, gk = k0 + lk - 1;
        const bool in = r < kHalo * kHalo * 4 && gj >= 0 && gk >= -g.ghostLo && gj < g.ny && gk < g.nz + g.ghostHi;
        xv[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) xv[m] = *reinterpret_cast<const float4 *>(x + (ptrdiff_t(gk) * g.ny + gj) * g.nx + i0 + 4 * q);
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int r = tid + m * 256;
        const int side = r & 1, row = r >> 1, lj = row % kHalo, lk = row / kHalo;
        const int gi = side ? i0 + kTile : i0 - 1, gj = j0 + lj - 1, gk = k0 + lk - 1;
        const bool in = r < kHalo * kHalo * 2 && gi >= 0 && gi < g.nx && gj >= 0 && gk >= -g.ghostLo && gj < g.ny && gk < g.nz + g.ghostHi;
        hv[m] = 0.f;
        if (in) hv[m] = x[(ptrdiff_t(gk) * g.ny + gj) * g.nx + gi];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int r = tid + m * 256;
        const int q = r & 3, lj = (r >> 2) % kTile, lk = (r >> 2) / kTile;
        bv[m] = *reinterpret_cast<const float4 *>(b + (size_t(k0 + lk) * g.ny + j0 + lj) * g.nx + i0 + 4 * q);
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int r = tid + m * 256;
        if (r < kHalo * kHalo * 4) {
            const int q = r & 3, row = r >> 2;
            float *dst = sx + row * kHalo + 1 + 4 * q;
            dst[0] = xv[m].x;
            dst[1] = xv[m].y;
            dst[2] = xv[m].z;
            dst[3] = xv[m].w;
        }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const int r = tid + m * 256;
        if (r < kHalo * kHalo * 2) sx[(r >> 1) * kHalo + ((r & 1) ? kHalo - 1 : 0)] = hv[m];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) *reinterpret_cast<float4 *>(sb + (tid + m * 256) * 4) = bv[m];
}
__global__ __launch_bounds__(256) void gsMemOnlyMLP(GridP g, float *__restrict__ x, const float *__restrict__ b, const int32_t *__restrict__ tiles)
{
    __shared__ float sx[kHalo3];
    __shared__ __attribute__((aligned(16))) float sb[kTile3];
    const int tilesX = (g.nx + kTile - 1) / kTile, tilesY = (g.ny + kTile - 1) / kTile;
    const int tile = tiles[blockIdx.x];
    const int i0 = (tile % tilesX) * kTile, j0 = ((tile / tilesX) % tilesY) * kTile, k0 = (tile / (tilesX * tilesY)) * kTile;
    gsLoadTileMLP(g, x, b, i0, j0, k0, sx, sb);
    __syncthreads();
    for (int r = threadIdx.x; r < kTile * kTile * 4; r += blockDim.x) {
        const int q = r & 3, cj = (r >> 2) % kTile, ck = (r >> 2) / kTile;
        const float *src = sx + haloIdx(4 * q, cj, ck);
        const float *bq = sb + (ck * kTile + cj) * kTile + 4 * q;
        *reinterpret_cast<float4 *>(x + (size_t(k0 + ck) * g.ny + j0 + cj) * g.nx + i0 + 4 * q) =
            make_float4(src[0] + 1e-9f * bq[0], src[1], src[2], src[3]);
    }
}
__global__ __launch_bounds__(256) void gsPureMLP(GridP g, float *__restrict__ x, const float *__restrict__ b, const int32_t *__restrict__ tiles, int forward)
{
    __shared__ float sx[kHalo3];
    __shared__ __attribute__((aligned(16))) float sb[kTile3];
    const int tilesX = (g.nx + kTile - 1) / kTile, tilesY = (g.ny + kTile - 1) / kTile;
    const int tile = tiles[blockIdx.x];
    const int i0 = (tile % tilesX) * kTile, j0 = ((tile / tilesX) % tilesY) * kTile, k0 = (tile / (tilesX * tilesY)) * kTile;
    gsLoadTileMLP(g, x, b, i0, j0, k0, sx, sb);
    __syncthreads();
    const int li = threadIdx.x % kTile, lj = threadIdx.x / kTile;
    for (int step = 0; step < kPlanes; ++step) {
        const int s = forward ? step : kPlanes - 1 - step;
        const int lk = s - li - lj;
        if (lk >= 0 && lk < kTile) {
            const int h = haloIdx(li, lj, lk);
            const float xc = sx[h];
            const float lap = 6.f * xc - (sx[h - 1] + sx[h + 1] + sx[h - kHalo] + sx[h + kHalo] + sx[h - kHalo * kHalo] + sx[h + kHalo * kHalo]);
            sx[h] = xc + (sb[(lk * kTile + lj) * kTile + li] - lap) * (1.f / 6.f);
        }
        __syncthreads();
    }
    for (int r = threadIdx.x; r < kTile * kTile * 4; r += blockDim.x) {
        const int q = r & 3, cj = (r >> 2) % kTile, ck = (r >> 2) / kTile;
        const float *src = sx + haloIdx(4 * q, cj, ck);
        *reinterpret_cast<float4 *>(x + (size_t(k0 + ck) * g.ny + j0 + cj) * g.nx + i0 + 4 * q) = make_float4(src[0], src[1], src[2], src[3]);
    }
}
}  // namespace
}  // namespace mgps

static float timeIt(const std::function<void()> &f, int reps = 20)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps * 1e3f;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const size_t cells = size_t(n) * n * n;
    float *x, *b;
    uint8_t *lab;
    CK(hipMalloc(&x, cells * 4));
    CK(hipMalloc(&b, cells * 4));
    CK(hipMalloc(&lab, cells));
    CK(hipMemset(x, 0, cells * 4));
    CK(hipMemset(b, 0, cells * 4));
    CK(hipMemset(lab, 0, cells));
    const int nt = n / 16;
    std::vector<int32_t> tl[2];
    for (int tz = 2; tz < nt - 2; ++tz)
        for (int ty = 2; ty < nt - 2; ++ty)
            for (int tx = 2; tx < nt - 2; ++tx) tl[(tx + ty + tz) & 1].push_back((tz * nt + ty) * nt + tx);
    int32_t *td[2];
    for (int c = 0; c < 2; ++c) {
        CK(hipMalloc(&td[c], tl[c].size() * 4));
        CK(hipMemcpy(td[c], tl[c].data(), tl[c].size() * 4, hipMemcpyHostToDevice));
    }
    GridP g{};
    g.nx = g.ny = g.nz = n;
    g.lab = lab;
    const unsigned nb0 = unsigned(tl[0].size()), nb1 = unsigned(tl[1].size());
    const double swept = double(nb0 + nb1) * 4096;
    printf("grid %d^3: %u + %u pure tiles, %.1f M cells per full sweep, 13 B/cell = %.3f GB\n", n, nb0, nb1, swept / 1e6, swept * 13 / 1e9);
    auto report = [&](const char *name, float us) { printf("  %-44s %8.1f us per full sweep  %6.2f TB/s algorithmic (%.1f %% of 8)\n", name, us, swept * 13 / us / 1e6, swept * 13 / us / 1e6 / 8 * 100); };
    report("shipped tiledGSPureKernel (2 launches)", timeIt([&] {
               tiledGSPureKernel<<<nb1, 256>>>(g, x, b, td[1], 1);
               tiledGSPureKernel<<<nb0, 256>>>(g, x, b, td[0], 1);
           }));
    report("  + XCD-aware block remap", timeIt([&] {
               gsPureRemap<<<nb1, 256>>>(g, x, b, td[1], 1);
               gsPureRemap<<<nb0, 256>>>(g, x, b, td[0], 1);
           }));
    report("load + store only", timeIt([&] {
               gsMemOnly<false><<<nb1, 256>>>(g, x, b, td[1]);
               gsMemOnly<false><<<nb0, 256>>>(g, x, b, td[0]);
           }));
    report("load + store only, remap", timeIt([&] {
               gsMemOnly<true><<<nb1, 256>>>(g, x, b, td[1]);
               gsMemOnly<true><<<nb0, 256>>>(g, x, b, td[0]);
           }));
    report("load + store only, all loads in flight", timeIt([&] {
               gsMemOnlyMLP<<<nb1, 256>>>(g, x, b, td[1]);
               gsMemOnlyMLP<<<nb0, 256>>>(g, x, b, td[0]);
           }));
    report("full kernel, all loads in flight", timeIt([&] {
               gsPureMLP<<<nb1, 256>>>(g, x, b, td[1], 1);
               gsPureMLP<<<nb0, 256>>>(g, x, b, td[0], 1);
           }));
#define PARTS_RUN(P, R, name) report(name, timeIt([&] { gsParts<P, R><<<nb1, 256>>>(g, x, b, td[1]); gsParts<P, R><<<nb0, 256>>>(g, x, b, td[0]); }))
    PARTS_RUN(0, false, "parts: x tile rows only (no store)");
    PARTS_RUN(8, false, "parts: x tile rows + store");
    PARTS_RUN(1, false, "parts: x tile rows + x halo cells");
    PARTS_RUN(1, true, "parts: x tile rows + x halo cells, remap");
    PARTS_RUN(2, false, "parts: x tile rows + y/z halo rows");
    PARTS_RUN(4, false, "parts: x tile rows + rhs");
    PARTS_RUN(15, false, "parts: everything");
    PARTS_RUN(15, true, "parts: everything, remap");
    PARTS_RUN(31, false, "parts: everything, nt rhs + nt store");
    PARTS_RUN(31, true, "parts: everything, nt rhs + nt store, remap");
    report("46 steps only (no tile loads)", timeIt([&] {
               gsComputeOnly<<<nb1, 256>>>(g, b, td[1], 1);
               gsComputeOnly<<<nb0, 256>>>(g, b, td[0], 1);
           }));
    return 0;
}
