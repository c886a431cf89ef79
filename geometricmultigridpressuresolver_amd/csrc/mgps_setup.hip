// mgps_setup.hip -- the multigrid hierarchy and every list of a whole-grid level, built on the device.
//
// What the reference's constructor does on the host (HDK_GeometricMultigridPoissonSolver.cpp:141-431 = "MG.cpp":
// buildCoarseCellLabels MG.cpp:238-253 / Ops.cpp:23-163, buildBoundaryCells MG.cpp:279-281 / Ops.cpp:165-469) and what
// this library adds around it (operator rows, cell codes, activity lists, Gauss-Seidel tile lists, the groups of the fused
// band stage) as kernels over device labels: a plugin that rebuilds its solver every sub-step (Plug.cpp:463) then never
// moves a label across PCIe.  Every array equals the host builder's (mgps_host.cpp) entry for entry -- that builder stays
// as the checker (tests/test_device_setup.py) and as the builder of slab runs.
//
// Labels are read as the reference's values (0 INTERIOR, 1 EXTERIOR, 2 DIRICHLET, 3 BOUNDARY); every test also accepts
// the patched device codes (>= 4: simple BOUNDARY cells) as BOUNDARY.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "mgps_internal.h"

namespace mgps {

namespace {

inline unsigned blocksFor(size_t work, unsigned per) { return unsigned((work + per - 1) / per); }
inline hipStream_t S(void *stream) { return static_cast<hipStream_t>(stream); }

__device__ __forceinline__ bool activeCode(unsigned c) { return c == MGPS_INTERIOR_CELL || c >= MGPS_BOUNDARY_CELL; }
__device__ __forceinline__ size_t cellIdx(const Dims &d, int i, int j, int k) { return (size_t(k) * d.ny + j) * d.nx + i; }
__device__ __forceinline__ size_t cellCount(const Dims &d) { return size_t(d.nx) * d.ny * d.nz; }

// ---- labels of the coarser levels (Ops.cpp:23-163) -------------------------------------------------------------------

// a coarse cell is DIRICHLET if any of its 8 children is, else INTERIOR if any child is active, else EXTERIOR
// flags[0] = 1 when the coarse level holds an active cell ("solvable", MG.cpp:241-246)
__global__ __launch_bounds__(256) void coarsenLabelsKernel(Dims fd, Dims cd, const uint8_t *__restrict__ fine, uint8_t *__restrict__ coarse,
                                                           int *__restrict__ flags)
{
    const size_t c = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    bool active = false;
    if (c < cellCount(cd)) {
        const int i = int(c % cd.nx), j = int((c / cd.nx) % cd.ny), k = int(c / (size_t(cd.nx) * cd.ny));
        const uint8_t *r0 = fine + cellIdx(fd, 2 * i, 2 * j, 2 * k);
        const size_t plane = size_t(fd.nx) * fd.ny;
        const unsigned v[4] = {*reinterpret_cast<const uint16_t *>(r0), *reinterpret_cast<const uint16_t *>(r0 + fd.nx),
                               *reinterpret_cast<const uint16_t *>(r0 + plane), *reinterpret_cast<const uint16_t *>(r0 + plane + fd.nx)};
        bool dir = false, act = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned a = v[q] & 0xffu, b = v[q] >> 8;
            dir = dir || a == MGPS_DIRICHLET_CELL || b == MGPS_DIRICHLET_CELL;
            act = act || activeCode(a) || activeCode(b);
        }
        coarse[c] = dir ? uint8_t(MGPS_DIRICHLET_CELL) : (act ? uint8_t(MGPS_INTERIOR_CELL) : uint8_t(MGPS_EXTERIOR_CELL));
        active = !dir && act;
    }
    if (__any(active) && (threadIdx.x & 63) == 0) flags[0] = 1;
}

// flags[0] = 1 when some cell is active (the fine level's "solvable" test, MG.cpp:233)
__global__ __launch_bounds__(256) void anyActiveKernel(const uint32_t *__restrict__ lab4, size_t nq, int *__restrict__ flags)
{
    bool any = false;
    for (size_t q = blockIdx.x * size_t(blockDim.x) + threadIdx.x; q < nq; q += size_t(gridDim.x) * blockDim.x) {
        const uint32_t v = lab4[q];
        any = any || activeCode(v & 0xffu) || activeCode((v >> 8) & 0xffu) || activeCode((v >> 16) & 0xffu) || activeCode(v >> 24);
    }
    if (__any(any) && (threadIdx.x & 63) == 0) flags[0] = 1;
}

// flags[0] = 1 when a cell of the outermost layer is not EXTERIOR (unitTestExteriorCells, MG.cpp:235, 252)
__global__ __launch_bounds__(256) void shellCheckKernel(Dims d, const uint8_t *__restrict__ lab, int *__restrict__ flags)
{
    const size_t fz = size_t(d.nx) * d.ny, fy = size_t(d.nx) * d.nz, fx = size_t(d.ny) * d.nz;
    size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    int i, j, k;
    if (t < 2 * fz) {
        k = t < fz ? 0 : d.nz - 1;
        t %= fz;
        i = int(t % d.nx);
        j = int(t / d.nx);
    } else if (t < 2 * fz + 2 * fy) {
        t -= 2 * fz;
        j = t < fy ? 0 : d.ny - 1;
        t %= fy;
        i = int(t % d.nx);
        k = int(t / d.nx);
    } else if (t < 2 * fz + 2 * fy + 2 * fx) {
        t -= 2 * fz + 2 * fy;
        i = t < fx ? 0 : d.nx - 1;
        t %= fx;
        j = int(t % d.ny);
        k = int(t / d.ny);
    } else
        return;
    if (lab[cellIdx(d, i, j, k)] != MGPS_EXTERIOR_CELL) flags[0] = 1;
}

// coarse levels (unit weights): an INTERIOR cell with an EXTERIOR or DIRICHLET face neighbour becomes BOUNDARY
// (Ops.cpp:112-160).  In place: the marking turns INTERIOR into BOUNDARY, neither of which the test looks for.
__global__ __launch_bounds__(256) void markBoundaryKernel(Dims d, uint8_t *__restrict__ lab)
{
    const size_t c = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (c >= cellCount(d) || lab[c] != MGPS_INTERIOR_CELL) return;
    const int i = int(c % d.nx), j = int((c / d.nx) % d.ny), k = int(c / (size_t(d.nx) * d.ny));
    const ptrdiff_t sy = d.nx, sz = ptrdiff_t(d.nx) * d.ny;
    auto outside = [&](bool inGrid, ptrdiff_t off) {  // past the grid counts as EXTERIOR (a level without shell is refused anyway)
        if (!inGrid) return true;
        const uint8_t l = lab[ptrdiff_t(c) + off];
        return l == MGPS_EXTERIOR_CELL || l == MGPS_DIRICHLET_CELL;
    };
    if (outside(i > 0, -1) || outside(i + 1 < d.nx, 1) || outside(j > 0, -sy) || outside(j + 1 < d.ny, sy) || outside(k > 0, -sz) ||
        outside(k + 1 < d.nz, sz))
        lab[c] = MGPS_BOUNDARY_CELL;
}

// ---- exclusive scan of int32 (n + 1 outputs: out[n] = the total) ------------------------------------------------------
constexpr int kScanThreads = 256, kScanPer = 8, kScanTile = kScanThreads * kScanPer;

__device__ __forceinline__ int waveInclusiveScan(int v)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int u = __shfl_up(v, s);
        if (int(threadIdx.x & 63) >= s) v += u;
    }
    return v;
}
// exclusive prefix of v over the 256 threads of a workgroup; *total = the sum (every thread gets it)
__device__ __forceinline__ int blockExclusiveScan(int v, int *total, int *scratch /* 4 ints of LDS */)
{
    const int inc = waveInclusiveScan(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // scratch may still be read from a previous call
    if ((threadIdx.x & 63) == 63) scratch[w] = inc;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int s = scratch[q];
        before += q < w ? s : 0;
        all += s;
    }
    *total = all;
    return before + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void scanReduceKernel(const int32_t *__restrict__ in, size_t n, int32_t *__restrict__ blockSums)
{
    __shared__ int scratch[4];
    const size_t base = size_t(blockIdx.x) * kScanTile + size_t(threadIdx.x) * kScanPer;
    int s = 0;
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) s += base + q < n ? in[base + q] : 0;
    int total;
    blockExclusiveScan(s, &total, scratch);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}
// one workgroup: exclusive scan of the nb block sums in place, blockSums[nb] = the total
__global__ __launch_bounds__(kScanThreads) void scanSumsKernel(int32_t *__restrict__ blockSums, size_t nb)
{
    __shared__ int scratch[4];
    int carry = 0;
    for (size_t b0 = 0; b0 < nb; b0 += kScanThreads) {
        const size_t q = b0 + threadIdx.x;
        const int v = q < nb ? blockSums[q] : 0;
        int total;
        const int ex = blockExclusiveScan(v, &total, scratch);
        if (q < nb) blockSums[q] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) blockSums[nb] = carry;
}
__global__ __launch_bounds__(kScanThreads) void scanDownKernel(const int32_t *__restrict__ in, int32_t *__restrict__ out, size_t n,
                                                               const int32_t *__restrict__ blockSums, size_t nb)
{
    __shared__ int scratch[4];
    const size_t base = size_t(blockIdx.x) * kScanTile + size_t(threadIdx.x) * kScanPer;
    int v[kScanPer], s = 0;
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) {
        v[q] = base + q < n ? in[base + q] : 0;
        s += v[q];
    }
    int total;
    int run = blockSums[blockIdx.x] + blockExclusiveScan(s, &total, scratch);
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) {
        if (base + q < n) out[base + q] = run;
        run += v[q];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = blockSums[nb];
}

// ---- which tiles need the band kernel ---------------------------------------------------------------------------------

// A streaming pass over the labels: per 16^3 tile its kind for the Gauss-Seidel lists ((active cells << 1) | all 4096
// cells INTERIOR) and three bits -- 1: holds a BOUNDARY cell, 2: holds an inactive cell (or reaches past the grid), 4: holds
// an INTERIOR cell.  One workgroup per row of tiles along x, a thread per tile: the lanes of a wave read consecutive
// 16-byte pieces of a grid row.
__global__ __launch_bounds__(64) void tileStatsKernel(Dims d, const uint8_t *__restrict__ lab, int tx, int ty, int32_t *__restrict__ tileKind,
                                                      uint8_t *__restrict__ tileBits)
{
    const int tj = blockIdx.x % ty, tk = blockIdx.x / ty;
    const bool wide = (d.nx & 15) == 0;
    for (int ti = threadIdx.x; ti < tx; ti += 64) {
        const int i0 = ti * kTile, w = min(kTile, d.nx - i0);
        int act = 0, inter = 0, bnd = 0;
        const int nk = min(kTile, d.nz - tk * kTile), nj = min(kTile, d.ny - tj * kTile);
        for (int lk = 0; lk < nk; ++lk) {
            const uint8_t *plane = lab + cellIdx(d, i0, tj * kTile, tk * kTile + lk);
            if (wide && nj == kTile) {
                uint4 v[kTile];
#pragma unroll
                for (int lj = 0; lj < kTile; ++lj) v[lj] = *reinterpret_cast<const uint4 *>(plane + size_t(lj) * d.nx);
#pragma unroll
                for (int lj = 0; lj < kTile; ++lj) {
                    const uint32_t q[4] = {v[lj].x, v[lj].y, v[lj].z, v[lj].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const unsigned c = (q[e] >> (8 * b)) & 0xffu;
                            inter += c == MGPS_INTERIOR_CELL;
                            bnd += c >= MGPS_BOUNDARY_CELL;
                        }
                }
            } else
                for (int lj = 0; lj < nj; ++lj)
                    for (int li = 0; li < w; ++li) {
                        const unsigned c = plane[size_t(lj) * d.nx + li];
                        inter += c == MGPS_INTERIOR_CELL;
                        bnd += c >= MGPS_BOUNDARY_CELL;
                    }
        }
        act = inter + bnd;
        const int cells = w * nj * nk, t = (tk * ty + tj) * tx + ti;
        tileKind[t] = (act << 1) | (inter == kTile * kTile * kTile ? 1 : 0);
        tileBits[t] = uint8_t((bnd ? 1 : 0) | ((act < cells || cells < kTile * kTile * kTile) ? 2 : 0) | (inter ? 4 : 0));
    }
}
// flags[t] = 1: the tile goes through bandMaskKernel -- a BOUNDARY cell in it or in one of its 26 neighbours can seed band
// cells in it (band_width - 1 <= 7 cells of reach), or it holds an INTERIOR cell that may touch an inactive one (its own or a
// face neighbour's: the INTERIOR rule of unitTestBoundaryCells is checked there).  Every other tile has no band cells.
// kLo / kHi (slab windows): only tiles that hold a plane of [kLo, kHi) are candidates -- the rest of the label buffer is there for
// the coarsening only
__global__ __launch_bounds__(256) void tileCandidateKernel(int tx, int ty, int tz, const uint8_t *__restrict__ bits, int32_t *__restrict__ flags, int kLo, int kHi)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= tx * ty * tz) return;
    const int ti = t % tx, tj = (t / tx) % ty, tk = t / (tx * ty);
    if (tk * kTile >= kHi || (tk + 1) * kTile <= kLo) {
        flags[t] = 0;
        return;
    }
    bool seed = false, inactive = false;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int ni = ti + dx, nj = tj + dy, nk = tk + dz;
                const bool face = (dx != 0) + (dy != 0) + (dz != 0) <= 1;
                if (ni < 0 || nj < 0 || nk < 0 || ni >= tx || nj >= ty || nk >= tz) {
                    inactive = inactive || face;  // past the grid: EXTERIOR
                    continue;
                }
                const unsigned b = bits[(nk * ty + nj) * tx + ni];
                seed = seed || (b & 1u);
                inactive = inactive || (face && (b & 2u));
            }
    flags[t] = (seed || ((bits[t] & 4u) && inactive)) ? 1 : 0;
}

// ---- band list (Ops.cpp:165-469) -------------------------------------------------------------------------------------

// One workgroup per 16^3 tile: the BOUNDARY cells of the tile and its halo of width-1 cells seed width-1 rings grown
// through INTERIOR face neighbours; what lands inside the tile is the tile's share of the band.  Left behind per tile:
// the 4096-bit membership mask (bit (k*16+j)*16+i, 128 words), the exclusive prefix count of every word, the count, and
// the tile's kind for the Gauss-Seidel lists ((active cells << 1) | all 4096 cells INTERIOR).  With the block in LDS the
// label half of unitTestBoundaryCells comes for free (interiorBad != nullptr: every INTERIOR cell of the tile must have
// six active neighbours, Ops.h:1771-1870).
__global__ __launch_bounds__(256) void bandMaskKernel(Dims d, const uint8_t *__restrict__ lab, int width, int tx, int ty, uint32_t *__restrict__ mask,
                                                      uint16_t *__restrict__ prefix, int32_t *__restrict__ tileCount, int32_t *__restrict__ tileKind,
                                                      int *__restrict__ interiorBad, const int32_t *__restrict__ tiles, int chk0, int chk1)
{
    extern __shared__ uint8_t sm[];
    const int halo = max(width - 1, 1), E = kTile + 2 * halo, E2 = E * E, E3 = E2 * E;
    uint8_t *sl = sm, *mk = sm + E3;
    __shared__ uint16_t rowBits[256];
    __shared__ int wordCount[128];
    __shared__ int sSeeds, sActive, sInterior;
    const int tid = threadIdx.x;
    const int t = tiles ? tiles[blockIdx.x] : int(blockIdx.x), ti = t % tx, tj = (t / tx) % ty, tk = t / (tx * ty);
    const int oi = ti * kTile - halo, oj = tj * kTile - halo, ok = tk * kTile - halo;
    if (tid == 0) sSeeds = sActive = sInterior = 0;
    __syncthreads();
    // the block of labels: EXTERIOR everywhere, then the part inside the grid row by row as aligned 4-byte words (eight
    // or sixteen consecutive lanes share a row: whole cache lines per wave instruction instead of 64 rows of one byte)
    for (int q = tid; q < E3 / 4; q += 256) reinterpret_cast<uint32_t *>(sl)[q] = 0x01010101u * MGPS_EXTERIOR_CELL;
    __syncthreads();
    {
        const int dshift = E + 7 <= 32 ? 3 : 4;  // words a row can touch: (E + 3) / 4 + 1 <= 8 up to E = 24, <= 16 beyond
        for (int w = tid; w < (E2 << dshift); w += 256) {
            const int r = w >> dshift, q = w & ((1 << dshift) - 1);
            const int lj = r % E, lk = r / E, gj = oj + lj, gk = ok + lk;
            if (gj < 0 || gj >= d.ny || gk < 0 || gk >= d.nz) continue;
            const ptrdiff_t rowBase = ptrdiff_t(cellIdx(d, 0, gj, gk));
            const ptrdiff_t a0 = rowBase + max(oi, 0), a1 = rowBase + min(oi + E, d.nx);
            const ptrdiff_t addr = (a0 & ~ptrdiff_t(3)) + 4 * q;
            if (addr >= a1) continue;
            const uint32_t word = *reinterpret_cast<const uint32_t *>(lab + addr);  // (the allocation has a spare plane on each side)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const ptrdiff_t at = addr + b;
                if (at >= a0 && at < a1) sl[r * E + int(at - rowBase) - oi] = uint8_t(word >> (8 * b));
            }
        }
    }
    __syncthreads();
    bool seeds = false;
    for (int q = tid; q < E3 / 4; q += 256) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(sl)[q];
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) m |= (((v >> (8 * b)) & 0xffu) >= MGPS_BOUNDARY_CELL ? 1u : 0u) << (8 * b);
        reinterpret_cast<uint32_t *>(mk)[q] = m;
        seeds = seeds || m != 0;
    }
    if (seeds) sSeeds = 1;
    __syncthreads();
    if (sSeeds)
        for (int ring = 1; ring < width; ++ring) {  // a cell marked in this ring carries ring + 1: readers of this ring skip it
            for (int r = tid; r < E2; r += 256) {
                const int lj = r % E, lk = r / E;
                for (int li = 0; li < E; ++li) {
                    const int c = r * E + li;
                    if (sl[c] != MGPS_INTERIOR_CELL || mk[c] != 0) continue;
                    auto hit = [&](bool in, int off) {
                        if (!in) return false;
                        const unsigned m = mk[c + off];
                        return m != 0 && m <= unsigned(ring);
                    };
                    if (hit(li > 0, -1) || hit(li + 1 < E, 1) || hit(lj > 0, -E) || hit(lj + 1 < E, E) || hit(lk > 0, -E2) || hit(lk + 1 < E, E2))
                        mk[c] = uint8_t(ring + 1);
                }
            }
            __syncthreads();
        }
    {  // thread = one x-row of the tile
        const int lk = tid >> 4, lj = tid & 15;
        const int c0 = ((lk + halo) * E + (lj + halo)) * E + halo;
        unsigned bits = 0;
        int act = 0, inter = 0;
        bool bad = false;
        const bool chk = interiorBad && tk * kTile + lk >= chk0 && tk * kTile + lk < chk1;  // (slab windows: the rank's own planes only)
#pragma unroll
        for (int li = 0; li < kTile; ++li) {
            const int c = c0 + li;
            bits |= (mk[c] != 0 ? 1u : 0u) << li;
            const unsigned v = sl[c];
            act += activeCode(v);
            inter += v == MGPS_INTERIOR_CELL;
            if (chk && v == MGPS_INTERIOR_CELL)  // (cells past the grid read EXTERIOR: not active)
                bad = bad || !activeCode(sl[c - 1]) || !activeCode(sl[c + 1]) || !activeCode(sl[c - E]) || !activeCode(sl[c + E]) ||
                      !activeCode(sl[c - E2]) || !activeCode(sl[c + E2]);
        }
        if (bad) *interiorBad = 1;
        rowBits[tid] = uint16_t(bits);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            act += __shfl_down(act, s);
            inter += __shfl_down(inter, s);
        }
        if ((tid & 63) == 0) {
            atomicAdd(&sActive, act);
            atomicAdd(&sInterior, inter);
        }
    }
    __syncthreads();
    uint32_t word = 0;
    if (tid < 128) {
        word = uint32_t(rowBits[2 * tid]) | (uint32_t(rowBits[2 * tid + 1]) << 16);
        wordCount[tid] = __popc(word);
    }
    __syncthreads();
    if (tid < 128) {
        int before = 0;
        for (int q = 0; q < tid; ++q) before += wordCount[q];
        mask[size_t(t) * 128 + tid] = word;
        prefix[size_t(t) * 128 + tid] = uint16_t(before);
        if (tid == 127) {
            tileCount[t] = before + wordCount[127];
            if (!tiles) tileKind[t] = (sActive << 1) | (sInterior == kTile * kTile * kTile ? 1 : 0);  // (with a list: tileStatsKernel's)
        }
    }
}

// the band list in the reference order (tile, k, j, i) from the masks and the scanned tile counts
__global__ __launch_bounds__(256) void bandFillKernel(Dims d, int tx, int ty, const uint32_t *__restrict__ mask, const uint16_t *__restrict__ prefix,
                                                      const int32_t *__restrict__ tileStart, int32_t *__restrict__ band)
{
    const int t = blockIdx.x;
    if (tileStart[t + 1] == tileStart[t]) return;
    const int tid = threadIdx.x, ti = t % tx, tj = (t / tx) % ty, tk = t / (tx * ty);
    const uint32_t word = mask[size_t(t) * 128 + (tid >> 1)];
    unsigned bits = (tid & 1) ? word >> 16 : word & 0xffffu;
    int at = tileStart[t] + prefix[size_t(t) * 128 + (tid >> 1)] + ((tid & 1) ? __popc(word & 0xffffu) : 0);
    const int lk = tid >> 4, lj = tid & 15;
    const size_t row = cellIdx(d, ti * kTile, tj * kTile + lj, tk * kTile + lk);
    while (bits) {
        const int li = __ffs(bits) - 1;
        bits &= bits - 1;
        band[at++] = int32_t(row + li);
    }
}

// ---- band split: general BOUNDARY cells first (their operator rows kept), the rest after ------------------------------

struct RowEval {
    float w[6], diag;
    bool simple, ruleOk;
};
// the host's rowOf (mgps_host.cpp) / boundaryRowsKernel term by term (Ops.h:208-256); wv.w[0] == nullptr: unit weights
// Slab windows (WeightView::gw > 0): the grid is the rank's label buffer, its plane k is plane k - wv.k0 of the weight arrays,
// which hold the owned planes; the gw planes below / above them come from the neighbours (wv.lo / wv.hi); farther out the weights
// read as 1 -- rows there are never used
__device__ __forceinline__ float faceWeight(const WeightView &wv, int a, int k, size_t inPlane, size_t planeSize, int nplanes)
{
    // nplanes: planes the owned array of axis a holds (nz, or nz + 1 for the z faces)
    if (k >= 0 && k < nplanes) return wv.w[a][size_t(k) * planeSize + inPlane];
    if (k < 0 && k >= -wv.gw && wv.lo[a]) return wv.lo[a][size_t(k + wv.gw) * planeSize + inPlane];
    if (k >= nplanes && k < nplanes + wv.gw && wv.hi[a]) return wv.hi[a][size_t(k - nplanes) * planeSize + inPlane];
    return 1.f;
}
__device__ __forceinline__ RowEval evalRow(const Dims &d, const uint8_t *__restrict__ lab, const WeightView &wv, size_t c)
{
    const int i = int(c % d.nx), j = int((c / d.nx) % d.ny), k = int(c / (size_t(d.nx) * d.ny));
    const size_t plane = size_t(d.nx) * d.ny;
    float w[6] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (wv.w[0]) {
        const int ko = k - wv.k0;
        const size_t px = size_t(d.nx + 1) * d.ny, py = size_t(d.nx) * (d.ny + 1);
        const size_t fx = size_t(j) * (d.nx + 1) + i, fy = size_t(j) * d.nx + i, fz = size_t(j) * d.nx + i;
        w[0] = faceWeight(wv, 0, ko, fx, px, wv.nz);
        w[1] = faceWeight(wv, 0, ko, fx + 1, px, wv.nz);
        w[2] = faceWeight(wv, 1, ko, fy, py, wv.nz);
        w[3] = faceWeight(wv, 1, ko, fy + d.nx, py, wv.nz);
        w[4] = faceWeight(wv, 2, ko, fz, plane, wv.nz + 1);
        w[5] = faceWeight(wv, 2, ko + 1, fz, plane, wv.nz + 1);
    }
    const ptrdiff_t off[6] = {-1, 1, -ptrdiff_t(d.nx), ptrdiff_t(d.nx), -ptrdiff_t(plane), ptrdiff_t(plane)};
    RowEval r;
    r.diag = 0.f;
    r.simple = true;
    r.ruleOk = false;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const uint8_t nl = lab[ptrdiff_t(c) + off[q]];
        r.w[q] = 0.f;
        if (nl == MGPS_INTERIOR_CELL) {
            r.w[q] = 1.f;
            r.diag += 1.f;
        } else if (nl >= MGPS_BOUNDARY_CELL) {
            r.w[q] = w[q];
            r.diag += w[q];
            r.simple = r.simple && w[q] == 1.f;
            r.ruleOk = r.ruleOk || w[q] != 1.f;
        } else if (nl == MGPS_DIRICHLET_CELL) {
            r.diag += w[q];
            r.simple = r.simple && w[q] == 1.f;
            r.ruleOk = true;
        } else
            r.ruleOk = true;
    }
    // a liquid cell without an open face (diagonal 0; the reference asserts diagonal > 0, Ops.h:354) is no simple cell: "simple with
    // diagonal 0" would read as "general" in the band diagonals.  As a general row it divides by zero like the reference's release build
    r.simple = r.simple && r.diag >= 1.f;
    return r;
}

// per entry s of the sorted band list: diagS = 0 for a general BOUNDARY cell, else the diagonal (6 for INTERIOR cells);
// general[s] = 1 / 0; *violations counts BOUNDARY cells that break the weight half of unitTestBoundaryCells
__global__ __launch_bounds__(256) void bandClassifyKernel(Dims d, const uint8_t *__restrict__ lab, WeightView wv, const int32_t *__restrict__ band,
                                                          int n, uint8_t *__restrict__ diagS, int32_t *__restrict__ general, int *__restrict__ violations,
                                                          size_t own0, size_t own1)
{
    const int s = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (s >= n) return;
    const size_t c = size_t(band[s]);
    uint8_t dg = 6;
    int gen = 0;
    if (lab[c] >= MGPS_BOUNDARY_CELL) {
        const RowEval r = evalRow(d, lab, wv, c);
        if (r.simple) dg = uint8_t(int(r.diag));
        else {
            dg = 0;
            gen = 1;
        }
        if (!r.ruleOk && violations && c >= own0 && c < own1) atomicAdd(violations, 1);  // (slab windows: every rank answers for its own cells)
    }
    diagS[s] = dg;
    general[s] = gen;
}

// the device order: entry = rank among the general cells, or nGeneral + rank among the rest
__global__ __launch_bounds__(256) void bandSplitKernel(Dims d, const uint8_t *__restrict__ lab, WeightView wv, const int32_t *__restrict__ band, int n,
                                                       const uint8_t *__restrict__ diagS, const int32_t *__restrict__ genRank, int32_t *__restrict__ bandDev,
                                                       uint8_t *__restrict__ bandDiag, int32_t *__restrict__ bandEntry, float *__restrict__ rows)
{
    const int s = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (s >= n) return;
    const int nGen = genRank[n], g = genRank[s];
    const bool gen = genRank[s + 1] != g;
    const int entry = gen ? g : nGen + (s - g);
    const int32_t c = band[s];
    bandDev[entry] = c;
    bandDiag[entry] = diagS[s];
    bandEntry[s] = entry;
    if (gen) {
        const RowEval r = evalRow(d, lab, wv, size_t(c));
#pragma unroll
        for (int q = 0; q < 6; ++q) rows[size_t(q) * nGen + g] = r.w[q];
        rows[size_t(6) * nGen + g] = r.diag;
    }
}

// out[t] = rank[start[t]] (per tile: first entry of its general BOUNDARY cells in the device order)
__global__ __launch_bounds__(256) void gatherKernel(const int32_t *__restrict__ rank, const int32_t *__restrict__ start, int n, int32_t *__restrict__ out)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n) out[t] = rank[start[t]];
}

// flags[t] = tile t holds band cells; then list[rank[t]] = t
__global__ __launch_bounds__(256) void tileFlagKernel(const int32_t *__restrict__ tileStart, int n, int32_t *__restrict__ flags)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n) flags[t] = tileStart[t + 1] > tileStart[t] ? 1 : 0;
}
__global__ __launch_bounds__(256) void tileListKernel(const int32_t *__restrict__ rank, int n, int32_t *__restrict__ list)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n && rank[t + 1] != rank[t]) list[rank[t]] = t;
}

// flags[t] = 1 for the tiles of one Gauss-Seidel list: active, of colour parity `odd` ((ti + tj + tk) & 1), pure (all 4096
// cells INTERIOR, kind bit 0) or mixed; which = 0 pure, 1 mixed (buildTileLists / tileListsFromKinds in mgps_host.cpp)
// tkOffset (slab windows): the global tile plane of the grid's first tile plane -- the colour of a tile is that of the whole grid
__global__ __launch_bounds__(256) void tileClassFlagKernel(const int32_t *__restrict__ kind, int n, int tx, int ty, int odd, int mixed,
                                                           int32_t *__restrict__ flags, int tkOffset)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n) return;
    const int kd = kind[t], ti = t % tx, tj = (t / tx) % ty, tk = t / (tx * ty) + tkOffset;
    flags[t] = ((kd >> 1) != 0 && ((ti + tj + tk) & 1) == odd && ((kd & 1) == 0) == (mixed != 0)) ? 1 : 0;
}
// flags of bytes (plane blocks) as ints for the scan
__global__ __launch_bounds__(256) void byteFlagKernel(const uint8_t *__restrict__ in, int n, int32_t *__restrict__ flags)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n) flags[t] = in[t] ? 1 : 0;
}

// ---- activity flags --------------------------------------------------------------------------------------------------

// chunkFlags[q] = the q-th run of kSegCells = 32 cells holds an active cell (8 lanes per run); planeFlags (optional, nx % 4 == 0):
// per block of the plane-marching sweep (256 cells in x, kPlaneRows rows, zc planes)
__global__ __launch_bounds__(256) void activityFlagsKernel(Dims d, const uint32_t *__restrict__ lab4, size_t nq, uint8_t *__restrict__ chunkFlags,
                                                           uint8_t *__restrict__ planeFlags, int zc, int nbx, int nby)
{
    const size_t q = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    bool any = false;
    if (q < nq) {
        const uint32_t v = lab4[q];
        any = activeCode(v & 0xffu) || activeCode((v >> 8) & 0xffu) || activeCode((v >> 16) & 0xffu) || activeCode(v >> 24);
        if (any && planeFlags) {
            const size_t c = q * 4;
            const int i = int(c % d.nx), j = int((c / d.nx) % d.ny), k = int(c / (size_t(d.nx) * d.ny));
            planeFlags[(size_t(k / zc) * nby + j / kPlaneRows) * nbx + i / 256] = 1;
        }
    }
    const unsigned long long votes = __ballot(any);
    static_assert(kSegCells == 32, "eight lanes of four cells per flag");
    if ((threadIdx.x & 7) == 0 && q < nq) chunkFlags[q >> 3] = ((votes >> (threadIdx.x & 56)) & 0xffull) ? 1 : 0;
}

// a thread takes the 32 flags of one 1024-cell run: how many of its runs of 1024 / 256 / 64 / 32 cells hold an active cell
__global__ __launch_bounds__(256) void countRunsKernel(const uint8_t *__restrict__ segFlags, size_t nseg, int *__restrict__ counts)
{
    const size_t q = blockIdx.x * size_t(blockDim.x) + threadIdx.x;  // 1024-cell run
    int c[4] = {0, 0, 0, 0};
    const size_t s0 = q * 32;
    if (s0 < nseg) {
        unsigned bits = 0;  // bit r = flag of the r-th 32-cell run
        for (int r = 0; r < 32; ++r)
            if (s0 + r < nseg && segFlags[s0 + r]) bits |= 1u << r;
        c[3] = __popc(bits);
        unsigned b64 = (bits | (bits >> 1)) & 0x55555555u;
        c[2] = __popc(b64);
        unsigned b256 = 0;
        for (int w = 0; w < 4; ++w) b256 |= ((bits >> (8 * w)) & 0xffu) ? 1u << w : 0u;
        c[1] = __popc(b256);
        c[0] = bits != 0;
    }
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        int v = c[z];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_down(v, s);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(counts + z, v);
    }
}
__global__ __launch_bounds__(256) void foldRunFlagsKernel(const uint8_t *__restrict__ segFlags, size_t nseg, int per, size_t nq, uint8_t *__restrict__ runFlags)
{
    const size_t q = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (q >= nq) return;
    bool any = false;
    for (int r = 0; r < per; ++r) any = any || (q * per + r < nseg && segFlags[q * per + r]);
    runFlags[q] = any ? 1 : 0;
}

// The activity list in launch order (runListFromFlags in mgps_host.cpp: by strip of 32 rows of the run's first cell, then by
// run index): one stable compaction per strip -- flags of the strip's active runs, scan, scatter behind the strips before.
__global__ __launch_bounds__(256) void stripFlagKernel(const uint8_t *__restrict__ runFlags, size_t nq, size_t cpr, size_t nx, size_t ny, int strip,
                                                       int32_t *__restrict__ out)
{
    const size_t q = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (q >= nq) return;
    bool on = runFlags[q] != 0;
    if (on && strip >= 0) on = int(((q * cpr / nx) % ny) >> 5) == strip;
    out[q] = on ? 1 : 0;
}
__global__ __launch_bounds__(256) void stripScatterKernel(const int32_t *__restrict__ rank, size_t nq, const int32_t *__restrict__ base,
                                                          int32_t *__restrict__ list)
{
    const size_t q = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (q < nq && rank[q + 1] != rank[q]) list[size_t(*base) + size_t(rank[q])] = int32_t(q);
}
// *base += total; when `last`, the tail of the list up to listLen is padding (-1)
__global__ void stripAdvanceKernel(int32_t *__restrict__ base, const int32_t *__restrict__ total, int32_t *__restrict__ list, int listLen, int last)
{
    const int b = *base + *total;
    if (last)
        for (int t = b + int(threadIdx.x); t < listLen; t += int(blockDim.x)) list[t] = -1;
    __syncthreads();
    if (threadIdx.x == 0) *base = b;
}

// ---- boxes of the fused band stage (BandBoxes in mgps_internal.h) -----------------------------------------------------

// flags[t] = 1: tile t or one of its six face neighbours holds band cells (only such a tile can hold a band-closure cell)
__global__ __launch_bounds__(256) void boxTileFlagKernel(const int32_t *__restrict__ tileStart, int tx, int ty, int tz, int32_t *__restrict__ flags)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= tx * ty * tz) return;
    const int ti = t % tx, tj = (t / tx) % ty, tk = t / (tx * ty);
    auto has = [&](int a, int b, int c) {
        if (a < 0 || b < 0 || c < 0 || a >= tx || b >= ty || c >= tz) return false;
        const int q = (c * ty + b) * tx + a;
        return tileStart[q + 1] > tileStart[q];
    };
    flags[t] = (has(ti, tj, tk) || has(ti - 1, tj, tk) || has(ti + 1, tj, tk) || has(ti, tj - 1, tk) || has(ti, tj + 1, tk) || has(ti, tj, tk - 1) ||
                has(ti, tj, tk + 1))
                   ? 1
                   : 0;
}

// One workgroup per listed 16^3 tile walks the recursion of the host builder (buildBandBoxes, mgps_host.cpp): a window of
// the tile grown by depth + 2 cells holds per cell "active" and "band member" (labels, band masks); a stack of sub-boxes of
// the tile is walked left half first: O = bounding box of the sub-box's closure cells (band cells and active cells next to
// one), every cell of O grown by depth + 1 is classified (BoxNode), R = bounding box of the classes != 0; a group whose R
// exceeds the band kernel's budget is halved along O's longest axis.  FILL = false: per tile the number of groups, code
// bytes and general entries; FILL = true: the arrays, at the tile's scanned offsets.
constexpr int kBoxBuildThreads = 256;

struct Box {
    int8_t lo[3], hi[3];
};

template <bool FILL>
__global__ __launch_bounds__(kBoxBuildThreads) void boxBuildKernel(Dims d, const uint8_t *__restrict__ lab, int tx, int ty, const uint32_t *__restrict__ mask,
                                                                   const uint16_t *__restrict__ prefix, const int32_t *__restrict__ tileStart,
                                                                   const int32_t *__restrict__ bandEntry, const uint8_t *__restrict__ bandDiag, int depth,
                                                                   const int32_t *__restrict__ tiles, int32_t *__restrict__ nGroups,
                                                                   int32_t *__restrict__ nList, int32_t *__restrict__ nGeneral,
                                                                   const int32_t *__restrict__ groupAt, const int32_t *__restrict__ listAt,
                                                                   const int32_t *__restrict__ generalAt, int32_t *__restrict__ info,
                                                                   uint32_t *__restrict__ list, int32_t *__restrict__ general, int *__restrict__ broken,
                                                                   int own0, int own1)
{
    extern __shared__ uint8_t sm[];
    const int D = depth, P = depth + 2, E = kTile + 2 * P, E2 = E * E, E3 = E2 * E;
    uint8_t *fl = sm;        // bit 0 active, bit 1 band
    uint8_t *cls = sm + E3;  // BoxNode class of the cells of O grown by depth + 1
    __shared__ Box stack[48];
    __shared__ int sp;
    __shared__ int bb[6], rb[6], sGen, sListed;
    __shared__ int scratch[4];
    const int tid = threadIdx.x;
    const int t = tiles[blockIdx.x], ti = t % tx, tj = (t / tx) % ty, tk = t / (tx * ty);
    const int oi = ti * kTile - P, oj = tj * kTile - P, ok = tk * kTile - P;  // grid cell of window cell (0, 0, 0)
    // entry (sorted band position) of the band cell at grid (gi, gj, gk)
    auto sortedOf = [&](int gi, int gj, int gk) {
        const int tile = ((gk >> 4) * ty + (gj >> 4)) * tx + (gi >> 4);
        const int b = (((gk & 15) << 4) | (gj & 15)) << 4 | (gi & 15);
        const uint32_t w = mask[size_t(tile) * 128 + (b >> 5)];
        return tileStart[tile] + int(prefix[size_t(tile) * 128 + (b >> 5)]) + __popc(w & ((1u << (b & 31)) - 1u));
    };
    // the window's labels: EXTERIOR everywhere, then the part inside the grid row by row as aligned 4-byte words
    for (int q = tid; q < (E3 + 3) / 4; q += kBoxBuildThreads) reinterpret_cast<uint32_t *>(fl)[q] = 0x01010101u * MGPS_EXTERIOR_CELL;
    if (tid == 0) {  // (slab windows: the owned boxes lie in the rank's own planes [own0, own1) of the label buffer)
        const int zl = max(0, own0 - tk * kTile), zh = min(kTile - 1, own1 - 1 - tk * kTile);
        sp = zl <= zh ? 1 : 0;
        for (int a = 0; a < 3; ++a) {
            stack[0].lo[a] = 0;
            stack[0].hi[a] = kTile - 1;
        }
        stack[0].lo[2] = int8_t(max(zl, 0));
        stack[0].hi[2] = int8_t(min(zh, kTile - 1));
    }
    __syncthreads();
    {
        const int dshift = E + 7 <= 32 ? 3 : 4;  // words a row can touch
        for (int w = tid; w < (E2 << dshift); w += kBoxBuildThreads) {
            const int r = w >> dshift, q = w & ((1 << dshift) - 1);
            const int lj = r % E, lk = r / E, gj = oj + lj, gk = ok + lk;
            if (gj < 0 || gj >= d.ny || gk < 0 || gk >= d.nz) continue;
            const ptrdiff_t rowBase = ptrdiff_t(cellIdx(d, 0, gj, gk));
            const ptrdiff_t a0 = rowBase + max(oi, 0), a1 = rowBase + min(oi + E, d.nx);
            const ptrdiff_t addr = (a0 & ~ptrdiff_t(3)) + 4 * q;
            if (addr >= a1) continue;
            const uint32_t word = *reinterpret_cast<const uint32_t *>(lab + addr);  // (the allocation has a spare plane on each side)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const ptrdiff_t at = addr + b;
                if (at >= a0 && at < a1) fl[r * E + int(at - rowBase) - oi] = uint8_t(word >> (8 * b));
            }
        }
    }
    __syncthreads();
    // labels -> flags, band membership from the masks of the (up to three) tiles a row crosses
    for (int r = tid; r < E2; r += kBoxBuildThreads) {
        const int lj = r % E, lk = r / E, gj = oj + lj, gk = ok + lk;
        const bool rowIn = gj >= 0 && gj < d.ny && gk >= 0 && gk < d.nz;
        uint32_t w0 = 0, w1 = 0, w2 = 0;
        const int tix0 = max(oi, 0) >> 4;
        if (rowIn) {
            const size_t trow = size_t((gk >> 4) * ty + (gj >> 4)) * tx;
            const int word = (((gk & 15) << 4) | (gj & 15)) >> 1, sh = (gj & 1) << 4;
            w0 = (mask[(trow + tix0) * 128 + word] >> sh) & 0xffffu;
            if (tix0 + 1 < tx) w1 = (mask[(trow + tix0 + 1) * 128 + word] >> sh) & 0xffffu;
            if (tix0 + 2 < tx) w2 = (mask[(trow + tix0 + 2) * 128 + word] >> sh) & 0xffffu;
        }
        for (int li = 0; li < E; ++li) {
            const int gi = oi + li, q = (gi >> 4) - tix0;
            uint8_t f = 0;
            if (rowIn && gi >= 0 && gi < d.nx && activeCode(fl[r * E + li])) {
                const uint32_t w = q == 0 ? w0 : (q == 1 ? w1 : w2);
                f = ((w >> (gi & 15)) & 1u) ? 3 : 1;
            }
            fl[r * E + li] = f;
        }
    }
    const int off[6] = {-1, 1, -E, E, -E2, E2};
    int groups = 0, lists = 0, gens = 0;  // running totals of this tile (uniform across the workgroup)
    const int gBase = FILL ? groupAt[blockIdx.x] : 0, uBase = FILL ? listAt[blockIdx.x] : 0, nBase = FILL ? generalAt[blockIdx.x] : 0;
    while (true) {
        __syncthreads();
        if (sp == 0) break;
        const Box B = stack[sp - 1];
        __syncthreads();
        if (tid == 0) {
            --sp;
            bb[0] = bb[1] = bb[2] = 99;
            bb[3] = bb[4] = bb[5] = -1;
            rb[0] = rb[1] = rb[2] = 99;
            rb[3] = rb[4] = rb[5] = -1;
            sGen = 0;
            sListed = 0;
        }
        __syncthreads();
        {  // O: tight bounding box of the closure cells of the tile inside B (thread = one x-row of the tile)
            const int lk = tid >> 4, lj = tid & 15;
            if (lk >= B.lo[2] && lk <= B.hi[2] && lj >= B.lo[1] && lj <= B.hi[1]) {
                int first = 99, lastI = -1;
                for (int li = B.lo[0]; li <= B.hi[0]; ++li) {
                    const int w = ((lk + P) * E + lj + P) * E + li + P;
                    const unsigned f = fl[w];
                    bool clo = (f & 2u) != 0;
                    if (!clo && (f & 1u)) clo = ((fl[w - 1] | fl[w + 1] | fl[w - E] | fl[w + E] | fl[w - E2] | fl[w + E2]) & 2u) != 0;
                    if (clo) {
                        first = min(first, li);
                        lastI = li;
                    }
                }
                if (lastI >= 0) {
                    atomicMin(&bb[0], first);
                    atomicMax(&bb[3], lastI);
                    atomicMin(&bb[1], lj);
                    atomicMax(&bb[4], lj);
                    atomicMin(&bb[2], lk);
                    atomicMax(&bb[5], lk);
                }
            }
        }
        __syncthreads();
        if (bb[3] < 0) continue;
        const int lo[3] = {bb[0], bb[1], bb[2]}, hi[3] = {bb[3], bb[4], bb[5]};
        const int olo[3] = {lo[0] + P, lo[1] + P, lo[2] + P}, ohi[3] = {hi[0] + P, hi[1] + P, hi[2] + P};      // O in window coordinates
        const int mlo[3] = {olo[0] - (D + 1), olo[1] - (D + 1), olo[2] - (D + 1)}, mhi[3] = {ohi[0] + D + 1, ohi[1] + D + 1, ohi[2] + D + 1};
        const int mx = mhi[0] - mlo[0] + 1, my = mhi[1] - mlo[1] + 1, mz = mhi[2] - mlo[2] + 1, mrows = my * mz;
        auto ringOf = [&](int wi, int wj, int wk) {
            return max(max(max(olo[0] - wi, wi - ohi[0]), max(olo[1] - wj, wj - ohi[1])), max(max(olo[2] - wk, wk - ohi[2]), 0));
        };
        const int dI[6] = {-1, 1, 0, 0, 0, 0}, dJ[6] = {0, 0, -1, 1, 0, 0}, dK[6] = {0, 0, 0, 0, -1, 1};
        for (int r = tid; r < mrows; r += kBoxBuildThreads) {
            const int wj = mlo[1] + r % my, wk = mlo[2] + r / my;
            for (int wi = mlo[0]; wi <= mhi[0]; ++wi) {
                const int w = (wk * E + wj) * E + wi;
                const unsigned f = fl[w];
                uint8_t c = kBoxSkip;
                if (f & 2u) c = kBoxGeneral;  // band cell (its code is looked up when the group is written)
                else {
                    int best = 99;  // smallest ring among the band face neighbours
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        if (fl[w + off[q]] & 2u) best = min(best, ringOf(wi + dI[q], wj + dJ[q], wk + dK[q]));
                    if (f & 1u) {
                        if (best < 99 && ringOf(wi, wj, wk) == 0) c = kBoxFrozenOut;
                        else if (best <= D - 1) c = kBoxFrozen;
                        else if (best <= D) c = kBoxFrozenFar;
                    } else if (best <= D)
                        c = kBoxZero;
                }
                cls[w] = c;
            }
        }
        __syncthreads();
        // active cells next to a closure-output cell: read by the closure pass (only class 11 seeds: no cascade)
        for (int r = tid; r < mrows; r += kBoxBuildThreads) {
            const int wj = mlo[1] + r % my, wk = mlo[2] + r / my;
            for (int wi = mlo[0]; wi <= mhi[0]; ++wi) {
                const int w = (wk * E + wj) * E + wi;
                if (cls[w] != kBoxSkip || !(fl[w] & 1u)) continue;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int ni = wi + dI[q], nj = wj + dJ[q], nk = wk + dK[q];
                    if (ni < mlo[0] || ni > mhi[0] || nj < mlo[1] || nj > mhi[1] || nk < mlo[2] || nk > mhi[2]) continue;
                    if (cls[w + off[q]] == kBoxFrozenOut) {
                        cls[w] = kBoxFrozenFar;
                        break;
                    }
                }
            }
        }
        __syncthreads();
        {  // R = bounding box of the classes != 0; general band cells with ring <= depth
            int l0 = 99, l1 = 99, l2 = 99, h0 = -1, h1 = -1, h2 = -1, gen = 0, listed = 0;
            for (int r = tid; r < mrows; r += kBoxBuildThreads) {
                const int wj = mlo[1] + r % my, wk = mlo[2] + r / my;
                for (int wi = mlo[0]; wi <= mhi[0]; ++wi) {
                    const int w = (wk * E + wj) * E + wi;
                    if (cls[w] == kBoxSkip) continue;
                    ++listed;
                    l0 = min(l0, wi);
                    h0 = max(h0, wi);
                    l1 = min(l1, wj);
                    h1 = max(h1, wj);
                    l2 = min(l2, wk);
                    h2 = max(h2, wk);
                    if ((fl[w] & 2u) && ringOf(wi, wj, wk) <= D && bandDiag[bandEntry[sortedOf(oi + wi, oj + wj, ok + wk)]] == 0) ++gen;
                }
            }
            if (h0 >= 0) {
                atomicMin(&rb[0], l0);
                atomicMax(&rb[3], h0);
                atomicMin(&rb[1], l1);
                atomicMax(&rb[4], h1);
                atomicMin(&rb[2], l2);
                atomicMax(&rb[5], h2);
                if (gen) atomicAdd(&sGen, gen);
                atomicAdd(&sListed, listed);
            }
        }
        __syncthreads();
        const int rlo[3] = {rb[0], rb[1], rb[2]}, rhi[3] = {rb[3], rb[4], rb[5]}, gen = sGen;
        const int rx = rhi[0] - rlo[0] + 1, ry = rhi[1] - rlo[1] + 1, rz = rhi[2] - rlo[2] + 1, nodes = rx * ry * rz;
        (void)mx;
        if (nodes > kBoxMaxNodes || sListed > kBoxMaxList || gen > kBoxMaxGeneral) {
            __syncthreads();
            if (tid == 0) {
                int axis = 0;
                for (int a = 1; a < 3; ++a)
                    if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
                if (hi[axis] == lo[axis] || sp + 2 > 48) *broken = 1;
                else {
                    const int mid = (lo[axis] + hi[axis] + 1) / 2;
                    Box Lh, Rh;
                    for (int a = 0; a < 3; ++a) {
                        Lh.lo[a] = Rh.lo[a] = int8_t(lo[a]);
                        Lh.hi[a] = Rh.hi[a] = int8_t(hi[a]);
                    }
                    Lh.hi[axis] = int8_t(mid - 1);
                    Rh.lo[axis] = int8_t(mid);
                    stack[sp++] = Rh;  // the left half is walked first
                    stack[sp++] = Lh;
                }
            }
            continue;
        }
        if (rx > 31 || ry > 31 || rz > 31) {  // (5 bits per coordinate; O grown by depth + 1 is at most 26 wide)
            if (tid == 0) *broken = 1;
            continue;
        }
        // Lists in region order (k, j, i) per category: band cells by ring 0 .. depth, closure-output cells, what the plain mode
        // reads besides, what only the closure mode reads.  A thread owns a run of consecutive x-rows of R: (thread, position)
        // order is the region's order.
        constexpr int kCatOut = kBandMaxDepth + 1, kCatReadPlain = kCatOut + 1, kCatReadFar = kCatOut + 2, kCats = kCatOut + 3;
        auto catOf = [&](int wi, int wj, int wk, bool &zero) {
            const int w = (wk * E + wj) * E + wi;
            zero = false;
            const uint8_t c = cls[w];
            if (c == kBoxSkip) return -1;
            if (fl[w] & 2u) {
                const int ring = ringOf(wi, wj, wk);
                return ring <= D ? ring : kCatReadFar;
            }
            if (c == kBoxFrozenOut) return kCatOut;
            if (c == kBoxFrozenFar) return kCatReadFar;
            zero = c == kBoxZero;
            return kCatReadPlain;
        };
        const int R = ry * rz, rp = (R + kBoxBuildThreads - 1) / kBoxBuildThreads, r0 = min(R, tid * rp), r1 = min(R, r0 + rp);
        int cnt[kCats], total[kCats], myGen = 0, myAll = 0;
#pragma unroll
        for (int q = 0; q < kCats; ++q) cnt[q] = 0;
        for (int r = r0; r < r1; ++r) {
            const int wj = rlo[1] + r % ry, wk = rlo[2] + r / ry;
            for (int wi = rlo[0]; wi <= rhi[0]; ++wi) {
                bool zero;
                const int c = catOf(wi, wj, wk, zero);
#pragma unroll
                for (int q = 0; q < kCats; ++q) cnt[q] += c == q ? 1 : 0;
                myAll += c >= 0 ? 1 : 0;
                if (c >= 0 && c <= kBandMaxDepth && bandDiag[bandEntry[sortedOf(oi + wi, oj + wj, ok + wk)]] == 0) ++myGen;
            }
        }
        int nListGroup = 0, genTotal = 0;
        const int allBefore = blockExclusiveScan(myAll, &nListGroup, scratch);
        const int genBefore = blockExclusiveScan(myGen, &genTotal, scratch);
        if (FILL) {
            // the statistics of the group (ring counts): only the fill pass needs them
            int dummy;
#pragma unroll
            for (int q = 0; q <= kCatOut; ++q) {
                dummy = blockExclusiveScan(cnt[q], &total[q], scratch);
                (void)dummy;
            }
            uint32_t *udst = list + size_t(uBase + lists);
            int32_t *gdst = general + 2 * size_t(nBase + gens + genBefore);
            int pos = allBefore;
            for (int r = r0; r < r1; ++r) {
                const int wj = rlo[1] + r % ry, wk = rlo[2] + r / ry;
                for (int wi = rlo[0]; wi <= rhi[0]; ++wi) {
                    bool zero;
                    const int c = catOf(wi, wj, wk, zero);
                    if (c < 0) continue;
                    const uint32_t coords = uint32_t(wi - rlo[0]) | (uint32_t(wj - rlo[1]) << 5) | (uint32_t(wk - rlo[2]) << 10);
                    uint32_t code = c == kCatOut ? uint32_t(kBoxFrozenOut) : c == kCatReadFar ? uint32_t(kBoxFrozenFar) : zero ? uint32_t(kBoxZero) : uint32_t(kBoxFrozen);
                    const uint32_t ring = uint32_t(min(ringOf(wi, wj, wk), 7));
                    if (c <= kBandMaxDepth) {
                        const int e = bandEntry[sortedOf(oi + wi, oj + wj, ok + wk)];
                        const int dg = bandDiag[e];
                        code = dg == 0 ? uint32_t(kBoxGeneral) : uint32_t(kBoxSimple + dg);
                        if (dg == 0) {
                            gdst[0] = int32_t(coords | (code << 16) | (ring << 20));
                            gdst[1] = e;
                            gdst += 2;
                        }
                    }
                    udst[pos++] = coords | (code << 16) | (ring << 20);
                }
            }
            if (tid == 0) {
                int32_t *g16 = info + kBoxInfoInts * size_t(gBase + groups);
                g16[0] = int32_t(cellIdx(d, oi + rlo[0], oj + rlo[1], ok + rlo[2]));
                g16[1] = rx | (ry << 8) | (rz << 16);
                g16[2] = uBase + lists;
                g16[3] = 0;
                g16[4] = nBase + gens;
                g16[5] = genTotal;
                g16[6] = 0;
                g16[7] = nListGroup;
                int run = 0;
#pragma unroll
                for (int q = 0; q <= kBandMaxDepth; ++q) {
                    run += total[q];
                    g16[8 + q] = run;
                }
                g16[13] = total[kCatOut];
                g16[14] = (olo[0] - rlo[0]) | ((olo[1] - rlo[1]) << 8) | ((olo[2] - rlo[2]) << 16);
                g16[15] = (ohi[0] - olo[0] + 1) | ((ohi[1] - olo[1] + 1) << 8) | ((ohi[2] - olo[2] + 1) << 16);
            }
        }
        ++groups;
        lists += nListGroup;
        gens += genTotal;
    }
    if (!FILL && tid == 0) {
        nGroups[blockIdx.x] = groups;
        nList[blockIdx.x] = lists;
        nGeneral[blockIdx.x] = gens;
    }
}

}  // namespace

// ---- launchers ------------------------------------------------------------------------------------------------------

int launchCoarsenLabels(void *stream, const Dims &fine, const uint8_t *fineLab, uint8_t *coarseLab, int *activeFlag)
{
    const Dims cd{fine.nx / 2, fine.ny / 2, fine.nz / 2};
    coarsenLabelsKernel<<<blocksFor(cd.cells(), 256), 256, 0, S(stream)>>>(fine, cd, fineLab, coarseLab, activeFlag);
    return int(hipGetLastError());
}
int launchAnyActive(void *stream, const Dims &d, const uint8_t *lab, int *activeFlag)
{
    const size_t nq = d.cells() / 4;  // extents are even: cells() is a multiple of 8
    anyActiveKernel<<<unsigned(std::min<size_t>(blocksFor(nq, 256), 8192)), 256, 0, S(stream)>>>(reinterpret_cast<const uint32_t *>(lab), nq, activeFlag);
    return int(hipGetLastError());
}
int launchShellCheck(void *stream, const Dims &d, const uint8_t *lab, int *badFlag)
{
    const size_t n = 2 * (size_t(d.nx) * d.ny + size_t(d.nx) * d.nz + size_t(d.ny) * d.nz);
    shellCheckKernel<<<blocksFor(n, 256), 256, 0, S(stream)>>>(d, lab, badFlag);
    return int(hipGetLastError());
}
int launchMarkBoundary(void *stream, const Dims &d, uint8_t *lab)
{
    markBoundaryKernel<<<blocksFor(d.cells(), 256), 256, 0, S(stream)>>>(d, lab);
    return int(hipGetLastError());
}

size_t scanScratchInts(size_t n) { return (n + kScanTile - 1) / kScanTile + 2; }
int launchExclusiveScan(void *stream, const int32_t *in, int32_t *out, size_t n, int32_t *scratch)
{
    const size_t nb = std::max<size_t>(1, (n + kScanTile - 1) / kScanTile);
    scanReduceKernel<<<unsigned(nb), kScanThreads, 0, S(stream)>>>(in, n, scratch);
    scanSumsKernel<<<1, kScanThreads, 0, S(stream)>>>(scratch, nb);
    scanDownKernel<<<unsigned(nb), kScanThreads, 0, S(stream)>>>(in, out, n, scratch, nb);
    return int(hipGetLastError());
}

// tileKind and tileBits of every tile, then flags / rank / list: the tiles that go through the band kernel (rank[nt] = count)
int launchBandCandidates(void *stream, const Dims &d, const uint8_t *lab, int32_t *tileKind, uint8_t *tileBits, int32_t *flags, int32_t *rank,
                         int32_t *list, int32_t *scanScratch, const SlabWindow *win)
{
    const int tx = (d.nx + kTile - 1) / kTile, ty = (d.ny + kTile - 1) / kTile, tz = (d.nz + kTile - 1) / kTile, nt = tx * ty * tz;
    tileStatsKernel<<<unsigned(ty * tz), 64, 0, S(stream)>>>(d, lab, tx, ty, tileKind, tileBits);
    tileCandidateKernel<<<blocksFor(size_t(nt), 256), 256, 0, S(stream)>>>(tx, ty, tz, tileBits, flags, win ? win->own0 - win->need : 0, win ? win->own1 + win->need : d.nz);
    const int e = launchExclusiveScan(stream, flags, rank, size_t(nt), scanScratch);
    if (e != 0) return e;
    tileListKernel<<<blocksFor(size_t(nt), 256), 256, 0, S(stream)>>>(rank, nt, list);
    return int(hipGetLastError());
}
// tiles == nullptr: every tile (and the kernel fills tileKind); else the ntiles listed ones -- mask / prefix / tileCount of
// the others must have been zeroed, their tileKind set by launchBandCandidates
int launchBandMasks(void *stream, const Dims &d, const uint8_t *lab, int width, uint32_t *mask, uint16_t *prefix, int32_t *tileCount, int32_t *tileKind,
                    int *interiorBad, const int32_t *tiles, int ntiles, const SlabWindow *win)
{
    const int tx = (d.nx + kTile - 1) / kTile, ty = (d.ny + kTile - 1) / kTile, tz = (d.nz + kTile - 1) / kTile;
    if (width < 1 || width > 8) return int(hipErrorInvalidValue);
    const int E = kTile + 2 * std::max(width - 1, 1);
    const size_t lds = 2 * size_t(E) * E * E;
    const unsigned nb = tiles ? unsigned(ntiles) : unsigned(tx * ty * tz);
    if (nb > 0)
        bandMaskKernel<<<nb, 256, lds, S(stream)>>>(d, lab, width, tx, ty, mask, prefix, tileCount, tileKind, interiorBad, tiles, win ? win->own0 : 0, win ? win->own1 : d.nz);
    return int(hipGetLastError());
}
int launchBandFill(void *stream, const Dims &d, const uint32_t *mask, const uint16_t *prefix, const int32_t *tileStart, int32_t *band)
{
    const int tx = (d.nx + kTile - 1) / kTile, ty = (d.ny + kTile - 1) / kTile, tz = (d.nz + kTile - 1) / kTile;
    bandFillKernel<<<unsigned(tx * ty * tz), 256, 0, S(stream)>>>(d, tx, ty, mask, prefix, tileStart, band);
    return int(hipGetLastError());
}
static WeightView wholeGridWeights(const Dims &d, const float *wx, const float *wy, const float *wz)
{
    WeightView wv;
    wv.w[0] = wx, wv.w[1] = wy, wv.w[2] = wz;
    wv.nz = d.nz;
    return wv;
}
int launchBandClassify(void *stream, const Dims &d, const uint8_t *lab, const float *wx, const float *wy, const float *wz, const int32_t *band, int n,
                       uint8_t *diagS, int32_t *general, int *violations)
{
    return launchBandClassify(stream, d, lab, wholeGridWeights(d, wx, wy, wz), band, n, diagS, general, violations, nullptr);
}
int launchBandClassify(void *stream, const Dims &d, const uint8_t *lab, const WeightView &wv, const int32_t *band, int n, uint8_t *diagS, int32_t *general,
                       int *violations, const SlabWindow *win)
{
    const size_t plane = size_t(d.nx) * d.ny;
    if (n > 0)
        bandClassifyKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(d, lab, wv, band, n, diagS, general, violations, win ? size_t(win->own0) * plane : 0,
                                                                           win ? size_t(win->own1) * plane : d.cells());
    return int(hipGetLastError());
}
int launchBandSplit(void *stream, const Dims &d, const uint8_t *lab, const float *wx, const float *wy, const float *wz, const int32_t *band, int n,
                    const uint8_t *diagS, const int32_t *genRank, int32_t *bandDev, uint8_t *bandDiag, int32_t *bandEntry, float *rows)
{
    return launchBandSplit(stream, d, lab, wholeGridWeights(d, wx, wy, wz), band, n, diagS, genRank, bandDev, bandDiag, bandEntry, rows);
}
int launchBandSplit(void *stream, const Dims &d, const uint8_t *lab, const WeightView &wv, const int32_t *band, int n, const uint8_t *diagS, const int32_t *genRank,
                    int32_t *bandDev, uint8_t *bandDiag, int32_t *bandEntry, float *rows)
{
    if (n > 0) bandSplitKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(d, lab, wv, band, n, diagS, genRank, bandDev, bandDiag, bandEntry, rows);
    return int(hipGetLastError());
}
int launchGather(void *stream, const int32_t *rank, const int32_t *start, int n, int32_t *out)
{
    if (n > 0) gatherKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(rank, start, n, out);
    return int(hipGetLastError());
}
int launchActivityFlags(void *stream, const Dims &d, const uint8_t *lab, uint8_t *chunkFlags, uint8_t *planeFlags, int zc)
{
    const size_t nq = d.cells() / 4;
    const int nbx = (d.nx + 255) / 256, nby = (d.ny + kPlaneRows - 1) / kPlaneRows;
    activityFlagsKernel<<<blocksFor(nq, 256), 256, 0, S(stream)>>>(d, reinterpret_cast<const uint32_t *>(lab), nq, chunkFlags, zc ? planeFlags : nullptr, zc ? zc : 1,
                                                                 nbx, nby);
    return int(hipGetLastError());
}

int launchCountRuns(void *stream, const uint8_t *segFlags, size_t nseg, int *counts)
{
    static_assert(kSegCells == 32 && kChunkCells == 1024, "countRunsKernel folds 32 flags per 1024-cell run");
    const size_t n1024 = (nseg + 31) / 32;
    countRunsKernel<<<blocksFor(n1024, 256), 256, 0, S(stream)>>>(segFlags, nseg, counts);
    return int(hipGetLastError());
}
int launchFoldRunFlags(void *stream, const uint8_t *segFlags, size_t nseg, int runCells, uint8_t *runFlags)
{
    const int per = runCells / kSegCells;
    const size_t nq = (nseg + per - 1) / per;
    foldRunFlagsKernel<<<blocksFor(nq, 256), 256, 0, S(stream)>>>(segFlags, nseg, per, nq, runFlags);
    return int(hipGetLastError());
}

// runFlags: nq flags of the runs of runCells cells; tmpFlags: nq ints, rank: nq + 1 ints, base: one int (scratch); list: listLen
// entries = the active runs in launch order, padded with -1
int launchRunList(void *stream, const Dims &d, const uint8_t *runFlags, size_t nq, int runCells, int32_t *tmpFlags, int32_t *rank, int32_t *scanScratch,
                  int32_t *base, int32_t *list, int listLen)
{
    const bool strips = size_t(d.nx) * d.ny * sizeof(float) > (size_t(256) << 10) && d.ny > 32;  // (the rule of runListFromFlags)
    const int nstrips = strips ? (d.ny + 31) / 32 : 1;
    hipError_t e = hipMemsetAsync(base, 0, sizeof(int32_t), S(stream));
    if (e != hipSuccess) return int(e);
    for (int sidx = 0; sidx < nstrips; ++sidx) {
        stripFlagKernel<<<blocksFor(nq, 256), 256, 0, S(stream)>>>(runFlags, nq, size_t(runCells), size_t(d.nx), size_t(d.ny), strips ? sidx : -1, tmpFlags);
        const int rc = launchExclusiveScan(stream, tmpFlags, rank, nq, scanScratch);
        if (rc != 0) return rc;
        stripScatterKernel<<<blocksFor(nq, 256), 256, 0, S(stream)>>>(rank, nq, base, list);
        stripAdvanceKernel<<<1, 256, 0, S(stream)>>>(base, rank + nq, list, listLen, sidx + 1 == nstrips ? 1 : 0);
    }
    return int(hipGetLastError());
}

// flags / rank: nt and nt + 1 ints of scratch; list: room for nt entries; rank[nt] = the number of band tiles afterwards
int launchBandTileList(void *stream, const int32_t *tileStart, int nt, int32_t *flags, int32_t *rank, int32_t *list, int32_t *scanScratch)
{
    tileFlagKernel<<<blocksFor(size_t(nt), 256), 256, 0, S(stream)>>>(tileStart, nt, flags);
    int e = launchExclusiveScan(stream, flags, rank, size_t(nt), scanScratch);
    if (e != 0) return e;
    tileListKernel<<<blocksFor(size_t(nt), 256), 256, 0, S(stream)>>>(rank, nt, list);
    return int(hipGetLastError());
}
// list = the tiles t with (kind[t] >> 1) != 0 of one colour and class, ascending; rank[nt] = their number afterwards
int launchTileClassList(void *stream, const Dims &d, const int32_t *kind, int odd, int mixed, int32_t *flags, int32_t *rank, int32_t *list,
                        int32_t *scanScratch, int tkOffset)
{
    const int tx = (d.nx + kTile - 1) / kTile, ty = (d.ny + kTile - 1) / kTile, tz = (d.nz + kTile - 1) / kTile, nt = tx * ty * tz;
    tileClassFlagKernel<<<blocksFor(size_t(nt), 256), 256, 0, S(stream)>>>(kind, nt, tx, ty, odd, mixed, flags, tkOffset);
    const int e = launchExclusiveScan(stream, flags, rank, size_t(nt), scanScratch);
    if (e != 0) return e;
    tileListKernel<<<blocksFor(size_t(nt), 256), 256, 0, S(stream)>>>(rank, nt, list);
    return int(hipGetLastError());
}
// list = the indices of the non-zero bytes, ascending (the active blocks of the plane-marching sweep); rank[n] = their number
int launchByteList(void *stream, const uint8_t *bytes, int n, int32_t *flags, int32_t *rank, int32_t *list, int32_t *scanScratch)
{
    if (n <= 0) return 0;
    byteFlagKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(bytes, n, flags);
    const int e = launchExclusiveScan(stream, flags, rank, size_t(n), scanScratch);
    if (e != 0) return e;
    tileListKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(rank, n, list);
    return int(hipGetLastError());
}
// list = the tiles that can hold a band-closure cell (a band cell in the tile or in a face neighbour); rank[nt] = their number
int launchBoxTileList(void *stream, const Dims &d, const int32_t *tileStart, int32_t *flags, int32_t *rank, int32_t *list, int32_t *scanScratch)
{
    const int tx = (d.nx + kTile - 1) / kTile, ty = (d.ny + kTile - 1) / kTile, tz = (d.nz + kTile - 1) / kTile, nt = tx * ty * tz;
    boxTileFlagKernel<<<blocksFor(size_t(nt), 256), 256, 0, S(stream)>>>(tileStart, tx, ty, tz, flags);
    const int e = launchExclusiveScan(stream, flags, rank, size_t(nt), scanScratch);
    if (e != 0) return e;
    tileListKernel<<<blocksFor(size_t(nt), 256), 256, 0, S(stream)>>>(rank, nt, list);
    return int(hipGetLastError());
}
static size_t boxBuildLds(int depth)
{
    const size_t E = size_t(kTile + 2 * (depth + 2));
    return 2 * E * E * E + 8;
}
int launchBandBoxesCount(void *stream, const Dims &d, const uint8_t *lab, const uint32_t *mask, const uint16_t *prefix, const int32_t *tileStart,
                         const int32_t *bandEntry, const uint8_t *bandDiag, int depth, const int32_t *tiles, int ntiles, int32_t *const counts[3], int *broken,
                         const SlabWindow *win)
{
    if (ntiles <= 0) return 0;
    const int tx = (d.nx + kTile - 1) / kTile, ty = (d.ny + kTile - 1) / kTile;
    boxBuildKernel<false><<<unsigned(ntiles), kBoxBuildThreads, boxBuildLds(depth), S(stream)>>>(d, lab, tx, ty, mask, prefix, tileStart, bandEntry, bandDiag, depth, tiles,
                                                                                                counts[0], counts[1], counts[2], nullptr, nullptr, nullptr, nullptr,
                                                                                                nullptr, nullptr, broken, win ? win->own0 : 0, win ? win->own1 : d.nz);
    return int(hipGetLastError());
}
int launchBandBoxesFill(void *stream, const Dims &d, const uint8_t *lab, const uint32_t *mask, const uint16_t *prefix, const int32_t *tileStart,
                        const int32_t *bandEntry, const uint8_t *bandDiag, int depth, const int32_t *tiles, int ntiles, const int32_t *const at[3],
                        int32_t *info, uint32_t *list, int32_t *general, int *broken, const SlabWindow *win)
{
    if (ntiles <= 0) return 0;
    const int tx = (d.nx + kTile - 1) / kTile, ty = (d.ny + kTile - 1) / kTile;
    boxBuildKernel<true><<<unsigned(ntiles), kBoxBuildThreads, boxBuildLds(depth), S(stream)>>>(d, lab, tx, ty, mask, prefix, tileStart, bandEntry, bandDiag, depth, tiles,
                                                                                               nullptr, nullptr, nullptr, at[0], at[1], at[2], info, list, general, broken,
                                                                                               win ? win->own0 : 0, win ? win->own1 : d.nz);
    return int(hipGetLastError());
}

// ---- launch order of the band boxes (round 4) -------------------------------------------------------------------------
// A group stages its owned box dilated by depth + 1 cells: on a face of the liquid the regions of neighbouring groups
// overlap by almost half (a 16 x 16 piece of the face reads 22 x 22 rows), and in the builders' order -- tile by tile, x
// fastest -- the z-neighbour of a group is a hundred groups away: its lines have left the chiplet's 4 MiB L2 by then, and the
// overlap is paid in fabric traffic (rocprofv3 PMC, round 3: 1.08 GB per closure launch at 1024^3 against ~0.55 GB of distinct
// lines).  The groups are therefore put in the Morton order of their tiles -- any run of 64 consecutive groups, what a chiplet
// has in flight, is a compact patch -- by a stable counting sort over the tile keys; only info moves (16 ints per group), the
// lists stay where their offsets point.
// Morton key of a tile with bx / by / bz bits per axis (a long or flat level interleaves only the bits every axis still has:
// the key space stays within 8 x the number of tiles)
__device__ __forceinline__ unsigned boxOrderKey(unsigned tx, unsigned ty, unsigned tz, int bx, int by, int bz)
{
    unsigned key = 0;
    int pos = 0;
    for (int p = 0; p < 10; ++p) {
        if (p < bx) key |= ((tx >> p) & 1u) << pos++;
        if (p < by) key |= ((ty >> p) & 1u) << pos++;
        if (p < bz) key |= ((tz >> p) & 1u) << pos++;
    }
    return key;
}
__global__ __launch_bounds__(256) void boxOrderKeysKernel(Dims d, const int32_t *__restrict__ info, int n, int bx, int by, int bz, int32_t *__restrict__ key,
                                                        int32_t *__restrict__ count, int32_t *__restrict__ first)
{
    const int gidx = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (gidx >= n) return;
    const int32_t *gi = info + kBoxInfoInts * size_t(gidx);
    const size_t c = size_t(gi[0]);
    const int ox = int(c % size_t(d.nx)) + (gi[14] & 255), oy = int((c / size_t(d.nx)) % size_t(d.ny)) + ((gi[14] >> 8) & 255),
              oz = int(c / (size_t(d.nx) * d.ny)) + ((gi[14] >> 16) & 255);
    const int k = int(boxOrderKey(unsigned(ox / kTile), unsigned(oy / kTile), unsigned(oz / kTile), bx, by, bz));
    key[gidx] = k;
    atomicAdd(count + k, 1);
    atomicMin(first + k, gidx);
}
__global__ __launch_bounds__(256) void boxOrderMoveKernel(const int32_t *__restrict__ info, int n, const int32_t *__restrict__ key,
                                                        const int32_t *__restrict__ start, const int32_t *__restrict__ first, int32_t *__restrict__ infoOut,
                                                        int *__restrict__ broken)
{
    // a thread per int: 16 per group
    const size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int gidx = int(t / kBoxInfoInts), q = int(t % kBoxInfoInts);
    if (gidx >= n) return;
    const int k = key[gidx], local = gidx - first[k];
    if (local < 0 || local >= start[k + 1] - start[k]) {  // (the groups of a tile are consecutive in both builders)
        *broken = 1;
        return;
    }
    const size_t pos = size_t(start[k]) + size_t(local);
    infoOut[pos * kBoxInfoInts + q] = info[size_t(gidx) * kBoxInfoInts + q];
}
__global__ __launch_bounds__(256) void fillIntKernel(int32_t *__restrict__ a, size_t n, int32_t v)
{
    const size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t < n) a[t] = v;
}
// infoOut: an array of the size of info; synchronises the stream (it owns temporaries)
int orderBandBoxes(void *stream, const Dims &d, const int32_t *info, int ngroups, int32_t *infoOut)
{
    if (ngroups <= 0) return 0;
    auto bitsFor = [](int tiles) {
        int b = 0;
        while ((1 << b) < tiles) ++b;
        return b;
    };
    const int bx = bitsFor((d.nx + kTile - 1) / kTile), by = bitsFor((d.ny + kTile - 1) / kTile), bz = bitsFor((d.nz + kTile - 1) / kTile);
    if (bx > 10 || by > 10 || bz > 10 || bx + by + bz > 28) return int(hipErrorInvalidValue);
    const size_t nbins = size_t(1) << (bx + by + bz);  // (at most 8 x the tiles of the level)
    int32_t *key = nullptr, *count = nullptr, *first = nullptr, *start = nullptr, *scratch = nullptr;
    int *broken = nullptr;
    auto release = [&]() {
        for (void *p : {(void *)key, (void *)count, (void *)first, (void *)start, (void *)scratch, (void *)broken})
            if (p) (void)deviceFree(p);
    };
    int rc = 0;
    if ((rc = deviceAlloc(reinterpret_cast<void **>(&key), size_t(ngroups) * 4)) || (rc = deviceAlloc(reinterpret_cast<void **>(&count), nbins * 4)) ||
        (rc = deviceAlloc(reinterpret_cast<void **>(&first), nbins * 4)) || (rc = deviceAlloc(reinterpret_cast<void **>(&start), (nbins + 1) * 4)) ||
        (rc = deviceAlloc(reinterpret_cast<void **>(&scratch), scanScratchInts(nbins) * 4)) || (rc = deviceAlloc(reinterpret_cast<void **>(&broken), 4))) {
        release();
        (void)hipGetLastError();
        return int(hipErrorOutOfMemory);  // (the caller keeps the builders' order: the ordering is an optimisation)
    }
    hipStream_t s = S(stream);
    (void)hipMemsetAsync(count, 0, nbins * 4, s);
    (void)hipMemsetAsync(broken, 0, 4, s);
    fillIntKernel<<<blocksFor(nbins, 256), 256, 0, s>>>(first, nbins, 0x7fffffff);
    boxOrderKeysKernel<<<blocksFor(size_t(ngroups), 256), 256, 0, s>>>(d, info, ngroups, bx, by, bz, key, count, first);
    rc = launchExclusiveScan(stream, count, start, nbins, scratch);
    boxOrderMoveKernel<<<blocksFor(size_t(ngroups) * kBoxInfoInts, 256), 256, 0, s>>>(info, ngroups, key, start, first, infoOut, broken);
    int bad = 0;
    if (!rc) rc = int(hipMemcpyAsync(&bad, broken, 4, hipMemcpyDeviceToHost, s));
    if (!rc) rc = int(hipStreamSynchronize(s));
    if (!rc) rc = int(hipGetLastError());
    release();
    if (!rc && bad) rc = int(hipErrorUnknown);
    return rc;
}

// ---- the region lists as the kernels walk them (round 4) -----------------------------------------------------------------
// The builders emit every region cell that matters, inactive neighbours (class 2: "the value is 0") and the cells only the
// closure mode reads (class 12) included -- 17 % and 21 % of the entries on the BASELINE cube (tools/box_stats.py).  The band
// stage follows its bytes (a second list made it 7 % slower), and the list is a quarter of them: here every group's list is
// rewritten without the class-2 entries (the kernel clears its LDS block instead) and with the class-12 entries moved to the
// end, stable otherwise -- info[3] = the entries the plain mode walks, info[7] = all of them, info[2] = the new offset.
constexpr int kCompactPer = kBoxMaxList / 256;
__device__ __forceinline__ int boxCompactKey(uint32_t e)
{
    const unsigned cls = (e >> 16) & 15u;
    return (cls == kBoxZero || cls == kBoxSkip) ? -1 : (cls == kBoxFrozenFar ? 1 : 0);
}
__global__ __launch_bounds__(256) void boxCompactCountKernel(const int32_t *__restrict__ info, const uint32_t *__restrict__ list, int32_t *__restrict__ cnt)
{
    __shared__ int part[4];
    const int32_t *gi = info + kBoxInfoInts * size_t(blockIdx.x);
    const uint32_t *U = list + gi[2];
    int mine = 0;
    for (int k = threadIdx.x; k < gi[7]; k += 256) mine += boxCompactKey(U[k]) >= 0 ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void boxCompactFillKernel(int32_t *__restrict__ info, const uint32_t *__restrict__ list, const int32_t *__restrict__ start,
                                                          uint32_t *__restrict__ listOut)
{
    __shared__ int counts[2][256];
    __shared__ int base[3];
    int32_t *gi = info + kBoxInfoInts * size_t(blockIdx.x);
    const int oldStart = gi[2], nList = gi[7], tid = threadIdx.x, k0 = tid * kCompactPer;
    const uint32_t *U = list + oldStart;
    uint32_t *O = listOut + start[blockIdx.x];
    uint32_t e[kCompactPer];
    int mine[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < kCompactPer; ++q) {
        e[q] = k0 + q < nList ? U[k0 + q] : (uint32_t(kBoxSkip) << 16);
        const int key = boxCompactKey(e[q]);
        mine[0] += key == 0 ? 1 : 0;
        mine[1] += key == 1 ? 1 : 0;
    }
    counts[0][tid] = mine[0];
    counts[1][tid] = mine[1];
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < 2; ++c) {
            base[c] = run;
            for (int t = 0; t < 256; ++t) {
                const int v = counts[c][t];
                counts[c][t] = run;
                run += v;
            }
        }
        base[2] = run;
        gi[2] = start[blockIdx.x];
        gi[3] = base[1];  // the plain mode's entries
        gi[7] = run;
    }
    __syncthreads();
    int at[2] = {counts[0][tid], counts[1][tid]};
#pragma unroll
    for (int q = 0; q < kCompactPer; ++q) {
        const int key = boxCompactKey(e[q]);
        if (key >= 0) O[at[key]++] = e[q];
    }
}
// listOut: an array of listCount entries (an upper bound); *newCount = the entries it holds afterwards.  Rewrites info in place;
// synchronises the stream
int compactBandBoxLists(void *stream, int32_t *info, const uint32_t *list, int ngroups, uint32_t *listOut, size_t *newCount)
{
    *newCount = 0;
    if (ngroups <= 0) return 0;
    int32_t *cnt = nullptr, *start = nullptr, *scratch = nullptr;
    auto release = [&]() {
        for (void *p : {(void *)cnt, (void *)start, (void *)scratch})
            if (p) (void)deviceFree(p);
    };
    int rc = 0;
    if ((rc = deviceAlloc(reinterpret_cast<void **>(&cnt), size_t(ngroups) * 4)) || (rc = deviceAlloc(reinterpret_cast<void **>(&start), (size_t(ngroups) + 1) * 4)) ||
        (rc = deviceAlloc(reinterpret_cast<void **>(&scratch), scanScratchInts(size_t(ngroups)) * 4))) {
        release();
        return rc;
    }
    hipStream_t s = S(stream);
    boxCompactCountKernel<<<unsigned(ngroups), 256, 0, s>>>(info, list, cnt);
    rc = launchExclusiveScan(stream, cnt, start, size_t(ngroups), scratch);
    boxCompactFillKernel<<<unsigned(ngroups), 256, 0, s>>>(info, list, start, listOut);
    int32_t total = 0;
    if (!rc) rc = int(hipMemcpyAsync(&total, start + ngroups, 4, hipMemcpyDeviceToHost, s));
    if (!rc) rc = int(hipStreamSynchronize(s));
    if (!rc) rc = int(hipGetLastError());
    release();
    *newCount = size_t(std::max(total, 0));
    return rc;
}


// ---- slab windows (round 5): a rank of a Z-slab run builds its levels on the device -----------------------------------------
// The rank holds a BUFFER of labels per distributed level: its owned planes [own0, own1) and label ghost planes on both sides
// (real labels as far as the band masks and the band boxes of the owned planes reach, EXTERIOR beyond).  The kernels above build
// band masks, band list, rows and boxes on that buffer as if it were a whole grid; what follows cuts the rank's own lists out of
// them and finds the cells of the neighbours' planes its boxes read.
namespace {

// the x / y faces of the rank's planes and, where the rank holds the first / last plane of the grid, that plane: flags[0] = 1 when
// a cell there is not EXTERIOR (shellCheckKernel for a slab: the ranks' verdicts are OR-ed by the caller)
__global__ __launch_bounds__(256) void shellCheckSlabKernel(Dims d, const uint8_t *__restrict__ lab, int zFaceLo, int zFaceHi, int *__restrict__ flags)
{
    const size_t fz = size_t(d.nx) * d.ny, fy = size_t(d.nx) * d.nz, fx = size_t(d.ny) * d.nz;
    size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    int i, j, k;
    if (t < 2 * fz) {
        const bool low = t < fz;
        if ((low && !zFaceLo) || (!low && !zFaceHi)) return;
        k = low ? 0 : d.nz - 1;
        t %= fz;
        i = int(t % d.nx);
        j = int(t / d.nx);
    } else if (t < 2 * fz + 2 * fy) {
        t -= 2 * fz;
        j = t < fy ? 0 : d.ny - 1;
        t %= fy;
        i = int(t % d.nx);
        k = int(t / d.nx);
    } else if (t < 2 * fz + 2 * fy + 2 * fx) {
        t -= 2 * fz + 2 * fy;
        i = t < fx ? 0 : d.nx - 1;
        t %= fx;
        j = int(t % d.ny);
        k = int(t / d.ny);
    } else
        return;
    if (lab[cellIdx(d, i, j, k)] != MGPS_EXTERIOR_CELL) flags[0] = 1;
}

// per entry s of the sorted band list of the buffer: own[s] = the cell lies in the rank's planes, ownGen[s] = ... and is a general cell
__global__ __launch_bounds__(256) void ownedFlagsKernel(const int32_t *__restrict__ band, const int32_t *__restrict__ general, int n, int32_t c0, int32_t c1,
                                                        int32_t *__restrict__ own, int32_t *__restrict__ ownGen)
{
    const int s = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (s >= n) return;
    const int32_t c = band[s];
    const int in = (c >= c0 && c < c1) ? 1 : 0;
    own[s] = in;
    ownGen[s] = in && general[s] ? 1 : 0;
}
// the rank's own band list in the device order (general cells first, each class in the buffer's order), cells as offsets from owned
// cell 0; the rows of its general cells copied from the buffer's (SoA, 7 x nGenOwn)
__global__ __launch_bounds__(256) void bandSplitOwnedKernel(const int32_t *__restrict__ band, int n, int32_t c0, const uint8_t *__restrict__ diagS,
                                                            const int32_t *__restrict__ ownRank, const int32_t *__restrict__ ownGenRank,
                                                            const int32_t *__restrict__ genRank, const float *__restrict__ rowsAll,
                                                            int32_t *__restrict__ bandOut, uint8_t *__restrict__ diagOut, float *__restrict__ rowsOut)
{
    const int s = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (s >= n || ownRank[s + 1] == ownRank[s]) return;
    const int nGenOwn = ownGenRank[n], nGenAll = genRank[n], g = ownGenRank[s];
    const bool gen = ownGenRank[s + 1] != g;
    const int entry = gen ? g : nGenOwn + (ownRank[s] - g);
    bandOut[entry] = band[s] - c0;
    diagOut[entry] = diagS[s];
    if (gen) {
        const int ga = genRank[s];
#pragma unroll
        for (int q = 0; q < 7; ++q) rowsOut[size_t(q) * nGenOwn + g] = rowsAll[size_t(q) * nGenAll + ga];
    }
}

// The cells of the neighbours' planes the rank's boxes read (closure mode: every list entry): a byte per cell of the `ghost`
// planes below (lo) and above (hi) the owned planes.  info[0] is the region's origin as an offset from owned cell 0 (rebased).
// broken: an entry beyond the ghost planes (cannot happen: a region reaches depth + 1 planes past its owned box)
__global__ __launch_bounds__(256) void haloMarkKernel(int nx, int ny, int nzOwn, int ghost, const int32_t *__restrict__ info, const uint32_t *__restrict__ list,
                                                      uint8_t *__restrict__ lo, uint8_t *__restrict__ hi, int *__restrict__ broken)
{
    const int32_t *gi = info + kBoxInfoInts * size_t(blockIdx.x);
    const ptrdiff_t sy = nx, sz = ptrdiff_t(nx) * ny, origin = gi[0], top = ptrdiff_t(nzOwn) * sz;
    const uint32_t *U = list + gi[2];
    for (int k = threadIdx.x; k < gi[7]; k += 256) {
        const uint32_t e = U[k];
        const ptrdiff_t c = origin + ptrdiff_t(e & 31u) + ptrdiff_t((e >> 5) & 31u) * sy + ptrdiff_t((e >> 10) & 31u) * sz;
        if (c < 0) {
            if (c < -ptrdiff_t(ghost) * sz) *broken = 1;
            else lo[c + ptrdiff_t(ghost) * sz] = 1;
        } else if (c >= top) {
            if (c >= top + ptrdiff_t(ghost) * sz) *broken = 1;
            else hi[c - top] = 1;
        }
    }
}
__global__ __launch_bounds__(256) void addIntKernel(int32_t *__restrict__ out, const int32_t *__restrict__ in, int n, int32_t delta)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n) out[t] = in[t] + delta;
}
__global__ __launch_bounds__(256) void rebaseBoxesKernel(int32_t *__restrict__ info, int n, int32_t delta)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n) info[kBoxInfoInts * size_t(t)] -= delta;
}
// *bad = 1 when an index lies outside [0, limit)
__global__ __launch_bounds__(256) void checkIndexKernel(const int32_t *__restrict__ idx, int n, int32_t limit, int *__restrict__ bad)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n && (idx[t] < 0 || idx[t] >= limit)) *bad = 1;
}
// a byte per block of the plane-marching sweep (256 x kPlaneRows x zc): set where plane k of the grid (a ghost plane: -1 or nz, clamped
// into the first / last layer of blocks) holds an active cell -- the residual + restriction pair of a cut level visits such blocks too
__global__ __launch_bounds__(256) void ghostPlaneBlockFlagsKernel(Dims d, const uint8_t *__restrict__ lab, int k, int zc, int nbx, int nby, uint8_t *__restrict__ flags)
{
    const size_t q = blockIdx.x * size_t(blockDim.x) + threadIdx.x, nq = size_t(d.nx) * d.ny / 4;
    if (q >= nq) return;
    const ptrdiff_t sz = ptrdiff_t(d.nx) * d.ny;
    const uint32_t v = *reinterpret_cast<const uint32_t *>(lab + ptrdiff_t(k) * sz + ptrdiff_t(q) * 4);
    if (!(activeCode(v & 0xffu) || activeCode((v >> 8) & 0xffu) || activeCode((v >> 16) & 0xffu) || activeCode(v >> 24))) return;
    const size_t c = q * 4;
    const int i = int(c % d.nx), j = int(c / d.nx), kb = min(max(k, 0), d.nz - 1) / zc;
    flags[(size_t(kb) * nby + j / kPlaneRows) * nbx + i / 256] = 1;
}

}  // namespace

int launchShellCheckSlab(void *stream, const Dims &d, const uint8_t *lab, bool zFaceLo, bool zFaceHi, int *badFlag)
{
    const size_t n = 2 * (size_t(d.nx) * d.ny + size_t(d.nx) * d.nz + size_t(d.ny) * d.nz);
    shellCheckSlabKernel<<<blocksFor(n, 256), 256, 0, S(stream)>>>(d, lab, zFaceLo ? 1 : 0, zFaceHi ? 1 : 0, badFlag);
    return int(hipGetLastError());
}
int launchOwnedFlags(void *stream, const int32_t *band, const int32_t *general, int n, int32_t c0, int32_t c1, int32_t *own, int32_t *ownGen)
{
    if (n > 0) ownedFlagsKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(band, general, n, c0, c1, own, ownGen);
    return int(hipGetLastError());
}
int launchBandSplitOwned(void *stream, const int32_t *band, int n, int32_t c0, const uint8_t *diagS, const int32_t *ownRank, const int32_t *ownGenRank,
                         const int32_t *genRank, const float *rowsAll, int32_t *bandOut, uint8_t *diagOut, float *rowsOut)
{
    if (n > 0)
        bandSplitOwnedKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(band, n, c0, diagS, ownRank, ownGenRank, genRank, rowsAll, bandOut, diagOut, rowsOut);
    return int(hipGetLastError());
}
int launchHaloMark(void *stream, int nx, int ny, int nzOwn, int ghost, const int32_t *info, const uint32_t *list, int ngroups, uint8_t *lo, uint8_t *hi, int *broken)
{
    if (ngroups > 0) haloMarkKernel<<<unsigned(ngroups), 256, 0, S(stream)>>>(nx, ny, nzOwn, ghost, info, list, lo, hi, broken);
    return int(hipGetLastError());
}
int launchAddInt(void *stream, int32_t *out, const int32_t *in, int n, int32_t delta)
{
    if (n > 0) addIntKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(out, in, n, delta);
    return int(hipGetLastError());
}
int launchRebaseBoxes(void *stream, int32_t *info, int ngroups, int32_t delta)
{
    if (ngroups > 0) rebaseBoxesKernel<<<blocksFor(size_t(ngroups), 256), 256, 0, S(stream)>>>(info, ngroups, delta);
    return int(hipGetLastError());
}
int launchCheckIndex(void *stream, const int32_t *idx, int n, int32_t limit, int *bad)
{
    if (n > 0) checkIndexKernel<<<blocksFor(size_t(n), 256), 256, 0, S(stream)>>>(idx, n, limit, bad);
    return int(hipGetLastError());
}
int launchGhostPlaneBlockFlags(void *stream, const Dims &d, const uint8_t *lab, int k, int zc, uint8_t *flags)
{
    if ((d.nx & 3) != 0 || zc <= 0) return 0;
    const int nbx = (d.nx + 255) / 256, nby = (d.ny + kPlaneRows - 1) / kPlaneRows;
    ghostPlaneBlockFlagsKernel<<<blocksFor(size_t(d.nx) * d.ny / 4, 256), 256, 0, S(stream)>>>(d, lab, k, zc, nbx, nby, flags);
    return int(hipGetLastError());
}

}  // namespace mgps
