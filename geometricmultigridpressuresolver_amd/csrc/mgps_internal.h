// Internal declarations shared by the host-side hierarchy builder (mgps_host.cpp), the HIP kernels
// (mgps_kernels.hip) and the solver / C-ABI layer (mgps_solver.hip).  Not part of the public ABI.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <cstddef>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "mgps.h"

namespace mgps {

constexpr int kTile = 16;  // UT_VoxelArray tile edge: decides the Gauss-Seidel colouring (Ops.h:436-448)

// activity-list granularity, chosen per level: 1024 cells (one workgroup of 256 threads x 4 cells) or, where
// that costs more (liquid that ends mid-row: free surfaces; chooseRunCells weighs the cells a length visits by its measured
// cost per cell), 256 cells (one wavefront; the list is then padded with -1 to whole workgroups of four entries)
constexpr int kChunkCells = 1024;
constexpr int kWaveChunkCells = 256;
// ... or 64 or 32 cells (16 / 8 lanes of a wavefront; the list is padded
// to whole workgroups): a free surface that cuts the x-rows -- the reference's own test domain,
// HDK_TestGeometricMultigrid.cpp:235, has its surface along x -- leaves most of a 256-cell run in the air.  The activity
// flags are kept per kSegCells cells.
constexpr int kSegCells = 32;

// std::vector that leaves trivially constructible elements uninitialised on resize(): the big set-up arrays are
// filled by all host threads right after, and a serial zero-fill of 100+ MB costs more than that fill
// Big blocks come from hostBigAlloc: page-locked memory once the solver layer has installed its allocator (a fresh
// pageable array uploads at ~3 GB/s on this platform, a page-locked one at ~55 GB/s), plain malloc before that and
// wherever no HIP device exists (the hierarchy alone needs none).
void *hostBigAlloc(size_t bytes);
void hostBigFree(void *p);
void setHostBigAllocator(void *(*alloc)(size_t), void (*release)(void *));
constexpr size_t kBigBlockBytes = size_t(1) << 20;

template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    T *allocate(size_t n)
    {
        if (n * sizeof(T) >= kBigBlockBytes) return static_cast<T *>(hostBigAlloc(n * sizeof(T)));
        return std::allocator<T>::allocate(n);
    }
    void deallocate(T *p, size_t n)
    {
        if (n * sizeof(T) >= kBigBlockBytes) hostBigFree(p);
        else std::allocator<T>::deallocate(p, n);
    }
    template <class U, class... Args>
    void construct(U *p, Args &&...args)
    {
        if constexpr (sizeof...(Args) == 0) ::new (static_cast<void *>(p)) U;
        else ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};
template <class T>
using RawVec = std::vector<T, DefaultInitAllocator<T>>;

inline bool isActive(uint8_t l) { return l == MGPS_INTERIOR_CELL || l == MGPS_BOUNDARY_CELL; }

// Device-side cell codes.  The kernels read one byte per cell; INTERIOR / EXTERIOR / DIRICHLET keep
// the reference's label values, BOUNDARY cells are split at set-up into
//   kCodeGeneral      (= MGPS_BOUNDARY_CELL): some face weight is neither 0 nor 1 -- the operator row is
//                     kept in the level's row list and applied by list kernels;
//   kCodeSimple + d   (d = 1..6): every face weight towards an active or DIRICHLET neighbour is exactly
//                     1, so the row is diag * x_c - sum of the six neighbours with diag = d = number of
//                     non-EXTERIOR neighbours.  Inactive neighbours contribute 0 to that sum because
//                     every grid holds exactly 0 outside active cells (the invariant asserted at
//                     Ops.h:821-823, 950-953), so the full-domain sweeps handle these cells in line with
//                     no extra loads.  All BOUNDARY cells of the unit-weight coarse levels are simple.
constexpr uint8_t kCodeGeneral = MGPS_BOUNDARY_CELL;
constexpr uint8_t kCodeSimple = 4;

struct Dims {
    int nx = 0, ny = 0, nz = 0;
    size_t cells() const { return size_t(nx) * ny * nz; }
    size_t idx(int i, int j, int k) const { return (size_t(k) * ny + j) * nx + i; }
};

// One level of the hierarchy, host side.
struct HostLevel {
    Dims d;
    RawVec<uint8_t> labels;
    std::vector<int32_t> band;       // linear indices of the band cells, reference order (tile,k,j,i)
    std::vector<int32_t> bandTileStart;  // hierarchy levels: per 16^3 tile (+1) the first entry of its cells in `band`
    // device order of the same set: the BOUNDARY cells first (each in reference order), then the
    // INTERIOR band cells.  Jacobi on the band is compute-then-scatter, so the order is free.
    RawVec<int32_t> bandDev;
    RawVec<uint8_t> bandDiag;        // diagonal (1..6) of every bandDev entry that is not a general cell
    int32_t numBoundary = 0;         // bandDev[0 .. numBoundary) are the *general* BOUNDARY cells
    // operator rows of the general BOUNDARY cells (Ops.h:208-256 evaluated once at set-up), SoA:
    // rows[q*numBoundary + t], q = 0..5 the off-diagonal weight towards -x,+x,-y,+y,-z,+z (0 when that
    // neighbour is not active), q = 6 the diagonal.
    std::vector<float> rows;
    // slab levels (buildSlabLevel) keep no label copy: views into the hierarchy level they were cut from -- the owned
    // planes and the plane below / above them (nullptr at the ends of the grid: EXTERIOR).  The device cell codes
    // (see kCodeSimple) are these labels with the simple BOUNDARY cells patched in by launchPatchSimpleCodes.
    const uint8_t *ownedLabels = nullptr, *ghostLoLabels = nullptr, *ghostHiLabels = nullptr;
    std::vector<int32_t> bandEntry;  // per entry of `band`: its index in bandDev
    // slab runs: band cells of the four planes a band-only ghost exchange touches, as offsets from
    // owned cell 0 in reference band order (both neighbours derive them from the same global band
    // list, so sender's pack order == receiver's unpack order): [0] owned plane 0 (sent down),
    // [1] ghost plane below (received), [2] owned top plane (sent up), [3] ghost plane above (received)
    std::vector<int32_t> bandPlane[4];
    // Activity lists: the reference skips constant (all-EXTERIOR / all-DIRICHLET) 16^3 tiles of its
    // tile-compressed UT_VoxelArray in every operator (`if (!vit.isTileConstant() || active)`, e.g.
    // Ops.h:322-324); the flat layout gets the same effect from lists of the 256-cell chunks (and, for
    // the plane-marching sweep, of the 256 x 16 x zc blocks) that hold at least one active cell.
    RawVec<int32_t> chunks;  // (page-locked and reused above 1 MB: in a process that holds GPU mappings fresh pages are slow, see runListFromFlags)
    int chunkCells = kChunkCells;
    std::vector<int32_t> planeBlocks;
    int planeZc = 0;
    // cut slab levels: the first edgeChunks / edgePlaneBlocks entries of the two lists touch the planes next to a cut (edgeFirst)
    int32_t edgeChunks = 0, edgePlaneBlocks = 0;
    // 16^3 tiles holding active cells, split by Gauss-Seidel colour ((tx+ty+tz) odd / even) and by
    // kind: "pure" = all 4096 cells INTERIOR (no label or weight look-ups needed), "mixed" = the rest
    std::vector<int32_t> tilesOdd, tilesEven;          // all active tiles of the colour (API / tests)
    std::vector<int32_t> pureOdd, pureEven, mixedOdd, mixedEven;
    std::vector<int32_t> tileBndStart;  // per tile (+1): first entry of its BOUNDARY cells in bandDev
    int64_t activeCells = 0;
};

constexpr int kBandMaxDepth = 4;  // band passes the fused stage takes (options.band_iterations beyond: pass by pass)
// Fused band stage, box form (round 3; on cut slab levels since round 5).  The graph form of rounds 1-2 (removed in round 5) spent 20 B of
// metadata per update node and gathered every value by index (PMC: 131 B per band cell at 1024^3 against 16 algorithmic).  Here a group is a BOX of the
// grid: the owned box O (a piece of a 16^3 tile: the bounding box of the tile's band-closure cells, halved until it fits)
// and the region R around it that its passes touch (inside O dilated by depth + 1, every extent < 32).  The values of R
// live in a dense LDS block -- the neighbours of region cell n are n +- 1, n +- rx, n +- rx*ry, no ids -- and the
// workgroup walks compact lists of the region cells that matter (4 B per update cell, 2 B per read-only cell):
//   ONE list per group, 4 B per entry, in the region's own order (k, j, i) -- so that consecutive lanes stage consecutive
//   cells of a row, like a dense block would, without the cells that need nothing:
//       li | lj << 5 | lk << 10 | class << 16 | ring << 20        (ring = Chebyshev distance from O, at most 7)
//   class 3      general band cell with ring <= depth (operator row: see `general`)
//   class 4 + d  simple band cell with ring <= depth, diagonal d (INTERIOR band cells: 10); pass p recomputes the band
//                cells with ring <= H - p
//   class 11     closure-output cell: active cell of O next to a band cell; the closure mode's last pass gives it its Jacobi value
//   class 1      frozen active cell next to a band cell of ring <= depth - 1: read by both modes
//   class 2      inactive neighbour of a band cell: the value is 0, nothing is loaded
//   class 12     read by the closure mode only: band cells of ring depth + 1, frozen cells next to a ring-depth band cell or
//                to a closure-output cell
//   A thread takes entries tid, tid + 1024, ...: all entries of a group are fetched in one batch of loads, all values in a
//   second one (a kernel of this size lives on its chain of dependent memory round trips: rocprofv3 SQ_WAIT_ANY 66 %).
//   (Tried: the same list sorted by ring, so that a pass walks a prefix -- the staging loads of a wave then hop between
//   rows and the kernel ran 1.5 x slower, 470 vs 312 us per closure stage at 1024^3.)
// Two modes of the kernel (launchBandBox):
//   plain    depth band passes (H = depth); the final values of O's band cells go to `dst`
//   closure  depth band passes + ONE full-domain Jacobi step evaluated on the band closure of O (H = depth + 1): the
//            stage "band passes, then sweep" becomes sweep(x -> y) over the whole grid followed by this kernel
//            overwriting y on the closure (band cells and their active face neighbours) -- nothing is scattered back
//            into x -- and the same values land in a snapshot grid from which the band stage AFTER the sweep reads
//            (that one then writes y in place: no workgroup reads what another one writes).
constexpr int kBoxMaxNodes = 8192;    // region cells of a group: two LDS copies of their values (64 KB: two workgroups per CU)
constexpr int kBoxMaxList = 4096;     // ... of which at most this many matter (list entries: kBoxSlots per thread, in registers)
constexpr int kBoxThreads = 1024;
constexpr int kBoxSlots = kBoxMaxList / kBoxThreads;
constexpr int kBoxMaxGeneral = 384;   // general band cells of a region (their rows sit in LDS: 13.5 KB)
constexpr int kBoxInfoInts = 16;
enum BoxNode : uint8_t { kBoxSkip = 0, kBoxFrozen = 1, kBoxZero = 2, kBoxGeneral = 3, kBoxSimple = 4, kBoxFrozenOut = 11, kBoxFrozenFar = 12 };
// info: [0] linear cell of the region's origin, [1] rx | ry << 8 | rz << 16, [2] first entry in `list`, [3] (device, after
// compactBandBoxLists) the entries the plain mode walks,
// [4] first entry in `general`, [5] general entries, [6] unused, [7] list entries, [8 + r] band cells with ring <= r
// (r = 0..4; r > depth repeats), [13] closure-output cells, [14] origin of O inside the region (packed like [1]),
// [15] extents of O (packed)
struct BandBoxes {
    int depth = 0;
    RawVec<int32_t> info;
    RawVec<uint32_t> list;
    RawVec<int32_t> general;   // per general band cell with ring <= depth two ints: its list entry (coordinates, class, ring), row index
    size_t groups() const { return info.size() / kBoxInfoInts; }
};
void buildBandBoxes(const HostLevel &L, int depth, BandBoxes &out);
struct BandBoxesDev {
    int depth = 0, ngroups = 0;
    int32_t *info = nullptr, *general = nullptr;
    uint32_t *list = nullptr;
    size_t listCount = 0, generalInts = 0;
    bool anyGeneral = false;
};
// the box kernels address a region cell by a 32-bit byte offset from the region's origin (31 planes at most) formed with 24-bit
// multiplies: a level whose x-y planes hold 2^24 cells or more (4096 x 4096) keeps the pass-by-pass band smoother
inline bool boxPlaneFits(const Dims &d) { return size_t(d.nx) * size_t(d.ny) < (size_t(1) << 24); }
// the region lists without the class-2 entries (the kernel clears its LDS block) and with the class-12 entries at the end of
// every group's list (info[3] = what the plain mode walks, info[7] = all, info[2] = new offsets); listOut: listCount entries;
// synchronises the stream
int compactBandBoxLists(void *stream, int32_t *info, const uint32_t *list, int ngroups, uint32_t *listOut, size_t *newCount);
// the groups in the Morton order of their tiles (launch order = L2 locality of the overlapping regions): info permuted into
// infoOut; synchronises the stream
int orderBandBoxes(void *stream, const Dims &d, const int32_t *info, int ngroups, int32_t *infoOut);

// rowsIn (optional, instead of wx / wy / wz): the operator rows of the BOUNDARY band cells of the planes
// [z0, z1) evaluated elsewhere (on the device, mgps_create_device_weights), 8 floats per cell in band order:
// the six off-diagonal weights, the diagonal, 1 / 0 for simple / general
void buildSlabLevel(const HostLevel &G, int z0, int z1, const float *wx, const float *wy, const float *wz,
                    HostLevel &L, const float *rowsIn = nullptr);
// device evaluation of those rows (Ops.h:208-256) for `n` cells of a grid of extents d: labels and weights on
// the device; rows = 8 floats per cell as above; *violations counts BOUNDARY cells that break the rule of
// unitTestBoundaryCells (Ops.h:1771-1870: some neighbour inactive, or a BOUNDARY neighbour across a face of weight != 1)
int launchBoundaryRows(void *stream, const Dims &d, const uint8_t *labels, const float *wx, const float *wy, const float *wz,
                       const int32_t *cells, int n, float *rows, int *violations);
// the label-only half of unitTestBoundaryCells (Ops.h:1771-1870): every INTERIOR cell has six active neighbours
void checkInteriorCells(const uint8_t *labels, int nx, int ny, int nz, int *pass);
int hierarchyCreate(mgps_hierarchy **out, int nx, int ny, int nz, const uint8_t *labels, int mg_levels,
                    const mgps_options *opt, bool forceCoarseSolver, bool requireShell, const int *window = nullptr);
// pieces of the host builder the device-side set-up shares: the activity list of a level from the flags of its
// runs of 64 cells, and the Gauss-Seidel tile lists from the per-tile kinds ((active cells << 1) | all INTERIOR)
void chunkListsFromFlags(HostLevel &L, const uint8_t *segAct, int64_t nseg);
constexpr int kRunSizes[4] = {kChunkCells, kWaveChunkCells, 64, kSegCells};  // the run lengths a level's list can have
int chooseRunCells(const int64_t nAct[4]);                                    // from the active runs of each length
// cost per visited cell of a sweep over runs of that length, relative to 1024-cell runs (measured: see chooseRunCells)
inline double runCostFactor(int cells) { return cells >= kChunkCells ? 1.0 : cells >= kWaveChunkCells ? 1.02 : cells >= 64 ? 1.13 : 1.22; }
void runListFromFlags(HostLevel &L, const uint8_t *runAct, int64_t nq, int runCells);
void tileListsFromKinds(HostLevel &L, const int64_t *kind, int tileZOffset);
void tileListsFromKinds(HostLevel &L, const int32_t *kind, int tileZOffset);
// A hierarchy that knows the extents of its levels and the labels of the coarsest one only (device-side set-up: the
// labels of the other levels live on the device) -- what the coarsest level's direct solver needs (MG.cpp:288-411).
int hierarchyLight(mgps_hierarchy **out, int nx, int ny, int nz, int levels, const uint8_t *coarsestLabels, const mgps_options &o,
                   bool needCoarseSolver);
void setLastGlobalError(const std::string &msg);
const char *lastGlobalError();
// No C++ exception crosses the C ABI: every extern "C" entry point that can allocate is a function-try-block ending in
// MGPS_API_CATCH(handle or nullptr).  apiException classifies the exception in flight (std::bad_alloc -> MGPS_ERR_ALLOC,
// anything else -> MGPS_ERR_INTERNAL) and leaves the text in the handle's (or the global) last-error slot.
int apiException(const mgps_solver *h) noexcept;
// how apiException stores a text in a handle: installed by the solver layer (the struct is defined in mgps_solver.hip;
// the host-only half of the library links without it)
extern void (*gSetHandleError)(const mgps_solver *h, const char *msg) noexcept;
#define MGPS_API_CATCH(h) catch (...) { return ::mgps::apiException(h); }

// ---- device side ------------------------------------------------------------------------------
// Read-only description of one level handed to the kernels.
struct GridP {
    int nx, ny, nz;
    const uint8_t *lab;
    const float *wx, *wy, *wz;  // fine level only; nullptr = unit weights (MG.cpp:572-575)
    // general BOUNDARY cells: cell list + 7 x nbnd SoA coefficients (see HostLevel::rows)
    const int32_t *bnd;
    const float *rows;
    int nbnd;
    const uint8_t *bandDiag;  // per band-list entry: diagonal of the simple / INTERIOR cells
    // Z-slab of a multi-GPU run: nz counts the planes this rank owns; when a flag is set the plane
    // just below (k = -1) / above (k = nz) the owned range is a ghost plane held in the same
    // allocation (every array pointer addresses owned plane 0), filled by the neighbour exchange.
    int ghostLo, ghostHi;
    // active 256-cell chunks of the flat array (one per wavefront, -1 = padding) / active blocks of the plane-marching sweep
    const int32_t *chunks;
    int nchunks, chunkCells;
    const int32_t *planeBlocks;
    int nplaneBlocks, planeZc;
    // the level's grids do not fit the 256 MiB Infinity Cache: sweeps use nontemporal loads for what they read once
    // (rhs, codes) and nontemporal stores for their output; smaller levels leave everything cacheable (the next
    // kernel finds it there)
    int streaming;
    // which sweep kernel launchStencil takes (options.stencil_path): 0 = by size, 1 = quad, 2 = plane-marching where it applies
    int sweepPath;
    // x extent of the active cells, quad-aligned: [xlo, xhi) holds every active cell of the level.  The reference's power-of-two
    // expansion pads every row with 2^(L-1) EXTERIOR cells per side (an eighth of a 1024-cell row at L = 7): activity-skipping
    // sweeps leave the quads outside the range alone -- nothing loaded, nothing stored, like the runs / blocks without active cells
    int xlo, xhi;
};
// xlo / xhi of a level from its cell codes (range[0] = min x, range[1] = max x over the active cells; range preset to {nx, -1})
int launchActiveXRange(void *stream, const uint8_t *lab, int nx, size_t cells, int *range);

// Scales of the mixed-precision V-cycle (options.precision = 1).  The cycle works on the rhs normalised by a power of
// two, sigma (device scalar: 2^-ceil(log2 max|b|), so that max |sigma b| is in (1/2, 1]); the fine-level iterate is stored
// in binary16 as x~ = x^ 2^-e (e from the level's size: |A^-1| can reach N^2 / 2) and the fine residual as r~ = 256 r^.
// A kernel sees the rhs as (*sigma * c1) * b and the operator term as c2 * (A x~).
struct MixScale {
    const float *sigma = nullptr;
    float c1 = 1.f, c2 = 1.f;
};

constexpr int kPlaneRows = 16;  // y extent of a plane-marching block (x extent 256)
constexpr size_t kPlaneSweepMinPlaneBytes = size_t(4) << 20;  // levels whose x-y planes are larger take the plane-marching sweep by size (launchStencil)
constexpr int kSlabEdgePlanes = kBandMaxDepth + 1;  // planes at either end of a slab whose sweep goes first: the band closure a stage message carries reaches band_iterations planes in, its face neighbours one more
// does the plane-marching sweep apply to a level of this shape, and with how many planes per block
inline int planeSweepZc(int nx, int ny, int nz)
{
    if ((nx & 3) != 0 || nx < 256 || ny < kPlaneRows) return 0;
    const size_t nbx = (nx + 255) / 256, nby = (ny + kPlaneRows - 1) / kPlaneRows;
    int zc = 32;  // fewer planes per workgroup on small grids so that the launch still fills 256 CUs
    while (zc > 4 && nbx * nby * size_t((nz + zc - 1) / zc) < 1024) zc >>= 1;
    return zc;
}

enum StencilOp { OP_JACOBI = 0, OP_RESIDUAL = 1, OP_APPLY = 2 };

// All launchers enqueue on `stream` (a hipStream_t passed as void*) and return a hipError_t as int.
// skipInactive: leave chunks without active cells untouched (callers whose `out` already holds the
// right values there: 0 for r / y, the unchanged iterate for Jacobi)
size_t stencilSweptCells(const GridP &g);
// which kernel launchStencil takes on this level: 1 = stencilQuadKernel, 2 = stencilPlaneKernel, 3 = stencilScalarKernel
int stencilKernelOf(const GridP &g);
int forcedStencilPath();  // MGPS_STENCIL=quad|plane (A/B switch of launchStencil): 0 none, 1 quad, 2 plane
bool setupTimingOn();     // MGPS_SETUP_TIMING=1: stage times of the set-up on stdout / stderr (mgps_host.cpp)
int launchStencil(void *stream, StencilOp op, const GridP &g, float *out, const float *x, const float *b, float omega,
                  bool skipInactive);
// out = A x and *resultDev = <x, A x> over the active cells of level g in one pass (the CG loop's A.p and its dot);
// `partials` holds applyDotPartialCount(g) doubles
size_t applyDotPartialCount(const GridP &g);
int launchApplyDot(void *stream, const GridP &g, float *out, const float *x, double *partials, double *resultDev);
// band = device-ordered band list (BOUNDARY cells first, g.nbnd of them)
// dotPartials (optional, here and below): the scatter leaves per-workgroup sums of (new - old) * b in
// bandScatterBlocks(nband) slots (see launchStencilDot)
int launchBandJacobi(void *stream, const GridP &g, float *x, const float *b, const int32_t *band, int nband,
                     float *bandTmp, float omega, double *dotPartials = nullptr);
unsigned bandScatterBlocks(int nband);
int launchStencilDot(void *stream, StencilOp op, const GridP &g, float *out, const float *x, const float *b, float omega,
                     double *partials, unsigned *nparts);
int launchFoldDot(void *stream, double *partials, unsigned nparts, double *resultDev);
// Box form of the fused band stage (BandBoxes).  src: where the region's values are read; dst: where the results go (the
// band cells of every owned box; closure mode: every closure cell of it), snap (closure mode, optional): a second copy.
// Grids are float, or binary16 when `half` is set (mixed precision: ms as in launchBandFusedMixed).
// dotPartials (optional): one slot per group receives sum (new - old) * b over the group's output cells, `old` read from
// dotOld (closure mode: the sweep's value in dst before it is overwritten; plain mode: the stage's input).
// Closure mode with dst == nullptr fills the snapshot only; the plain launch after the sweep, with outClosure, then writes the
// closure-output cells from the snapshot as well as the band cells -- the closure launch needs nothing of the sweep's output and
// can run beside the sweep (smoothStroke).
int launchBandBox(void *stream, const GridP &g, const BandBoxesDev &bx, bool closure, const void *src, const float *b, void *dst, void *snap,
                  float omega, bool half = false, const MixScale &ms = MixScale{}, double *dotPartials = nullptr, const void *dotOld = nullptr,
                  bool outClosure = false);
// dst = src on the band cells of every owned box (the legacy form of the plain stage: out of place into a scratch grid,
// then this copy -- used where no snapshot of the input exists)
// The closure launch and the sweep of a stroke as one launch (levels that take the quad sweep; x == nullptr: the zero iterate): the
// sweep leaves the cells whose bit is set in `keep` alone (launchMarkClosure: the owned band / closure-output cells of the boxes, one
// bit per cell, cells / 32 words zeroed by the caller), which the plain launch writes afterwards
int launchStrokeFront(void *stream, const GridP &g, const BandBoxesDev &bx, float *out, const float *x, const float *b, float *snap, float omega, const uint32_t *keep);
int launchMarkClosure(void *stream, const GridP &g, const BandBoxesDev &bx, uint32_t *bits);
int launchBandBoxCopy(void *stream, const GridP &g, const BandBoxesDev &bx, const void *src, void *dst, bool half = false);
// one side of the list part of a cut level's band-stage message (box form): n cells at base + idx[t] (offsets from owned cell 0);
// buf: 2 n floats.  buf == nullptr: no neighbour on that side
struct HaloList {
    float *buf = nullptr;
    const int32_t *idx = nullptr;
    int n = 0;
    ptrdiff_t base = 0;
};
int launchHaloListPack(void *stream, const HaloList &lo, const HaloList &hi, const float *a0, const float *a1);      // a1 may be nullptr (one array)
int launchHaloListUnpack(void *stream, const HaloList &lo, const HaloList &hi, float *a0, float *a1);
// pure = tiles whose 4096 cells are all INTERIOR; mixed = every other tile with active cells
// snap / snapTile (optional): the tiles flagged in snapTile (launchMarkSnapTiles) leave a second copy of their result in `snap`
int launchTiledGS(void *stream, const GridP &g, float *x, const float *b, const int32_t *pureTiles, int npure,
                  const int32_t *mixedTiles, int nmixed, const int32_t *tileBndStart, int forward, double *dotPartials = nullptr,
                  float *snap = nullptr, const uint8_t *snapTile = nullptr);
// a byte per 16^3 tile (zeroed by the caller): set where some box group of the fused band stage reads a cell of the tile
int launchMarkSnapTiles(void *stream, const GridP &g, const BandBoxesDev &bx, uint8_t *tiles);
int launchTiledGSMixed(void *stream, const GridP &g, void *xH, const float *b, const int32_t *pureTiles, int npure, const int32_t *mixedTiles, int nmixed,
                       const int32_t *tileBndStart, int forward, const MixScale &ms);
// codes[band[t]] = kCodeSimple + bandDiag[t] for the BOUNDARY cells among the entries t >= nbnd (the simple ones)
int launchPatchSimpleCodes(void *stream, uint8_t *codes, const int32_t *band, const uint8_t *bandDiag, int nbnd, int nband);
int launchRestrict(void *stream, const GridP &coarse, float *coarseOut, const float *fine);
// residual of `fine` (no ghost planes, plane blocks) restricted to `coarse` without writing the residual: launchResidualZ folds
// r = b - A x along z into rz (fine.nx x fine.ny x fine.nz / 2 floats that nothing else writes: zero wherever the launches do not
// go), general BOUNDARY cells by a patch in four conflict-free launches; launchRestrictXY does the x-y half.  edges (device, may
// be empty): the blocks off the activity list that lie below / above a listed one, as planeBlockEdges makes them from a host copy
// of the block flags (launchPlaneBlockFlags) -- they owe rz one plane of terms each.  residualRestrictFits: the shapes the pair takes
bool residualRestrictFits(const GridP &fine, const GridP &coarse);
std::vector<int32_t> planeBlockEdges(const GridP &g, const std::vector<uint8_t> &flags);
// rEdge (cut levels): the level's residual grid, whose ghost planes hold the neighbours' r on the planes next to the slab
int launchResidualZ(void *stream, const GridP &fine, float *rz, const float *x, const float *b, const int32_t *edges, int nedges, const float *rEdge = nullptr);
int launchResidualEdgePlanes(void *stream, const GridP &g, float *r, const float *x, const float *b);
int launchRestrictXY(void *stream, const GridP &coarse, float *coarseOut, const float *rz);
// ---- mixed precision (options.precision = 1): binary16 grids of the fine level, passed as void* -----------------------
bool mixedPrecisionShapeOk(int nx, int ny, int nz);  // the fine level must take the quad sweep and the block prolongation
int launchStencilMixed(void *stream, StencilOp op, const GridP &g, void *outH, const void *xH, const float *b, float omega, const MixScale &ms,
                       double *dotPartials = nullptr, unsigned *nparts = nullptr);
int launchScaleResult(void *stream, double *resultDev, const float *sigmaDev, float mul);  // *result *= mul / *sigma
int launchRestrictMixed(void *stream, const GridP &coarse, float *coarseOut, const void *fineH, float fm);
int launchProlongAddMixed(void *stream, const GridP &fine, void *fineH, const float *coarse, float pm);
int launchFromHalf(void *stream, float *dst, const void *srcH, const float *sigmaDev, float mul, size_t cells);
int launchMixSigma(void *stream, const double *maxAbsDev, float *sigmaDev);
int launchZeroActiveHalf(void *stream, const GridP &g, void *aH);
int launchProlongAdd(void *stream, const GridP &fine, float *fineInOut, const float *coarse, float *snap = nullptr, const uint8_t *snapTile = nullptr);
size_t planeBlockCount(const GridP &g);
int launchPlaneBlockFlags(void *stream, const GridP &g, uint8_t *flags);
// dense coarsest matrix (n x n doubles, zeroed by the caller) from the level's labels; fp32 inverse from the triangle potri left
int launchCoarseAssemble(void *stream, int n, int nx, int ny, const int32_t *cells, const int32_t *index, const uint8_t *lab, double *A);
int launchCoarseNarrow(void *stream, int n, const double *A, float *inv);
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
// x += alpha p, r -= alpha t, *resultDev = sum of the new r^2 over active cells (one pass, CG.h:132-153)
// alphaDev (optional): alpha = float(alphaDev[0] / alphaDev[1]) read on the device instead of the host's value
// maxAbsDev (optional; `partials` then holds 2 x kReducePartials doubles): *maxAbsDev = max |r| of the new residual
// xWide (optional): the iterate in fp64 (options.pcg_fp64_vectors = 2) -- updated instead of x, with alpha in double
// xFirst (fp32 x): x = alpha p instead of x += alpha p -- x is the sum of the updates since the last flush of the fp64-iterate loop
int launchCgUpdate(void *stream, const GridP &g, float *x, const float *p, float *r, const float *t, float alpha, double *partials,
                   double *resultDev, const double *alphaDev = nullptr, double *maxAbsDev = nullptr, double *xWide = nullptr, bool xFirst = false);
// CG steps that read the mixed-precision V-cycle's binary16 result in place, z = (mul / *sigma) x~ (level g = the fine level;
// the grid's dimensions need n % 4 == 0): *resultDev = <z, r>; p = z + beta p
int launchHalfDot(void *stream, const GridP &g, const void *xH, const float *r, const float *sigmaDev, float mul, double *partials, double *resultDev);
int launchXpayHalf(void *stream, const GridP &g, float *p, const void *xH, const float *sigmaDev, float mul, const float *betaDev, float betaHost);
int launchCgScalars(void *stream, double *scal, float *beta, int init);
// fp64 CG vectors (options.pcg_fp64_vectors), level g = the fine level of a single-device solver:
// mode 0: out = A x, *resultDev = <x, A x>; mode 1: out = b - A x, out32 = float(out), *resultDev = |out|^2
// (`partials` holds `capacity` doubles: the per-workgroup sums and launchFoldDot's scratch)
// mode 2 (dx != nullptr, out != x): the iterate is x + dx (fp32 sum of the updates since the last flush): out32 = float(b - A (x + dx)),
// *resultDev its squared norm, out = x + dx on the active cells (x elsewhere)
int launchStencil64(void *stream, int mode, const GridP &g, double *out, const double *x, const float *b, float *out32,
                    double *partials, size_t capacity, double *resultDev, const float *dx = nullptr);
int launchNarrowSum(void *stream, const GridP &g, float *x32, const double *x64, bool addDx);  // x32 = float(x64 (+ x32 on active cells))
int launchCgUpdate64(void *stream, const GridP &g, double *x, const double *p, double *r, const double *t, double alpha, float *r32,
                     double *partials, size_t capacity, double *resultDev);
int launchXpay64(void *stream, const GridP &g, double *p, const float *z, double beta, int first);
int launchWiden(void *stream, double *dst, const float *src, size_t n);
int launchNarrow(void *stream, float *dst, const double *src, size_t n);
// device memory with a per-device cache of released blocks (mgps_solver.hip); both return a hipError_t as int;
// deviceFree does not synchronise
int deviceAlloc(void **p, size_t bytes);
int deviceFree(void *p);
void deviceTrim();
// ---- device-side set-up (mgps_setup.hip): see the kernels there ------------------------------------------------------
// Slab runs (round 5): a rank builds a distributed level on a BUFFER of labels -- its owned planes [own0, own1) of the buffer and
// label ghost planes on both sides -- with the whole-grid kernels; `need` = the planes beyond the owned range in which band masks
// are still wanted (the tiles farther out are skipped).  nullptr = a whole grid.
struct SlabWindow {
    int own0 = 0, own1 = 0, need = 0;
};
// the face weights of a level: whole-grid arrays, or (gw > 0) a slab's arrays -- plane k0 of the label buffer is their plane 0 -- with
// gw planes of the neighbours' weights below (lo) and above (hi); farther out the weights read as 1
struct WeightView {
    const float *w[3] = {nullptr, nullptr, nullptr}, *lo[3] = {nullptr, nullptr, nullptr}, *hi[3] = {nullptr, nullptr, nullptr};
    int nz = 0, gw = 0, k0 = 0;
};
int launchCoarsenLabels(void *stream, const Dims &fine, const uint8_t *fineLab, uint8_t *coarseLab, int *activeFlag);
int launchAnyActive(void *stream, const Dims &d, const uint8_t *lab, int *activeFlag);
int launchShellCheck(void *stream, const Dims &d, const uint8_t *lab, int *badFlag);
int launchMarkBoundary(void *stream, const Dims &d, uint8_t *lab);
size_t scanScratchInts(size_t n);
int launchExclusiveScan(void *stream, const int32_t *in, int32_t *out, size_t n, int32_t *scratch);  // out: n + 1 entries
int launchBandCandidates(void *stream, const Dims &d, const uint8_t *lab, int32_t *tileKind, uint8_t *tileBits, int32_t *flags, int32_t *rank,
                         int32_t *list, int32_t *scanScratch, const SlabWindow *win = nullptr);  // the tiles that can hold band cells (or need the INTERIOR check)
int launchBandMasks(void *stream, const Dims &d, const uint8_t *lab, int width, uint32_t *mask, uint16_t *prefix, int32_t *tileCount, int32_t *tileKind,
                    int *interiorBad, const int32_t *tiles, int ntiles, const SlabWindow *win = nullptr);  // interiorBad (optional): set when an INTERIOR cell (of the owned planes) has an inactive neighbour
int launchBandFill(void *stream, const Dims &d, const uint32_t *mask, const uint16_t *prefix, const int32_t *tileStart, int32_t *band);
int launchBandClassify(void *stream, const Dims &d, const uint8_t *lab, const float *wx, const float *wy, const float *wz, const int32_t *band, int n,
                       uint8_t *diagS, int32_t *general, int *violations);
int launchBandSplit(void *stream, const Dims &d, const uint8_t *lab, const float *wx, const float *wy, const float *wz, const int32_t *band, int n,
                    const uint8_t *diagS, const int32_t *genRank, int32_t *bandDev, uint8_t *bandDiag, int32_t *bandEntry, float *rows);
// the same two over a slab's label buffer (violations: counted on the owned planes only)
int launchBandClassify(void *stream, const Dims &d, const uint8_t *lab, const WeightView &wv, const int32_t *band, int n, uint8_t *diagS, int32_t *general,
                       int *violations, const SlabWindow *win);
int launchBandSplit(void *stream, const Dims &d, const uint8_t *lab, const WeightView &wv, const int32_t *band, int n, const uint8_t *diagS, const int32_t *genRank,
                    int32_t *bandDev, uint8_t *bandDiag, int32_t *bandEntry, float *rows);
// slab windows: see the end of mgps_setup.hip
int launchShellCheckSlab(void *stream, const Dims &d, const uint8_t *lab, bool zFaceLo, bool zFaceHi, int *badFlag);
int launchOwnedFlags(void *stream, const int32_t *band, const int32_t *general, int n, int32_t c0, int32_t c1, int32_t *own, int32_t *ownGen);
int launchBandSplitOwned(void *stream, const int32_t *band, int n, int32_t c0, const uint8_t *diagS, const int32_t *ownRank, const int32_t *ownGenRank,
                         const int32_t *genRank, const float *rowsAll, int32_t *bandOut, uint8_t *diagOut, float *rowsOut);
int launchHaloMark(void *stream, int nx, int ny, int nzOwn, int ghost, const int32_t *info, const uint32_t *list, int ngroups, uint8_t *lo, uint8_t *hi, int *broken);
int launchAddInt(void *stream, int32_t *out, const int32_t *in, int n, int32_t delta);
int launchRebaseBoxes(void *stream, int32_t *info, int ngroups, int32_t delta);  // info[0] -= delta for every group
int launchCheckIndex(void *stream, const int32_t *idx, int n, int32_t limit, int *bad);
int launchGhostPlaneBlockFlags(void *stream, const Dims &d, const uint8_t *lab, int k, int zc, uint8_t *flags);
int launchGather(void *stream, const int32_t *rank, const int32_t *start, int n, int32_t *out);
int launchActivityFlags(void *stream, const Dims &d, const uint8_t *lab, uint8_t *chunkFlags, uint8_t *planeFlags, int zc);
// counts[z] += the runs of kRunSizes[z] cells that hold an active cell (counts zeroed by the caller); then the flags folded
// to runs of runCells cells
int launchCountRuns(void *stream, const uint8_t *segFlags, size_t nseg, int *counts);
int launchFoldRunFlags(void *stream, const uint8_t *segFlags, size_t nseg, int runCells, uint8_t *runFlags);
// the activity list itself (launch order of runListFromFlags), listLen = active runs rounded up to whole workgroups
int launchRunList(void *stream, const Dims &d, const uint8_t *runFlags, size_t nq, int runCells, int32_t *tmpFlags, int32_t *rank, int32_t *scanScratch,
                  int32_t *base, int32_t *list, int listLen);
int launchBandTileList(void *stream, const int32_t *tileStart, int nt, int32_t *flags, int32_t *rank, int32_t *list, int32_t *scanScratch);
// one Gauss-Seidel tile list (colour parity `odd`, pure or mixed tiles) from the per-tile kinds; the active plane blocks
// from their flags: list entries ascending, rank[n] = the count
int launchTileClassList(void *stream, const Dims &d, const int32_t *kind, int odd, int mixed, int32_t *flags, int32_t *rank, int32_t *list,
                        int32_t *scanScratch, int tkOffset = 0);  // tkOffset (slab windows): global tile plane of the grid's first one (tile colours)
int launchByteList(void *stream, const uint8_t *bytes, int n, int32_t *flags, int32_t *rank, int32_t *list, int32_t *scanScratch);
// the boxes of the fused band stage (BandBoxes), built over the list of tiles that can hold a band-closure cell; counts
// and offsets are indexed by list position
int launchBoxTileList(void *stream, const Dims &d, const int32_t *tileStart, int32_t *flags, int32_t *rank, int32_t *list, int32_t *scanScratch);
// counts / at: three arrays each (groups, list entries, general entries)
int launchBandBoxesCount(void *stream, const Dims &d, const uint8_t *lab, const uint32_t *mask, const uint16_t *prefix, const int32_t *tileStart,
                         const int32_t *bandEntry, const uint8_t *bandDiag, int depth, const int32_t *tiles, int ntiles, int32_t *const counts[3], int *broken,
                         const SlabWindow *win = nullptr);
int launchBandBoxesFill(void *stream, const Dims &d, const uint8_t *lab, const uint32_t *mask, const uint16_t *prefix, const int32_t *tileStart,
                        const int32_t *bandEntry, const uint8_t *bandDiag, int depth, const int32_t *tiles, int ntiles, const int32_t *const at[3],
                        int32_t *info, uint32_t *list, int32_t *general, int *broken, const SlabWindow *win = nullptr);
int launchZero(void *stream, float *a, size_t count);
int launchZeroInactive(void *stream, const GridP &g, float *a);  // a = 0 on the cells of level g that are not active
// the same for a grid of level g whose chunks without active cells already hold 0 (solver-owned grids)
// ghostPlanes: also the plane below and the plane above the grid (slab runs)
int launchZeroActive(void *stream, const GridP &g, float *a, bool ghostPlanes = false);
// buf[t] = a[idx[t]] / a[idx[t]] = buf[t]; idx are offsets from owned cell 0 (negative in the lower ghost plane)
int launchPack(void *stream, float *buf, const float *a, const int32_t *idx, int n);
int launchUnpack(void *stream, float *a, const float *buf, const int32_t *idx, int n);

}  // namespace mgps

namespace mgps {
constexpr int kHostCoarseMax = 8192;
constexpr double kHostFactorFlops = 4e9;  // n x bandwidth^2 of the host's banded Cholesky factor (buildCoarseSolver): beyond it the device factorises
}

struct mgps_hierarchy;
namespace mgps {
int hostCoarseFallback(mgps_hierarchy *H);  // coarseOnDevice -> a host factor after all (<= kHostCoarseMax unknowns; mgps_host.cpp)
}

// Host-only hierarchy (C-ABI opaque type).
struct mgps_hierarchy {
    int levels = 0;
    int bandWidth = 3;
    bool light = false;  // hierarchyLight: only the extents of the levels and the labels of the coarsest one
    bool windowed = false;  // slab runs: band lists of the rank's z-window only, no coarsest-level factor
    std::vector<mgps::HostLevel> lv;
    // coarsest-level direct solver
    int coarseN = 0;
    int coarseBW = 0;
    std::vector<int32_t> coarseCell;    // unknown id -> linear cell of the coarsest grid
    std::vector<int32_t> coarseIndex;   // linear cell -> unknown id or -1
    std::vector<double> coarseL;        // banded Cholesky factor, coarseN x (coarseBW+1)
    std::vector<float> coarseInverse;   // dense coarseN x coarseN inverse (built on demand for the GPU)
    // more unknowns than kHostCoarseMax or a banded factor past kHostFactorFlops: no factor on the host (the reference's tile
    // numbering gives the banded factor a width of thousands there); the solver factorises and inverts the dense matrix on the device (hipSOLVER potrf / potri in fp64)
    bool coarseOnDevice = false;
    void bandedSolve(double *v) const;
    void buildDenseInverse();
};
