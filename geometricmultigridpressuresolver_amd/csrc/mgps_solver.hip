// The device-resident solver object and the C ABI (include/mgps.h): level storage in HBM, the
// V-cycle schedule of GeometricMultigridPoissonSolver::applyVCycle (MG.cpp:420-881), the PCG driver
// of solveGeometricConjugateGradient (CG.h:18-207), and the Z-slab multi-GPU orchestration (one ghost
// exchange per band stage and per whole-grid operator that reads across a cut, collapse of the coarse
// tail to rank 0).  Host orchestration only -- every arithmetic step is a HIP kernel from
// mgps_kernels.hip; there is no CPU fallback: without a HIP device the constructors fail with
// MGPS_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <atomic>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <map>
#include <memory>
#include <vector>

#include "mgps_internal.h"

using namespace mgps;

namespace {
// the library's device memory (deviceAlloc / deviceFree below: released blocks are cached) behind hipMalloc's signature
inline hipError_t cacheMalloc(void **p, size_t bytes) { return hipError_t(mgps::deviceAlloc(p, bytes)); }
inline hipError_t cacheFree(void *p) { return hipError_t(mgps::deviceFree(p)); }
}  // namespace

namespace {

// One level on the device.  Every grid of a level is allocated with one spare plane below and one
// above the owned planes (the ghost planes of a slab run; unused otherwise) and addressed through
// the pointer to owned plane 0.
struct DevLevel {
    Dims d;  // owned planes
    GridP g{};
    uint8_t *codes = nullptr;  // allocation base (ghost plane first); g.lab = codes + plane
    float *x = nullptr, *b = nullptr;  // levels > 0 only; level 0 works on the caller's grids
    float *r = nullptr, *tmp = nullptr;
    int32_t *band = nullptr;  // device-ordered band list: general BOUNDARY cells first
    int nband = 0;
    int nbndGeneral = 0;  // general BOUNDARY cells: the first entries of band
    float *bandTmp = nullptr;
    float *rows = nullptr;  // 7 x numBoundary operator rows of the general BOUNDARY cells
    uint8_t *bandDiag = nullptr;
    // Gauss-Seidel tile lists per colour: [0] = even, [1] = odd tiles
    int32_t *pure[2] = {nullptr, nullptr}, *mixed[2] = {nullptr, nullptr};
    int npure[2] = {0, 0}, nmixed[2] = {0, 0};
    int32_t *tileBndStart = nullptr;
    int z0 = 0, z1 = 0;  // owned global plane range
    // band-only ghost exchange (slab runs): index lists + staging buffers, see HostLevel::bandPlane
    int32_t *bandPlane[4] = {nullptr, nullptr, nullptr, nullptr};
    int nbandPlane[4] = {0, 0, 0, 0};
    float *packBuf[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t *chunks = nullptr, *planeBlocks = nullptr;  // activity lists
    int edgeChunks = 0, edgePlaneBlocks = 0;            // cut slab levels: the leading entries that touch the planes next to a cut
    BandBoxesDev bandBoxes;  // fused band stage, box form (levels that are not cut into slabs)
    uint8_t *planeFlags = nullptr;  // a byte per block of the plane-marching sweep: on its activity list or not
    int32_t *rzEdges = nullptr;     // launchResidualZ: the blocks without active cells below / above a block with some (planeBlockEdges), made with rz
    int nrzEdges = 0;
    float *rz = nullptr;            // the residual folded along z (residualRestrictFuses; nx x ny x nz / 2, made on first use, zero where nothing writes)
    uint8_t *snapTile = nullptr;    // Gauss-Seidel strokes: a byte per 16^3 tile, set where a box group reads (launchMarkSnapTiles; made on first use)
    uint32_t *keepBits = nullptr;   // launchStrokeFront: one bit per cell, the owned band / closure-output cells of the boxes (made on first use)
    // A cut level of a slab run built on the device (round 5): the box form of the band stage with the neighbours' cells its
    // regions read kept in the deep ghost planes of the grids (mgps_solver::ghost planes on either side of every grid).  recvIdx:
    // those cells, offsets into the ghost planes below [0] / above [1] the owned planes (ascending); sendIdx: the cells of this
    // rank's planes the lower [0] / upper [1] neighbour asked for at set-up (offsets from owned cell 0).  A stage message is the
    // boundary plane straight from the grid plus the packed lists (x and rhs: closure launch; snapshot: plain launch).
    struct BoxHalo {
        int32_t *sendIdx[2] = {nullptr, nullptr}, *recvIdx[2] = {nullptr, nullptr};
        int nsend[2] = {0, 0}, nrecv[2] = {0, 0};
        float *sendBuf[2] = {nullptr, nullptr}, *recvBuf[2] = {nullptr, nullptr};
    } halo;
    bool boxForm = false;      // cut level: every rank runs the box protocol on this level (agreed at set-up)
    bool hasBandPlanes = false;  // host-built slab levels: packed band cells of the planes at the cuts (GHOST_BAND exchanges)
    GridP gBox{};              // the level as the box launches of a cut level see it: rows of the general cells of the whole label buffer
    float *extRows = nullptr;
    int elo = 0;               // label ghost planes below owned plane 0 in the level's label buffer (codes allocation: spare plane | buffer | spare plane)
};

}  // namespace

struct mgps_solver {
    mgps_hierarchy *hier = nullptr;  // global hierarchy (all levels, whole grid)
    std::vector<mgps_hierarchy *> retiredHier;  // (mgps_get_hierarchy replaced a slab rank's windowed hierarchy: freed with the solver)
    mgps_options opt{};
    bool useGS = false;
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<DevLevel> lv;  // single GPU: all levels; slab run: the distributed levels + the collapse level
    float *w[3] = {nullptr, nullptr, nullptr};
    bool weightsBorrowed = false;  // options.borrow_device_weights: w[] are the caller's arrays
    // coarsest-level dense inverse
    int cn = 0;
    float *cinv = nullptr, *cvec = nullptr;
    std::shared_ptr<void> cinvShared;  // set when cinv is the device-built inverse, shared with the cache of the last one (buildDeviceInverse)
    int32_t *ccells = nullptr;
    bool tailOfSlabRun = false;  // this solver is the collapsed tail owned by rank 0 of a slab run
    // reductions
    double *partials = nullptr, *resultDev = nullptr, *resultHost = nullptr;
    double *dotPartials = nullptr;  // per-workgroup shares of <p, A p> from the fused A.p launch of the CG loop
    double *cg64[4] = {nullptr, nullptr, nullptr, nullptr};  // x, r, p, A p in fp64 (options.pcg_fp64_vectors), with ghost planes
    // the last stroke of a preconditioning V-cycle also gathers <x, b> (= <z, r>, CG.h:86 / 180) when asked to:
    // the sweep leaves <x', b>, every band scatter after it the correction sum (new - old) b
    bool gatherDot = false;
    unsigned dotUsed = 0;
    size_t dotCapacity = 0;       // doubles behind dotPartials
    double *dotTarget = nullptr;  // where the gathered <x, b> goes (nullptr: resultDev)
    double *cgScal = nullptr;     // scalars of the CG loop kept on the device (launchCgScalars); [4] holds beta as a float
    // PCG work grids (allocated on first use): r, p, z, t (CG.h:43, 67, 92, 96) and 1/diag
    float *pcg[4] = {nullptr, nullptr, nullptr, nullptr};
    float *dinv = nullptr;
    // mixed precision (options.precision = 1): binary16 grids of the fine level -- iterate, its Jacobi partner, residual --
    // the power-of-two normalisation of the rhs (device scalar) and the exponent e of the iterate's storage scale 2^-e
    void *mixX = nullptr, *mixTmp = nullptr, *mixR = nullptr;
    float *mixSigma = nullptr;
    int mixExp = 0;
    double *mixMax = nullptr;   // max |r| left by the CG update pass: the next preconditioning cycle's normalisation
    void *mixResult = nullptr;  // the binary16 grid that holds the last cycle's result (scale 2^mixExp / *mixSigma)
    std::vector<void *> userGrids;  // allocation bases handed out by mgps_grid_alloc (ghost planes first)
    // planes every grid of this solver carries below and above its owned planes: 1 (the ghost plane of a slab run; spare otherwise),
    // kSlabGhostPlanes on slab ranks set up on the device (the neighbours' cells their band boxes read live there)
    int ghost = 1;
    // slab run
    bool dist = false;
    mgps_comm comm{};
    int64_t exchanges = 0;         // ghost / stage exchanges issued so far (mgps_exchange_count: the bench line's exchanges_per_cycle)
    std::vector<int> splits;       // slab run: rank r owns the fine planes [splits[r], splits[r + 1])
    int distLevels = 0;            // levels 0 .. distLevels-1 are distributed, lv[distLevels] is the collapse level
    int totalLevels = 0;           // levels of the whole hierarchy
    int requestedLevels = 0;       // device-side set-up: mg_levels as asked for (the host hierarchy is built on demand)
    mgps_solver *tail = nullptr;   // rank 0: solver of levels distLevels .. totalLevels-1 on the whole grid
    float *tailX = nullptr, *tailB = nullptr;
    // measurement hooks: event pairs around the fine-level full-domain smoother
    bool profiling = false;
    bool stageProfiling = false;  // mgps_profile_enable(h, 2): also an event pair around every stage of every level
    std::vector<hipEvent_t> profEvents;
    size_t profUsed = 0;
    int profSweeps = 0;  // full-domain sweeps the event pairs cover (a Gauss-Seidel sweep is two pairs: one per colour)
    // per-stage timing of the V-cycle (the reference's UT_StopWatch scopes, MG.cpp:436-878): event pairs per (stage, level),
    // read back at the end of the cycle; on while options.print_stats or mgps_profile_enable
    struct StageMark {
        int stage, level;
    };
    std::vector<StageMark> stageMarks;
    std::vector<hipEvent_t> stageEvents;  // 2 per mark
    double stageMs[6] = {0, 0, 0, 0, 0, 0};
    double stageMsFine[6] = {0, 0, 0, 0, 0, 0};  // the same, level 0 only
    double stageMsFineLast[6] = {0, 0, 0, 0, 0, 0};  // what the last mgps_stage_times call covered (mgps_stage_times_fine)
    int stageCycles = 0;
    std::string lastError = "";
};

namespace {
void setHandleError(const mgps_solver *h, const char *msg) noexcept
{
    try {
        const_cast<mgps_solver *>(h)->lastError = msg;
    } catch (...) {
    }
}
const bool handleErrorHookInstalled = [] {
    mgps::gSetHandleError = setHandleError;
    return true;
}();
}  // namespace

namespace {

int failH(mgps_solver *h, int code, const std::string &msg)
{
    static std::mutex guard;  // set-up reports from more than one thread
    std::lock_guard<std::mutex> lock(guard);
    if (h) h->lastError = msg;
    else setLastGlobalError(msg);
    return code;
}

#define MGPS_HIP(h, call)                                                                                 \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            return failH(h, MGPS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));             \
    } while (0)
#define MGPS_LAUNCH(h, call)                                                                              \
    do {                                                                                                  \
        int e_ = (call);                                                                                  \
        if (e_ != 0)                                                                                      \
            return failH(h, MGPS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(hipError_t(e_))); \
    } while (0)
#define MGPS_TRY(call)                \
    do {                              \
        int s_ = (call);              \
        if (s_ != MGPS_OK) return s_; \
    } while (0)
#define MGPS_COMM(h, call)                                                           \
    do {                                                                             \
        int s_ = (call);                                                             \
        if (s_ != 0) return failH(h, MGPS_ERR_COMM, std::string(#call) + " failed"); \
    } while (0)


// ---- device memory of the library -------------------------------------------------------------------------------------
// hipMalloc is cheap while the process holds little (0.3 ms for 4 GiB) and expensive once it holds much: ten 4 GiB blocks
// cost 0.6-1.3 s on this platform (tools/allocbench.hip), and every hipFree of a block >= 16 MiB 0.2 ms -- a solver of a
// 1024^3 grid holds ~45 GiB in ~300 blocks, and a plugin that rebuilds it every sub-step (Plug.cpp:463) would pay more for
// memory than for the solve.  Released blocks are therefore kept, per device, and handed to the next solver (sizes are
// rounded up by at most 1/8 so that lists whose length moves with the liquid find their block again); the cap is
// MGPS_DEVICE_CACHE_MB (default: a quarter of the device's memory), the oldest blocks leave first once it is reached,
// mgps_trim_device_cache returns everything, and an allocation that fails trims the cache and tries again.
struct DeviceCache {
    struct Block {
        void *p;
        uint64_t age;  // release order: the oldest cached block is the first to go when the cap is reached
    };
    using FreeMap = std::multimap<std::pair<int, size_t>, Block>;
    std::mutex guard;
    FreeMap free;                                                 // (device, bytes) -> block not in use
    std::unordered_map<void *, std::pair<int, size_t>> live;     // every block of the library, in use or cached
    uint64_t clock = 0;
    // cached bytes, age list and cap PER DEVICE (a process that drives several GPUs: one device's releases never evict
    // another device's blocks, and every device gets the whole cap).  cap: MGPS_DEVICE_CACHE_MB, else a quarter of that
    // device's memory (72 of 288 GiB: a 1024^3 solver holds ~45 GiB, so the plugin's solver-per-sub-step pattern still finds
    // all of its blocks again) -- the rest stays with whoever else allocates in the process (torch, RCCL, Houdini's own GPU
    // users).  It is read on the device's first allocation, with that device current and outside the lock.
    struct PerDevice {
        size_t cached = 0, cap = 0;
        bool capKnown = false;
        std::map<uint64_t, FreeMap::iterator> byAge;
    };
    std::map<int, PerDevice> dev;
};
DeviceCache &deviceCache()
{
    static DeviceCache *c = new DeviceCache();  // never destroyed (process tear-down order)
    return *c;
}
size_t deviceCapOfCurrent()
{
    if (const char *e = getenv("MGPS_DEVICE_CACHE_MB")) return size_t(std::max(0, atoi(e))) << 20;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) return totalB / 4;
    (void)hipGetLastError();
    return size_t(16) << 30;
}
size_t deviceRounded(size_t bytes)
{
    bytes = std::max<size_t>(bytes, 256);
    size_t p2 = 256;
    while (p2 * 2 <= bytes) p2 *= 2;
    const size_t step = std::max<size_t>(256, p2 / 8);
    return (bytes + step - 1) / step * step;
}
void deviceTrimLocked(DeviceCache &c, std::vector<void *> &drop)
{
    for (auto &b : c.free) {
        drop.push_back(b.second.p);
        c.live.erase(b.second.p);
    }
    c.free.clear();
    for (auto &d : c.dev) {
        d.second.byAge.clear();
        d.second.cached = 0;
    }
}
}  // namespace
namespace mgps {
int deviceAlloc(void **p, size_t bytes)
{
    DeviceCache &c = deviceCache();
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t want = deviceRounded(bytes);
    bool needCap = false;
    {
        std::lock_guard<std::mutex> lock(c.guard);
        DeviceCache::PerDevice &D = c.dev[dev];
        needCap = !D.capKnown;
        const auto it = c.free.find({dev, want});
        if (it != c.free.end()) {
            *p = it->second.p;
            D.cached -= want;
            D.byAge.erase(it->second.age);
            c.free.erase(it);
            return hipSuccess;
        }
    }
    if (needCap) {
        const size_t cap = deviceCapOfCurrent();
        std::lock_guard<std::mutex> lock(c.guard);
        DeviceCache::PerDevice &D = c.dev[dev];
        if (!D.capKnown) {
            D.cap = cap;
            D.capKnown = true;
        }
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {  // give the cached blocks back and try once more
        (void)hipGetLastError();
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lock(c.guard);
            deviceTrimLocked(c, drop);
        }
        for (void *b : drop) (void)hipFree(b);
        e = hipMalloc(p, want);
    }
    if (e != hipSuccess) return int(e);
    std::lock_guard<std::mutex> lock(c.guard);
    c.live[*p] = {dev, want};
    return hipSuccess;
}
// (unlike hipFree this does not wait for the device: callers release only what no queued kernel touches any more)
// A block larger than the cap goes straight back to the driver; otherwise it is kept and the OLDEST cached blocks of ITS device
// are returned until that device's cache fits its cap again (sizes nobody asks for any more age out instead of pinning the
// cache full).
int deviceFree(void *p)
{
    if (!p) return hipSuccess;
    DeviceCache &c = deviceCache();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lock(c.guard);
        const auto it = c.live.find(p);
        if (it == c.live.end()) drop.push_back(p);  // (not ours: plain hipFree)
        else {
            DeviceCache::PerDevice &D = c.dev[it->second.first];  // (its cap was read when the block was allocated)
            if (it->second.second > D.cap) {
                c.live.erase(it);
                drop.push_back(p);
            } else {
                const uint64_t age = c.clock++;
                D.byAge[age] = c.free.emplace(it->second, DeviceCache::Block{p, age});
                D.cached += it->second.second;
                while (D.cached > D.cap && !D.byAge.empty()) {
                    const auto oldest = D.byAge.begin()->second;
                    drop.push_back(oldest->second.p);
                    D.cached -= oldest->first.second;
                    c.live.erase(oldest->second.p);
                    c.free.erase(oldest);
                    D.byAge.erase(D.byAge.begin());
                }
            }
        }
    }
    hipError_t e = hipSuccess;
    for (void *b : drop) {
        const hipError_t eb = hipFree(b);
        if (eb != hipSuccess) e = eb;
    }
    return int(e);
}
void deviceTrim()
{
    DeviceCache &c = deviceCache();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lock(c.guard);
        deviceTrimLocked(c, drop);
    }
    for (void *b : drop) (void)hipFree(b);
}
}  // namespace mgps
namespace {

template <class T>
int devAlloc(mgps_solver *h, T **p, size_t count, bool zero)
{
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = cacheMalloc(reinterpret_cast<void **>(p), count * sizeof(T));
    if (e != hipSuccess) return failH(h, MGPS_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (zero) MGPS_HIP(h, hipMemsetAsync(*p, 0, count * sizeof(T), h->stream));
    return MGPS_OK;
}

template <class T, class A>
int devUpload(mgps_solver *h, T **p, const std::vector<T, A> &v)
{
    MGPS_TRY(devAlloc(h, p, v.size(), false));
    if (!v.empty()) MGPS_HIP(h, hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return MGPS_OK;
}

// a grid of d.nz owned planes with h->ghost zeroed ghost planes on each side; *p addresses owned plane 0
int gridAlloc(mgps_solver *h, float **p, const Dims &d)
{
    const size_t plane = size_t(d.nx) * d.ny;
    float *base = nullptr;
    MGPS_TRY(devAlloc(h, &base, (size_t(d.nz) + 2 * size_t(h->ghost)) * plane, true));
    *p = base + size_t(h->ghost) * plane;
    return MGPS_OK;
}
void gridFree(const mgps_solver *h, float *p, const Dims &d)
{
    if (p) (void)cacheFree(p - size_t(h->ghost) * size_t(d.nx) * d.ny);
}

void freeAll(mgps_solver *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->tail) freeAll(h->tail);
    for (auto &L : h->lv) {
        (void)cacheFree(L.codes);
        gridFree(h, L.x, L.d);
        gridFree(h, L.b, L.d);
        gridFree(h, L.r, L.d);
        gridFree(h, L.tmp, L.d);
        (void)cacheFree(L.band);
        (void)cacheFree(L.bandTmp);
        (void)cacheFree(L.rows);
        (void)cacheFree(L.bandDiag);
        for (int c = 0; c < 2; ++c) {
            (void)cacheFree(L.pure[c]);
            (void)cacheFree(L.mixed[c]);
        }
        (void)cacheFree(L.tileBndStart);
        for (int q = 0; q < 4; ++q) {
            (void)cacheFree(L.bandPlane[q]);
            (void)cacheFree(L.packBuf[q]);
        }
        (void)cacheFree(L.chunks);
        (void)cacheFree(L.planeBlocks);
        for (int q = 0; q < 2; ++q) {
            (void)cacheFree(L.halo.sendIdx[q]);
            (void)cacheFree(L.halo.recvIdx[q]);
            (void)cacheFree(L.halo.sendBuf[q]);
            (void)cacheFree(L.halo.recvBuf[q]);
        }
        (void)cacheFree(L.extRows);
        (void)cacheFree(L.bandBoxes.info);
        (void)cacheFree(L.bandBoxes.list);
        (void)cacheFree(L.planeFlags);
        (void)cacheFree(L.keepBits);
        (void)cacheFree(L.snapTile);
        (void)cacheFree(L.rz);
        (void)cacheFree(L.rzEdges);
        (void)cacheFree(L.bandBoxes.general);
    }
    for (int a = 0; a < 3 && !h->weightsBorrowed; ++a) (void)cacheFree(h->w[a]);
    if (!h->cinvShared) (void)cacheFree(h->cinv);  // (a device-built inverse belongs to its shared holder)
    (void)cacheFree(h->cvec);
    (void)cacheFree(h->ccells);
    (void)cacheFree(h->partials);
    (void)cacheFree(h->resultDev);
    (void)cacheFree(h->dotPartials);
    (void)cacheFree(h->cgScal);
    (void)cacheFree(h->mixX);
    (void)cacheFree(h->mixTmp);
    (void)cacheFree(h->mixR);
    (void)cacheFree(h->mixSigma);
    (void)cacheFree(h->mixMax);
    for (double *g64 : h->cg64)
        if (g64) (void)cacheFree(g64 - size_t(h->lv[0].d.nx) * h->lv[0].d.ny);
    if (h->resultHost) (void)hipHostFree(h->resultHost);
    if (!h->lv.empty()) {
        for (int q = 0; q < 4; ++q) gridFree(h, h->pcg[q], h->lv[0].d);
        gridFree(h, h->dinv, h->lv[0].d);
    }
    for (void *p : h->userGrids) (void)cacheFree(p);
    for (hipEvent_t e : h->profEvents) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->stageEvents) (void)hipEventDestroy(e);
    mgps_hierarchy_destroy(h->hier);
    for (mgps_hierarchy *r : h->retiredHier) mgps_hierarchy_destroy(r);
    delete h;
}

int checkLevel(mgps_solver *h, int level, const char *who)
{
    if (!h) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, std::string(who) + ": NULL handle");
    if (level < 0 || level >= int(h->lv.size()))
        return failH(h, MGPS_ERR_INVALID_ARGUMENT, std::string(who) + ": level out of range");
    (void)hipSetDevice(h->device);
    return MGPS_OK;
}

// ---- slab plumbing -------------------------------------------------------------------------------

// Refresh the two ghost planes of grid `a` of level l from the Z-neighbours.  GHOST_FULL moves the
// whole planes; GHOST_BAND only their band cells, packed (valid when nothing but band passes touched
// `a` since its ghosts were last complete -- a band pass changes band cells only).
enum GhostMode { GHOST_NONE = 0, GHOST_FULL = 1, GHOST_BAND = 2 };

// on: the stream the transfer is queued on (nullptr: the solver's)
int exchangeGhosts(mgps_solver *h, int l, float *a, GhostMode mode = GHOST_FULL, hipStream_t on = nullptr)
{
    if (!h->dist || mode == GHOST_NONE) return MGPS_OK;
    DevLevel &L = h->lv[l];
    const size_t plane = size_t(L.d.nx) * L.d.ny;
    const bool lo = h->comm.rank > 0, hi = h->comm.rank < h->comm.size - 1;
    if (mode == GHOST_BAND && !L.hasBandPlanes) mode = GHOST_FULL;  // (levels built on the device keep no packed lists of the cut planes' band cells)
    if (mode == GHOST_FULL) {
        const size_t bytes = plane * sizeof(float);
        ++h->exchanges;
        MGPS_COMM(h, h->comm.exchange(h->comm.user, lo ? a : nullptr, bytes, lo ? a - plane : nullptr, bytes,
                                      hi ? a + (size_t(L.d.nz) - 1) * plane : nullptr, bytes,
                                      hi ? a + size_t(L.d.nz) * plane : nullptr, bytes, on ? on : h->stream));
        return MGPS_OK;
    }
    const int *n = L.nbandPlane;
    if (lo) MGPS_LAUNCH(h, launchPack(h->stream, L.packBuf[0], a, L.bandPlane[0], n[0]));
    if (hi) MGPS_LAUNCH(h, launchPack(h->stream, L.packBuf[2], a, L.bandPlane[2], n[2]));
    ++h->exchanges;
    MGPS_COMM(h, h->comm.exchange(h->comm.user, lo ? L.packBuf[0] : nullptr, size_t(n[0]) * sizeof(float),
                                  lo ? L.packBuf[1] : nullptr, size_t(n[1]) * sizeof(float),
                                  hi ? L.packBuf[2] : nullptr, size_t(n[2]) * sizeof(float),
                                  hi ? L.packBuf[3] : nullptr, size_t(n[3]) * sizeof(float), h->stream));
    if (lo) MGPS_LAUNCH(h, launchUnpack(h->stream, a, L.packBuf[1], L.bandPlane[1], n[1]));
    if (hi) MGPS_LAUNCH(h, launchUnpack(h->stream, a, L.packBuf[3], L.bandPlane[3], n[3]));
    return MGPS_OK;
}

// ---- level operators -----------------------------------------------------------------------------

// The list part of a cut level's band-stage messages (DevLevel::BoxHalo): a0 (and a1) at the cells the neighbours' boxes read,
// packed, exchanged, and put into the deep ghost planes of the same grids on the other side.  withPlane: the boundary plane of a0
// travels with it, straight from the grid into the neighbour's ghost plane (mgps_comm::exchange2; without that entry: a plane
// exchange of its own first).  One message per neighbour.
int haloListExchange(mgps_solver *h, int l, float *a0, float *a1, bool withPlane)
{
    DevLevel &L = h->lv[l];
    DevLevel::BoxHalo &H = L.halo;
    const size_t plane = size_t(L.d.nx) * L.d.ny, G = size_t(h->ghost);
    const bool lo = h->comm.rank > 0, hi = h->comm.rank < h->comm.size - 1;
    if (!lo && !hi) return MGPS_OK;
    const size_t per = a1 ? 2 : 1;
    HaloList sLo, sHi, rLo, rHi;
    if (lo) {
        sLo = HaloList{H.sendBuf[0], H.sendIdx[0], H.nsend[0], 0};
        rLo = HaloList{H.recvBuf[0], H.recvIdx[0], H.nrecv[0], -ptrdiff_t(G * plane)};
    }
    if (hi) {
        sHi = HaloList{H.sendBuf[1], H.sendIdx[1], H.nsend[1], 0};
        rHi = HaloList{H.recvBuf[1], H.recvIdx[1], H.nrecv[1], ptrdiff_t(size_t(L.d.nz) * plane)};
    }
    MGPS_LAUNCH(h, launchHaloListPack(h->stream, sLo, sHi, a0, a1));
    const size_t pb = plane * sizeof(float);
    const size_t sb[2] = {size_t(H.nsend[0]) * per * sizeof(float), size_t(H.nsend[1]) * per * sizeof(float)};
    const size_t rb[2] = {size_t(H.nrecv[0]) * per * sizeof(float), size_t(H.nrecv[1]) * per * sizeof(float)};
    ++h->exchanges;
    if (withPlane && h->comm.exchange2) {
        mgps_xfer2 seg[4];  // send lo, recv lo, send hi, recv hi: [plane | lists]
        if (lo) {
            seg[0] = mgps_xfer2{{a0, H.sendBuf[0]}, {pb, sb[0]}};
            seg[1] = mgps_xfer2{{a0 - plane, H.recvBuf[0]}, {pb, rb[0]}};
        }
        if (hi) {
            seg[2] = mgps_xfer2{{a0 + (size_t(L.d.nz) - 1) * plane, H.sendBuf[1]}, {pb, sb[1]}};
            seg[3] = mgps_xfer2{{a0 + size_t(L.d.nz) * plane, H.recvBuf[1]}, {pb, rb[1]}};
        }
        MGPS_COMM(h, h->comm.exchange2(h->comm.user, lo ? &seg[0] : nullptr, lo ? &seg[1] : nullptr, hi ? &seg[2] : nullptr, hi ? &seg[3] : nullptr, h->stream));
    } else {
        if (withPlane) MGPS_TRY(exchangeGhosts(h, l, a0, GHOST_FULL));
        MGPS_COMM(h, h->comm.exchange(h->comm.user, lo ? H.sendBuf[0] : nullptr, sb[0], lo ? H.recvBuf[0] : nullptr, rb[0], hi ? H.sendBuf[1] : nullptr, sb[1],
                                      hi ? H.recvBuf[1] : nullptr, rb[1], h->stream));
    }
    MGPS_LAUNCH(h, launchHaloListUnpack(h->stream, rLo, rHi, a0, a1));
    return MGPS_OK;
}

// true: the band stage of level l leaves the ghost planes of x complete (never since round 5: the box form of a cut level exchanges
// the planes the next operator reads when it needs them)
bool bandStageCompletesGhosts(const mgps_solver *, int) { return false; }

// the level runs the box form of the fused band stage (BandBoxes: whole-grid levels, options.fuse_band_passes)
// (a cut level of a slab run: every rank runs the box protocol, also a rank whose planes hold no band cell)
bool levelHasBoxes(const mgps_solver *h, int l)
{
    if (h->lv[l].boxForm) return true;
    return h->lv[l].bandBoxes.ngroups > 0 && h->lv[l].bandBoxes.depth == h->opt.band_iterations;
}
// the level as the band boxes see it (a cut level: operator rows of the whole label buffer)
const GridP &boxGrid(const mgps_solver *h, int l) { return h->lv[l].boxForm ? h->lv[l].gBox : h->lv[l].g; }

// `first`: what the ghosts of x need before the first pass; the later passes follow a band pass
// dot (single-device runs only): the scatters append their corrections to h->dotPartials (see mgps_solver::gatherDot)
int bandPasses(mgps_solver *h, int l, float *x, const float *b, GhostMode first, bool dot = false)
{
    DevLevel &L = h->lv[l];
    auto sink = [&]() -> double * {
        if (!dot || L.nband <= 0) return nullptr;
        double *p = h->dotPartials + h->dotUsed;
        h->dotUsed += bandScatterBlocks(L.nband);
        return p;
    };
    if (levelHasBoxes(h, l)) {  // level is not cut: no exchanges.  No snapshot of x exists here: out of place into the level's
                                // residual grid (free during a stroke), then the band cells copied back
        double *s = nullptr;
        if (dot) {
            s = h->dotPartials + h->dotUsed;
            h->dotUsed += unsigned(L.bandBoxes.ngroups);
        }
        if (x == L.r || b == L.r) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "band stage: the level's residual grid is its scratch");
        // (a cut level: the boundary plane and the neighbours' cells the boxes read arrive first, x and rhs in one message)
        if (L.boxForm) MGPS_TRY(haloListExchange(h, l, x, const_cast<float *>(b), true));
        MGPS_LAUNCH(h, launchBandBox(h->stream, boxGrid(h, l), L.bandBoxes, false, x, b, L.r, nullptr, h->opt.jacobi_weight, false, MixScale{}, s, x));
        MGPS_LAUNCH(h, launchBandBoxCopy(h->stream, L.g, L.bandBoxes, L.r, x));
        return MGPS_OK;
    }
    for (int it = 0; it < h->opt.band_iterations; ++it) {
        MGPS_TRY(exchangeGhosts(h, l, x, it == 0 ? first : GHOST_BAND));
        MGPS_LAUNCH(h, launchBandJacobi(h->stream, L.g, x, b, L.band, L.nband, L.bandTmp, h->opt.jacobi_weight, sink()));
    }
    return MGPS_OK;
}

// ---- per-stage timing (doPrintStats, MG.h:24 / MG.cpp:436-878) --------------------------------------------------------
enum Stage { ST_BAND = 0, ST_SMOOTH = 1, ST_RESIDUAL = 2, ST_RESTRICT = 3, ST_COARSE = 4, ST_PROLONG = 5 };
const char *const kStageNames[6] = {"Boundary smoother time", "Smoother time", "Compute residual time", "Downsample time", "Direct solve time",
                                    "Upsample and add time"};
inline bool stageTimingOn(const mgps_solver *h) { return (h->opt.print_stats || h->stageProfiling) && !h->tailOfSlabRun; }
void stageFlush(mgps_solver *h);
// RAII scope: records an event pair around the launches of one stage of one level
struct StageScope {
    mgps_solver *h;
    bool on;
    size_t slot = 0;
    StageScope(mgps_solver *hh, int stage, int level) : h(hh), on(stageTimingOn(hh))
    {
        if (!on) return;
        if (h->stageMarks.size() >= 65536) stageFlush(h);  // nobody reads them: fold into the totals
        slot = h->stageMarks.size();
        while (h->stageEvents.size() < 2 * (slot + 1)) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) {
                on = false;
                return;
            }
            h->stageEvents.push_back(e);
        }
        h->stageMarks.push_back({stage, level});
        (void)hipEventRecord(h->stageEvents[2 * slot], h->stream);
    }
    ~StageScope()
    {
        if (on) (void)hipEventRecord(h->stageEvents[2 * slot + 1], h->stream);
    }
};
// end of a cycle: read the marks back (synchronises the stream), print them in the reference's wording when print_stats is
// set, add them to the per-stage totals that mgps_stage_times returns
void stageFlush(mgps_solver *h)
{
    if (h->stageMarks.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    for (size_t q = 0; q < h->stageMarks.size(); ++q) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->stageEvents[2 * q], h->stageEvents[2 * q + 1]) != hipSuccess) continue;
        h->stageMs[h->stageMarks[q].stage] += ms;
        if (h->stageMarks[q].level == 0) h->stageMsFine[h->stageMarks[q].stage] += ms;
        if (h->opt.print_stats && (!h->dist || h->comm.rank == 0))
            std::printf("      level %d  %s: %.4f ms\n", h->stageMarks[q].level, kStageNames[h->stageMarks[q].stage], double(ms));
    }
    h->stageMarks.clear();
}

// measurement hook: an event pair strictly around the launches of a fine-level full-domain sweep (after its ghost
// exchange, so that on slab runs the figure is kernel time, not kernel + communication time)
int profMark(mgps_solver *h, bool begin)
{
    if (begin && h->profUsed + 2 > h->profEvents.size()) {
        hipEvent_t e0, e1;
        MGPS_HIP(h, hipEventCreate(&e0));
        MGPS_HIP(h, hipEventCreate(&e1));
        h->profEvents.push_back(e0);
        h->profEvents.push_back(e1);
    }
    MGPS_HIP(h, hipEventRecord(h->profEvents[h->profUsed + (begin ? 0 : 1)], h->stream));
    if (!begin) h->profUsed += 2;
    return MGPS_OK;
}

// snap: the tiles the band boxes read also leave their result in the level's residual grid (see gsStrokeSnapshots)
int gsHalfSweep(mgps_solver *h, int l, float *x, const float *b, int odd, int forward, GhostMode ghosts = GHOST_FULL, bool dot = false,
                bool timed = false, bool snap = false)
{
    DevLevel &L = h->lv[l];
    MGPS_TRY(exchangeGhosts(h, l, x, ghosts));  // the other colour's tiles across the cut changed in the previous pass
    double *sink = nullptr;
    if (dot) {  // every tile is swept once per sweep: its values after its colour's pass are the sweep's
        sink = h->dotPartials + h->dotUsed;
        h->dotUsed += unsigned(L.npure[odd] + L.nmixed[odd]);
    }
    if (timed) MGPS_TRY(profMark(h, true));
    MGPS_LAUNCH(h, launchTiledGS(h->stream, L.g, x, b, L.pure[odd], L.npure[odd], L.mixed[odd], L.nmixed[odd],
                                 L.tileBndStart, forward, sink, snap ? L.r : nullptr, snap ? L.snapTile : nullptr));
    if (timed) MGPS_TRY(profMark(h, false));
    return MGPS_OK;
}

// Gauss-Seidel strokes on a level with band boxes (round 4).  The sweep runs in place, so no closure launch can compute "what the
// sweep will hold"; but the box launch only must not read what another group writes.  Whoever writes the iterate last before a
// band stage therefore leaves a copy of the cells the boxes read in the level's residual grid (free during a stroke) -- the
// Gauss-Seidel tile kernels after their colour pass, the prolongation in front of an up-stroke; only tiles flagged in
// L.snapTile, a fifth of the tiles on the cube -- and the stage reads that snapshot and writes the iterate IN PLACE: one launch
// per stage instead of "out of place + copy" (round 3: bandBoxCopyKernel 39 us per stage at 512^3, and the plugin's own
// smoother 10 % slower than in round 2).  A stroke that starts from the cleared iterate needs no snapshot: nothing is read.
// MGPS_GS_SNAPSHOT=0: the round-3 form (A/B).
bool gsStrokeSnapshots(const mgps_solver *h, int l, const float *cur, const float *b)
{
    static const bool allowed = [] {
        const char *e = getenv("MGPS_GS_SNAPSHOT");
        return !(e && e[0] == '0');
    }();
    return allowed && h->useGS && !h->dist && h->opt.band_iterations > 0 && levelHasBoxes(h, l) && cur != h->lv[l].r && b != h->lv[l].r;
}
int ensureSnapTiles(mgps_solver *h, int l)
{
    DevLevel &L = h->lv[l];
    if (L.snapTile) return MGPS_OK;
    const size_t nt = size_t((L.d.nx + kTile - 1) / kTile) * size_t((L.d.ny + kTile - 1) / kTile) * size_t((L.d.nz + kTile - 1) / kTile);
    MGPS_TRY(devAlloc(h, &L.snapTile, nt, true));
    MGPS_LAUNCH(h, launchMarkSnapTiles(h->stream, L.g, L.bandBoxes, L.snapTile));
    return MGPS_OK;
}

// A down-stroke that starts from the cleared iterate (MG.cpp:439-440, 566) can skip the clearing launch when it runs as "sweep,
// closure launch, plain launch" on a level that takes the quad or the plane-marching sweep: both readers of the iterate take it as zero, and the grid
// itself becomes the stroke's spare.  Own grids only: their chunks without active cells hold 0 already.
bool strokeTakesZero(const mgps_solver *h, int l, const float *cur, const float *other, const float *b, bool dot)
{
    static const bool allowed = [] {  // MGPS_ZERO_START=0: clear and read the grid (A/B timing)
        const char *e = getenv("MGPS_ZERO_START");
        return !(e && e[0] == '0');
    }();
    const DevLevel &L = h->lv[l];
    // (a slab run: cut levels in box form -- the stroke's first message then carries the rhs at the neighbours' cells and nothing of the iterate)
    return allowed && !dot && (!h->dist || L.boxForm) && !h->useGS && h->opt.pre_sweeps == 1 && h->opt.band_iterations > 0 && levelHasBoxes(h, l) && stencilKernelOf(L.g) != 3 &&
           !(h->profiling && l == 0) && cur != L.r && other != L.r && b != L.r;
}

// Residual + restriction of a down-stroke without the residual grid (launchResidualZ + launchRestrictXY): whole-grid fp32 levels
// that kept their plane blocks (planeZcFor: the liquid fills most of the blocks it touches) and whose x-y planes are 4 MiB or more.
// Measured on MI355X, separate passes against the pair, ms per cycle: 1024^3 fine level
// residual 1.79 + restriction 0.80 against 1.75 + 0.44 (99.2 -> 104.0 cycles/s); 512^3 0.22 + 0.11 against 0.25 + 0.064 (a wash: 784
// workgroups of 1024 threads are one and a half rounds of the chip, and the residual it never writes would have stayed in the
// Infinity Cache); 256^3 0.027 + 0.019 against 0.048 + 0.014.  MGPS_FUSE_RR=0: never; =1: every level that fits (tests).  The
// terms of a coarse cell are added along z first instead of last: the last bits of the coarse rhs differ from the separate passes'.
bool residualRestrictFuses(const mgps_solver *h, int l)
{
    static const int mode = [] {  // -1: by size
        const char *e = getenv("MGPS_FUSE_RR");
        return !e ? -1 : (e[0] == '0' ? 0 : 1);
    }();
    if (mode == 0 || l + 1 >= int(h->lv.size())) return false;
    const GridP &F = h->lv[l].g;
    if (mode < 0 && size_t(F.nx) * F.ny * sizeof(float) < kPlaneSweepMinPlaneBytes) return false;
    return residualRestrictFits(F, h->lv[l + 1].g);
}
// out (optional): where the coarse rhs goes instead of the coarse level's own rhs grid (mgps_residual_downsample)
int residualRestrict(mgps_solver *h, int l, const float *x, const float *rhs, float *out = nullptr)
{
    DevLevel &F = h->lv[l], &C = h->lv[l + 1];
    if (!F.rz) {
        MGPS_TRY(devAlloc(h, &F.rz, F.d.cells() / 2, true));
        if (F.g.planeBlocks) {  // once: which blocks off the activity list sit below / above a listed one
            if (!F.planeFlags) {
                MGPS_TRY(devAlloc(h, &F.planeFlags, planeBlockCount(F.g), true));
                MGPS_LAUNCH(h, launchPlaneBlockFlags(h->stream, F.g, F.planeFlags));
            }
            std::vector<uint8_t> flags(planeBlockCount(F.g));
            MGPS_HIP(h, hipMemcpyAsync(flags.data(), F.planeFlags, flags.size(), hipMemcpyDeviceToHost, h->stream));
            MGPS_HIP(h, hipStreamSynchronize(h->stream));
            const std::vector<int32_t> edges = planeBlockEdges(F.g, flags);
            F.nrzEdges = int(edges.size());
            if (F.nrzEdges) MGPS_TRY(devUpload(h, &F.rzEdges, edges));
        }
    }
    {
        StageScope scope(h, ST_RESIDUAL, l);
        const bool cut = h->dist && (F.g.ghostLo || F.g.ghostHi);
        if (cut) {  // r on the planes at the cuts first: the neighbours' marches fold it in as their edge terms (one message, like the restriction's)
            MGPS_TRY(exchangeGhosts(h, l, const_cast<float *>(x), F.boxForm ? GHOST_FULL : h->opt.band_iterations > 0 ? GHOST_BAND : GHOST_FULL));  // (see vcycle)
            MGPS_LAUNCH(h, launchResidualEdgePlanes(h->stream, F.g, F.r, x, rhs));
            MGPS_TRY(exchangeGhosts(h, l, F.r, GHOST_FULL));
        }
        MGPS_LAUNCH(h, launchResidualZ(h->stream, F.g, F.rz, x, rhs, F.rzEdges, F.nrzEdges, cut ? F.r : nullptr));
    }
    StageScope scope(h, ST_RESTRICT, l);
    MGPS_LAUNCH(h, launchRestrictXY(h->stream, C.g, out ? out : C.b, F.rz));
    return MGPS_OK;
}

// The closure launch and the sweep of a stroke in one launch (launchStrokeFront): whole-grid levels that take the quad sweep, up to
// 2^24 cells (a 256^3 level) -- where a launch is a latency chain, one chain instead of two
bool strokeFrontMerges(const mgps_solver *h, int l)
{
    constexpr size_t maxCells = size_t(1) << 24;
    const DevLevel &L = h->lv[l];
    return !h->dist && L.d.cells() <= maxCells && stencilKernelOf(L.g) == 1 && (L.d.cells() & 31) == 0;  // (512^3 level: 618-629 -> 565-593 cycles/s merged)
}
int ensureKeepBits(mgps_solver *h, int l)
{
    DevLevel &L = h->lv[l];
    if (L.keepBits) return MGPS_OK;
    MGPS_TRY(devAlloc(h, &L.keepBits, L.d.cells() / 32, true));
    MGPS_LAUNCH(h, launchMarkClosure(h->stream, L.g, L.bandBoxes, L.keepBits));
    return MGPS_OK;
}

// 3 x band Jacobi -> full-domain smoother -> 3 x band Jacobi (MG.cpp:445-513 down, 806-879 up).
// The smoother runs options.pre_sweeps (down) / post_sweeps (up) times; the reference's count is one.
// Jacobi runs out of place: `cur` holds the current iterate, `other` the spare grid; they swap.
// ghostsFresh: the ghosts of `cur` are known to be complete on entry (all zero after a clear).
// Ghost traffic of a stroke: whole planes after whatever rewrote the whole grid (the caller's
// prolongation / initial guess, the full-domain smoother), packed band cells after band passes.
// dot: see mgps_solver::gatherDot (the caller folds the partials afterwards)
// xZero: `cur` is known to be zero everywhere and was NOT cleared (strokeTakesZero said the stroke needs no copy of it): the sweep
// and the closure launch take the iterate as zero
// preSnap (Gauss-Seidel up-strokes): the prolongation left the band boxes' input in L.r (gsStrokeSnapshots)
int smoothStroke(mgps_solver *h, int l, float *&cur, float *&other, const float *b, bool down, bool ghostsFresh, bool dot = false, bool xZero = false,
                 bool preSnap = false)
{
    DevLevel &L = h->lv[l];
    const bool bands = h->opt.band_iterations > 0;
    if (gsStrokeSnapshots(h, l, cur, b)) {
        MGPS_TRY(ensureSnapTiles(h, l));
        const bool timed = h->profiling && l == 0;
        {
            StageScope scope(h, ST_BAND, l);
            if (down && ghostsFresh)  // the iterate was cleared (MG.cpp:439-440, 566): nothing is read, the stage writes it in place
                MGPS_LAUNCH(h, launchBandBox(h->stream, L.g, L.bandBoxes, false, nullptr, b, cur, nullptr, h->opt.jacobi_weight));
            else if (preSnap)
                MGPS_LAUNCH(h, launchBandBox(h->stream, L.g, L.bandBoxes, false, L.r, b, cur, nullptr, h->opt.jacobi_weight));
            else
                MGPS_TRY(bandPasses(h, l, cur, b, GHOST_NONE));
        }
        const int reps = down ? h->opt.pre_sweeps : h->opt.post_sweeps;
        for (int rep = 0; rep < reps; ++rep) {
            StageScope scope(h, ST_SMOOTH, l);
            const bool d = dot && rep == reps - 1, sn = rep == reps - 1;  // (the last sweep's values are the stage's input)
            if (down) {  // odd tiles forward, then even tiles forward (MG.cpp:466-479)
                MGPS_TRY(gsHalfSweep(h, l, cur, b, 1, 1, GHOST_NONE, d, timed, sn));
                MGPS_TRY(gsHalfSweep(h, l, cur, b, 0, 1, GHOST_NONE, d, timed, sn));
            } else {  // even tiles backward, then odd tiles backward (MG.cpp:740-751)
                MGPS_TRY(gsHalfSweep(h, l, cur, b, 0, 0, GHOST_NONE, d, timed, sn));
                MGPS_TRY(gsHalfSweep(h, l, cur, b, 1, 0, GHOST_NONE, d, timed, sn));
            }
            if (timed) ++h->profSweeps;
        }
        StageScope scope(h, ST_BAND, l);
        double *sink = nullptr;
        if (dot) {
            sink = h->dotPartials + h->dotUsed;
            h->dotUsed += unsigned(L.bandBoxes.ngroups);
        }
        if (reps > 0)
            MGPS_LAUNCH(h, launchBandBox(h->stream, L.g, L.bandBoxes, false, L.r, b, cur, nullptr, h->opt.jacobi_weight, false, MixScale{}, sink, L.r));  // (old values: the snapshot holds every band cell of a flagged tile)
        else
            MGPS_TRY(bandPasses(h, l, cur, b, GHOST_NONE, dot));
        return MGPS_OK;
    }
    if (bands && !h->useGS && (down ? h->opt.pre_sweeps : h->opt.post_sweeps) == 1 && levelHasBoxes(h, l) && cur != L.r && other != L.r && b != L.r) {
        // "band passes, sweep, band passes" in three launches, nothing scattered (launchBandBox): the closure launch, which
        // computes on the band closure what the sweep would write there after the band passes and leaves it as a snapshot in
        // the level's residual grid (free during a stroke); the sweep over the un-smoothed grid; the second band stage, which
        // reads the snapshot and writes the band cells AND the closure-output cells into the sweep's output (4 bytes less per
        // closure cell than a closure launch that patches the sweep's output itself: +1.5 % at 512^3, +0.7 % at 1024^3).  The two
        // launches before the plain one are independent of each other; running the closure launch on a second stream beside the
        // sweep was measured and lost -- 1024^3: sweep 1.72 -> 2.02 ms with the closure launch at 1.57 ms beside it, cycle 10.81 ->
        // 10.91 ms; on small levels the two event hops cost ~17 us per stroke)
        const bool timed = h->profiling && l == 0;
        double *sinkB = nullptr;
        const float *src = xZero ? nullptr : cur;
        if (!dot && !timed && !stageTimingOn(h) && strokeFrontMerges(h, l)) {  // (stage timers and the sweep timer want the launches apart)
            MGPS_TRY(ensureKeepBits(h, l));
            MGPS_LAUNCH(h, launchStrokeFront(h->stream, L.g, L.bandBoxes, other, src, b, L.r, h->opt.jacobi_weight, L.keepBits));
            std::swap(cur, other);
            MGPS_LAUNCH(h, launchBandBox(h->stream, L.g, L.bandBoxes, false, L.r, b, cur, nullptr, h->opt.jacobi_weight, false, MixScale{}, nullptr, nullptr, true));
            return MGPS_OK;
        }
        // A cut level of a slab run (DevLevel::boxForm) runs the same three launches on its own planes with two messages: before
        // the closure launch the boundary plane of the iterate (the sweep's ghost plane) and, packed, iterate and rhs at the
        // neighbours' cells its regions read (up to depth + 1 planes deep: they live in the deep ghost planes of the grids); before
        // the plain launch the snapshot at the same cells.  The stroke leaves the ghost planes of its result stale: whoever reads
        // across the cut next exchanges them (the residual, the finer level's prolongation).
        {
            StageScope scope(h, ST_BAND, l);
            if (L.boxForm && xZero) MGPS_TRY(haloListExchange(h, l, const_cast<float *>(b), nullptr, false));  // (the iterate is zero on every rank: the rhs alone)
            else if (L.boxForm) MGPS_TRY(haloListExchange(h, l, cur, const_cast<float *>(b), true));
            MGPS_LAUNCH(h, launchBandBox(h->stream, boxGrid(h, l), L.bandBoxes, true, src, b, nullptr, L.r, h->opt.jacobi_weight));
        }
        {
            StageScope scope(h, ST_SMOOTH, l);
            if (timed) MGPS_TRY(profMark(h, true));
            if (dot) {
                unsigned used = 0;
                MGPS_LAUNCH(h, launchStencilDot(h->stream, OP_JACOBI, L.g, other, cur, b, h->opt.jacobi_weight, h->dotPartials + h->dotUsed, &used));
                h->dotUsed += used;
                sinkB = h->dotPartials + h->dotUsed;
                h->dotUsed += unsigned(L.bandBoxes.ngroups);
            } else {
                GridP gs = L.g;
                gs.nbnd = 0;  // every BOUNDARY cell lies in the band closure: the box launches compute the general ones as well
                MGPS_LAUNCH(h, launchStencil(h->stream, OP_JACOBI, gs, other, src, b, h->opt.jacobi_weight, true));
            }
            if (timed) {
                MGPS_TRY(profMark(h, false));
                ++h->profSweeps;
            }
        }
        StageScope scope(h, ST_BAND, l);
        std::swap(cur, other);
        if (L.boxForm) MGPS_TRY(haloListExchange(h, l, L.r, nullptr, false));
        // (the gathered dot: the sweep left sum x' b with its own values everywhere; this launch adds (new - sweep's) b on every cell it writes)
        MGPS_LAUNCH(h, launchBandBox(h->stream, boxGrid(h, l), L.bandBoxes, false, L.r, b, cur, nullptr, h->opt.jacobi_weight, false, MixScale{}, sinkB, cur, true));
        return MGPS_OK;
    }
    const int reps = down ? h->opt.pre_sweeps : h->opt.post_sweeps;
    // Several Jacobi sweeps per stroke (options.pre_sweeps / post_sweeps > 1, BASELINE config 1's "2 + 2") on a level with boxes: the
    // first band stage and the first sweep as above -- sweep over the un-smoothed grid, then the closure launch writes what the
    // sweep should hold on the band closure straight into its output (no snapshot: the next sweep reads the grid) -- two launches
    // instead of three (band stage out of place, copy, sweep); the remaining sweeps and the last band stage as below
    const bool firstFused = bands && !h->useGS && reps > 1 && levelHasBoxes(h, l) && !L.boxForm && cur != L.r && other != L.r && b != L.r;
    if (firstFused) {
        {
            StageScope scope(h, ST_SMOOTH, l);
            GridP gs = L.g;
            gs.nbnd = 0;
            const bool timedFirst = h->profiling && l == 0;  // (the sweep is a launch of its own here: the sweep timer takes it)
            if (timedFirst) MGPS_TRY(profMark(h, true));
            MGPS_LAUNCH(h, launchStencil(h->stream, OP_JACOBI, gs, other, cur, b, h->opt.jacobi_weight, true));
            if (timedFirst) {
                MGPS_TRY(profMark(h, false));
                ++h->profSweeps;
            }
        }
        StageScope scope(h, ST_BAND, l);
        MGPS_LAUNCH(h, launchBandBox(h->stream, L.g, L.bandBoxes, true, cur, b, other, nullptr, h->opt.jacobi_weight));
        std::swap(cur, other);
    } else {
        StageScope scope(h, ST_BAND, l);
        MGPS_TRY(bandPasses(h, l, cur, b, ghostsFresh ? GHOST_NONE : GHOST_FULL));
    }
    // after the band passes only band cells are stale across the cut -- unless there were none
    const GhostMode afterBands = bands ? GHOST_BAND : (ghostsFresh ? GHOST_NONE : GHOST_FULL);
    const bool timed = h->profiling && l == 0;
    for (int rep = firstFused ? 1 : 0; rep < reps; ++rep) {
        StageScope scope(h, ST_SMOOTH, l);
        const GhostMode before = rep == 0 ? afterBands : GHOST_FULL;  // a sweep rewrote everything
        const bool d = dot && rep == reps - 1;  // <x, b> of the stroke's result: the last sweep's values
        if (h->useGS) {
            if (down) {  // odd tiles forward, then even tiles forward (MG.cpp:466-479)
                MGPS_TRY(gsHalfSweep(h, l, cur, b, 1, 1, before, d, timed));
                MGPS_TRY(gsHalfSweep(h, l, cur, b, 0, 1, GHOST_FULL, d, timed));
            } else {  // even tiles backward, then odd tiles backward (MG.cpp:740-751)
                MGPS_TRY(gsHalfSweep(h, l, cur, b, 0, 0, before, d, timed));
                MGPS_TRY(gsHalfSweep(h, l, cur, b, 1, 0, GHOST_FULL, d, timed));
            }
        } else {
            MGPS_TRY(exchangeGhosts(h, l, cur, before));
            if (timed) MGPS_TRY(profMark(h, true));
            if (d) {
                unsigned used = 0;
                MGPS_LAUNCH(h, launchStencilDot(h->stream, OP_JACOBI, L.g, other, cur, b, h->opt.jacobi_weight, h->dotPartials + h->dotUsed, &used));
                h->dotUsed += used;
            } else
                MGPS_LAUNCH(h, launchStencil(h->stream, OP_JACOBI, L.g, other, cur, b, h->opt.jacobi_weight, true));
            if (timed) MGPS_TRY(profMark(h, false));
            std::swap(cur, other);
        }
        if (timed) ++h->profSweeps;
    }
    StageScope scope(h, ST_BAND, l);
    MGPS_TRY(bandPasses(h, l, cur, b, GHOST_FULL, dot));  // the full-domain smoother rewrote everything
    return MGPS_OK;
}

int vcycle(mgps_solver *h, float *x, const float *b, bool useInitialGuess, bool ownGrid = false, bool wantDot = false);

// MGPS_POISON_SPARES=1 (tests): the grids a zero-start stroke neither clears nor is supposed to read -- the never-cleared iterate
// and its Jacobi partner -- get NaN in every active cell first.  The shortcut rests on "nobody reads them before writing them";
// a stale read then poisons the result instead of hiding behind the previous cycle's plausible values
int poisonSpares(mgps_solver *h, int l, float *a, float *b2)
{
    static const bool on = [] {
        const char *e = getenv("MGPS_POISON_SPARES");
        return e && e[0] == '1';
    }();
    if (!on) return MGPS_OK;
    const float nan = std::numeric_limits<float>::quiet_NaN();
    MGPS_LAUNCH(h, launchScale(h->stream, h->lv[l].g, a, nan));
    MGPS_LAUNCH(h, launchScale(h->stream, h->lv[l].g, b2, nan));
    return MGPS_OK;
}

// levels distLevels .. totalLevels-1 of a slab run: gather the rhs of the collapse level to rank 0,
// run the rest of the cycle there on the whole grid, scatter the correction back
int collapsedTail(mgps_solver *h)
{
    DevLevel &C = h->lv[h->distLevels];
    const size_t bytes = C.d.cells() * sizeof(float);
    const int P = h->comm.size;
    bool uniform = true;
    for (int r = 1; r < P; ++r) uniform = uniform && (h->splits[size_t(r) + 1] - h->splits[size_t(r)]) == (h->splits[1] - h->splits[0]);
    std::vector<size_t> counts, displs;
    if (!uniform) {  // slabs balanced by active planes: rank r's share of the collapse level is its planes of that level
        const size_t planeBytes = size_t(C.d.nx) * C.d.ny * sizeof(float);
        for (int r = 0; r < P; ++r) {
            displs.push_back(size_t(h->splits[size_t(r)] >> h->distLevels) * planeBytes);
            counts.push_back(size_t((h->splits[size_t(r) + 1] - h->splits[size_t(r)]) >> h->distLevels) * planeBytes);
        }
    }
    if (uniform) MGPS_COMM(h, h->comm.gather(h->comm.user, C.b, h->comm.rank == 0 ? h->tailB : nullptr, bytes, 0, h->stream));
    else MGPS_COMM(h, h->comm.gatherv(h->comm.user, C.b, bytes, h->comm.rank == 0 ? h->tailB : nullptr, counts.data(), displs.data(), 0, h->stream));
    if (h->comm.rank == 0) {
        h->tail->stream = h->stream;
        const int rc = vcycle(h->tail, h->tailX, h->tailB, false);
        if (rc != MGPS_OK) return failH(h, rc, "collapsed tail: " + h->tail->lastError);
    }
    if (uniform) MGPS_COMM(h, h->comm.scatter(h->comm.user, h->comm.rank == 0 ? h->tailX : nullptr, C.x, bytes, 0, h->stream));
    else MGPS_COMM(h, h->comm.scatterv(h->comm.user, h->comm.rank == 0 ? h->tailX : nullptr, counts.data(), displs.data(), C.x, bytes, 0, h->stream));
    return MGPS_OK;
}

int zeroGrid(mgps_solver *h, float *a, const Dims &d, bool withGhosts)
{
    const size_t plane = size_t(d.nx) * d.ny;
    if (withGhosts) MGPS_LAUNCH(h, launchZero(h->stream, a - plane, d.cells() + 2 * plane));
    else MGPS_LAUNCH(h, launchZero(h->stream, a, d.cells()));
    return MGPS_OK;
}

// a grid of the solver's own (level l): chunks without active cells are never written, they still hold their initial 0
int zeroOwnGrid(mgps_solver *h, int l, float *a, bool withGhosts)
{
    DevLevel &L = h->lv[l];
    MGPS_LAUNCH(h, launchZeroActive(h->stream, L.g, a, withGhosts && h->dist));
    return MGPS_OK;
}

// ownGrid: x is one of the solver's own grids (see zeroOwnGrid).  wantDot (single-device runs, after ensurePcgGrids):
// <x, b> of the result is left in h->resultDev as a by-product of the last stroke (mgps_solver::gatherDot)
int vcycle(mgps_solver *h, float *x, const float *b, bool useInitialGuess, bool ownGrid, bool wantDot)
{
    h->gatherDot = wantDot && !h->dist && h->dotPartials != nullptr;
    h->dotUsed = 0;
    const int nlv = int(h->lv.size());
    if (h->tailOfSlabRun && nlv == 1) {  // the tail of a slab run can be the direct solve alone
        MGPS_TRY(zeroGrid(h, x, h->lv[0].d, false));
        MGPS_LAUNCH(h, launchCoarseSolve(h->stream, h->cn, h->cinv, h->ccells, x, b, h->cvec));
        return MGPS_OK;
    }
    // levels this object smooths on: all but its last one (direct solve, or the collapse level of a
    // slab run), unless the whole hierarchy is a single level (MG.cpp:516-517)
    const int nsmooth = nlv > 1 ? nlv - 1 : 1;
    const bool hasBottom = nlv > 1;
    std::vector<float *> cur(nlv, nullptr), other(nlv, nullptr);
    cur[0] = x;
    other[0] = h->lv[0].tmp;
    bool fresh = false, zero0 = false;
    if (!useInitialGuess) {  // MG.cpp:439-440
        zero0 = ownGrid && strokeTakesZero(h, 0, cur[0], other[0], b, h->gatherDot && !hasBottom);
        if (zero0) MGPS_TRY(poisonSpares(h, 0, cur[0], other[0]));
        else if (ownGrid) MGPS_TRY(zeroOwnGrid(h, 0, x, true));
        else MGPS_TRY(zeroGrid(h, x, h->lv[0].d, h->dist));
        fresh = true;
    }
    MGPS_TRY(smoothStroke(h, 0, cur[0], other[0], b, true, fresh, h->gatherDot && !hasBottom, zero0));
    if (hasBottom) {
        const float *rhs = b;
        // options.interrupt is also polled once per level and stroke of a single-device cycle (the reference polls inside
        // every operator loop, e.g. Ops.h:319); slab runs poll between CG iterations only, where the ranks can agree
        auto stopRequested = [&] { return !h->dist && !h->tailOfSlabRun && h->opt.interrupt && h->opt.interrupt(h->opt.interrupt_user); };
        for (int l = 0; l < nsmooth; ++l) {  // MG.cpp:519-553 (fine), 557-667 (coarser)
            DevLevel &F = h->lv[l], &C = h->lv[l + 1];
            if (l > 0 && stopRequested()) return failH(h, MGPS_ERR_INTERRUPTED, "mgps_apply_vcycle: interrupted");
            if (l > 0) {
                cur[l] = F.x;
                other[l] = F.tmp;
                rhs = F.b;
                const bool zl = strokeTakesZero(h, l, cur[l], other[l], rhs, false);
                if (!zl) MGPS_TRY(zeroOwnGrid(h, l, F.x, true));  // MG.cpp:566
                else MGPS_TRY(poisonSpares(h, l, cur[l], other[l]));
                MGPS_TRY(smoothStroke(h, l, cur[l], other[l], rhs, true, true, false, zl));
            }
            if (residualRestrictFuses(h, l)) {
                MGPS_TRY(residualRestrict(h, l, cur[l], rhs));
                continue;
            }
            {
                StageScope scope(h, ST_RESIDUAL, l);
                // (a cut level in box form: the stroke's last launch rewrote band and closure cells of a grid whose ghost planes are
                // the sweep's input -- whole planes; per-pass band smoothing: only band cells changed since the last whole plane)
                MGPS_TRY(exchangeGhosts(h, l, cur[l], F.boxForm ? GHOST_FULL : h->opt.band_iterations > 0 ? GHOST_BAND : GHOST_FULL));
                MGPS_LAUNCH(h, launchStencil(h->stream, OP_RESIDUAL, F.g, F.r, cur[l], rhs, 0.f, true));
            }
            StageScope scope(h, ST_RESTRICT, l);
            MGPS_TRY(exchangeGhosts(h, l, F.r));
            MGPS_LAUNCH(h, launchRestrict(h->stream, C.g, C.b, F.r));
        }
        DevLevel &B = h->lv[nsmooth];
        {
            StageScope scope(h, ST_COARSE, nsmooth);
            if (h->dist) MGPS_TRY(collapsedTail(h));
            else MGPS_LAUNCH(h, launchCoarseSolve(h->stream, h->cn, h->cinv, h->ccells, B.x, B.b, h->cvec));  // MG.cpp:669-692
        }
        cur[nsmooth] = B.x;
        for (int l = nsmooth - 1; l >= 0; --l) {  // MG.cpp:695-784 (coarser), 787-880 (fine)
            DevLevel &F = h->lv[l];
            if (stopRequested()) return failH(h, MGPS_ERR_INTERRUPTED, "mgps_apply_vcycle: interrupted");
            const float *rhsUp = l == 0 ? b : F.b;
            bool upSnap = false;
            {
                StageScope scope(h, ST_PROLONG, l);
                MGPS_TRY(exchangeGhosts(h, l + 1, cur[l + 1]));
                const bool snap = gsStrokeSnapshots(h, l, cur[l], rhsUp);
                if (snap) MGPS_TRY(ensureSnapTiles(h, l));
                MGPS_LAUNCH(h, launchProlongAdd(h->stream, F.g, cur[l], cur[l + 1], snap ? F.r : nullptr, snap ? F.snapTile : nullptr));
                upSnap = snap;
            }
            MGPS_TRY(smoothStroke(h, l, cur[l], other[l], rhsUp, false, false, h->gatherDot && l == 0, false, upSnap));
        }
    }
    if (cur[0] != x)  // single-level Jacobi cycle: the iterate ended in the spare grid
        MGPS_HIP(h, hipMemcpyAsync(x, cur[0], h->lv[0].d.cells() * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    if (h->gatherDot) MGPS_LAUNCH(h, launchFoldDot(h->stream, h->dotPartials, h->dotUsed, h->dotTarget ? h->dotTarget : h->resultDev));
    if (stageTimingOn(h)) ++h->stageCycles;
    if (h->opt.print_stats) stageFlush(h);  // (profiling without print_stats: the marks pile up until mgps_stage_times reads them)
    return MGPS_OK;
}

// levels first .. last of a single-device solver: rhs in lv[first].b, *result = the grid that ends up holding the
// correction (MG.cpp:557-784 from level `first` on)
int innerCycle(mgps_solver *h, int first, float **result)
{
    const int nlv = int(h->lv.size());
    const int nsmooth = nlv - 1;  // the last level is the direct solve
    std::vector<float *> cur(nlv, nullptr), other(nlv, nullptr);
    for (int l = first; l < nsmooth; ++l) {
        DevLevel &F = h->lv[l], &C = h->lv[l + 1];
        cur[l] = F.x;
        other[l] = F.tmp;
        const bool zl = strokeTakesZero(h, l, cur[l], other[l], F.b, false);
        if (!zl) MGPS_TRY(zeroOwnGrid(h, l, F.x, true));  // MG.cpp:566
        else MGPS_TRY(poisonSpares(h, l, cur[l], other[l]));
        MGPS_TRY(smoothStroke(h, l, cur[l], other[l], F.b, true, true, false, zl));
        if (residualRestrictFuses(h, l)) {
            MGPS_TRY(residualRestrict(h, l, cur[l], F.b));
            continue;
        }
        MGPS_LAUNCH(h, launchStencil(h->stream, OP_RESIDUAL, F.g, F.r, cur[l], F.b, 0.f, true));
        MGPS_LAUNCH(h, launchRestrict(h->stream, C.g, C.b, F.r));
    }
    DevLevel &B = h->lv[nsmooth];
    MGPS_LAUNCH(h, launchCoarseSolve(h->stream, h->cn, h->cinv, h->ccells, B.x, B.b, h->cvec));  // MG.cpp:669-692
    cur[nsmooth] = B.x;
    for (int l = nsmooth - 1; l >= first; --l) {
        DevLevel &F = h->lv[l];
        const bool snap = gsStrokeSnapshots(h, l, cur[l], F.b);
        if (snap) MGPS_TRY(ensureSnapTiles(h, l));
        MGPS_LAUNCH(h, launchProlongAdd(h->stream, F.g, cur[l], cur[l + 1], snap ? F.r : nullptr, snap ? F.snapTile : nullptr));
        MGPS_TRY(smoothStroke(h, l, cur[l], other[l], F.b, false, false, false, false, snap));
    }
    *result = cur[first];
    return MGPS_OK;
}

int ensureMixedGrids(mgps_solver *h)
{
    if (h->mixX) return MGPS_OK;
    const size_t n = h->lv[0].d.cells();
    uint16_t *p[3] = {nullptr, nullptr, nullptr};
    for (int q = 0; q < 3; ++q) MGPS_TRY(devAlloc(h, &p[q], n, true));
    h->mixX = p[0];
    h->mixTmp = p[1];
    h->mixR = p[2];
    MGPS_TRY(devAlloc(h, &h->mixSigma, 1, true));
    MGPS_TRY(devAlloc(h, &h->mixMax, 1, true));
    // |A^-1| of the h-free operator can reach N^2 / 2 (a column of liquid under a free surface): keep that below 2^14
    const Dims d = h->lv[0].d;
    const double bound = 0.5 * double(std::max(d.nx, std::max(d.ny, d.nz))) * double(std::max(d.nx, std::max(d.ny, d.nz)));
    h->mixExp = std::max(0, int(std::ceil(std::log2(bound / 16384.0))));
    return MGPS_OK;
}

// The V-cycle with the fine level's iterate and residual in binary16 (options.precision = 1, BASELINE config 5; Jacobi
// smoother, single device).  Same schedule as vcycle() (MG.cpp:420-881); x and b are fp32 grids in the caller's units.
// x == nullptr: the result stays in binary16 (h->mixResult; z = 2^mixExp / *mixSigma times it) for launchHalfDot /
// launchXpayHalf.  maxAbsDev: max |b| if a previous pass already left it on the device (else a reduction runs here).
// dotDev (x == nullptr only): *dotDev = <z, b> gathered on the way by the last stroke (the last Jacobi sweep leaves
// sum x~' b, the band scatters after it the corrections), as vcycle() does for the fp32 cycle
int vcycleMixed(mgps_solver *h, float *x, const float *b, bool useInitialGuess, const double *maxAbsDev = nullptr, double *dotDev = nullptr)
{
    MGPS_TRY(ensureMixedGrids(h));
    DevLevel &F = h->lv[0];
    if (useInitialGuess && x) {
        // A cycle from an initial guess is x + M (b - A x) (the cycle is linear).  In that form the residual is taken
        // in fp32 and only the correction passes through binary16 -- storing the iterate itself there would put rounding
        // noise of 2^-11 |x| into every residual, which the coarse correction amplifies by |A^-1|: chained cycles would stall
        MGPS_LAUNCH(h, launchStencil(h->stream, OP_RESIDUAL, F.g, F.r, x, b, 0.f, true));
        MGPS_TRY(vcycleMixed(h, nullptr, F.r, false));
        MGPS_LAUNCH(h, launchXpayHalf(h->stream, F.g, x, h->mixResult, h->mixSigma, std::ldexp(1.f, h->mixExp), nullptr, 1.f));
        return MGPS_OK;
    }
    const int nlv = int(h->lv.size());
    const size_t n = F.d.cells();
    const float xs = std::ldexp(1.f, -h->mixExp), xsInv = std::ldexp(1.f, h->mixExp);
    const float omega = h->opt.jacobi_weight;
    // sigma = the power of two that brings max |b| into (1/2, 1]
    if (!maxAbsDev) {
        MGPS_LAUNCH(h, launchReduce(h->stream, 3, F.g, b, nullptr, h->partials, h->mixMax));
        maxAbsDev = h->mixMax;
    }
    MGPS_LAUNCH(h, launchMixSigma(h->stream, maxAbsDev, h->mixSigma));
    void *cur = h->mixX, *other = h->mixTmp;
    // (the first stroke starts from zero, MG.cpp:439-440: like strokeTakesZero, it can take the iterate as zero instead of clearing and reading it)
    const bool gatherDown = dotDev != nullptr && h->dotPartials != nullptr && nlv == 1;
    const bool zeroStart = strokeTakesZero(h, 0, h->lv[0].x, h->lv[0].tmp, b, gatherDown) && (F.g.nx & 3) == 0;
    if (!zeroStart) MGPS_LAUNCH(h, launchZeroActiveHalf(h->stream, F.g, cur));
    const MixScale smooth{h->mixSigma, xs, 1.f};  // the iterate's units: rhs sigma 2^-e b
    const bool gather = dotDev != nullptr && h->dotPartials != nullptr && !h->useGS;  // (the binary16 tile kernels gather nothing: a separate pass below)
    h->dotUsed = 0;
    // (the box form of the band stage, binary16 grids: see smoothStroke; the scratch / snapshot grid is the binary16 residual)
    auto bandStage = [&](bool dot) -> int {  // no snapshot: out of place, band cells copied back
        double *sink = nullptr;
        if (dot && F.bandBoxes.ngroups > 0) {
            sink = h->dotPartials + h->dotUsed;
            h->dotUsed += unsigned(F.bandBoxes.ngroups);
        }
        MGPS_LAUNCH(h, launchBandBox(h->stream, F.g, F.bandBoxes, false, cur, b, h->mixR, nullptr, omega, true, smooth, sink, cur));
        MGPS_LAUNCH(h, launchBandBoxCopy(h->stream, F.g, F.bandBoxes, h->mixR, cur, true));
        return MGPS_OK;
    };
    auto sweep = [&](bool d, bool patchGeneral, bool xZero) -> int {
        unsigned used = 0;
        GridP gs = F.g;
        if (!patchGeneral) gs.nbnd = 0;
        if (h->profiling) MGPS_TRY(profMark(h, true));  // the measurement hook of smoothStroke: event pair around the fine sweep
        MGPS_LAUNCH(h, launchStencilMixed(h->stream, OP_JACOBI, gs, other, xZero ? nullptr : cur, b, omega, smooth, d ? h->dotPartials + h->dotUsed : nullptr, &used));
        if (h->profiling) {
            MGPS_TRY(profMark(h, false));
            ++h->profSweeps;
        }
        h->dotUsed += used;
        return MGPS_OK;
    };
    auto stroke = [&](bool down, bool dot, bool xZero) -> int {
        const int reps = down ? h->opt.pre_sweeps : h->opt.post_sweeps;
        if (reps == 1 && F.bandBoxes.ngroups > 0 && !h->useGS) {  // closure launch (snapshot only), sweep, plain launch: see smoothStroke
            MGPS_LAUNCH(h, launchBandBox(h->stream, F.g, F.bandBoxes, true, xZero ? nullptr : cur, b, nullptr, h->mixR, omega, true, smooth));
            MGPS_TRY(sweep(dot, dot, xZero));
            double *sinkB = nullptr;
            if (dot) {
                sinkB = h->dotPartials + h->dotUsed;
                h->dotUsed += unsigned(F.bandBoxes.ngroups);
            }
            std::swap(cur, other);
            MGPS_LAUNCH(h, launchBandBox(h->stream, F.g, F.bandBoxes, false, h->mixR, b, cur, nullptr, omega, true, smooth, sinkB, cur, true));
            return MGPS_OK;
        }
        MGPS_TRY(bandStage(false));
        for (int rep = 0; rep < reps; ++rep) {
            if (h->useGS) {  // the colour passes of smoothStroke (MG.cpp:466-479 down, 740-751 up), in place on the binary16 iterate
                const int first = down ? 1 : 0, forward = down ? 1 : 0;
                if (h->profiling) MGPS_TRY(profMark(h, true));
                for (int pass = 0; pass < 2; ++pass) {
                    const int odd = pass == 0 ? first : 1 - first;
                    MGPS_LAUNCH(h, launchTiledGSMixed(h->stream, F.g, cur, b, F.pure[odd], F.npure[odd], F.mixed[odd], F.nmixed[odd], F.tileBndStart, forward, smooth));
                }
                if (h->profiling) {
                    MGPS_TRY(profMark(h, false));
                    ++h->profSweeps;
                }
                continue;
            }
            MGPS_TRY(sweep(dot && rep == reps - 1, true, false));
            std::swap(cur, other);
        }
        MGPS_TRY(bandStage(dot));
        return MGPS_OK;
    };
    MGPS_TRY(stroke(true, gather && nlv == 1, zeroStart));
    if (nlv > 1) {
        DevLevel &C = h->lv[1];
        // r~ = 256 r^ = 256 sigma b - 256 2^e (A x~); level 1 receives Restrict(r^) in fp32
        const MixScale res{h->mixSigma, 256.f, 256.f * xsInv};
        MGPS_LAUNCH(h, launchStencilMixed(h->stream, OP_RESIDUAL, F.g, h->mixR, cur, b, 0.f, res));
        MGPS_LAUNCH(h, launchRestrictMixed(h->stream, C.g, C.b, h->mixR, 1.f / 256.f));
        float *corr = nullptr;
        MGPS_TRY(innerCycle(h, 1, &corr));
        MGPS_LAUNCH(h, launchProlongAddMixed(h->stream, F.g, cur, corr, xs));
        MGPS_TRY(stroke(false, gather, false));
    }
    if (gather) {
        MGPS_LAUNCH(h, launchFoldDot(h->stream, h->dotPartials, h->dotUsed, dotDev));
        MGPS_LAUNCH(h, launchScaleResult(h->stream, dotDev, h->mixSigma, xsInv));
    }
    h->mixResult = cur;
    if (dotDev && !gather) MGPS_LAUNCH(h, launchHalfDot(h->stream, F.g, cur, b, h->mixSigma, xsInv, h->partials, dotDev));
    if (x) MGPS_LAUNCH(h, launchFromHalf(h->stream, x, cur, h->mixSigma, xsInv, n));
    return MGPS_OK;
}

// the value a reduction launch left in resultDev, on the host (summed / maximised over the ranks of a slab run)
int fetchReduction(mgps_solver *h, int kind, double *out)
{
    MGPS_HIP(h, hipMemcpyAsync(h->resultHost, h->resultDev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    *out = *h->resultHost;
    if (h->dist) MGPS_COMM(h, h->comm.allreduce(h->comm.user, out, 1, kind <= 1 ? 0 : 1));
    return MGPS_OK;
}

int reduceToHost(mgps_solver *h, int kind, int level, const float *a, const float *b, double *out)
{
    DevLevel &L = h->lv[level];
    MGPS_LAUNCH(h, launchReduce(h->stream, kind, L.g, a, b, h->partials, h->resultDev));
    return fetchReduction(h, kind, out);
}

int ensurePcgGrids(mgps_solver *h, bool needDiag)
{
    for (int q = 0; q < 4; ++q)
        if (!h->pcg[q]) MGPS_TRY(gridAlloc(h, &h->pcg[q], h->lv[0].d));
    if (!h->cgScal) MGPS_TRY(devAlloc(h, &h->cgScal, 8, true));
    if (!h->dotPartials) {  // the fused A.p launch, or the last stroke of a V-cycle: sweep workgroups / tiles + band scatters
        const DevLevel &F = h->lv[0];
        const size_t tiles = size_t(F.npure[0]) + F.npure[1] + F.nmixed[0] + F.nmixed[1];
        const size_t scatters = size_t(std::max(1, h->opt.band_iterations)) * bandScatterBlocks(F.nband) + 2 * size_t(F.bandBoxes.ngroups);
        h->dotCapacity = applyDotPartialCount(F.g) + tiles + scatters + 64 + 2048;
        MGPS_TRY(devAlloc(h, &h->dotPartials, h->dotCapacity, true));
    }
    if (needDiag && !h->dinv) {
        MGPS_TRY(gridAlloc(h, &h->dinv, h->lv[0].d));
        MGPS_LAUNCH(h, launchDiagInverse(h->stream, h->lv[0].g, h->dinv));
    }
    return MGPS_OK;
}

// skipInactive: `out` is one of the solver's own grids, already 0 (r, y) / equal to x (Jacobi) on chunks
// without active cells; the public entry points pass false and get every cell of `out` written
int applyOp(mgps_solver *h, StencilOp op, int level, float *out, float *x, const float *b, bool skipInactive = false)
{
    MGPS_TRY(exchangeGhosts(h, level, x));
    MGPS_LAUNCH(h, launchStencil(h->stream, op, h->lv[level].g, out, x, b, h->opt.jacobi_weight, skipInactive));
    return MGPS_OK;
}

// whole ghost planes of an fp64 grid of the fine level (slab runs)
int exchangeGhosts64(mgps_solver *h, double *a)
{
    if (!h->dist) return MGPS_OK;
    DevLevel &L = h->lv[0];
    const size_t plane = size_t(L.d.nx) * L.d.ny, bytes = plane * sizeof(double);
    const bool lo = h->comm.rank > 0, hi = h->comm.rank < h->comm.size - 1;
    ++h->exchanges;
    MGPS_COMM(h, h->comm.exchange(h->comm.user, lo ? a : nullptr, bytes, lo ? a - plane : nullptr, bytes,
                                  hi ? a + (size_t(L.d.nz) - 1) * plane : nullptr, bytes, hi ? a + size_t(L.d.nz) * plane : nullptr, bytes,
                                  h->stream));
    return MGPS_OK;
}

int interruptRequested(mgps_solver *h, bool *stop);

// device time of a solve: an event pair whose lifetime is the scope's (every early return of pcg / pcg64 passes through it)
struct SolveClock {
    mgps_solver *h;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = false;
    explicit SolveClock(mgps_solver *hh) : h(hh)
    {
        ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventRecord(e0, h->stream) == hipSuccess;
    }
    double stop()
    {
        float ms = 0.f;
        if (ok && hipEventRecord(e1, h->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
        return ms;
    }
    ~SolveClock()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        h->dotTarget = nullptr;
    }
};

// MG-PCG with the CG vectors in fp64 (options.pcg_fp64_vectors): CG.h:18-207 step by step like pcg() below; the
// preconditioner is the same fp32 V-cycle (or diagonal) applied to float(r), x and b are fp32 at the boundary
int pcg64(mgps_solver *h, float *x, const float *b, double tol, int maxIt, bool useMG, mgps_pcg_stats *st)
{
    DevLevel &F = h->lv[0];
    const size_t n = F.d.cells(), plane = size_t(F.d.nx) * F.d.ny;
    MGPS_TRY(ensurePcgGrids(h, !useMG));
    for (double *&g64 : h->cg64)
        if (!g64) {
            double *base = nullptr;
            MGPS_TRY(devAlloc(h, &base, n + 2 * plane, true));
            g64 = base + plane;
        }
    h->dotTarget = nullptr;  // (the gathered <z, r> goes to resultDev here)
    double *x64 = h->cg64[0], *r64 = h->cg64[1], *p64 = h->cg64[2], *t64 = h->cg64[3];
    float *r32 = h->pcg[0], *z = h->pcg[2];
    SolveClock clock(h);  // (destroys its events and resets h->dotTarget on every way out)
    if (!clock.ok) return failH(h, MGPS_ERR_HIP, "hipEventCreate failed");
    auto finish = [&](int outcome) {
        st->outcome = outcome;
        st->solve_ms = clock.stop();
        return MGPS_OK;
    };
    bool gathered = false;
    auto precondition = [&]() -> int {  // z = M float(r)
        gathered = false;
        if (useMG && h->opt.precision == 1) return vcycleMixed(h, z, r32, false);
        if (useMG) {
            MGPS_TRY(vcycle(h, z, r32, false, true, true));
            gathered = h->gatherDot;
            return MGPS_OK;
        }
        MGPS_LAUNCH(h, launchMulMasked(h->stream, F.g, z, r32, h->dinv));
        return MGPS_OK;
    };
    auto zDotR = [&](double *out) -> int {  // <z, float(r)>: r enters the V-cycle rounded, the same rounded r is used here
        if (gathered) return fetchReduction(h, 0, out);
        return reduceToHost(h, 0, 0, z, r32, out);
    };
    double rhs2 = 0;
    MGPS_TRY(reduceToHost(h, 1, 0, b, nullptr, &rhs2));  // CG.h:35
    st->rhs_norm2 = rhs2;
    if (rhs2 == 0) return finish(MGPS_PCG_RHS_ZERO);  // CG.h:36-40
    MGPS_LAUNCH(h, launchWiden(h->stream, x64, x, n));
    double res2 = 0;
    MGPS_TRY(exchangeGhosts64(h, x64));
    MGPS_LAUNCH(h, launchStencil64(h->stream, 1, F.g, r64, x64, b, r32, h->dotPartials, h->dotCapacity, h->resultDev));  // CG.h:50-57
    MGPS_TRY(fetchReduction(h, 1, &res2));
    const double threshold = tol * tol * rhs2;  // CG.h:58
    if (res2 < threshold) {                     // CG.h:60-64
        st->rel_residual = st->rel_residual_recomputed = std::sqrt(res2 / rhs2);
        return finish(MGPS_PCG_ALREADY_CONVERGED);
    }
    MGPS_TRY(precondition());                                              // CG.h:75
    MGPS_LAUNCH(h, launchXpay64(h->stream, F.g, p64, z, 0.0, 1));          // p = z
    double absNew = 0;
    MGPS_TRY(zDotR(&absNew));                                              // CG.h:86
    int it = 0;
    bool converged = false;
    for (; it < maxIt; ++it) {
        bool stop = false;
        MGPS_TRY(interruptRequested(h, &stop));
        if (stop) {
            (void)launchNarrow(h->stream, x, x64, n);  // what the iterations reached so far, as the fp32 loop leaves it
            finish(MGPS_PCG_MAX_ITERATIONS);
            st->iterations = it;
            return failH(h, MGPS_ERR_INTERRUPTED, "mgps_solve_pcg: interrupted");
        }
        double pAp = 0;
        MGPS_TRY(exchangeGhosts64(h, p64));
        MGPS_LAUNCH(h, launchStencil64(h->stream, 0, F.g, t64, p64, nullptr, nullptr, h->dotPartials, h->dotCapacity, h->resultDev));  // CG.h:110-121
        MGPS_TRY(fetchReduction(h, 0, &pAp));
        const double alpha = absNew / pAp;
        MGPS_LAUNCH(h, launchCgUpdate64(h->stream, F.g, x64, p64, r64, t64, alpha, r32, h->dotPartials, h->dotCapacity, h->resultDev));  // CG.h:132-153
        MGPS_TRY(fetchReduction(h, 1, &res2));
        if (h->opt.print_stats && (!h->dist || h->comm.rank == 0)) std::printf("  Iteration: %d  Relative error: %.10g\n", it, std::sqrt(res2 / rhs2));
        if (res2 < threshold) {  // CG.h:161
            converged = true;
            break;
        }
        if (const int prc = precondition()) {  // CG.h:168
            if (prc == MGPS_ERR_INTERRUPTED) {  // (polled inside the V-cycle): hand back what the iterations reached
                (void)launchNarrow(h->stream, x, x64, n);
                finish(MGPS_PCG_MAX_ITERATIONS);
                st->iterations = it;
            }
            return prc;
        }
        const double absOld = absNew;
        MGPS_TRY(zDotR(&absNew));  // CG.h:180
        MGPS_LAUNCH(h, launchXpay64(h->stream, F.g, p64, z, absNew / absOld, 0));  // CG.h:191
    }
    st->iterations = it;
    st->rel_residual = std::sqrt(res2 / rhs2);  // CG.h:199
    double rec2 = 0;
    MGPS_TRY(exchangeGhosts64(h, x64));
    MGPS_LAUNCH(h, launchStencil64(h->stream, 1, F.g, r64, x64, b, r32, h->dotPartials, h->dotCapacity, h->resultDev));  // CG.h:203-205, in fp64
    MGPS_TRY(fetchReduction(h, 1, &rec2));
    st->rel_residual_recomputed = std::sqrt(rec2 / rhs2);
    MGPS_LAUNCH(h, launchNarrow(h->stream, x, x64, n));
    return finish(converged ? MGPS_PCG_CONVERGED : MGPS_PCG_MAX_ITERATIONS);
}

// options.interrupt, polled once per CG iteration; in a slab run the ranks agree (max over ranks) so that nobody is left
// waiting in an exchange for a neighbour that stopped
int interruptRequested(mgps_solver *h, bool *stop)
{
    *stop = false;
    if (!h->opt.interrupt) return MGPS_OK;
    double flag = h->opt.interrupt(h->opt.interrupt_user) ? 1.0 : 0.0;
    if (h->dist) MGPS_COMM(h, h->comm.allreduce(h->comm.user, &flag, 1, 1));
    *stop = flag != 0.0;
    return MGPS_OK;
}

int pcg(mgps_solver *h, float *x, const float *b, double tol, int maxIt, bool useMG, mgps_pcg_stats *st)
{
    DevLevel &F = h->lv[0];
    if (h->opt.pcg_fp64_vectors == 1) {
        mgps_pcg_stats local64{};
        if (!st) st = &local64;
        std::memset(st, 0, sizeof(*st));
        return pcg64(h, x, b, tol, maxIt, useMG, st);
    }
    mgps_pcg_stats local{};
    if (!st) st = &local;
    std::memset(st, 0, sizeof(*st));
    MGPS_TRY(ensurePcgGrids(h, !useMG));
    float *r = h->pcg[0], *p = h->pcg[1], *z = h->pcg[2], *t = h->pcg[3];
    // options.pcg_fp64_vectors = 2: the iterate alone in fp64 (round 4).  What keeps the recomputed residual b - A x of the fp32
    // loop at eps * cond (2.8e-3 at 512^3, 2e-2 at 1024^3 on the free-surface pool, while the pressure is 4e-7 from the oracle's)
    // is the fp32 STORAGE of x: the loop then carries x in fp64 (x += alpha p, 8 B more per cell and iteration), takes the
    // residuals of CG.h:50-51 and 203-205 in fp64 from it and hands back the rounded x; r, p, A p and the V-cycle stay fp32
    // Round 5: the fp64 iterate is touched only where a true residual is taken.  Between two such points the updates alpha p are
    // summed in fp32 IN THE CALLER'S x (free during the solve: the iterate is x64 + x) -- the group-wise update of van der Vorst &
    // Ye: the rounding of that sum is relative to the sum, eight updates small against the iterate -- so an iteration moves the
    // bytes of the fp32 loop; the replacement pass reads x64 + x at its seven points, leaves float(b - A (x64 + x)) and the flushed
    // iterate in the second fp64 grid (the two swap), and the next group starts with x = alpha p.  Per iteration 3 B per cell
    // more than the fp32 loop instead of 10.
    const bool wideX = h->opt.pcg_fp64_vectors == 2 && !(useMG && h->opt.precision == 1);
    double *x64 = nullptr, *x64Spare = nullptr;
    int grouped = 0;  // updates summed in x since the last flush
    if (wideX) {
        const size_t plane = size_t(F.d.nx) * F.d.ny;
        for (int q = 0; q < 2; ++q)
            if (!h->cg64[q]) {
                double *base = nullptr;
                MGPS_TRY(devAlloc(h, &base, F.d.cells() + 2 * plane, true));
                h->cg64[q] = base + plane;
            }
        x64 = h->cg64[0];
        x64Spare = h->cg64[1];
    }
    SolveClock clock(h);  // (destroys its events and resets h->dotTarget on every way out)
    if (!clock.ok) return failH(h, MGPS_ERR_HIP, "hipEventCreate failed");
    bool widened = false;  // (x64 holds the caller's iterate: from then on x is the sum of the pending updates)
    auto finish = [&](int outcome) {
        if (wideX && widened) (void)launchNarrowSum(h->stream, F.g, x, x64, grouped > 0);  // (what the iterations reached, rounded once)
        st->outcome = outcome;
        st->solve_ms = clock.stop();
        return MGPS_OK;
    };
    // r = float(b - A x) from the wide iterate x64 (+ the pending updates in x, which are flushed on the way): it leaves float(r) in r
    // and |r|^2 on the device
    auto wideResidual = [&](double *res2) -> int {
        MGPS_TRY(exchangeGhosts64(h, x64));
        if (grouped > 0) {
            MGPS_TRY(exchangeGhosts(h, 0, x));
            MGPS_LAUNCH(h, launchStencil64(h->stream, 2, F.g, x64Spare, x64, b, r, h->dotPartials, h->dotCapacity, h->resultDev, x));
            std::swap(x64, x64Spare);
            grouped = 0;
        } else
            MGPS_LAUNCH(h, launchStencil64(h->stream, 1, F.g, nullptr, x64, b, r, h->dotPartials, h->dotCapacity, h->resultDev));
        return fetchReduction(h, 1, res2);
    };
    // dst = M src; gathered: <dst, src> is already in h->resultDev (a by-product of the V-cycle's last stroke)
    bool gathered = false;
    auto precondition = [&](float *dst, const float *src) -> int {
        gathered = false;
        if (useMG && h->opt.precision == 1) return vcycleMixed(h, dst, src, false);  // (the first application, p = M r: CG.h:75)
        if (useMG) {
            MGPS_TRY(vcycle(h, dst, src, false, true, true));  // Plug.cpp:468-472 (dst = p or z: grids of the solver)
            gathered = h->gatherDot;
            return MGPS_OK;
        }
        MGPS_LAUNCH(h, launchMulMasked(h->stream, F.g, dst, src, h->dinv));  // Plug.cpp:555-606
        return MGPS_OK;
    };
    static const bool checkGathered = [] {  // MGPS_CHECK_FUSED_DOT=1: compare every gathered <z, r> with a separate reduction
        const char *e = getenv("MGPS_CHECK_FUSED_DOT");
        return e && e[0] == '1';
    }();
    auto dotWithResidual = [&](const float *v, double *out) -> int {  // <v, r> right after precondition(v, r)
        if (!gathered) return reduceToHost(h, 0, 0, v, r, out);
        MGPS_TRY(fetchReduction(h, 0, out));
        if (checkGathered) {
            double ref = 0;
            MGPS_TRY(reduceToHost(h, 0, 0, v, r, &ref));
            if (!(std::fabs(*out - ref) <= 1e-9 * std::fabs(ref) + 1e-300))
                return failH(h, MGPS_ERR_HIP, "gathered <z, r> = " + std::to_string(*out) + " differs from the separate reduction " + std::to_string(ref));
        }
        return MGPS_OK;
    };

    double rhs2 = 0;
    MGPS_TRY(reduceToHost(h, 1, 0, b, nullptr, &rhs2));  // CG.h:35
    st->rhs_norm2 = rhs2;
    if (rhs2 == 0) return finish(MGPS_PCG_RHS_ZERO);  // CG.h:36-40
    double res2 = 0;
    if (wideX) {
        MGPS_LAUNCH(h, launchWiden(h->stream, x64, x, F.d.cells()));
        // (the second fp64 grid: zero since its allocation wherever no pass writes -- inactive cells, which read as 0 in every grid)
        widened = true;
        MGPS_TRY(wideResidual(&res2));  // CG.h:50-57 (r = float of the fp64 residual)
    } else {
        MGPS_TRY(applyOp(h, OP_RESIDUAL, 0, r, x, b, true));  // CG.h:50-51
        MGPS_TRY(reduceToHost(h, 1, 0, r, nullptr, &res2));  // CG.h:57
    }
    const double threshold = tol * tol * rhs2;           // CG.h:58
    if (res2 < threshold) {                              // CG.h:60-64
        st->rel_residual = st->rel_residual_recomputed = std::sqrt(res2 / rhs2);
        return finish(MGPS_PCG_ALREADY_CONVERGED);
    }
    // Single-device runs keep alpha and beta on the device (launchCgScalars): the reductions leave <p, A p> and
    // <z, r> there, the update and xpay kernels read them, and only |r|^2 -- the convergence test of CG.h:161 -- is
    // fetched: one host round trip per iteration instead of three.  Slab runs sum over ranks through the host.
    // Slab runs do the same when the transport can all-reduce device doubles on the solver's stream (mgps_comm::
    // allreduce_device: ncclAllReduce for the RCCL transport): every rank's reduction leaves its share on the device, the
    // collective sums it there, the scalar kernel divides.
    const bool devScal = (!h->dist || h->comm.allreduce_device != nullptr) && !checkGathered;
    double *scal = h->cgScal;
    float *betaDev = reinterpret_cast<float *>(h->cgScal + 4);
    auto sumOverRanks = [&](double *dev) -> int {  // (slab runs: the value a reduction just left on the device, summed in place)
        if (h->dist) MGPS_COMM(h, h->comm.allreduce_device(h->comm.user, dev, 1, 0, h->stream));
        return MGPS_OK;
    };
    auto dotToDevice = [&](const float *v) -> int {  // scal[3] = <v, r> right after precondition(v, r)
        if (!gathered) MGPS_LAUNCH(h, launchReduce(h->stream, 0, F.g, v, r, h->partials, scal + 3));
        return sumOverRanks(scal + 3);
    };
    h->dotTarget = devScal ? scal + 3 : nullptr;
    MGPS_LAUNCH(h, launchZero(h->stream, p, F.d.cells()));  // CG.h:69
    MGPS_TRY(precondition(p, r));                         // CG.h:75
    double absNew = 0;
    if (devScal) {
        MGPS_TRY(dotToDevice(p));
        MGPS_LAUNCH(h, launchCgScalars(h->stream, scal, betaDev, 1));
    } else
        MGPS_TRY(dotWithResidual(p, &absNew));  // CG.h:86
    MGPS_LAUNCH(h, launchZero(h->stream, z, F.d.cells()));
    MGPS_LAUNCH(h, launchZero(h->stream, t, F.d.cells()));
    int it = 0;
    bool converged = false, rFresh = false;
    double groupStart2 = res2;  // |r|^2 (true) where the current group of fp32 updates began
    constexpr int wideReplaceEvery = 8;  // iterations between two residual replacements (every 4: +2 % time; every 16: one more iteration -- LABNOTES R4)
    for (; it < maxIt; ++it) {
        bool stop = false;
        MGPS_TRY(interruptRequested(h, &stop));
        if (stop) {
            finish(MGPS_PCG_MAX_ITERATIONS);
            st->iterations = it;
            return failH(h, MGPS_ERR_INTERRUPTED, "mgps_solve_pcg: interrupted");
        }
        // t = A p (CG.h:110) and <p, A p> (CG.h:121) in one pass over p
        MGPS_TRY(exchangeGhosts(h, 0, p));
        MGPS_LAUNCH(h, launchApplyDot(h->stream, F.g, t, p, h->dotPartials, devScal ? scal + 1 : h->resultDev));
        if (devScal) MGPS_TRY(sumOverRanks(scal + 1));
        double alpha = 0;
        if (!devScal) {
            double pAp = 0;
            MGPS_TRY(fetchReduction(h, 0, &pAp));
            alpha = absNew / pAp;  // CG.h:121
        }
        // x += alpha p (CG.h:132), r -= alpha t (143) and |r|^2 (153) in one pass over the grids
        const bool mixed = useMG && h->opt.precision == 1;  // the pass also leaves max |r| for the cycle's normalisation
        MGPS_LAUNCH(h, launchCgUpdate(h->stream, F.g, x, p, r, t, float(alpha), h->partials, h->resultDev, devScal ? scal : nullptr,
                                      mixed ? h->mixMax : nullptr, nullptr, wideX && grouped == 0));
        if (wideX) ++grouped;
        if (devScal && h->dist) {  // |r|^2 summed on the device too: the fetch below is then the iteration's only host round trip
            MGPS_TRY(sumOverRanks(h->resultDev));
            MGPS_HIP(h, hipMemcpyAsync(h->resultHost, h->resultDev, sizeof(double), hipMemcpyDeviceToHost, h->stream));
            MGPS_HIP(h, hipStreamSynchronize(h->stream));
            res2 = *h->resultHost;
        } else
            MGPS_TRY(fetchReduction(h, 1, &res2));
        // fp64 iterate: the recurrence r -= alpha A p drifts from b - A x by the rounding of the fp32 products it sums (5e-4 of |b|
        // after 23 iterations on the 512^3 pool).  Every few iterations, and whenever the recurrence claims convergence, r is
        // REPLACED by the true residual float(b - A x) taken in fp64 from the wide iterate (residual replacement, van der Vorst
        // & Ye 2000): the drift a solve ends with is what the last few -- by then tiny -- updates added, and "converged" is
        // only ever said of a true residual.  One fp64 stencil pass per replacement.
        rFresh = false;
        bool restartDir = false;
        // (a group ends after wideReplaceEvery updates, when the recurrence claims convergence, or when the residual has dropped a
        // hundredfold since the group began: the rounding of a group's fp32 sum is relative to its FIRST update, so a group
        // that spans many decades of the residual leaves the true residual at eps * cond times where it began -- fast solves,
        // the cube from the zero guess at 0.15 per iteration, flush every other iteration; the pool's 0.66 never triggers it)
        if (wideX && (res2 < threshold || grouped >= wideReplaceEvery || res2 < 1e-4 * groupStart2)) {
            const double recur2 = res2;
            MGPS_TRY(wideResidual(&res2));
            groupStart2 = res2;
            rFresh = true;
            // The true residual far above the recurrence's (more than twice its norm): the direction p was built for a residual
            // that no longer exists, and beta = <z, r>_new / <z, r>_old would blow it up -- restart with p = z.  It happens when
            // a group of fp32 updates was large against what fp32 resolves of it: the first group of a solve from the zero
            // guess IS the solution, so the true residual after it sits at eps * cond * |b| whatever the recurrence says
            // (64^3 box, tolerance 1e-6: 4.5e-5 against 9e-7 at the first claim; without the restart the loop then crawls
            // for hundreds of iterations).  The groups after it start from that residual and are that much smaller.
            restartDir = res2 > 4.0 * recur2 && res2 >= threshold;
        }
        if (h->opt.print_stats && (!h->dist || h->comm.rank == 0))
            std::printf("  Iteration: %d  Relative error: %.10g\n", it, std::sqrt(res2 / rhs2));
        if (res2 < threshold) {  // CG.h:161 -- the counter is not advanced on the exit pass
            converged = true;
            break;
        }
        if (mixed) {  // z = M r stays in binary16: <z, r> and p = z + beta p read it there (CG.h:168-191)
            const float zScale = std::ldexp(1.f, h->mixExp);
            if (devScal) {  // <z, r> gathered by the cycle's last stroke
                MGPS_TRY(vcycleMixed(h, nullptr, r, false, h->mixMax, scal + 3));
                MGPS_LAUNCH(h, launchCgScalars(h->stream, scal, betaDev, 0));
                MGPS_LAUNCH(h, launchXpayHalf(h->stream, F.g, p, h->mixResult, h->mixSigma, zScale, betaDev, 0.f));
            } else {
                MGPS_TRY(vcycleMixed(h, nullptr, r, false, h->mixMax));
                const double absOld = absNew;
                MGPS_LAUNCH(h, launchHalfDot(h->stream, F.g, h->mixResult, r, h->mixSigma, zScale, h->partials, h->resultDev));
                MGPS_TRY(fetchReduction(h, 0, &absNew));
                MGPS_LAUNCH(h, launchXpayHalf(h->stream, F.g, p, h->mixResult, h->mixSigma, zScale, nullptr, float(absNew / absOld)));
            }
            continue;
        }
        MGPS_TRY(precondition(z, r));  // CG.h:168
        if (devScal) {
            MGPS_TRY(dotToDevice(z));                                                // CG.h:180
            MGPS_LAUNCH(h, launchCgScalars(h->stream, scal, betaDev, restartDir ? 2 : 0));
            MGPS_LAUNCH(h, launchXpay(h->stream, F.g, p, z, p, betaDev, 0.f));       // CG.h:191
        } else {
            const double absOld = absNew;
            MGPS_TRY(dotWithResidual(z, &absNew));  // CG.h:180
            const double beta = restartDir ? 0.0 : absNew / absOld;
            MGPS_LAUNCH(h, launchXpay(h->stream, F.g, p, z, p, nullptr, float(beta)));  // CG.h:191
        }
    }
    h->dotTarget = nullptr;
    st->iterations = it;
    st->rel_residual = std::sqrt(res2 / rhs2);      // CG.h:199
    double rec2 = 0;
    if (wideX && rFresh) rec2 = res2;  // (the loop's last test was on the true residual of this iterate)
    else if (wideX) MGPS_TRY(wideResidual(&rec2));  // CG.h:203-205 on the iterate the loop carried
    else {
        MGPS_TRY(applyOp(h, OP_RESIDUAL, 0, r, x, b, true));  // CG.h:203-204
        MGPS_TRY(reduceToHost(h, 1, 0, r, nullptr, &rec2));
    }
    st->rel_residual_recomputed = std::sqrt(rec2 / rhs2);  // CG.h:205
    return finish(converged ? MGPS_PCG_CONVERGED : MGPS_PCG_MAX_ITERATIONS);
}

// ---- construction --------------------------------------------------------------------------------

// The plane-marching sweep visits blocks of 256 x 16 x zc cells, the quad sweep runs of 1024 .. 32 cells: where the liquid
// fills a small part of the grid (a 480^3 simulation inside the reference's 1024^3 power-of-two expansion: 57 M of 1074 M
// cells) the blocks hold twice the cells of the runs, and the few percent the plane kernel gains per visited cell on planes
// beyond 4 MiB (round 1 measured 9 % at 1024^2, before the quad kernel caught up there) are lost many times over: measured on
// that case, V-cycle 2.48 ms through the quad kernel, 3.07 ms through the plane kernel.  The level goes to whichever costs less, cells visited x cost per cell (runCostFactor; plane 0.92).  0 = the quad
// sweep (options.stencil_path = 2 still forces the plane kernel).
int planeZcFor(const mgps_options &o, int planeZc, size_t nplaneBlocks, size_t nchunks, int chunkCells)
{
    if (!planeZc || o.stencil_path == 2 || forcedStencilPath() == 2) return planeZc;  // (MGPS_STENCIL=plane: the A/B switch of launchStencil)
    const double planeCost = double(nplaneBlocks) * 256.0 * kPlaneRows * planeZc * 0.92, runCost = double(nchunks) * chunkCells * runCostFactor(chunkCells);
    return planeCost > runCost ? 0 : planeZc;
}

// upload one level built by buildSlabLevel
// set-up stage timings on stdout when options.print_stats is set (the reference's ctor prints its four
// stage times unconditionally, MG.cpp:183, 256, 284, 415)
struct StageClock {
    bool on;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit StageClock(bool enabled) : on(enabled) {}
    void lap(const char *what, int level = -1)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        if (level >= 0) std::printf("  mgps set-up: %-28s level %d  %8.1f ms\n", what, level, std::chrono::duration<double, std::milli>(t1 - t0).count());
        else std::printf("  mgps set-up: %-28s          %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = std::chrono::steady_clock::now();
    }
};

// codesPreloaded: the caller allocated L.codes, copies the labels into it itself and patches the simple cells afterwards
// The boxes of a level once info / list / general are on the device, whoever built them: the lists compacted
// (compactBandBoxLists) and the launch order of the groups (orderBandBoxes)
// dims: the grid info[0] counts cells of (a cut slab level: its label buffer)
int finishBandBoxes(mgps_solver *h, DevLevel &L, hipStream_t s, const Dims *dims = nullptr)
{
    BandBoxesDev &bx = L.bandBoxes;
    {  // the lists as the kernels walk them: no class-2 entries, class-12 entries last (compactBandBoxLists)
        uint32_t *list2 = nullptr;
        MGPS_TRY(devAlloc(h, &list2, bx.listCount, false));
        size_t count2 = 0;
        const int e = compactBandBoxLists(s, bx.info, bx.list, bx.ngroups, list2, &count2);
        if (e != 0) {
            (void)cacheFree(list2);
            return failH(h, MGPS_ERR_HIP, std::string("compactBandBoxLists: ") + hipGetErrorString(hipError_t(e)));
        }
        (void)cacheFree(bx.list);
        bx.list = list2;
        bx.listCount = count2;
    }
    if (bx.ngroups < 64) return MGPS_OK;  // (fewer groups than a chiplet has in flight)
    int32_t *info2 = nullptr;
    MGPS_TRY(devAlloc(h, &info2, size_t(kBoxInfoInts) * size_t(bx.ngroups), false));
    const int e = orderBandBoxes(s, dims ? *dims : L.d, bx.info, bx.ngroups, info2);
    if (e == int(hipErrorOutOfMemory) || e == int(hipErrorInvalidValue)) {  // no room for the sort (or a level past its key space): the builders' order stays
        (void)cacheFree(info2);
        return MGPS_OK;
    }
    if (e != 0) {
        (void)cacheFree(info2);
        return failH(h, MGPS_ERR_HIP, std::string("orderBandBoxes: ") + hipGetErrorString(hipError_t(e)));
    }
    (void)cacheFree(bx.info);
    bx.info = info2;
    return MGPS_OK;
}

int uploadLevel(mgps_solver *h, DevLevel &L, const HostLevel &HL, int z0, int z1, int globalNz, bool withWeights,
                bool workGrids, bool xbGrids, bool codesPreloaded = false)
{
    L.d = HL.d;
    L.z0 = z0;
    L.z1 = z1;
    const size_t plane = size_t(L.d.nx) * L.d.ny;
    if (!codesPreloaded) {  // cell codes: ghost plane | owned planes | ghost plane of labels, then the simple BOUNDARY cells patched in
        const size_t owned = size_t(L.d.nz) * plane;
        MGPS_TRY(devAlloc(h, &L.codes, owned + 2 * plane, false));
        MGPS_HIP(h, hipMemcpy(L.codes + plane, HL.ownedLabels, owned, hipMemcpyHostToDevice));
        if (HL.ghostLoLabels) MGPS_HIP(h, hipMemcpy(L.codes, HL.ghostLoLabels, plane, hipMemcpyHostToDevice));
        else MGPS_HIP(h, hipMemset(L.codes, MGPS_EXTERIOR_CELL, plane));
        if (HL.ghostHiLabels) MGPS_HIP(h, hipMemcpy(L.codes + plane + owned, HL.ghostHiLabels, plane, hipMemcpyHostToDevice));
        else MGPS_HIP(h, hipMemset(L.codes + plane + owned, MGPS_EXTERIOR_CELL, plane));
    }
    MGPS_TRY(devUpload(h, &L.band, HL.bandDev));
    MGPS_TRY(devUpload(h, &L.rows, HL.rows));
    MGPS_TRY(devUpload(h, &L.bandDiag, HL.bandDiag));
    L.nband = int(HL.bandDev.size());
    L.nbndGeneral = int(HL.numBoundary);
    if (!codesPreloaded) MGPS_LAUNCH(h, launchPatchSimpleCodes(nullptr, L.codes + plane, L.band, L.bandDiag, L.nbndGeneral, L.nband));
    MGPS_TRY(devAlloc(h, &L.bandTmp, HL.bandDev.size(), false));
    MGPS_TRY(devUpload(h, &L.pure[0], HL.pureEven));
    MGPS_TRY(devUpload(h, &L.pure[1], HL.pureOdd));
    MGPS_TRY(devUpload(h, &L.mixed[0], HL.mixedEven));
    MGPS_TRY(devUpload(h, &L.mixed[1], HL.mixedOdd));
    L.npure[0] = int(HL.pureEven.size());
    L.npure[1] = int(HL.pureOdd.size());
    L.nmixed[0] = int(HL.mixedEven.size());
    L.nmixed[1] = int(HL.mixedOdd.size());
    MGPS_TRY(devUpload(h, &L.tileBndStart, HL.tileBndStart));
    if (h->dist) {
        for (int q = 0; q < 4; ++q) {
            MGPS_TRY(devUpload(h, &L.bandPlane[q], HL.bandPlane[q]));
            L.nbandPlane[q] = int(HL.bandPlane[q].size());
            MGPS_TRY(devAlloc(h, &L.packBuf[q], HL.bandPlane[q].size(), true));
        }
        L.hasBandPlanes = true;
    }
    MGPS_TRY(devUpload(h, &L.chunks, HL.chunks));
    MGPS_TRY(devUpload(h, &L.planeBlocks, HL.planeBlocks));
    // band passes fuse only where no ghost exchange has to happen between them
    const bool cut = h->dist && (z0 > 0 || z1 < globalNz);
    if (!cut && h->opt.fuse_band_passes && h->opt.band_iterations >= 1 && h->opt.band_iterations <= kBandMaxDepth && !HL.bandDev.empty() && boxPlaneFits(L.d)) {
        StageClock gclock(h->opt.print_stats != 0);
        BandBoxes bx;
        buildBandBoxes(HL, h->opt.band_iterations, bx);
        gclock.lap("  (band boxes alone)");
        if (bx.groups() == 0) return failH(h, MGPS_ERR_INTERNAL, "band boxes: the builder failed on level of " + std::to_string(L.d.nx) + " cells in x");
        L.bandBoxes.depth = bx.depth;
        L.bandBoxes.ngroups = int(bx.groups());
        L.bandBoxes.listCount = bx.list.size();
        L.bandBoxes.generalInts = bx.general.size();
        L.bandBoxes.anyGeneral = !bx.general.empty();
        MGPS_TRY(devUpload(h, &L.bandBoxes.info, bx.info));
        MGPS_TRY(devUpload(h, &L.bandBoxes.list, bx.list));
        MGPS_TRY(devUpload(h, &L.bandBoxes.general, bx.general));
        MGPS_TRY(finishBandBoxes(h, L, h->stream));
    }
    if (xbGrids) {
        MGPS_TRY(gridAlloc(h, &L.x, L.d));
        MGPS_TRY(gridAlloc(h, &L.b, L.d));
    }
    if (workGrids) {
        MGPS_TRY(gridAlloc(h, &L.r, L.d));
        MGPS_TRY(gridAlloc(h, &L.tmp, L.d));
    }
    L.g = GridP{L.d.nx,
                L.d.ny,
                L.d.nz,
                L.codes + plane,
                withWeights ? h->w[0] : nullptr,
                withWeights ? h->w[1] : nullptr,
                withWeights ? h->w[2] : nullptr,
                L.band,
                L.rows,
                int(HL.numBoundary),
                L.bandDiag,
                (h->dist && z0 > 0) ? 1 : 0,
                (h->dist && z1 < globalNz) ? 1 : 0,
                L.chunks,
                int(HL.chunks.size()),
                HL.chunkCells,
                HL.planeZc ? L.planeBlocks : nullptr,
                int(HL.planeBlocks.size()),
                HL.planeZc,
                L.d.cells() * 3 * sizeof(float) > (size_t(256) << 20) ? 1 : 0,
                h->opt.stencil_path};
    L.g.planeZc = planeZcFor(h->opt, HL.planeZc, HL.planeBlocks.size(), HL.chunks.size(), HL.chunkCells);
    if (!L.g.planeZc) L.g.planeBlocks = nullptr;
    L.edgeChunks = HL.edgeChunks;
    L.edgePlaneBlocks = HL.edgePlaneBlocks;
    return MGPS_OK;
}

// ---- coarsest level past kHostCoarseMax unknowns: factorised and inverted on the device ---------------------------------
// hipSOLVER through dlopen (like librccl: the library links against nothing but the HIP runtime): potrf + potri in fp64 on the
// dense matrix.  BASELINE configs 3 / 5 as SURVEY 8(d) states them have 5 levels at 512^3: a 32^3 coarsest level, ~27 000
// unknowns, 5.8 GB of doubles for the factorisation, 2.9 GB for the fp32 inverse the solve multiplies with.  The last inverse
// is kept by label pattern (a plugin that rebuilds its solver every sub-step, Plug.cpp:463, rarely changes the coarsest level).
struct HipSolver {
    void *lib = nullptr;
    int (*create)(void **) = nullptr;
    int (*destroy)(void *) = nullptr;
    int (*setStream)(void *, hipStream_t) = nullptr;
    int (*potrfSize)(void *, int, int, double *, int, int *) = nullptr;
    int (*potrf)(void *, int, int, double *, int, double *, int, int *) = nullptr;
    int (*potriSize)(void *, int, int, double *, int, int *) = nullptr;
    int (*potri)(void *, int, int, double *, int, double *, int, int *) = nullptr;
    bool ok = false;
    HipSolver()
    {
        for (const char *name : {"libhipsolver.so", "libhipsolver.so.1", "/opt/rocm/lib/libhipsolver.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return;
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        create = reinterpret_cast<decltype(create)>(sym("hipsolverCreate"));
        destroy = reinterpret_cast<decltype(destroy)>(sym("hipsolverDestroy"));
        setStream = reinterpret_cast<decltype(setStream)>(sym("hipsolverSetStream"));
        potrfSize = reinterpret_cast<decltype(potrfSize)>(sym("hipsolverDpotrf_bufferSize"));
        potrf = reinterpret_cast<decltype(potrf)>(sym("hipsolverDpotrf"));
        potriSize = reinterpret_cast<decltype(potriSize)>(sym("hipsolverDpotri_bufferSize"));
        potri = reinterpret_cast<decltype(potri)>(sym("hipsolverDpotri"));
        ok = create && destroy && setStream && potrfSize && potrf && potriSize && potri;
    }
};
struct DeviceInverse {
    float *p = nullptr;
    int n = 0, device = 0;
    Dims d;
    std::vector<uint8_t> labels;
    ~DeviceInverse() { (void)cacheFree(p); }
};
// The device-built inverses are kept past their solvers (the plugin makes a solver per sub-step, and a 32^3 coarsest level costs a
// potrf + potri): the most recent kDevInverseKeptPerDevice label patterns of every device, most recent first.  They are
// outside the block cache while kept (n * n floats: 0.7 GB for config 5, up to 4.3 GB at the 32768-unknown cap), so
// mgps_trim_device_cache drops them too -- a solver that still uses one keeps it alive through its shared_ptr -- and an entry
// pushed out of the list goes back to the block cache, where the LRU eviction and the cap see it.
constexpr size_t kDevInverseKeptPerDevice = 2;
std::mutex gDevInverseGuard;
std::vector<std::shared_ptr<DeviceInverse>> gDevInverseKept;
void dropKeptInverses()
{
    std::vector<std::shared_ptr<DeviceInverse>> drop;
    {
        std::lock_guard<std::mutex> lock(gDevInverseGuard);
        drop.swap(gDevInverseKept);
    }
}  // (the destructors run here, outside the lock: cacheFree takes the block cache's own)

int buildDeviceInverse(mgps_solver *h)
{
    mgps_hierarchy *hier = h->hier;
    const HostLevel &C = hier->lv[size_t(hier->levels - 1)];
    const int n = hier->coarseN;
    {
        std::lock_guard<std::mutex> lock(gDevInverseGuard);
        for (size_t q = 0; q < gDevInverseKept.size(); ++q) {
            const std::shared_ptr<DeviceInverse> k = gDevInverseKept[q];
            if (k->n == n && k->device == h->device && k->d.nx == C.d.nx && k->d.ny == C.d.ny && k->d.nz == C.d.nz && k->labels.size() == C.labels.size() &&
                std::memcmp(k->labels.data(), C.labels.data(), C.labels.size()) == 0) {
                std::rotate(gDevInverseKept.begin(), gDevInverseKept.begin() + ptrdiff_t(q), gDevInverseKept.begin() + ptrdiff_t(q) + 1);  // most recent first
                h->cinvShared = k;
                h->cinv = k->p;
                return MGPS_OK;
            }
        }
    }
    static HipSolver *solver = new HipSolver();  // (never unloaded: process tear-down order)
    if (!solver->ok) {  // no libhipsolver: the host's banded factor after all, slowly, while the level is small enough for it
        const int rc = hostCoarseFallback(hier);
        if (rc != MGPS_OK) return failH(h, rc, lastGlobalError());
        hier->buildDenseInverse();
        return devUpload(h, &h->cinv, hier->coarseInverse);
    }
    const int kLower = 122;  // HIPSOLVER_FILL_MODE_LOWER
    double *A = nullptr, *work = nullptr;
    int32_t *index = nullptr, *cells = nullptr;
    uint8_t *lab = nullptr;
    int *info = nullptr;
    void *handle = nullptr;
    auto cleanup = [&] {
        (void)hipStreamSynchronize(h->stream);
        (void)cacheFree(A);
        (void)cacheFree(work);
        (void)cacheFree(index);
        (void)cacheFree(cells);
        (void)cacheFree(lab);
        (void)cacheFree(info);
        if (handle) (void)solver->destroy(handle);
    };
    auto fail = [&](const std::string &what) {
        cleanup();
        return failH(h, MGPS_ERR_COARSE_FACTOR, "coarsest-level factorisation on the device: " + what);
    };
    const size_t plane = size_t(C.d.nx) * C.d.ny;
    if (devAlloc(h, &A, size_t(n) * n, true) != MGPS_OK || devUpload(h, &index, hier->coarseIndex) != MGPS_OK || devUpload(h, &cells, hier->coarseCell) != MGPS_OK ||
        devAlloc(h, &info, 1, true) != MGPS_OK || devAlloc(h, &lab, C.labels.size() + 2 * plane, false) != MGPS_OK)
        return fail("allocation");
    if (hipMemsetAsync(lab, MGPS_EXTERIOR_CELL, C.labels.size() + 2 * plane, h->stream) != hipSuccess ||
        hipMemcpyAsync(lab + plane, C.labels.data(), C.labels.size(), hipMemcpyHostToDevice, h->stream) != hipSuccess)
        return fail("label upload");
    if (launchCoarseAssemble(h->stream, n, C.d.nx, C.d.ny, cells, index, lab + plane, A) != 0) return fail("assembly launch");
    if (solver->create(&handle) != 0 || solver->setStream(handle, h->stream) != 0) return fail("hipsolverCreate");
    int lw1 = 0, lw2 = 0;
    if (solver->potrfSize(handle, kLower, n, A, n, &lw1) != 0 || solver->potriSize(handle, kLower, n, A, n, &lw2) != 0) return fail("workspace query");
    const int lwork = std::max(std::max(lw1, lw2), 1);
    if (devAlloc(h, &work, size_t(lwork), false) != MGPS_OK) return fail("workspace allocation");
    int hinfo = 0;
    if (solver->potrf(handle, kLower, n, A, n, work, lwork, info) != 0) return fail("potrf");
    if (hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess || hinfo != 0) {
        cleanup();
        return failH(h, MGPS_ERR_COARSE_FACTOR, "coarsest-level matrix is not positive definite");
    }
    if (solver->potri(handle, kLower, n, A, n, work, lwork, info) != 0) return fail("potri");
    auto inv = std::make_shared<DeviceInverse>();
    if (devAlloc(h, &inv->p, size_t(n) * n, false) != MGPS_OK) return fail("inverse allocation");
    if (launchCoarseNarrow(h->stream, n, A, inv->p) != 0) return fail("narrow launch");
    if (hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess || hinfo != 0)
        return fail("potri reported " + std::to_string(hinfo));
    cleanup();
    inv->n = n;
    inv->device = h->device;
    inv->d = C.d;
    inv->labels.assign(C.labels.begin(), C.labels.end());
    h->cinvShared = inv;
    h->cinv = inv->p;
    std::vector<std::shared_ptr<DeviceInverse>> pushedOut;  // (released after the lock)
    {
        std::lock_guard<std::mutex> lock(gDevInverseGuard);
        gDevInverseKept.insert(gDevInverseKept.begin(), inv);
        size_t sameDevice = 0;
        for (size_t q = 0; q < gDevInverseKept.size();) {
            if (gDevInverseKept[q]->device == h->device && ++sameDevice > kDevInverseKeptPerDevice) {
                pushedOut.push_back(gDevInverseKept[q]);
                gDevInverseKept.erase(gDevInverseKept.begin() + ptrdiff_t(q));
            } else
                ++q;
        }
    }
    return MGPS_OK;
}

int commonDeviceState(mgps_solver *h, bool needCoarseSolver)
{
    mgps_hierarchy *hier = h->hier;
    if (needCoarseSolver) {
        h->cn = hier->coarseN;
        if (hier->coarseOnDevice) MGPS_TRY(buildDeviceInverse(h));
        else {
            hier->buildDenseInverse();
            MGPS_TRY(devUpload(h, &h->cinv, hier->coarseInverse));
        }
        MGPS_TRY(devUpload(h, &h->ccells, hier->coarseCell));
        MGPS_TRY(devAlloc(h, &h->cvec, size_t(h->cn), true));
    }
    MGPS_TRY(devAlloc(h, &h->partials, size_t(2 * kReducePartials), true));
    MGPS_TRY(devAlloc(h, &h->resultDev, 1, true));
    if (hipHostMalloc(reinterpret_cast<void **>(&h->resultHost), sizeof(double)) != hipSuccess)
        return failH(h, MGPS_ERR_ALLOC, "pinned allocation failed");
    // the active x range of every level (GridP::xlo): one pass over the codes, the answers come back with the set-up's last sync.
    // (Tried on top of it and dropped: 128 x 32 tiles from the start of the range for the plane-marching sweep, so that no lane owns
    // only padding -- 896 active cells are 7 such tiles against 3.5 of the 256 x 16 ones: 1.949 -> 1.919 ms per sweep at 1024^3,
    // 94.0 -> 94.4 cycles/s.  The idle lanes of a partly active tile cost next to nothing; what the range saves is the bytes.)
    static const bool rangeSkip = [] {  // MGPS_X_RANGE=0: sweeps visit the whole rows of their runs / blocks (A/B timing)
        const char *e = getenv("MGPS_X_RANGE");
        return !(e && e[0] == '0');
    }();
    const size_t nlv = h->lv.size();
    std::vector<int> range(2 * nlv);
    int *rangeDev = nullptr;
    for (size_t l = 0; l < nlv; ++l) {
        range[2 * l] = h->lv[l].g.nx;
        range[2 * l + 1] = -1;
        h->lv[l].g.xlo = 0;
        h->lv[l].g.xhi = h->lv[l].g.nx;
    }
    if (rangeSkip && nlv > 0) {
        MGPS_TRY(devAlloc(h, &rangeDev, 2 * nlv, false));
        // (the null stream, like the set-up kernels that wrote the codes)
        MGPS_HIP(h, hipMemcpyAsync(rangeDev, range.data(), range.size() * sizeof(int), hipMemcpyHostToDevice, nullptr));
        for (size_t l = 0; l < nlv; ++l) {
            const GridP &g = h->lv[l].g;
            if (g.lab && (g.nx & 3) == 0) MGPS_LAUNCH(h, launchActiveXRange(nullptr, g.lab, g.nx, size_t(g.nx) * g.ny * g.nz, rangeDev + 2 * l));
        }
        MGPS_HIP(h, hipMemcpyAsync(range.data(), rangeDev, range.size() * sizeof(int), hipMemcpyDeviceToHost, nullptr));
    }
    MGPS_HIP(h, hipDeviceSynchronize());
    if (rangeDev) {
        (void)cacheFree(rangeDev);
        for (size_t l = 0; l < nlv; ++l)
            if (range[2 * l + 1] >= range[2 * l]) {  // (a level without active cells keeps the whole row)
                h->lv[l].g.xlo = range[2 * l] & ~3;
                h->lv[l].g.xhi = (range[2 * l + 1] | 3) + 1;
            }
    }
    return MGPS_OK;
}

// page-locked blocks for the big set-up arrays (see hostBigAlloc); nullptr sends the caller to malloc
// Page-locking costs ~45 us per MB each way, and a time-stepping caller builds a solver of much the same size every
// step: released blocks are kept (up to a cap, MGPS_PINNED_CACHE_MB, default 4096) and handed out again.
struct PinnedCache {
    std::mutex guard;
    std::vector<std::pair<size_t, void *>> free;                // (bytes, block) not in use
    std::unordered_map<void *, size_t> size;                     // every live block, in use or cached
    size_t cached = 0;
    size_t cap = [] {
        const char *e = getenv("MGPS_PINNED_CACHE_MB");
        return size_t(e ? std::max(0, atoi(e)) : 4096) << 20;
    }();
};
PinnedCache &pinnedCache()
{
    static PinnedCache *c = new PinnedCache();  // never destroyed: blocks may be released during process tear-down
    return *c;
}
void *pinnedAlloc(size_t bytes)
{
    PinnedCache &c = pinnedCache();
    {
        std::lock_guard<std::mutex> lock(c.guard);
        size_t best = c.free.size();
        for (size_t q = 0; q < c.free.size(); ++q)  // smallest cached block that fits without wasting more than half
            if (c.free[q].first >= bytes && c.free[q].first <= bytes + bytes / 2 && (best == c.free.size() || c.free[q].first < c.free[best].first)) best = q;
        if (best < c.free.size()) {
            void *p = c.free[best].second;
            c.cached -= c.free[best].first;
            c.free.erase(c.free.begin() + ptrdiff_t(best));
            return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(c.guard);
    c.size[p] = bytes;
    return p;
}
void pinnedFree(void *p)
{
    PinnedCache &c = pinnedCache();
    {
        std::lock_guard<std::mutex> lock(c.guard);
        const auto it = c.size.find(p);
        if (it != c.size.end() && c.cached + it->second <= c.cap) {
            c.free.emplace_back(it->second, p);
            c.cached += it->second;
            return;
        }
        if (it != c.size.end()) c.size.erase(it);
    }
    (void)hipHostFree(p);
}
void pinnedTrim()
{
    PinnedCache &c = pinnedCache();
    std::vector<std::pair<size_t, void *>> drop;
    {
        std::lock_guard<std::mutex> lock(c.guard);
        drop.swap(c.free);
        c.cached = 0;
        for (auto &b : drop) c.size.erase(b.second);
    }
    for (auto &b : drop) (void)hipHostFree(b.second);
}

int pickDevice(const mgps_options &o, int *device)
{
    static std::once_flag once;
    std::call_once(once, [] { setHostBigAllocator(pinnedAlloc, pinnedFree); });
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return failH(nullptr, MGPS_ERR_NO_DEVICE, "no HIP device is visible (this library has no CPU path)");
    int dev = o.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev >= ndev) return failH(nullptr, MGPS_ERR_NO_DEVICE, "device ordinal out of range");
    if (hipSetDevice(dev) != hipSuccess) return failH(nullptr, MGPS_ERR_NO_DEVICE, "hipSetDevice failed");
    *device = dev;
    return MGPS_OK;
}

int readOptions(const mgps_options *opt, mgps_options *o)
{
    mgps_default_options(o);
    if (opt) {
        if (opt->struct_size != int(sizeof(mgps_options)))
            return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_options.struct_size mismatch: call mgps_default_options first");
        *o = *opt;
    }
    if (o->pre_sweeps < 1 || o->post_sweeps < 1 || o->stencil_path < 0 || o->stencil_path > 2 || o->precision < 0 || o->precision > 1)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT,
                     "mgps_options: pre_sweeps / post_sweeps must be >= 1, stencil_path 0, 1 or 2, precision 0 or 1");
    return MGPS_OK;
}

// options.host_setup, or MGPS_HOST_SETUP=1 in the environment (A/B timing of the two builders)
bool hostSetup(const mgps_options &o)
{
    static const bool env = [] {
        const char *e = getenv("MGPS_HOST_SETUP");
        return e && e[0] == '1';
    }();
    return o.host_setup != 0 || env;
}

// whole-grid solver on one device.  weights may be nullptr (unit weights: the collapsed tail)
const float kNoRows = 0.f;  // a non-null "no BOUNDARY cells" row array

// rowsL0 != nullptr: wx / wy / wz are DEVICE pointers and the fine level's BOUNDARY rows come precomputed
int createWhole(mgps_solver **out, mgps_hierarchy *hier, const float *wx, const float *wy, const float *wz,
                bool useGS, const mgps_options &o, int device, bool tailOfSlabRun, const float *rowsL0 = nullptr)
{
    auto *h = new mgps_solver();
    h->hier = hier;
    h->opt = o;
    h->useGS = useGS;
    h->device = device;
    h->tailOfSlabRun = tailOfSlabRun;
    h->totalLevels = hier->levels;
    auto bail = [&](int code) {
        setLastGlobalError(h->lastError);
        freeAll(h);
        return code;
    };
    const Dims d0 = hier->lv[0].d;
    if (tailOfSlabRun) h->opt.precision = 0;
    if (h->opt.precision == 1) {
        const bool fused = o.fuse_band_passes && o.band_iterations >= 1 && o.band_iterations <= kBandMaxDepth;
        if (!fused || !mixedPrecisionShapeOk(d0.nx, d0.ny, d0.nz))
            return bail(failH(h, MGPS_ERR_INVALID_ARGUMENT,
                              "options.precision = 1 (mixed precision) needs the fused band stage "
                              "(fuse_band_passes, 1 <= band_iterations <= 4) and a fine grid with nx % 4 == 0 and even ny, nz"));
    }
    StageClock clock(h->opt.print_stats != 0);
    // two jobs nothing below waits for until the end run beside the level set-up: the dense inverse of the coarsest
    // matrix (host threads) and the copy of the face weights (a blocking hipMemcpy of 3 x 4 B per cell)
    const bool needCoarse = hier->levels > 1 || tailOfSlabRun;
    std::thread inverseJob, weightJob;
    std::atomic<bool> weightsFailed{false};
    auto joinJobs = [&] {
        if (inverseJob.joinable()) inverseJob.join();
        if (weightJob.joinable()) weightJob.join();
    };
    if (needCoarse) inverseJob = std::thread([hier] { hier->buildDenseInverse(); });
    h->lv.resize(hier->levels);
    struct CodeCopy {
        uint8_t *dst;
        const uint8_t *src;
        size_t plane, owned;
    };
    std::vector<CodeCopy> codeCopies;
    for (int l = 0; l < hier->levels; ++l) {  // ghost plane | the level's labels | ghost plane
        const Dims d = hier->lv[l].d;
        const size_t plane = size_t(d.nx) * d.ny;
        int rc = devAlloc(h, &h->lv[l].codes, d.cells() + 2 * plane, false);
        if (rc != MGPS_OK) {
            joinJobs();
            return bail(rc);
        }
        codeCopies.push_back({h->lv[l].codes, hier->lv[l].labels.data(), plane, d.cells()});
    }
    if (wx) {
        const size_t wn[3] = {size_t(d0.nx + 1) * d0.ny * d0.nz, size_t(d0.nx) * (d0.ny + 1) * d0.nz,
                              size_t(d0.nx) * d0.ny * (d0.nz + 1)};
        const float *wh[3] = {wx, wy, wz};
        for (int a = 0; a < 3; ++a) {
            int rc = devAlloc(h, &h->w[a], wn[a], false);
            if (rc != MGPS_OK) {
                joinJobs();
                return bail(rc);
            }
        }
        float *dst[3] = {h->w[0], h->w[1], h->w[2]};
        const hipMemcpyKind kind = rowsL0 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        weightJob = std::thread([=, &weightsFailed] {
            if (hipSetDevice(device) != hipSuccess) weightsFailed = true;
            for (const CodeCopy &c : codeCopies) {
                if (weightsFailed) break;
                if (hipMemset(c.dst, MGPS_EXTERIOR_CELL, c.plane) != hipSuccess || hipMemset(c.dst + c.plane + c.owned, MGPS_EXTERIOR_CELL, c.plane) != hipSuccess ||
                    hipMemcpy(c.dst + c.plane, c.src, c.owned, hipMemcpyHostToDevice) != hipSuccess)
                    weightsFailed = true;
            }
            for (int a = 0; a < 3 && !weightsFailed; ++a)
                if (hipMemcpy(dst[a], wh[a], wn[a] * sizeof(float), kind) != hipSuccess) weightsFailed = true;
        });
    } else {
        weightJob = std::thread([=, &weightsFailed] {
            if (hipSetDevice(device) != hipSuccess) weightsFailed = true;
            for (const CodeCopy &c : codeCopies) {
                if (weightsFailed) break;
                if (hipMemset(c.dst, MGPS_EXTERIOR_CELL, c.plane) != hipSuccess || hipMemset(c.dst + c.plane + c.owned, MGPS_EXTERIOR_CELL, c.plane) != hipSuccess ||
                    hipMemcpy(c.dst + c.plane, c.src, c.owned, hipMemcpyHostToDevice) != hipSuccess)
                    weightsFailed = true;
            }
        });
    }
    clock.lap("allocations, copy jobs started");
    // the fine level on this thread, the coarser ones (together about a third of its work) on another
    auto buildLevel = [&](int l, StageClock &clk) {
        HostLevel HL;
        const Dims d = hier->lv[l].d;
        const bool hostW = l == 0 && !rowsL0;
        buildSlabLevel(hier->lv[l], 0, d.nz, hostW ? wx : nullptr, hostW ? wy : nullptr, hostW ? wz : nullptr, HL,
                       l == 0 ? rowsL0 : nullptr);
        clk.lap("codes, rows, lists", l);
        const int rc = uploadLevel(h, h->lv[l], HL, 0, d.nz, d.nz, l == 0 && wx, true, l > 0, true);
        clk.lap("band groups + upload", l);
        return rc;
    };
    int rcCoarse = MGPS_OK;
    std::thread coarseLevels([&] {
        if (hipSetDevice(device) != hipSuccess) {
            rcCoarse = MGPS_ERR_HIP;
            return;
        }
        StageClock clk(h->opt.print_stats != 0);
        for (int l = 1; l < hier->levels && rcCoarse == MGPS_OK; ++l) rcCoarse = buildLevel(l, clk);
    });
    const int rcFine = buildLevel(0, clock);
    coarseLevels.join();
    if (rcFine != MGPS_OK || rcCoarse != MGPS_OK) {
        joinJobs();
        return bail(rcFine != MGPS_OK ? rcFine : rcCoarse);
    }
    joinJobs();
    if (weightsFailed) return bail(failH(h, MGPS_ERR_HIP, "label / weight upload failed"));
    for (int l = 0; l < hier->levels; ++l) {
        DevLevel &L = h->lv[l];
        int e = launchPatchSimpleCodes(nullptr, L.codes + size_t(L.d.nx) * L.d.ny, L.band, L.bandDiag, L.nbndGeneral, L.nband);
        if (e != 0) return bail(failH(h, MGPS_ERR_HIP, "launchPatchSimpleCodes failed"));
    }
    clock.lap("wait for inverse + copies");
    int rc = commonDeviceState(h, needCoarse);
    if (rc != MGPS_OK) return bail(rc);
    clock.lap("coarse solver upload + scratch");
    *out = h;
    return MGPS_OK;
}


// ---- the same solver with the hierarchy and every list built on the device (mgps_setup.hip) ---------------------------
// labels: nx*ny*nz bytes, weights: the three face grids; `kind` says where they live (hipMemcpyHostToDevice or
// hipMemcpyDeviceToDevice).  The labels of all levels stay on the device; the host sees per-level counts, the activity
// flags (one byte per 256 cells), the tile kinds and the coarsest level's labels (for the direct solver) -- nothing of
// O(cells).  mgps_get_hierarchy builds the host-side hierarchy on demand from the fine labels.
struct DevScratch {
    std::vector<void *> ptrs;
    template <class T>
    int get(mgps_solver *h, T **p, size_t count)
    {
        MGPS_TRY(devAlloc(h, p, count, false));
        ptrs.push_back(*p);
        return MGPS_OK;
    }
    ~DevScratch()
    {
        (void)hipDeviceSynchronize();  // (deviceFree does not wait for the kernels that still use a block)
        for (void *p : ptrs) (void)cacheFree(p);
    }
};

void fillGridP(mgps_solver *h, DevLevel &L, bool withWeights, int nchunks, int chunkCells, int nplaneBlocks, int planeZc)
{
    const size_t plane = size_t(L.d.nx) * L.d.ny;
    L.g = GridP{L.d.nx,
                L.d.ny,
                L.d.nz,
                L.codes + plane,
                withWeights ? h->w[0] : nullptr,
                withWeights ? h->w[1] : nullptr,
                withWeights ? h->w[2] : nullptr,
                L.band,
                L.rows,
                L.nbndGeneral,
                L.bandDiag,
                0,
                0,
                L.chunks,
                nchunks,
                chunkCells,
                planeZc ? L.planeBlocks : nullptr,
                nplaneBlocks,
                planeZc,
                L.d.cells() * 3 * sizeof(float) > (size_t(256) << 20) ? 1 : 0,
                h->opt.stencil_path};
    L.g.planeZc = planeZcFor(h->opt, planeZc, size_t(nplaneBlocks), size_t(nchunks), chunkCells);
    if (!L.g.planeZc) L.g.planeBlocks = nullptr;
}

// tail: the collapsed tail of a slab run on rank 0 -- unit weights (wx == nullptr), the coarsest solver even for a single level, no
// shell test of its finest level (the level cap of the whole hierarchy has been applied by the ranks together)
int createWholeOnDevice(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels, const float *wx, const float *wy, const float *wz,
                        hipMemcpyKind kind, int mgLevels, bool useGS, const mgps_options &o, int device, bool tail = false)
{
    // the argument rules of mgps_hierarchy_create (MG.cpp:159-161)
    if (mgLevels < 1 || nx < 2 || ny < 2 || nz < 2 || (nx & 1) || (ny & 1) || (nz & 1))
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_hierarchy_create: extents must be even and >= 2, mg_levels >= 1");
    if (size_t(nx) * ny * nz > size_t(0x7fffffff)) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_hierarchy_create: more than 2^31-1 cells per grid");
    for (int l = 1; l < mgLevels; ++l)
        if (((nx >> (l - 1)) & 1) || ((ny >> (l - 1)) & 1) || ((nz >> (l - 1)) & 1) || (nx >> l) < 1 || (ny >> l) < 1 || (nz >> l) < 1)
            return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_hierarchy_create: extents are not divisible by 2^(levels-1)");
    if (o.band_width < 1 || o.band_width > 8 || o.band_iterations < 0)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_hierarchy_create: band_width >= 1 (device-side set-up: <= 8), band_iterations >= 0");
    auto *h = new mgps_solver();
    h->opt = o;
    h->useGS = useGS;
    h->device = device;
    h->requestedLevels = mgLevels;
    h->tailOfSlabRun = tail;
    if (tail) h->opt.precision = 0;
    std::thread inverseJob;  // the coarsest level's direct solver, on host threads beside the device work
    struct Joiner {          // (an exception on the way out -- std::bad_alloc -- must not meet a joinable thread)
        std::thread &t;
        ~Joiner()
        {
            if (t.joinable()) t.join();
        }
    } joiner{inverseJob};
    auto bail = [&](int code) {
        if (inverseJob.joinable()) inverseJob.join();
        setLastGlobalError(h->lastError);
        freeAll(h);
        return code;
    };
    const Dims d0{nx, ny, nz};
    if (h->opt.precision == 1) {
        const bool fused = o.fuse_band_passes && o.band_iterations >= 1 && o.band_iterations <= kBandMaxDepth;
        if (!fused || !mixedPrecisionShapeOk(d0.nx, d0.ny, d0.nz))
            return bail(failH(h, MGPS_ERR_INVALID_ARGUMENT,
                              "options.precision = 1 (mixed precision) needs the fused band stage "
                              "(fuse_band_passes, 1 <= band_iterations <= 4) and a fine grid with nx % 4 == 0 and even ny, nz"));
    }
    StageClock clock(h->opt.print_stats != 0);
    DevScratch tmp;
#define ODS_TRY(call)                        \
    do {                                     \
        int s_ = (call);                     \
        if (s_ != MGPS_OK) return bail(s_);  \
    } while (0)
#define ODS_HIP(call)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return bail(failH(h, MGPS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)));  \
    } while (0)
#define ODS_LAUNCH(call)                                                                                                       \
    do {                                                                                                                       \
        int e_ = (call);                                                                                                       \
        if (e_ != 0) return bail(failH(h, MGPS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(hipError_t(e_))));       \
    } while (0)

    // ---- labels of every level: ghost plane | the level | ghost plane
    h->lv.resize(size_t(mgLevels));
    for (int l = 0; l < mgLevels; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        L.d = Dims{nx >> l, ny >> l, nz >> l};
        L.z0 = 0;
        L.z1 = L.d.nz;
        const size_t plane = size_t(L.d.nx) * L.d.ny;
        ODS_TRY(devAlloc(h, &L.codes, L.d.cells() + 2 * plane, false));
        ODS_HIP(hipMemsetAsync(L.codes, MGPS_EXTERIOR_CELL, plane, nullptr));
        ODS_HIP(hipMemsetAsync(L.codes + plane + L.d.cells(), MGPS_EXTERIOR_CELL, plane, nullptr));
    }
    auto labOf = [&](int l) { return h->lv[size_t(l)].codes + size_t(h->lv[size_t(l)].d.nx) * h->lv[size_t(l)].d.ny; };
    if (labels != labOf(0)) ODS_HIP(hipMemcpy(labOf(0), labels, d0.cells(), kind));
    // flags per level: [0] shell broken, [1] holds an active cell; then [2 L] interior rule broken, [2 L + 1] weight rule broken,
    // [2 L + 2 ...] band groups broken per level
    int *flags = nullptr;
    const size_t nflags = size_t(3 * mgLevels + 2);
    ODS_TRY(tmp.get(h, &flags, nflags));
    ODS_HIP(hipMemsetAsync(flags, 0, nflags * sizeof(int), nullptr));
    ODS_LAUNCH(launchShellCheck(nullptr, d0, labOf(0), flags));
    ODS_LAUNCH(launchAnyActive(nullptr, d0, labOf(0), flags + 1));
    for (int l = 1; l < mgLevels; ++l) {
        ODS_LAUNCH(launchCoarsenLabels(nullptr, h->lv[size_t(l - 1)].d, labOf(l - 1), labOf(l), flags + 2 * l + 1));
        ODS_LAUNCH(launchShellCheck(nullptr, h->lv[size_t(l)].d, labOf(l), flags + 2 * l));
        ODS_LAUNCH(launchMarkBoundary(nullptr, h->lv[size_t(l)].d, labOf(l)));
    }
    std::vector<int> hflags(nflags);
    ODS_HIP(hipMemcpy(hflags.data(), flags, nflags * sizeof(int), hipMemcpyDeviceToHost));
    if (hflags[0] && !tail) return bail(failH(h, MGPS_ERR_HIERARCHY, "labels need an EXTERIOR shell on all six sides (unitTestExteriorCells)"));
    if (!hflags[1]) return bail(failH(h, MGPS_ERR_HIERARCHY, "no INTERIOR or BOUNDARY cell in the domain"));
    int levels = mgLevels;
    for (int l = 1; l < mgLevels; ++l) {  // MG.cpp:238-253
        if (hflags[size_t(2 * l)])
            return bail(failH(h, MGPS_ERR_HIERARCHY,
                              "level " + std::to_string(l) + " has no EXTERIOR shell (unitTestExteriorCells, MG.cpp:252): " + std::to_string(mgLevels) +
                                  " levels need 2^(levels-1) = " + std::to_string(1 << (mgLevels - 1)) +
                                  " EXTERIOR cells on every side of the solver grid (mgps_expanded_layout pads that much)"));
        if (!hflags[size_t(2 * l + 1)]) {
            levels = l - 1;  // the reference drops the last solvable level too (MG.cpp:245)
            break;
        }
    }
    if (levels < 1) return bail(failH(h, MGPS_ERR_HIERARCHY, "level cap left no multigrid level (first coarse level has no solvable cell)"));
    for (int l = levels; l < mgLevels; ++l) {
        (void)cacheFree(h->lv[size_t(l)].codes);
        h->lv[size_t(l)].codes = nullptr;
    }
    h->lv.resize(size_t(levels));
    h->totalLevels = levels;
    clock.lap("labels of all levels (device)");

    // ---- the coarsest level's direct solver on host threads, beside everything below
    const bool needCoarse = levels > 1 || tail;
    int rcInverse = MGPS_OK;
    {
        const Dims cd = h->lv[size_t(levels - 1)].d;
        std::vector<uint8_t> coarsest(cd.cells());
        ODS_HIP(hipMemcpy(coarsest.data(), labOf(levels - 1), cd.cells(), hipMemcpyDeviceToHost));
        inverseJob = std::thread([h, nx, ny, nz, levels, needCoarse, &rcInverse, lab = std::move(coarsest)] {
            rcInverse = hierarchyLight(&h->hier, nx, ny, nz, levels, lab.data(), h->opt, needCoarse);
            if (rcInverse == MGPS_OK && needCoarse) h->hier->buildDenseInverse();
        });
    }
    // ---- face weights
    {
        const size_t wn[3] = {size_t(d0.nx + 1) * d0.ny * d0.nz, size_t(d0.nx) * (d0.ny + 1) * d0.nz, size_t(d0.nx) * d0.ny * (d0.nz + 1)};
        const float *src[3] = {wx, wy, wz};
        h->weightsBorrowed = kind == hipMemcpyDeviceToDevice && o.borrow_device_weights != 0;
        for (int a = 0; a < 3 && wx; ++a) {
            if (h->weightsBorrowed) {
                h->w[a] = const_cast<float *>(src[a]);
                continue;
            }
            ODS_TRY(devAlloc(h, &h->w[a], wn[a], false));
            ODS_HIP(hipMemcpyAsync(h->w[a], src[a], wn[a] * sizeof(float), kind, nullptr));
        }
    }

    // ---- band masks and tile counts of every level, then one round of counts to the host
    struct LevelTmp {
        int nt = 0, tx = 0, ty = 0, tz = 0;
        uint32_t *mask = nullptr;
        uint16_t *prefix = nullptr;
        int32_t *tileCount = nullptr, *tileKind = nullptr, *tileStart = nullptr, *scan = nullptr;
        int32_t *sorted = nullptr, *general = nullptr, *genRank = nullptr, *bandEntry = nullptr;
        uint8_t *diagS = nullptr, *chunkFlags = nullptr, *planeFlags = nullptr;
        int32_t *gcount[4] = {nullptr, nullptr, nullptr, nullptr}, *gat[4] = {nullptr, nullptr, nullptr, nullptr};
        int32_t *tileFlags = nullptr, *tileRank = nullptr, *bandTiles = nullptr, *boxTiles = nullptr;
        uint8_t *tileBits = nullptr;
        int *runCounts = nullptr;
        int32_t *listCounts = nullptr;  // pure even / odd, mixed even / odd tiles, plane blocks
        int runCells = 0, listLen = 0;
        int nband = 0, nGen = 0, planeZc = 0, nBandTiles = 0, nBoxTiles = 0;
        size_t nfine = 0, nplane = 0;
    };
    std::vector<LevelTmp> T{size_t(levels)};
    for (int l = 0; l < levels; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        LevelTmp &t = T[size_t(l)];
        t.tx = (L.d.nx + kTile - 1) / kTile;
        t.ty = (L.d.ny + kTile - 1) / kTile;
        t.tz = (L.d.nz + kTile - 1) / kTile;
        t.nt = t.tx * t.ty * t.tz;
        ODS_TRY(tmp.get(h, &t.mask, size_t(t.nt) * 128));
        ODS_TRY(tmp.get(h, &t.prefix, size_t(t.nt) * 128));
        ODS_TRY(tmp.get(h, &t.tileCount, size_t(t.nt)));
        ODS_TRY(tmp.get(h, &t.tileKind, size_t(t.nt)));
        ODS_TRY(tmp.get(h, &t.tileStart, size_t(t.nt) + 1));
        ODS_TRY(tmp.get(h, &t.scan, scanScratchInts(L.d.cells())));
        ODS_TRY(tmp.get(h, &t.tileFlags, size_t(t.nt)));
        ODS_TRY(tmp.get(h, &t.tileRank, size_t(t.nt) + 1));
        ODS_TRY(tmp.get(h, &t.bandTiles, size_t(t.nt)));
        ODS_TRY(tmp.get(h, &t.tileBits, size_t(t.nt)));
        // which tiles can hold band cells at all: a streaming pass over the labels; the band kernel then runs on those only
        ODS_HIP(hipMemsetAsync(t.mask, 0, size_t(t.nt) * 128 * sizeof(uint32_t), nullptr));
        ODS_HIP(hipMemsetAsync(t.prefix, 0, size_t(t.nt) * 128 * sizeof(uint16_t), nullptr));
        ODS_HIP(hipMemsetAsync(t.tileCount, 0, size_t(t.nt) * sizeof(int32_t), nullptr));
        ODS_LAUNCH(launchBandCandidates(nullptr, L.d, labOf(l), t.tileKind, t.tileBits, t.tileFlags, t.tileRank, t.bandTiles, t.scan));
    }
    for (int l = 0; l < levels; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        LevelTmp &t = T[size_t(l)];
        int ncand = 0;
        ODS_HIP(hipMemcpy(&ncand, t.tileRank + t.nt, sizeof(int), hipMemcpyDeviceToHost));
        ODS_LAUNCH(launchBandMasks(nullptr, L.d, labOf(l), o.band_width, t.mask, t.prefix, t.tileCount, t.tileKind, l == 0 ? flags + 2 * mgLevels : nullptr,
                                   t.bandTiles, ncand));
        ODS_LAUNCH(launchExclusiveScan(nullptr, t.tileCount, t.tileStart, size_t(t.nt), t.scan));
        ODS_LAUNCH(launchBandTileList(nullptr, t.tileStart, t.nt, t.tileFlags, t.tileRank, t.bandTiles, t.scan));
    }
    for (int l = 0; l < levels; ++l) {
        ODS_HIP(hipMemcpy(&T[size_t(l)].nband, T[size_t(l)].tileStart + T[size_t(l)].nt, sizeof(int), hipMemcpyDeviceToHost));
        ODS_HIP(hipMemcpy(&T[size_t(l)].nBandTiles, T[size_t(l)].tileRank + T[size_t(l)].nt, sizeof(int), hipMemcpyDeviceToHost));
    }
    {
        int interiorBad = 0;  // (the mask kernel of the fine level checks the INTERIOR cells on the way)
        ODS_HIP(hipMemcpy(&interiorBad, flags + 2 * mgLevels, sizeof(int), hipMemcpyDeviceToHost));
        if (interiorBad)
            return bail(failH(h, MGPS_ERR_HIERARCHY,
                              "labels/weights violate the BOUNDARY-cell rules (unitTestBoundaryCells): run mgps_fields_set_boundary_labels"));
    }
    clock.lap("band masks + counts");

    // ---- band lists in reference order, classification, scan of the general cells
    for (int l = 0; l < levels; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        LevelTmp &t = T[size_t(l)];
        const size_t nb = size_t(t.nband);
        L.nband = t.nband;
        ODS_TRY(tmp.get(h, &t.sorted, nb));
        ODS_TRY(tmp.get(h, &t.general, nb));
        ODS_TRY(tmp.get(h, &t.genRank, nb + 1));
        ODS_TRY(tmp.get(h, &t.bandEntry, nb));
        ODS_TRY(tmp.get(h, &t.diagS, nb));
        ODS_TRY(devAlloc(h, &L.band, nb, false));
        ODS_TRY(devAlloc(h, &L.bandDiag, nb, false));
        ODS_TRY(devAlloc(h, &L.bandTmp, nb, false));
        if (t.nband > 0) {
            ODS_LAUNCH(launchBandFill(nullptr, L.d, t.mask, t.prefix, t.tileStart, t.sorted));
            ODS_LAUNCH(launchBandClassify(nullptr, L.d, labOf(l), l == 0 ? h->w[0] : nullptr, l == 0 ? h->w[1] : nullptr, l == 0 ? h->w[2] : nullptr, t.sorted,
                                          t.nband, t.diagS, t.general, l == 0 ? flags + 2 * mgLevels + 1 : nullptr));
        }
        ODS_LAUNCH(launchExclusiveScan(nullptr, t.general, t.genRank, nb, t.scan));
    }
    for (int l = 0; l < levels; ++l) ODS_HIP(hipMemcpy(&T[size_t(l)].nGen, T[size_t(l)].genRank + T[size_t(l)].nband, sizeof(int), hipMemcpyDeviceToHost));
    {
        int violations = 0;
        ODS_HIP(hipMemcpy(&violations, flags + 2 * mgLevels + 1, sizeof(int), hipMemcpyDeviceToHost));
        if (violations)
            return bail(failH(h, MGPS_ERR_HIERARCHY,
                              "labels/weights violate the BOUNDARY-cell rules (unitTestBoundaryCells): run mgps_fields_set_boundary_labels"));
    }
    clock.lap("band lists + classification");

    // ---- device order, rows, codes, activity flags
    for (int l = 0; l < levels; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        LevelTmp &t = T[size_t(l)];
        L.nbndGeneral = t.nGen;
        ODS_TRY(devAlloc(h, &L.rows, size_t(7) * size_t(t.nGen), false));
        ODS_TRY(devAlloc(h, &L.tileBndStart, size_t(t.nt) + 1, false));
        ODS_LAUNCH(launchBandSplit(nullptr, L.d, labOf(l), l == 0 ? h->w[0] : nullptr, l == 0 ? h->w[1] : nullptr, l == 0 ? h->w[2] : nullptr, t.sorted, t.nband,
                                   t.diagS, t.genRank, L.band, L.bandDiag, t.bandEntry, L.rows));
        ODS_LAUNCH(launchGather(nullptr, t.genRank, t.tileStart, t.nt + 1, L.tileBndStart));
        t.nfine = (L.d.cells() + kSegCells - 1) / kSegCells;
        t.planeZc = planeSweepZc(L.d.nx, L.d.ny, L.d.nz);
        ODS_TRY(tmp.get(h, &t.chunkFlags, t.nfine));
        if (t.planeZc) {
            t.nplane = size_t((L.d.nx + 255) / 256) * size_t((L.d.ny + kPlaneRows - 1) / kPlaneRows) * size_t((L.d.nz + t.planeZc - 1) / t.planeZc);
            ODS_TRY(tmp.get(h, &t.planeFlags, t.nplane));
            ODS_HIP(hipMemsetAsync(t.planeFlags, 0, t.nplane, nullptr));
        }
        ODS_LAUNCH(launchActivityFlags(nullptr, L.d, labOf(l), t.chunkFlags, t.planeFlags, t.planeZc));
        ODS_TRY(tmp.get(h, &t.runCounts, 4));
        ODS_HIP(hipMemsetAsync(t.runCounts, 0, 4 * sizeof(int), nullptr));
        ODS_LAUNCH(launchCountRuns(nullptr, t.chunkFlags, t.nfine, t.runCounts));
        // (the groups below read the labels for activity only: patched or not makes no difference)
        ODS_LAUNCH(launchPatchSimpleCodes(nullptr, labOf(l), L.band, L.bandDiag, L.nbndGeneral, L.nband));
    }
    // ---- boxes of the fused band stage: counts
    const bool wantGroups = o.fuse_band_passes && o.band_iterations >= 1 && o.band_iterations <= kBandMaxDepth;
    std::vector<int> haveGroups(size_t(levels), 0);
    for (int l = 0; l < levels && wantGroups; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        LevelTmp &t = T[size_t(l)];
        if (t.nband == 0 || !boxPlaneFits(L.d)) continue;
        haveGroups[size_t(l)] = 1;
        ODS_TRY(tmp.get(h, &t.boxTiles, size_t(t.nt)));
        ODS_LAUNCH(launchBoxTileList(nullptr, L.d, t.tileStart, t.tileFlags, t.tileRank, t.boxTiles, t.scan));
        ODS_HIP(hipMemcpy(&t.nBoxTiles, t.tileRank + t.nt, sizeof(int), hipMemcpyDeviceToHost));
        for (int q = 0; q < 3; ++q) {
            ODS_TRY(tmp.get(h, &t.gcount[q], size_t(t.nBoxTiles)));
            ODS_TRY(tmp.get(h, &t.gat[q], size_t(t.nBoxTiles) + 1));
        }
        ODS_LAUNCH(launchBandBoxesCount(nullptr, L.d, labOf(l), t.mask, t.prefix, t.tileStart, t.bandEntry, L.bandDiag, o.band_iterations, t.boxTiles, t.nBoxTiles,
                                        t.gcount, flags + 2 * mgLevels + 2 + l));
        for (int q = 0; q < 3; ++q) ODS_LAUNCH(launchExclusiveScan(nullptr, t.gcount[q], t.gat[q], size_t(t.nBoxTiles), t.scan));
    }
    // ---- host side of the lists: flags and kinds come back, lists go up
    for (int l = 0; l < levels; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        LevelTmp &t = T[size_t(l)];
        StageClock sub(h->opt.print_stats > 1);
        // run length of the activity list from the device's counts; the flags come back folded to that length
        int rc4[4] = {0, 0, 0, 0};
        ODS_HIP(hipMemcpy(rc4, t.runCounts, sizeof(rc4), hipMemcpyDeviceToHost));
        const int64_t rc64[4] = {rc4[0], rc4[1], rc4[2], rc4[3]};
        const int runCells = chooseRunCells(rc64);
        const size_t nruns = (t.nfine + size_t(runCells / kSegCells) - 1) / size_t(runCells / kSegCells);
        uint8_t *runFlags = t.chunkFlags;
        if (runCells != kSegCells) {
            ODS_TRY(tmp.get(h, &runFlags, nruns));
            ODS_LAUNCH(launchFoldRunFlags(nullptr, t.chunkFlags, t.nfine, runCells, runFlags));
        }
        // the list on the device: its length is the count of active runs, rounded up to whole workgroups
        int activeRuns = 0;
        for (int z = 0; z < 4; ++z)
            if (kRunSizes[z] == runCells) activeRuns = rc4[z];
        const int perGroup = kChunkCells / runCells, listLen = (activeRuns + perGroup - 1) / perGroup * perGroup;
        {
            int32_t *tmpFlags = nullptr, *rank = nullptr, *base = nullptr;
            ODS_TRY(tmp.get(h, &tmpFlags, nruns));
            ODS_TRY(tmp.get(h, &rank, nruns + 1));
            ODS_TRY(tmp.get(h, &base, 1));
            ODS_TRY(devAlloc(h, &L.chunks, size_t(listLen), false));
            ODS_LAUNCH(launchRunList(nullptr, L.d, runFlags, nruns, runCells, tmpFlags, rank, t.scan, base, L.chunks, listLen));
        }
        // the four Gauss-Seidel tile lists and the plane-block list: compacted on the device too; the five counts come back
        // together below
        ODS_TRY(tmp.get(h, &t.listCounts, 5));
        ODS_HIP(hipMemsetAsync(t.listCounts, 0, 5 * sizeof(int32_t), nullptr));
        {
            int32_t **cls[4] = {&L.pure[0], &L.pure[1], &L.mixed[0], &L.mixed[1]};  // pure even, pure odd, mixed even, mixed odd
            for (int q = 0; q < 4; ++q) {
                ODS_TRY(devAlloc(h, cls[q], size_t(t.nt) / 2 + 1, false));
                ODS_LAUNCH(launchTileClassList(nullptr, L.d, t.tileKind, q & 1, q >> 1, t.tileFlags, t.tileRank, *cls[q], t.scan));
                ODS_HIP(hipMemcpyAsync(t.listCounts + q, t.tileRank + t.nt, sizeof(int32_t), hipMemcpyDeviceToDevice, nullptr));
            }
            if (t.nplane) {
                int32_t *pf = nullptr, *pr = nullptr;
                ODS_TRY(tmp.get(h, &pf, t.nplane));
                ODS_TRY(tmp.get(h, &pr, t.nplane + 1));
                ODS_TRY(devAlloc(h, &L.planeBlocks, t.nplane, false));
                ODS_LAUNCH(launchByteList(nullptr, t.planeFlags, int(t.nplane), pf, pr, L.planeBlocks, t.scan));
                ODS_HIP(hipMemcpyAsync(t.listCounts + 4, pr + t.nplane, sizeof(int32_t), hipMemcpyDeviceToDevice, nullptr));
            }
        }
        t.runCells = runCells;
        t.listLen = listLen;
        sub.lap("   lists: enqueued", l);
        if (l > 0) {
            ODS_TRY(gridAlloc(h, &L.x, L.d));
            ODS_TRY(gridAlloc(h, &L.b, L.d));
        }
        ODS_TRY(gridAlloc(h, &L.r, L.d));
        ODS_TRY(gridAlloc(h, &L.tmp, L.d));
    }
    for (int l = 0; l < levels; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        LevelTmp &t = T[size_t(l)];
        int32_t counts[5] = {0, 0, 0, 0, 0};
        ODS_HIP(hipMemcpy(counts, t.listCounts, sizeof(counts), hipMemcpyDeviceToHost));
        L.npure[0] = counts[0];
        L.npure[1] = counts[1];
        L.nmixed[0] = counts[2];
        L.nmixed[1] = counts[3];
        fillGridP(h, L, l == 0 && wx, t.listLen, t.runCells, counts[4], t.planeZc);
    }
    clock.lap("device order, codes, activity + tile lists");
    // ---- boxes: totals to the host, arrays filled
    for (int l = 0; l < levels; ++l) {
        if (!haveGroups[size_t(l)]) continue;
        DevLevel &L = h->lv[size_t(l)];
        LevelTmp &t = T[size_t(l)];
        int tot[3] = {0, 0, 0}, brokenL = 0;
        for (int q = 0; q < 3; ++q) ODS_HIP(hipMemcpy(&tot[q], t.gat[q] + t.nBoxTiles, sizeof(int), hipMemcpyDeviceToHost));
        ODS_HIP(hipMemcpy(&brokenL, flags + 2 * mgLevels + 2 + l, sizeof(int), hipMemcpyDeviceToHost));
        if (brokenL || tot[0] == 0) return bail(failH(h, MGPS_ERR_INTERNAL, "band boxes: the builder failed on level " + std::to_string(l)));
        L.bandBoxes.depth = o.band_iterations;
        L.bandBoxes.ngroups = tot[0];
        L.bandBoxes.listCount = size_t(tot[1]);
        L.bandBoxes.generalInts = 2 * size_t(tot[2]);
        L.bandBoxes.anyGeneral = tot[2] > 0;
        ODS_TRY(devAlloc(h, &L.bandBoxes.info, size_t(kBoxInfoInts) * size_t(tot[0]), false));
        ODS_TRY(devAlloc(h, &L.bandBoxes.list, size_t(tot[1]), false));
        ODS_TRY(devAlloc(h, &L.bandBoxes.general, 2 * size_t(tot[2]), false));
        ODS_LAUNCH(launchBandBoxesFill(nullptr, L.d, labOf(l), t.mask, t.prefix, t.tileStart, t.bandEntry, L.bandDiag, o.band_iterations, t.boxTiles, t.nBoxTiles,
                                       t.gat, L.bandBoxes.info, L.bandBoxes.list, L.bandBoxes.general, flags + 2 * mgLevels + 2 + l));
        ODS_TRY(finishBandBoxes(h, L, nullptr));
    }
    ODS_HIP(hipDeviceSynchronize());
    clock.lap("band boxes");
    inverseJob.join();
    if (rcInverse != MGPS_OK) {
        h->lastError = lastGlobalError();
        return bail(rcInverse);
    }
    clock.lap("wait for the coarse inverse");
    int rc = commonDeviceState(h, needCoarse);
    if (rc != MGPS_OK) return bail(rc);
    clock.lap("coarse solver upload + scratch");
#undef ODS_TRY
#undef ODS_HIP
#undef ODS_LAUNCH
    *out = h;
    return MGPS_OK;
}

}  // namespace

extern "C" {

const char *mgps_last_error(const mgps_solver *h) { return h ? h->lastError.c_str() : lastGlobalError(); }

void mgps_trim_host_cache(void) { pinnedTrim(); }
void mgps_trim_device_cache(void)
{
    (void)hipDeviceSynchronize();
    dropKeptInverses();  // (their blocks land in the cache that is emptied next)
    deviceTrim();
}

void *mgps_host_alloc(size_t bytes)
try {
    int ndev = 0;
    if (bytes == 0 || hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return nullptr;
    }
    return pinnedAlloc(bytes);
} catch (...) {
    return nullptr;
}

void mgps_host_free(void *p)
try {
    if (p) pinnedFree(p);
} catch (...) {
}

int mgps_device_count(int *count)
try {
    if (!count) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_device_count: NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return MGPS_OK;
}
MGPS_API_CATCH(nullptr)

int mgps_create(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels_host, const float *wx_host,
                const float *wy_host, const float *wz_host, int mg_levels, int use_gauss_seidel,
                const mgps_options *opt)
try {
    if (!out) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create: out is NULL");
    *out = nullptr;
    if (!labels_host || !wx_host || !wy_host || !wz_host)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create: labels and the three weight grids are required");
    mgps_options o;
    MGPS_TRY(readOptions(opt, &o));
    int device = 0;
    MGPS_TRY(pickDevice(o, &device));
    if (!hostSetup(o)) return createWholeOnDevice(out, nx, ny, nz, labels_host, wx_host, wy_host, wz_host, hipMemcpyHostToDevice, mg_levels, use_gauss_seidel != 0, o, device);
    mgps_hierarchy *hier = nullptr;
    MGPS_TRY(mgps_hierarchy_create(&hier, nx, ny, nz, labels_host, mg_levels, &o));
    {  // the fine-level invariants the reference asserts in debug builds (MG.cpp:234)
        int pass = 0;
        mgps_check_boundary_cells(labels_host, wx_host, wy_host, wz_host, nx, ny, nz, &pass);
        if (!pass) {
            mgps_hierarchy_destroy(hier);
            return failH(nullptr, MGPS_ERR_HIERARCHY,
                         "labels/weights violate the BOUNDARY-cell rules (unitTestBoundaryCells): run mgps_set_boundary_labels");
        }
    }
    return createWhole(out, hier, wx_host, wy_host, wz_host, use_gauss_seidel != 0, o, device, false);
}
MGPS_API_CATCH(nullptr)

}  // extern "C"

namespace {
// labels_dev: the same labels on the device when the caller has them there (mgps_create_device), else nullptr
int createFromDeviceWeights(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels_host, const uint8_t *labels_dev,
                            const float *wx_dev, const float *wy_dev, const float *wz_dev, int mg_levels, int use_gauss_seidel,
                            const mgps_options *opt)
{
    mgps_options o;
    MGPS_TRY(readOptions(opt, &o));
    int device = 0;
    MGPS_TRY(pickDevice(o, &device));
    mgps_hierarchy *hier = nullptr;
    MGPS_TRY(mgps_hierarchy_create(&hier, nx, ny, nz, labels_host, mg_levels, &o));
    auto drop = [&](int code, const std::string &msg) {
        mgps_hierarchy_destroy(hier);
        return failH(nullptr, code, msg);
    };
    // the BOUNDARY cells of the fine level in band order (every BOUNDARY cell is a band cell, layer 0 of Ops.cpp:192-224)
    const HostLevel &G = hier->lv[0];
    std::vector<int32_t> cells;
    for (int32_t c : G.band)
        if (G.labels[size_t(c)] == MGPS_BOUNDARY_CELL) cells.push_back(c);
    std::vector<float> rows(8 * cells.size());
    int violations = 0;
    {
        uint8_t *labDev = nullptr;
        int32_t *cellsDev = nullptr;
        float *rowsDev = nullptr;
        int *violDev = nullptr;
        const size_t n = G.d.cells();
        hipError_t e = labels_dev ? hipSuccess : cacheMalloc(reinterpret_cast<void **>(&labDev), n);
        if (e == hipSuccess) e = cacheMalloc(reinterpret_cast<void **>(&cellsDev), std::max<size_t>(1, cells.size()) * sizeof(int32_t));
        if (e == hipSuccess) e = cacheMalloc(reinterpret_cast<void **>(&rowsDev), std::max<size_t>(1, rows.size()) * sizeof(float));
        if (e == hipSuccess) e = cacheMalloc(reinterpret_cast<void **>(&violDev), sizeof(int));
        if (e == hipSuccess && !labels_dev) e = hipMemcpy(labDev, G.labels.data(), n, hipMemcpyHostToDevice);
        if (e == hipSuccess && !cells.empty()) e = hipMemcpy(cellsDev, cells.data(), cells.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(violDev, 0, sizeof(int));
        if (e == hipSuccess) e = hipError_t(launchBoundaryRows(nullptr, G.d, labels_dev ? labels_dev : labDev, wx_dev, wy_dev, wz_dev, cellsDev, int(cells.size()), rowsDev, violDev));
        if (e == hipSuccess && !rows.empty()) e = hipMemcpy(rows.data(), rowsDev, rows.size() * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&violations, violDev, sizeof(int), hipMemcpyDeviceToHost);
        (void)cacheFree(labDev);
        (void)cacheFree(cellsDev);
        (void)cacheFree(rowsDev);
        (void)cacheFree(violDev);
        if (e != hipSuccess) return drop(MGPS_ERR_HIP, std::string("mgps_create_device_weights: ") + hipGetErrorString(e));
    }
    int interiorOk = 0;
    checkInteriorCells(G.labels.data(), nx, ny, nz, &interiorOk);
    if (violations != 0 || !interiorOk)
        return drop(MGPS_ERR_HIERARCHY,
                    "labels/weights violate the BOUNDARY-cell rules (unitTestBoundaryCells): run mgps_fields_set_boundary_labels");
    return createWhole(out, hier, wx_dev, wy_dev, wz_dev, use_gauss_seidel != 0, o, device, false, rows.empty() ? &kNoRows : rows.data());
}
}  // namespace

extern "C" {

int mgps_create_device_weights(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels_host, const float *wx_dev,
                               const float *wy_dev, const float *wz_dev, int mg_levels, int use_gauss_seidel,
                               const mgps_options *opt)
try {
    if (!out) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_device_weights: out is NULL");
    *out = nullptr;
    if (!labels_host || !wx_dev || !wy_dev || !wz_dev)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_device_weights: labels and the three weight grids are required");
    {
        mgps_options o;
        MGPS_TRY(readOptions(opt, &o));
        if (!hostSetup(o)) {  // the labels go up once (1 B per cell), everything else happens where the weights are
            int device = 0;
            MGPS_TRY(pickDevice(o, &device));
            uint8_t *lab = nullptr;
            const size_t n = size_t(nx) * ny * nz;
            if (nx < 1 || ny < 1 || nz < 1 || cacheMalloc(reinterpret_cast<void **>(&lab), n) != hipSuccess)
                return failH(nullptr, MGPS_ERR_ALLOC, "mgps_create_device_weights: label staging allocation failed");
            int rc = hipMemcpy(lab, labels_host, n, hipMemcpyHostToDevice) == hipSuccess
                         ? createWholeOnDevice(out, nx, ny, nz, lab, wx_dev, wy_dev, wz_dev, hipMemcpyDeviceToDevice, mg_levels, use_gauss_seidel != 0, o, device)
                         : failH(nullptr, MGPS_ERR_HIP, "mgps_create_device_weights: label upload failed");
            (void)cacheFree(lab);
            return rc;
        }
    }
    return createFromDeviceWeights(out, nx, ny, nz, labels_host, nullptr, wx_dev, wy_dev, wz_dev, mg_levels, use_gauss_seidel, opt);
}
MGPS_API_CATCH(nullptr)

int mgps_create_device(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels_dev, const float *wx_dev,
                       const float *wy_dev, const float *wz_dev, int mg_levels, int use_gauss_seidel, const mgps_options *opt)
try {
    if (!out) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_device: out is NULL");
    *out = nullptr;
    if (!labels_dev || !wx_dev || !wy_dev || !wz_dev || nx < 1 || ny < 1 || nz < 1)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_device: labels and the three weight grids are required");
    mgps_options o;
    MGPS_TRY(readOptions(opt, &o));
    int device = 0;
    MGPS_TRY(pickDevice(o, &device));
    if (!hostSetup(o)) return createWholeOnDevice(out, nx, ny, nz, labels_dev, wx_dev, wy_dev, wz_dev, hipMemcpyDeviceToDevice, mg_levels, use_gauss_seidel != 0, o, device);
    // options.host_setup: the hierarchy and the lists are built on the host from one byte per cell
    RawVec<uint8_t> labels(size_t(nx) * ny * nz);
    if (hipMemcpy(labels.data(), labels_dev, labels.size(), hipMemcpyDeviceToHost) != hipSuccess)
        return failH(nullptr, MGPS_ERR_HIP, "mgps_create_device: copying the labels to the host failed");
    return createFromDeviceWeights(out, nx, ny, nz, labels.data(), labels_dev, wx_dev, wy_dev, wz_dev, mg_levels, use_gauss_seidel, opt);
}
MGPS_API_CATCH(nullptr)

}  // extern "C"

namespace {
// How many levels of a hierarchy of `levels` levels stay distributed for the cuts `splits` (fine planes, size + 1 entries).
// Level l stays distributed while every cut is a whole plane of level l + 1 (restriction / prolongation stay rank-local up to
// one ghost plane), every rank owns at least 16 planes of it, with Gauss-Seidel every cut is a multiple of 16 planes of
// level l (the 16^3 tile colouring of the single-GPU run), and -- below the finest -- the ranks own min_cells_per_rank
// cells of it on average.  The last level is always collapsed.  0: the cuts do not allow a slab run.
int distributedLevelsFor(const int *splits, int size, int nx, int ny, int levels, bool useGS, const mgps_options &o)
{
    int D = 0;
    const int nz = splits[size];
    for (int l = 0; l < levels - 1; ++l) {
        bool ok = true;
        for (int r = 0; r <= size && ok; ++r) {
            ok = splits[r] % (2 << l) == 0;
            if (ok && useGS) ok = (splits[r] >> l) % kTile == 0;
            if (ok && r < size) ok = ((splits[r + 1] - splits[r]) >> l) >= kTile;
        }
        if (ok && l > 0) ok = (size_t(nx >> l) * size_t(ny >> l) * size_t(nz >> l)) / size_t(size) >= size_t(std::max(o.min_cells_per_rank, 0));
        if (!ok) break;
        ++D;
    }
    return D;
}
}  // namespace

extern "C" {

int mgps_slab_partition(int nx, int ny, int nz, const uint8_t *labels, int mg_levels, int size, int use_gauss_seidel,
                        const mgps_options *opt, int *out_splits)
try {
    if (!labels || !out_splits || size < 1 || nx < 1 || ny < 1 || nz < 1 || mg_levels < 1)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_slab_partition: bad arguments");
    mgps_options o;
    MGPS_TRY(readOptions(opt, &o));
    auto uniform = [&] {
        for (int r = 0; r <= size; ++r) out_splits[r] = int(int64_t(nz) * r / size);
    };
    uniform();
    if (size == 1 || nz % size != 0) return MGPS_OK;  // (a grid that does not divide is refused by the constructor)
    // depth of the distributed part as the even cut gives it; the balanced cuts keep it
    const int D = distributedLevelsFor(out_splits, size, nx, ny, mg_levels, use_gauss_seidel != 0, o);
    if (D < 1 || use_gauss_seidel) return MGPS_OK;  // Gauss-Seidel: cuts on multiples of 16 planes of EVERY distributed level -- the even cut
    const int unit = 1 << D;                         // Jacobi: whole planes of the collapse level (round 5: the device-side set-up takes cuts that are not
                                                     // multiples of 16 planes; at 1024^3 / 8 ranks the unit of 8 planes is what lets the middle ranks shed load)
    if (nz % unit != 0) return MGPS_OK;
    const int units = nz / unit, minUnits = std::max(1, ((kTile << (D - 1)) + unit - 1) / unit);  // (every rank: 16 planes of the coarsest distributed level)
    if (units < size * minUnits) return MGPS_OK;
    // Load of a unit of planes: active cells + w x BOUNDARY cells.  The weights are measured (tools/slab_compute_bound.py, 1024^3,
    // one rank at a time with a null transport, round 5: P = 1, 2, 4, 8 fitted by cycle = c x active planes + F x faces + T on
    // rank 0 + a constant): c = 10.8 us per 1024^2 plane of liquid, F = 60 us for a z face of 894^2 BOUNDARY cells (the box form
    // of the band stage: 5.6 planes' worth, w = 6 -- the graph form of rounds 2-4 cost five times that, w = 30, which is what the
    // pass-by-pass stage still gets), T = 0.23 ms for the collapsed tail on rank 0 = 2.3 % of the whole load, which everybody
    // waits for
    const double kBoundaryWeight = (o.fuse_band_passes && o.deep_band_halo && o.band_iterations >= 1 && o.band_iterations <= kBandMaxDepth) ? 6.0 : 30.0;
    constexpr double kTailShare = 0.032;  // (2.3 % by the fit; rank 0 also pays for the EXTERIOR planes in front of the liquid: measured, 3.2 % levels rank 0 with the middle ranks at 1024^3 / 8)
    std::vector<double> load(size_t(units), 0.0);
    const size_t plane = size_t(nx) * ny;
    {
        std::vector<std::thread> pool;
        const int nt = std::max(1, std::min(units, int(std::thread::hardware_concurrency() ? std::thread::hardware_concurrency() : 4)));
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&, t] {
                for (int u = t; u < units; u += nt) {
                    const uint8_t *p = labels + size_t(u) * unit * plane;
                    size_t n = 0, nb = 0;
                    for (size_t c = 0; c < size_t(unit) * plane; ++c) {
                        n += isActive(p[c]);
                        nb += p[c] == MGPS_BOUNDARY_CELL;
                    }
                    load[size_t(u)] = double(n) + kBoundaryWeight * double(nb);
                }
            });
        for (auto &th : pool) th.join();
    }
    double total = 0;
    for (double v : load) total += v;
    if (total == 0) return MGPS_OK;
    const double kTailLoad = kTailShare * total;
    // the cuts that minimise the largest per-rank load (dynamic programme over unit boundaries: units <= nz / 16, ranks <= 8),
    // every rank at least minUnits; ties go to the more even plane counts
    std::vector<double> prefix(size_t(units) + 1, 0.0);
    for (int u = 0; u < units; ++u) prefix[size_t(u) + 1] = prefix[size_t(u)] + load[size_t(u)];
    const double kInf = 1e300;
    std::vector<std::vector<double>> best(size_t(size) + 1, std::vector<double>(size_t(units) + 1, kInf));
    std::vector<std::vector<int>> from(size_t(size) + 1, std::vector<int>(size_t(units) + 1, -1));
    best[0][0] = 0.0;
    for (int r = 1; r <= size; ++r)
        for (int u = r * minUnits; u <= units - (size - r) * minUnits; ++u)
            for (int v = (r - 1) * minUnits; v <= u - minUnits; ++v) {
                if (best[size_t(r) - 1][size_t(v)] >= kInf) continue;
                // (a whisker per plane keeps slabs of equal load equal in size as well)
                const double cost = std::max(best[size_t(r) - 1][size_t(v)], prefix[size_t(u)] - prefix[size_t(v)] + (r == 1 ? kTailLoad : 0.0) +
                                                                                1e-9 * total * double(u - v) / units);
                if (cost < best[size_t(r)][size_t(u)]) {
                    best[size_t(r)][size_t(u)] = cost;
                    from[size_t(r)][size_t(u)] = v;
                }
            }
    if (best[size_t(size)][size_t(units)] >= kInf) return MGPS_OK;
    double evenWorst = 0.0;  // what the even cut costs
    for (int r = 0; r < size; ++r)
        evenWorst = std::max(evenWorst, prefix[size_t(out_splits[r + 1] / unit)] - prefix[size_t(out_splits[r] / unit)] + (r == 0 ? kTailLoad : 0.0));
    if (best[size_t(size)][size_t(units)] >= 0.98 * evenWorst) return MGPS_OK;  // nothing to gain: keep the even cut
    for (int r = size, u = units; r > 0; --r) {
        out_splits[r] = u * unit;
        u = from[size_t(r)][size_t(u)];
    }
    out_splits[0] = 0;
    if (distributedLevelsFor(out_splits, size, nx, ny, mg_levels, false, o) != D) uniform();  // (never observed; the even cut always stands)
    return MGPS_OK;
}
MGPS_API_CATCH(nullptr)

}  // extern "C"
namespace {
// ---- a slab rank set up on the device (round 5) ---------------------------------------------------------------------------
// What createSlabImpl below does through the host builder -- 217-280 ms for a 1024^3 / 8 rank, of which 90 ms went into coarsening
// the GLOBAL labels on every rank's host and 80 ms into the halo groups of the graph-form band stage -- with the kernels of
// createWholeOnDevice on a window of the labels:
//   * per level a label BUFFER: the owned planes and, on both sides, ghost planes of labels (kLabelGhost0 x 2^(Dmax-1-l) planes: the
//     buffers of two levels nest, the coarsening kernel runs on them unchanged).  Real labels as far as the coarser levels' windows
//     need them (the rank's window of the global labels goes up once, 1 B per cell), EXTERIOR beyond;
//   * band masks, band list, operator rows and band boxes are built on the buffer as if it were a whole grid; the boxes are
//     clipped to the owned planes, their regions reach into the ghost zone -- the neighbours' cells they read are found from the
//     box lists themselves and ASKED FOR at set-up (the ranks trade index lists once; nobody has to derive a neighbour's
//     structure), and live in the deep ghost planes of the grids from then on (DevLevel::BoxHalo);
//   * the rank's own lists (band list with rows, activity runs, plane blocks, Gauss-Seidel tiles) are cut out of the buffer's;
//   * the face weights of the kSlabGhostPlanes planes next to a cut come from the neighbour (operator rows of the ghost zone's
//     general cells);
//   * level cap and shell tests are agreed by an all-reduce of the ranks' flags; rank 0 receives the collapse level's labels and
//     builds the tail with createWholeOnDevice.
// Every failure is folded into an all-reduce that all ranks take part in before the next collective: no rank leaves alone.
constexpr int kSlabGhostPlanes = kBandMaxDepth + 1;  // planes of every grid beyond its owned ones: a box region reaches depth + 1 past its owned box
constexpr int kLabelGhost0 = 16;                     // label ghost planes of the coarsest distributed level's buffer (>= the reach below, tile-aligned)

// host -> device from pageable memory: slices copied into a page-locked block by several threads, each slice sent as soon as
// it is there (a pageable array uploads at ~3 GB/s on this platform, a page-locked one at ~55)
int uploadPageable(mgps_solver *h, void *dst, const void *src, size_t bytes, int device)
{
    if (bytes == 0) return MGPS_OK;
    void *stage = bytes >= (size_t(4) << 20) ? pinnedAlloc(bytes) : nullptr;
    if (!stage) {
        MGPS_HIP(h, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return MGPS_OK;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = int(std::max<size_t>(1, std::min<size_t>(hw ? hw : 4, std::min<size_t>(16, bytes >> 22))));
    std::atomic<int> failed{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t)
        pool.emplace_back([&, t] {
            if (hipSetDevice(device) != hipSuccess) {
                failed = 1;
                return;
            }
            const size_t per = ((bytes + size_t(nt) - 1) / size_t(nt) + 4095) & ~size_t(4095);
            for (size_t off = size_t(t) * per, end = std::min(bytes, off + per); off < end;) {
                const size_t n = std::min<size_t>(end - off, size_t(8) << 20);
                std::memcpy(static_cast<char *>(stage) + off, static_cast<const char *>(src) + off, n);
                if (hipMemcpyAsync(static_cast<char *>(dst) + off, static_cast<char *>(stage) + off, n, hipMemcpyHostToDevice, nullptr) != hipSuccess) failed = 1;
                off += n;
            }
        });
    for (auto &th : pool) th.join();
    const hipError_t e = hipStreamSynchronize(nullptr);
    pinnedFree(stage);
    if (failed || e != hipSuccess) return failH(h, MGPS_ERR_HIP, "label upload failed");
    return MGPS_OK;
}

int createSlabOnDevice(mgps_solver **out, int nx, int ny, int nzg, const uint8_t *labels_global_host, const float *wx_slab, const float *wy_slab,
                       const float *wz_slab, bool weightsOnDevice, int mgLevels, bool useGS, const mgps_options &o, const mgps_comm *comm, const int *splits, int device)
{
    if (mgLevels < 2 || nx < 2 || ny < 2 || nzg < 2 || (nx & 1) || (ny & 1) || (nzg & 1))
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: extents must be even and >= 2, mg_levels >= 2");
    if (size_t(nx) * ny * nzg > size_t(0x7fffffff)) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: more than 2^31-1 cells per grid");
    for (int l = 1; l < mgLevels; ++l)
        if (((nx >> (l - 1)) & 1) || ((ny >> (l - 1)) & 1) || ((nzg >> (l - 1)) & 1) || (nx >> l) < 1 || (ny >> l) < 1 || (nzg >> l) < 1)
            return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: extents are not divisible by 2^(levels-1)");
    if (o.band_width < 1 || o.band_width > 8 || o.band_iterations < 0)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: band_width 1 .. 8, band_iterations >= 0");
    const int P = comm->size, rank = comm->rank;
    const bool lo = rank > 0, hi = rank < P - 1;
    // the deepest the distributed part can be (the level cap can only shorten the hierarchy): label windows are sized for it
    const int Dmax = distributedLevelsFor(splits, P, nx, ny, mgLevels, useGS, o);
    if (Dmax < 1)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT,
                     "mgps_create_slab: every rank needs at least 16 planes cut on even planes (multiples of 16 with Gauss-Seidel) and the "
                     "hierarchy at least 2 levels");
    auto *h = new mgps_solver();
    h->opt = o;
    h->useGS = useGS;
    h->device = device;
    h->dist = true;
    h->ghost = kSlabGhostPlanes;
    h->requestedLevels = mgLevels;
    h->comm = mgps_comm{};
    std::memcpy(&h->comm, comm, size_t(comm->struct_size));  // (struct_size bytes are the caller's; the rest stays NULL)
    h->comm.struct_size = int(sizeof(mgps_comm));
    h->splits.assign(splits, splits + P + 1);
    auto bail = [&](int code) {
        setLastGlobalError(h->lastError);
        freeAll(h);
        return code;
    };
    StageClock sclock(setupTimingOn());
    DevScratch tmp;
    // a verdict all ranks share: the largest status (0 = fine).  Every rank calls it at the same places.
    auto agree = [&](int mine, int *all) -> int {
        double v = double(mine);
        if (h->comm.allreduce(h->comm.user, &v, 1, 1) != 0) return failH(h, MGPS_ERR_COMM, "all-reduce failed during set-up");
        *all = int(v);
        return MGPS_OK;
    };
    const int z0 = splits[rank], z1 = splits[rank + 1];
    const bool wantBoxes = o.fuse_band_passes && o.deep_band_halo && o.band_iterations >= 1 && o.band_iterations <= kBandMaxDepth;
    const int depth = o.band_iterations;
    // planes beyond the owned ones in which band masks must be right: a box region reaches depth + 1 planes out, classifying its
    // cells looks one plane farther, band membership there depends on BOUNDARY cells band_width - 1 planes farther; + 1 spare
    const int reach = (o.band_width - 1) + (wantBoxes ? depth + 2 : 1) + 1;
    if (reach > kLabelGhost0) return bail(failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: band_width + band_iterations reach past 16 planes"));

    // ---- label buffers of levels 0 .. Dmax: spare plane | elo ghost planes | owned planes | elo ghost planes | spare plane
    struct Buf {
        Dims d;          // the buffer as a grid
        Dims own;        // the owned planes as a grid
        int elo = 0;     // ghost planes on either side
        int real = 0;    // of which this many next to the owned planes hold real labels (inside the global grid)
        uint8_t *base = nullptr;
        uint8_t *lab() const { return base + size_t(d.nx) * d.ny; }                      // buffer plane 0
        uint8_t *owned() const { return base + size_t(1 + elo) * size_t(d.nx) * d.ny; }  // owned plane 0
    };
    std::vector<Buf> B;
    B.resize(size_t(Dmax) + 1);
    for (int l = 0; l <= Dmax; ++l) {
        Buf &b = B[size_t(l)];
        b.elo = l < Dmax ? kLabelGhost0 << (Dmax - 1 - l) : kLabelGhost0 / 2;
        b.own = Dims{nx >> l, ny >> l, (z1 - z0) >> l};
        b.d = Dims{nx >> l, ny >> l, b.own.nz + 2 * b.elo};
    }
    B[size_t(Dmax)].real = 3;  // (the collapse level is marked on its owned planes: one plane of neighbours, and one for theirs)
    if (Dmax >= 1) B[size_t(Dmax) - 1].real = std::max(reach, 2 * B[size_t(Dmax)].real + 2);
    for (int l = Dmax - 2; l >= 0; --l) B[size_t(l)].real = 2 * B[size_t(l) + 1].real + 2;
    for (int l = 0; l <= Dmax; ++l) {
        Buf &b = B[size_t(l)];
        if (b.real > b.elo) return bail(failH(h, MGPS_ERR_INTERNAL, "slab set-up: label window past its buffer"));
        const size_t plane = size_t(b.d.nx) * b.d.ny;
        int rc = tmp.get(h, &b.base, (size_t(b.d.nz) + 2) * plane);
        if (rc != MGPS_OK) return bail(rc);
        if (hipMemsetAsync(b.base, MGPS_EXTERIOR_CELL, (size_t(b.d.nz) + 2) * plane, nullptr) != hipSuccess) return bail(failH(h, MGPS_ERR_HIP, "memset failed"));
    }
    int status = MGPS_OK;
    {  // the rank's window of the fine labels
        const Buf &b = B[0];
        const int a = std::max(0, z0 - b.real), e = std::min(nzg, z1 + b.real);
        const size_t plane = size_t(nx) * ny;
        status = uploadPageable(h, b.owned() + ptrdiff_t(a - z0) * ptrdiff_t(plane), labels_global_host + size_t(a) * plane, size_t(e - a) * plane, device);
    }
    sclock.lap("slab: label window upload");
    // flags: per level l [2 l] shell broken, [2 l + 1] holds an active cell; then interior rule, weight rule, boxes broken per level
    const size_t nflags = size_t(2 * (mgLevels + 1) + 2 + mgLevels);
    int *flags = nullptr;
    if (status == MGPS_OK) status = tmp.get(h, &flags, nflags);
    const int fInterior = 2 * (mgLevels + 1), fWeights = fInterior + 1, fBoxes = fInterior + 2;
    std::vector<int> hflags(nflags, 0);
    // (levels beyond Dmax are the tail's: rank 0 applies cap and shell tests to them when it builds the tail, and tells the others)
    if (status == MGPS_OK) status = [&]() -> int {
        MGPS_HIP(h, hipMemsetAsync(flags, 0, nflags * sizeof(int), nullptr));
        MGPS_LAUNCH(h, launchShellCheckSlab(nullptr, B[0].own, B[0].owned(), !lo, !hi, flags));
        MGPS_LAUNCH(h, launchAnyActive(nullptr, B[0].own, B[0].owned(), flags + 1));
        for (int l = 1; l <= Dmax; ++l) {
            int scratchFlag = 0;
            (void)scratchFlag;
            MGPS_LAUNCH(h, launchCoarsenLabels(nullptr, B[size_t(l) - 1].d, B[size_t(l) - 1].lab(), B[size_t(l)].lab(), flags + 2 * l + 1));
            MGPS_LAUNCH(h, launchShellCheckSlab(nullptr, B[size_t(l)].own, B[size_t(l)].owned(), !lo, !hi, flags + 2 * l));
        }
        // (the buffers' "active" flags cover ghost planes too: an OR over ranks either way)
        for (int l = 1; l <= Dmax; ++l) MGPS_LAUNCH(h, launchMarkBoundary(nullptr, B[size_t(l)].d, B[size_t(l)].lab()));
        MGPS_HIP(h, hipMemcpy(hflags.data(), flags, nflags * sizeof(int), hipMemcpyDeviceToHost));
        return MGPS_OK;
    }();
    sclock.lap("slab: labels of all levels");
    // ---- the ranks' flags together: level cap (MG.cpp:238-253), shell tests
    int levels = mgLevels;
    {
        std::vector<double> v(size_t(2 * mgLevels) + 1, 0.0);
        for (int q = 0; q < 2 * mgLevels; ++q) v[size_t(q)] = hflags[size_t(q)] ? 1.0 : 0.0;
        v[size_t(2 * mgLevels)] = double(status);
        if (h->comm.allreduce(h->comm.user, v.data(), int(v.size()), 1) != 0) return bail(failH(h, MGPS_ERR_COMM, "all-reduce failed during set-up"));
        if (int(v[size_t(2 * mgLevels)]) != MGPS_OK) {
            if (status == MGPS_OK) failH(h, int(v[size_t(2 * mgLevels)]), "slab set-up failed on another rank");
            return bail(status != MGPS_OK ? status : int(v[size_t(2 * mgLevels)]));
        }
        if (v[0] != 0.0) return bail(failH(h, MGPS_ERR_HIERARCHY, "labels need an EXTERIOR shell on all six sides (unitTestExteriorCells)"));
        if (v[1] == 0.0) return bail(failH(h, MGPS_ERR_HIERARCHY, "no INTERIOR or BOUNDARY cell in the domain"));
        for (int l = 1; l <= Dmax && l < mgLevels; ++l) {
            if (v[size_t(2 * l)] != 0.0)
                return bail(failH(h, MGPS_ERR_HIERARCHY,
                                  "level " + std::to_string(l) + " has no EXTERIOR shell (unitTestExteriorCells, MG.cpp:252): " + std::to_string(mgLevels) +
                                      " levels need 2^(levels-1) = " + std::to_string(1 << (mgLevels - 1)) + " EXTERIOR cells on every side of the solver grid"));
            if (v[size_t(2 * l + 1)] == 0.0) {
                levels = l - 1;  // the reference drops the last solvable level too (MG.cpp:245)
                break;
            }
        }
        if (levels < 1) return bail(failH(h, MGPS_ERR_HIERARCHY, "level cap left no multigrid level (first coarse level has no solvable cell)"));
    }
    const int D = distributedLevelsFor(splits, P, nx, ny, levels, useGS, o);
    if (D < 1)
        return bail(failH(h, MGPS_ERR_INVALID_ARGUMENT,
                          "mgps_create_slab: every rank needs at least 16 planes cut on even planes (multiples of 16 with Gauss-Seidel) and the "
                          "hierarchy at least 2 levels"));
    h->distLevels = D;
    h->totalLevels = levels;  // (the tail may still shorten it: the level cap below the collapse level)
    h->lv.resize(size_t(D) + 1);

    // ---- face weights: the slab's own, and kSlabGhostPlanes planes of each neighbour's next to the cuts
    const int nzl = z1 - z0, gw = kSlabGhostPlanes;
    const size_t wplane[3] = {size_t(nx + 1) * ny, size_t(nx) * (ny + 1), size_t(nx) * ny};
    const size_t wn[3] = {wplane[0] * size_t(nzl), wplane[1] * size_t(nzl), wplane[2] * size_t(nzl + 1)};
    float *wlo[3] = {nullptr, nullptr, nullptr}, *whi[3] = {nullptr, nullptr, nullptr};
    status = [&]() -> int {
        const float *wh[3] = {wx_slab, wy_slab, wz_slab};
        if (weightsOnDevice && o.borrow_device_weights) {
            for (int a = 0; a < 3; ++a) h->w[a] = const_cast<float *>(wh[a]);
            h->weightsBorrowed = true;
        } else
            for (int a = 0; a < 3; ++a) {
                MGPS_TRY(devAlloc(h, &h->w[a], wn[a], false));
                MGPS_HIP(h, hipMemcpyAsync(h->w[a], wh[a], wn[a] * sizeof(float), weightsOnDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, nullptr));
            }
        for (int a = 0; a < 3; ++a) {
            MGPS_TRY(tmp.get(h, &wlo[a], wplane[a] * size_t(gw)));
            MGPS_TRY(tmp.get(h, &whi[a], wplane[a] * size_t(gw)));
        }
        MGPS_HIP(h, hipStreamSynchronize(nullptr));
        return MGPS_OK;
    }();
    {
        int all = MGPS_OK;
        const int rc = agree(status, &all);
        if (rc != MGPS_OK) return bail(rc);
        if (all != MGPS_OK) {
            if (status == MGPS_OK) failH(h, all, "slab set-up failed on another rank");
            return bail(status != MGPS_OK ? status : all);
        }
    }
    // what travels: wx / wy planes [0, gw) down and [nzl - gw, nzl) up; of wz the faces [1, gw] down and [nzl - gw, nzl - 1] up
    // (before the exchange the ghost planes hold a mirror image of the rank's own: a transport that moves nothing -- the
    // compute-bound tools -- leaves plausible weights there)
    status = MGPS_OK;
    for (int a = 0; a < 3 && nzl >= gw; ++a) {
        const size_t bytes = wplane[a] * size_t(gw) * sizeof(float);
        const float *down = h->w[a] + (a == 2 ? wplane[a] : 0), *up = h->w[a] + wplane[a] * size_t(nzl - gw);
        if (hipMemcpyAsync(wlo[a], up, bytes, hipMemcpyDeviceToDevice, nullptr) != hipSuccess || hipMemcpyAsync(whi[a], down, bytes, hipMemcpyDeviceToDevice, nullptr) != hipSuccess)
            status = failH(h, MGPS_ERR_HIP, "weight ghost planes: copy failed");
    }
    if (hipStreamSynchronize(nullptr) != hipSuccess && status == MGPS_OK) status = failH(h, MGPS_ERR_HIP, "weight ghost planes: copy failed");
    {  // (a rank whose copies failed must not leave its neighbours waiting in the exchange below)
        int all = MGPS_OK;
        const int rc = agree(status, &all);
        if (rc != MGPS_OK) return bail(rc);
        if (all != MGPS_OK) {
            if (status == MGPS_OK) failH(h, all, "slab set-up failed on another rank");
            return bail(status != MGPS_OK ? status : all);
        }
    }
    for (int a = 0; a < 3 && nzl >= gw; ++a) {
        const size_t bytes = wplane[a] * size_t(gw) * sizeof(float);
        const float *down = h->w[a] + (a == 2 ? wplane[a] : 0), *up = h->w[a] + wplane[a] * size_t(nzl - gw);
        if (P > 1 && h->comm.exchange(h->comm.user, lo ? down : nullptr, bytes, lo ? wlo[a] : nullptr, bytes, hi ? up : nullptr, bytes, hi ? whi[a] : nullptr, bytes, nullptr) != 0)
            return bail(failH(h, MGPS_ERR_COMM, "weight ghost planes: exchange failed"));  // (a transport failure: nobody is left waiting for this rank's data only)
    }
    (void)hipStreamSynchronize(nullptr);
    sclock.lap("slab: weights + their ghost planes");

    // ---- the distributed levels on their buffers
    struct LevelTmp {
        int nt = 0, tx = 0, ty = 0, tz = 0;
        uint32_t *mask = nullptr;
        uint16_t *prefix = nullptr;
        int32_t *tileCount = nullptr, *tileKind = nullptr, *tileStart = nullptr, *scan = nullptr;
        int32_t *sorted = nullptr, *general = nullptr, *genRank = nullptr, *bandEntry = nullptr, *extBand = nullptr;
        int32_t *own = nullptr, *ownGen = nullptr, *ownRank = nullptr, *ownGenRank = nullptr;
        uint8_t *diagS = nullptr, *extDiag = nullptr, *chunkFlags = nullptr, *planeFlags = nullptr;
        int32_t *gcount[3] = {nullptr, nullptr, nullptr}, *gat[3] = {nullptr, nullptr, nullptr};
        int32_t *tileFlags = nullptr, *tileRank = nullptr, *bandTiles = nullptr, *boxTiles = nullptr;
        uint8_t *tileBits = nullptr;
        int *runCounts = nullptr;
        int32_t *listCounts = nullptr;
        int runCells = 0, listLen = 0, nbandExt = 0, nGenExt = 0, planeZc = 0, nBoxTiles = 0, nplaneBlocks = 0;
        size_t nfine = 0, nplane = 0;
        SlabWindow win;
    };
    std::vector<LevelTmp> T;
    T.resize(size_t(D));
    std::vector<int> boxesOk(size_t(D), 0);
    status = [&]() -> int {
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            LevelTmp &t = T[size_t(l)];
            const Buf &b = B[size_t(l)];
            L.d = b.own;
            L.z0 = z0 >> l;
            L.z1 = z1 >> l;
            L.elo = b.elo;
            t.win = SlabWindow{b.elo, b.elo + b.own.nz, reach};
            t.tx = (b.d.nx + kTile - 1) / kTile;
            t.ty = (b.d.ny + kTile - 1) / kTile;
            t.tz = (b.d.nz + kTile - 1) / kTile;
            t.nt = t.tx * t.ty * t.tz;
            MGPS_TRY(tmp.get(h, &t.mask, size_t(t.nt) * 128));
            MGPS_TRY(tmp.get(h, &t.prefix, size_t(t.nt) * 128));
            MGPS_TRY(tmp.get(h, &t.tileCount, size_t(t.nt)));
            MGPS_TRY(tmp.get(h, &t.tileKind, size_t(t.nt)));
            MGPS_TRY(tmp.get(h, &t.tileStart, size_t(t.nt) + 1));
            MGPS_TRY(tmp.get(h, &t.scan, scanScratchInts(b.d.cells())));
            MGPS_TRY(tmp.get(h, &t.tileFlags, size_t(t.nt)));
            MGPS_TRY(tmp.get(h, &t.tileRank, size_t(t.nt) + 1));
            MGPS_TRY(tmp.get(h, &t.bandTiles, size_t(t.nt)));
            MGPS_TRY(tmp.get(h, &t.tileBits, size_t(t.nt)));
            MGPS_HIP(h, hipMemsetAsync(t.mask, 0, size_t(t.nt) * 128 * sizeof(uint32_t), nullptr));
            MGPS_HIP(h, hipMemsetAsync(t.prefix, 0, size_t(t.nt) * 128 * sizeof(uint16_t), nullptr));
            MGPS_HIP(h, hipMemsetAsync(t.tileCount, 0, size_t(t.nt) * sizeof(int32_t), nullptr));
            MGPS_LAUNCH(h, launchBandCandidates(nullptr, b.d, b.lab(), t.tileKind, t.tileBits, t.tileFlags, t.tileRank, t.bandTiles, t.scan, &t.win));
        }
        for (int l = 0; l < D; ++l) {
            LevelTmp &t = T[size_t(l)];
            const Buf &b = B[size_t(l)];
            int ncand = 0;
            MGPS_HIP(h, hipMemcpy(&ncand, t.tileRank + t.nt, sizeof(int), hipMemcpyDeviceToHost));
            MGPS_LAUNCH(h, launchBandMasks(nullptr, b.d, b.lab(), o.band_width, t.mask, t.prefix, t.tileCount, t.tileKind, l == 0 ? flags + fInterior : nullptr, t.bandTiles,
                                           ncand, &t.win));
            MGPS_LAUNCH(h, launchExclusiveScan(nullptr, t.tileCount, t.tileStart, size_t(t.nt), t.scan));
        }
        for (int l = 0; l < D; ++l) MGPS_HIP(h, hipMemcpy(&T[size_t(l)].nbandExt, T[size_t(l)].tileStart + T[size_t(l)].nt, sizeof(int), hipMemcpyDeviceToHost));
        // band lists of the buffers in reference order, classification (level 0: with the weights and their ghost planes), rows
        WeightView wv;
        for (int a = 0; a < 3; ++a) {
            wv.w[a] = h->w[a];
            wv.lo[a] = wlo[a];
            wv.hi[a] = whi[a];
        }
        wv.nz = nzl;
        wv.gw = nzl >= gw ? gw : 0;
        wv.k0 = B[0].elo;
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            LevelTmp &t = T[size_t(l)];
            const Buf &b = B[size_t(l)];
            const size_t nb = size_t(t.nbandExt);
            MGPS_TRY(tmp.get(h, &t.sorted, nb));
            MGPS_TRY(tmp.get(h, &t.general, nb));
            MGPS_TRY(tmp.get(h, &t.genRank, nb + 1));
            MGPS_TRY(tmp.get(h, &t.bandEntry, nb));
            MGPS_TRY(tmp.get(h, &t.diagS, nb));
            MGPS_TRY(tmp.get(h, &t.extBand, nb));
            MGPS_TRY(tmp.get(h, &t.extDiag, nb));
            MGPS_TRY(tmp.get(h, &t.own, nb));
            MGPS_TRY(tmp.get(h, &t.ownGen, nb));
            MGPS_TRY(tmp.get(h, &t.ownRank, nb + 1));
            MGPS_TRY(tmp.get(h, &t.ownGenRank, nb + 1));
            if (t.nbandExt > 0) {
                MGPS_LAUNCH(h, launchBandFill(nullptr, b.d, t.mask, t.prefix, t.tileStart, t.sorted));
                MGPS_LAUNCH(h, launchBandClassify(nullptr, b.d, b.lab(), l == 0 ? wv : WeightView{}, t.sorted, t.nbandExt, t.diagS, t.general, l == 0 ? flags + fWeights : nullptr,
                                                  &t.win));
            }
            MGPS_LAUNCH(h, launchExclusiveScan(nullptr, t.general, t.genRank, nb, t.scan));
            const size_t plane = size_t(b.d.nx) * b.d.ny;
            MGPS_LAUNCH(h, launchOwnedFlags(nullptr, t.sorted, t.general, t.nbandExt, int32_t(size_t(t.win.own0) * plane), int32_t(size_t(t.win.own1) * plane), t.own, t.ownGen));
            MGPS_LAUNCH(h, launchExclusiveScan(nullptr, t.own, t.ownRank, nb, t.scan));
            MGPS_LAUNCH(h, launchExclusiveScan(nullptr, t.ownGen, t.ownGenRank, nb, t.scan));
            (void)L;
        }
        for (int l = 0; l < D; ++l) {
            LevelTmp &t = T[size_t(l)];
            DevLevel &L = h->lv[size_t(l)];
            MGPS_HIP(h, hipMemcpy(&t.nGenExt, t.genRank + t.nbandExt, sizeof(int), hipMemcpyDeviceToHost));
            MGPS_HIP(h, hipMemcpy(&L.nband, t.ownRank + t.nbandExt, sizeof(int), hipMemcpyDeviceToHost));
            MGPS_HIP(h, hipMemcpy(&L.nbndGeneral, t.ownGenRank + t.nbandExt, sizeof(int), hipMemcpyDeviceToHost));
        }
        MGPS_HIP(h, hipMemcpy(hflags.data() + fInterior, flags + fInterior, 2 * sizeof(int), hipMemcpyDeviceToHost));
        if (hflags[size_t(fInterior)] || hflags[size_t(fWeights)])
            return failH(h, MGPS_ERR_HIERARCHY, "labels/weights violate the BOUNDARY-cell rules (unitTestBoundaryCells): run mgps_fields_set_boundary_labels");
        sclock.lap("slab: band lists + classification");
        // the buffer's device order and rows, the codes, the rank's own lists, activity
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            LevelTmp &t = T[size_t(l)];
            const Buf &b = B[size_t(l)];
            const size_t plane = size_t(b.d.nx) * b.d.ny;
            MGPS_TRY(devAlloc(h, &L.extRows, size_t(7) * size_t(t.nGenExt), false));
            MGPS_LAUNCH(h, launchBandSplit(nullptr, b.d, b.lab(), l == 0 ? wv : WeightView{}, t.sorted, t.nbandExt, t.diagS, t.genRank, t.extBand, t.extDiag, t.bandEntry,
                                           L.extRows));
            MGPS_LAUNCH(h, launchPatchSimpleCodes(nullptr, b.lab(), t.extBand, t.extDiag, t.nGenExt, t.nbandExt));
            MGPS_TRY(devAlloc(h, &L.band, size_t(L.nband), false));
            MGPS_TRY(devAlloc(h, &L.bandDiag, size_t(L.nband), false));
            MGPS_TRY(devAlloc(h, &L.bandTmp, size_t(L.nband), false));
            MGPS_TRY(devAlloc(h, &L.rows, size_t(7) * size_t(L.nbndGeneral), false));
            MGPS_LAUNCH(h, launchBandSplitOwned(nullptr, t.sorted, t.nbandExt, int32_t(size_t(t.win.own0) * plane), t.diagS, t.ownRank, t.ownGenRank, t.genRank, L.extRows, L.band,
                                                L.bandDiag, L.rows));
            // per owned tile (+ 1) the first of its general cells in the rank's list (the mixed Gauss-Seidel tiles look their rows up there)
            const int ntOwn = t.tx * t.ty * ((b.own.nz + kTile - 1) / kTile), tileOff = (b.elo / kTile) * t.tx * t.ty;
            MGPS_TRY(devAlloc(h, &L.tileBndStart, size_t(ntOwn) + 1, false));
            MGPS_LAUNCH(h, launchGather(nullptr, t.ownGenRank, t.tileStart + tileOff, ntOwn + 1, L.tileBndStart));
            // activity of the owned planes
            t.nfine = (b.own.cells() + kSegCells - 1) / kSegCells;
            t.planeZc = planeSweepZc(b.own.nx, b.own.ny, b.own.nz);
            MGPS_TRY(tmp.get(h, &t.chunkFlags, t.nfine));
            if (t.planeZc) {
                t.nplane = size_t((b.own.nx + 255) / 256) * size_t((b.own.ny + kPlaneRows - 1) / kPlaneRows) * size_t((b.own.nz + t.planeZc - 1) / t.planeZc);
                MGPS_TRY(tmp.get(h, &t.planeFlags, t.nplane));
                MGPS_HIP(h, hipMemsetAsync(t.planeFlags, 0, t.nplane, nullptr));
            }
            MGPS_LAUNCH(h, launchActivityFlags(nullptr, b.own, b.owned(), t.chunkFlags, t.planeFlags, t.planeZc));
            if (t.planeZc) {  // (a block next to a cut with active cells only across it still takes part in the residual + restriction pair)
                if (lo) MGPS_LAUNCH(h, launchGhostPlaneBlockFlags(nullptr, b.own, b.owned(), -1, t.planeZc, t.planeFlags));
                if (hi) MGPS_LAUNCH(h, launchGhostPlaneBlockFlags(nullptr, b.own, b.owned(), b.own.nz, t.planeZc, t.planeFlags));
            }
            MGPS_TRY(tmp.get(h, &t.runCounts, 4));
            MGPS_HIP(h, hipMemsetAsync(t.runCounts, 0, 4 * sizeof(int), nullptr));
            MGPS_LAUNCH(h, launchCountRuns(nullptr, t.chunkFlags, t.nfine, t.runCounts));
        }
        // boxes of the fused band stage, clipped to the owned planes: counts
        for (int l = 0; l < D && wantBoxes; ++l) {
            LevelTmp &t = T[size_t(l)];
            const Buf &b = B[size_t(l)];
            if (!boxPlaneFits(b.d)) continue;
            boxesOk[size_t(l)] = 1;
            if (t.nbandExt == 0) continue;
            MGPS_TRY(tmp.get(h, &t.boxTiles, size_t(t.nt)));
            MGPS_LAUNCH(h, launchBoxTileList(nullptr, b.d, t.tileStart, t.tileFlags, t.tileRank, t.boxTiles, t.scan));
            MGPS_HIP(h, hipMemcpy(&t.nBoxTiles, t.tileRank + t.nt, sizeof(int), hipMemcpyDeviceToHost));
            for (int q = 0; q < 3; ++q) {
                MGPS_TRY(tmp.get(h, &t.gcount[q], size_t(t.nBoxTiles)));
                MGPS_TRY(tmp.get(h, &t.gat[q], size_t(t.nBoxTiles) + 1));
            }
            MGPS_LAUNCH(h, launchBandBoxesCount(nullptr, b.d, b.lab(), t.mask, t.prefix, t.tileStart, t.bandEntry, t.extDiag, depth, t.boxTiles, t.nBoxTiles, t.gcount,
                                                flags + fBoxes + l, &t.win));
            for (int q = 0; q < 3; ++q) MGPS_LAUNCH(h, launchExclusiveScan(nullptr, t.gcount[q], t.gat[q], size_t(t.nBoxTiles), t.scan));
        }
        // lists of the owned planes
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            LevelTmp &t = T[size_t(l)];
            const Buf &b = B[size_t(l)];
            int rc4[4] = {0, 0, 0, 0};
            MGPS_HIP(h, hipMemcpy(rc4, t.runCounts, sizeof(rc4), hipMemcpyDeviceToHost));
            const int64_t rc64[4] = {rc4[0], rc4[1], rc4[2], rc4[3]};
            const int runCells = chooseRunCells(rc64);
            const size_t nruns = (t.nfine + size_t(runCells / kSegCells) - 1) / size_t(runCells / kSegCells);
            uint8_t *runFlags = t.chunkFlags;
            if (runCells != kSegCells) {
                MGPS_TRY(tmp.get(h, &runFlags, nruns));
                MGPS_LAUNCH(h, launchFoldRunFlags(nullptr, t.chunkFlags, t.nfine, runCells, runFlags));
            }
            int activeRuns = 0;
            for (int z = 0; z < 4; ++z)
                if (kRunSizes[z] == runCells) activeRuns = rc4[z];
            const int perGroup = kChunkCells / runCells, listLen = (activeRuns + perGroup - 1) / perGroup * perGroup;
            {
                int32_t *tmpFlags = nullptr, *rnk = nullptr, *base = nullptr;
                MGPS_TRY(tmp.get(h, &tmpFlags, nruns));
                MGPS_TRY(tmp.get(h, &rnk, nruns + 1));
                MGPS_TRY(tmp.get(h, &base, 1));
                MGPS_TRY(devAlloc(h, &L.chunks, size_t(listLen), false));
                MGPS_LAUNCH(h, launchRunList(nullptr, b.own, runFlags, nruns, runCells, tmpFlags, rnk, t.scan, base, L.chunks, listLen));
            }
            MGPS_TRY(tmp.get(h, &t.listCounts, 5));
            MGPS_HIP(h, hipMemsetAsync(t.listCounts, 0, 5 * sizeof(int32_t), nullptr));
            {
                const int tileOff = (b.elo / kTile) * t.tx * t.ty, ntOwn = t.tx * t.ty * ((b.own.nz + kTile - 1) / kTile);
                const int tkOffset = ((z0 >> l) / kTile) & 1;  // (tile colours are the whole grid's; with Gauss-Seidel the cuts are multiples of 16 planes)
                int32_t **cls[4] = {&L.pure[0], &L.pure[1], &L.mixed[0], &L.mixed[1]};
                for (int q = 0; q < 4; ++q) {
                    MGPS_TRY(devAlloc(h, cls[q], size_t(ntOwn) / 2 + 1, false));
                    MGPS_LAUNCH(h, launchTileClassList(nullptr, b.own, t.tileKind + tileOff, q & 1, q >> 1, t.tileFlags, t.tileRank, *cls[q], t.scan, tkOffset));
                    MGPS_HIP(h, hipMemcpyAsync(t.listCounts + q, t.tileRank + ntOwn, sizeof(int32_t), hipMemcpyDeviceToDevice, nullptr));
                }
                if (t.nplane) {
                    int32_t *pf = nullptr, *pr = nullptr;
                    MGPS_TRY(tmp.get(h, &pf, t.nplane));
                    MGPS_TRY(tmp.get(h, &pr, t.nplane + 1));
                    MGPS_TRY(devAlloc(h, &L.planeBlocks, t.nplane, false));
                    MGPS_LAUNCH(h, launchByteList(nullptr, t.planeFlags, int(t.nplane), pf, pr, L.planeBlocks, t.scan));
                    MGPS_HIP(h, hipMemcpyAsync(t.listCounts + 4, pr + t.nplane, sizeof(int32_t), hipMemcpyDeviceToDevice, nullptr));
                }
            }
            t.runCells = runCells;
            t.listLen = listLen;
            if (l > 0) {
                MGPS_TRY(gridAlloc(h, &L.x, L.d));
                MGPS_TRY(gridAlloc(h, &L.b, L.d));
            }
            MGPS_TRY(gridAlloc(h, &L.r, L.d));
            MGPS_TRY(gridAlloc(h, &L.tmp, L.d));
        }
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            LevelTmp &t = T[size_t(l)];
            int32_t counts[5] = {0, 0, 0, 0, 0};
            MGPS_HIP(h, hipMemcpy(counts, t.listCounts, sizeof(counts), hipMemcpyDeviceToHost));
            L.npure[0] = counts[0];
            L.npure[1] = counts[1];
            L.nmixed[0] = counts[2];
            L.nmixed[1] = counts[3];
            t.nplaneBlocks = counts[4];
        }
        sclock.lap("slab: own lists");
        // boxes: arrays
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            LevelTmp &t = T[size_t(l)];
            const Buf &b = B[size_t(l)];
            if (!boxesOk[size_t(l)] || t.nbandExt == 0) continue;
            int tot[3] = {0, 0, 0}, brokenL = 0;
            for (int q = 0; q < 3; ++q) MGPS_HIP(h, hipMemcpy(&tot[q], t.gat[q] + t.nBoxTiles, sizeof(int), hipMemcpyDeviceToHost));
            MGPS_HIP(h, hipMemcpy(&brokenL, flags + fBoxes + l, sizeof(int), hipMemcpyDeviceToHost));
            if (brokenL) {
                boxesOk[size_t(l)] = 0;
                continue;
            }
            L.bandBoxes.depth = depth;
            L.bandBoxes.ngroups = tot[0];
            L.bandBoxes.listCount = size_t(tot[1]);
            L.bandBoxes.generalInts = 2 * size_t(tot[2]);
            L.bandBoxes.anyGeneral = tot[2] > 0;
            if (tot[0] == 0) continue;  // (no band-closure cell in the rank's planes)
            MGPS_TRY(devAlloc(h, &L.bandBoxes.info, size_t(kBoxInfoInts) * size_t(tot[0]), false));
            MGPS_TRY(devAlloc(h, &L.bandBoxes.list, size_t(tot[1]), false));
            MGPS_TRY(devAlloc(h, &L.bandBoxes.general, 2 * size_t(tot[2]), false));
            MGPS_LAUNCH(h, launchBandBoxesFill(nullptr, b.d, b.lab(), t.mask, t.prefix, t.tileStart, t.bandEntry, t.extDiag, depth, t.boxTiles, t.nBoxTiles, t.gat, L.bandBoxes.info,
                                               L.bandBoxes.list, L.bandBoxes.general, flags + fBoxes + l, &t.win));
            MGPS_TRY(finishBandBoxes(h, L, nullptr, &b.d));
            // cells as offsets from owned cell 0 (a region below the owned planes starts at a negative one)
            MGPS_LAUNCH(h, launchRebaseBoxes(nullptr, L.bandBoxes.info, L.bandBoxes.ngroups, int32_t(size_t(b.elo) * size_t(b.d.nx) * b.d.ny)));
            MGPS_HIP(h, hipMemcpy(&brokenL, flags + fBoxes + l, sizeof(int), hipMemcpyDeviceToHost));
            if (brokenL) boxesOk[size_t(l)] = 0;
        }
        MGPS_HIP(h, hipDeviceSynchronize());
        sclock.lap("slab: band boxes");
        return MGPS_OK;
    }();
    // ---- every rank takes the same form of the band stage on a level; every rank knows whether all are still fine
    {
        std::vector<double> v(size_t(D) + 1, 0.0);
        for (int l = 0; l < D; ++l) v[size_t(l)] = boxesOk[size_t(l)] ? 0.0 : 1.0;
        v[size_t(D)] = double(status);
        if (h->comm.allreduce(h->comm.user, v.data(), int(v.size()), 1) != 0) return bail(failH(h, MGPS_ERR_COMM, "all-reduce failed during set-up"));
        if (int(v[size_t(D)]) != MGPS_OK) {
            if (status == MGPS_OK) failH(h, int(v[size_t(D)]), "slab set-up failed on another rank");
            return bail(status != MGPS_OK ? status : int(v[size_t(D)]));
        }
        for (int l = 0; l < D; ++l) h->lv[size_t(l)].boxForm = v[size_t(l)] == 0.0;
    }
    // ---- level descriptions; the cells the neighbours' planes must deliver; the index lists traded
    const int G = h->ghost;
    auto keepBuffer = [&](int l) {  // the label buffer of level l leaves the scratch set: the level's codes (allocation base: the spare plane in front)
        h->lv[size_t(l)].codes = B[size_t(l)].base;
        for (auto it = tmp.ptrs.begin(); it != tmp.ptrs.end(); ++it)
            if (*it == B[size_t(l)].base) {
                tmp.ptrs.erase(it);
                break;
            }
    };
    std::vector<int32_t *> wantDev(size_t(2 * D), nullptr);  // per level and side: the rank's request in the neighbour's own cell offsets
    int32_t *countsDev = nullptr;                             // per level: [my lo request, my hi request, lo neighbour's request, hi neighbour's request]
    status = [&]() -> int {
        MGPS_TRY(tmp.get(h, &countsDev, size_t(4 * D)));
        std::vector<int32_t> hc(size_t(4 * D), 0);
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            LevelTmp &t = T[size_t(l)];
            const Buf &b = B[size_t(l)];
            const size_t plane = size_t(b.d.nx) * b.d.ny;
            keepBuffer(l);  // (the label buffer is the level's codes from here on)
            fillGridP(h, L, l == 0, t.listLen, t.runCells, t.nplaneBlocks, t.planeZc);
            L.g.lab = b.owned();
            L.g.ghostLo = lo ? 1 : 0;
            L.g.ghostHi = hi ? 1 : 0;
            L.gBox = L.g;
            L.gBox.rows = L.extRows;
            L.gBox.nbnd = t.nGenExt;
            if (!L.boxForm) {  // (per-pass band smoothing on this level: no boxes kept)
                (void)cacheFree(L.bandBoxes.info);
                (void)cacheFree(L.bandBoxes.list);
                (void)cacheFree(L.bandBoxes.general);
                L.bandBoxes = BandBoxesDev{};
                continue;
            }
            L.bandBoxes.depth = depth;
            uint8_t *mk[2] = {nullptr, nullptr};
            int32_t *fl = nullptr, *rk = nullptr;
            const size_t zone = size_t(G) * plane;
            int *brokenDev = flags + fBoxes + l;
            for (int q = 0; q < 2; ++q) {
                MGPS_TRY(tmp.get(h, &mk[q], zone));
                MGPS_HIP(h, hipMemsetAsync(mk[q], 0, zone, nullptr));
            }
            MGPS_LAUNCH(h, launchHaloMark(nullptr, b.d.nx, b.d.ny, b.own.nz, G, L.bandBoxes.info, L.bandBoxes.list, L.bandBoxes.ngroups, mk[0], mk[1], brokenDev));
            MGPS_TRY(tmp.get(h, &fl, zone));
            MGPS_TRY(tmp.get(h, &rk, zone + 1));
            for (int q = 0; q < 2; ++q) {
                const bool nb = q == 0 ? lo : hi;
                if (!nb) continue;  // (no neighbour: beyond the grid, EXTERIOR -- nothing is read there)
                int32_t *list = nullptr;
                MGPS_TRY(tmp.get(h, &list, zone));
                MGPS_LAUNCH(h, launchByteList(nullptr, mk[q], int(zone), fl, rk, list, t.scan));
                int n = 0;
                MGPS_HIP(h, hipMemcpy(&n, rk + zone, sizeof(int), hipMemcpyDeviceToHost));
                L.halo.nrecv[q] = n;
                MGPS_TRY(devAlloc(h, &L.halo.recvIdx[q], size_t(n), false));
                MGPS_HIP(h, hipMemcpyAsync(L.halo.recvIdx[q], list, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToDevice, nullptr));
                // the same cells as the neighbour counts them: below, its top planes (nzNb - G .. nzNb - 1); above, its planes 0 .. G - 1
                const int nzNb = q == 0 ? (splits[rank] - splits[rank - 1]) >> l : 0;
                MGPS_TRY(tmp.get(h, &wantDev[size_t(2 * l + q)], size_t(n)));
                MGPS_LAUNCH(h, launchAddInt(nullptr, wantDev[size_t(2 * l + q)], list, n, q == 0 ? int32_t(size_t(nzNb - G) * plane) : 0));
                hc[size_t(4 * l + q)] = n;
            }
            int brokenL = 0;
            MGPS_HIP(h, hipMemcpy(&brokenL, brokenDev, sizeof(int), hipMemcpyDeviceToHost));
            if (brokenL) return failH(h, MGPS_ERR_INTERNAL, "slab set-up: a box region reaches past the ghost planes on level " + std::to_string(l));
        }
        // what the neighbours ask of this rank: until a transport says otherwise, the mirror image of the rank's own request (the
        // compute-bound tools run with a transport that moves nothing)
        for (int l = 0; l < D; ++l) {
            hc[size_t(4 * l + 2)] = hc[size_t(4 * l + 1)];
            hc[size_t(4 * l + 3)] = hc[size_t(4 * l + 0)];
        }
        MGPS_HIP(h, hipMemcpy(countsDev, hc.data(), hc.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        return MGPS_OK;
    }();
    {
        int all = MGPS_OK;
        const int rc = agree(status, &all);
        if (rc != MGPS_OK) return bail(rc);
        if (all != MGPS_OK) {
            if (status == MGPS_OK) failH(h, all, "slab set-up failed on another rank");
            return bail(status != MGPS_OK ? status : all);
        }
    }
    // counts first (one int per level and side), then the lists
    std::vector<int32_t> hc(size_t(4 * D), 0);
    for (int l = 0; l < D && P > 1; ++l) {
        int32_t *c = countsDev + 4 * l;
        if (h->comm.exchange(h->comm.user, lo ? c : nullptr, sizeof(int32_t), lo ? c + 2 : nullptr, sizeof(int32_t), hi ? c + 1 : nullptr, sizeof(int32_t), hi ? c + 3 : nullptr,
                             sizeof(int32_t), nullptr) != 0)
            return bail(failH(h, MGPS_ERR_COMM, "halo list counts: exchange failed"));
    }
    status = [&]() -> int {
        MGPS_HIP(h, hipMemcpy(hc.data(), countsDev, hc.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            if (!L.boxForm) continue;
            const size_t plane = size_t(L.d.nx) * L.d.ny;
            for (int q = 0; q < 2; ++q) {
                const bool nb = q == 0 ? lo : hi;
                if (!nb) continue;
                const int n = hc[size_t(4 * l + 2 + q)];
                if (n < 0 || size_t(n) > size_t(G) * plane) return failH(h, MGPS_ERR_COMM, "halo list counts: a neighbour asked for more cells than the ghost planes hold");
                L.halo.nsend[q] = n;
                MGPS_TRY(devAlloc(h, &L.halo.sendIdx[q], size_t(n), false));
                // (the default: the mirror image of the other side's request)
                const int other = 1 - q, no = hc[size_t(4 * l + other)];
                if (no == n && wantDev[size_t(2 * l + other)])
                    MGPS_LAUNCH(h, launchAddInt(nullptr, L.halo.sendIdx[q], L.halo.recvIdx[other], n, q == 0 ? 0 : int32_t(size_t(L.d.nz - G) * plane)));
                else if (n > 0)
                    MGPS_HIP(h, hipMemsetAsync(L.halo.sendIdx[q], 0, size_t(n) * sizeof(int32_t), nullptr));
                MGPS_TRY(devAlloc(h, &L.halo.sendBuf[q], 2 * size_t(n), true));
                MGPS_TRY(devAlloc(h, &L.halo.recvBuf[q], 2 * size_t(L.halo.nrecv[q]), true));
            }
        }
        MGPS_HIP(h, hipDeviceSynchronize());
        return MGPS_OK;
    }();
    {
        int all = MGPS_OK;
        const int rc = agree(status, &all);
        if (rc != MGPS_OK) return bail(rc);
        if (all != MGPS_OK) {
            if (status == MGPS_OK) failH(h, all, "slab set-up failed on another rank");
            return bail(status != MGPS_OK ? status : all);
        }
    }
    for (int l = 0; l < D && P > 1; ++l) {
        DevLevel &L = h->lv[size_t(l)];
        if (!L.boxForm) continue;
        const DevLevel::BoxHalo &H = L.halo;
        if (h->comm.exchange(h->comm.user, lo ? wantDev[size_t(2 * l)] : nullptr, size_t(H.nrecv[0]) * sizeof(int32_t), lo ? H.sendIdx[0] : nullptr,
                             size_t(H.nsend[0]) * sizeof(int32_t), hi ? wantDev[size_t(2 * l + 1)] : nullptr, size_t(H.nrecv[1]) * sizeof(int32_t),
                             hi ? H.sendIdx[1] : nullptr, size_t(H.nsend[1]) * sizeof(int32_t), nullptr) != 0)
            return bail(failH(h, MGPS_ERR_COMM, "halo lists: exchange failed"));
    }
    status = [&]() -> int {
        // what a neighbour asked for must lie in this rank's planes
        int *bad = flags;
        MGPS_HIP(h, hipMemsetAsync(bad, 0, sizeof(int), nullptr));
        for (int l = 0; l < D; ++l) {
            DevLevel &L = h->lv[size_t(l)];
            for (int q = 0; q < 2; ++q) MGPS_LAUNCH(h, launchCheckIndex(nullptr, L.halo.sendIdx[q], L.halo.nsend[q], int32_t(L.d.cells()), bad));
        }
        int hbad = 0;
        MGPS_HIP(h, hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost));
        if (hbad) return failH(h, MGPS_ERR_COMM, "halo lists: a neighbour asked for cells outside this rank's planes");
        // the collapse level on this rank: its owned planes (labels, activity runs) and the two grids the collapse moves
        DevLevel &C = h->lv[size_t(D)];
        const Buf &b = B[size_t(D)];
        C.d = b.own;
        C.z0 = z0 >> D;
        C.z1 = z1 >> D;
        C.elo = b.elo;
        if (D == Dmax) MGPS_LAUNCH(h, launchMarkBoundary(nullptr, b.d, b.lab()));  // (levels below Dmax were marked with the others)
        {
            const size_t nfine = (b.own.cells() + kSegCells - 1) / kSegCells;
            uint8_t *cf = nullptr;
            int *rcnt = nullptr;
            MGPS_TRY(tmp.get(h, &cf, nfine));
            MGPS_TRY(tmp.get(h, &rcnt, 4));
            MGPS_HIP(h, hipMemsetAsync(rcnt, 0, 4 * sizeof(int), nullptr));
            MGPS_LAUNCH(h, launchActivityFlags(nullptr, b.own, b.owned(), cf, nullptr, 0));
            MGPS_LAUNCH(h, launchCountRuns(nullptr, cf, nfine, rcnt));
            int rc4[4] = {0, 0, 0, 0};
            MGPS_HIP(h, hipMemcpy(rc4, rcnt, sizeof(rc4), hipMemcpyDeviceToHost));
            const int64_t rc64[4] = {rc4[0], rc4[1], rc4[2], rc4[3]};
            const int runCells = chooseRunCells(rc64);
            const size_t nruns = (nfine + size_t(runCells / kSegCells) - 1) / size_t(runCells / kSegCells);
            uint8_t *runFlags = cf;
            if (runCells != kSegCells) {
                MGPS_TRY(tmp.get(h, &runFlags, nruns));
                MGPS_LAUNCH(h, launchFoldRunFlags(nullptr, cf, nfine, runCells, runFlags));
            }
            int activeRuns = 0;
            for (int z = 0; z < 4; ++z)
                if (kRunSizes[z] == runCells) activeRuns = rc4[z];
            const int perGroup = kChunkCells / runCells, listLen = (activeRuns + perGroup - 1) / perGroup * perGroup;
            int32_t *tmpFlags = nullptr, *rnk = nullptr, *base = nullptr, *scan = nullptr;
            MGPS_TRY(tmp.get(h, &tmpFlags, nruns));
            MGPS_TRY(tmp.get(h, &rnk, nruns + 1));
            MGPS_TRY(tmp.get(h, &base, 1));
            MGPS_TRY(tmp.get(h, &scan, scanScratchInts(b.d.cells())));
            MGPS_TRY(devAlloc(h, &C.chunks, size_t(listLen), false));
            MGPS_LAUNCH(h, launchRunList(nullptr, b.own, runFlags, nruns, runCells, tmpFlags, rnk, scan, base, C.chunks, listLen));
            MGPS_TRY(gridAlloc(h, &C.x, C.d));
            MGPS_TRY(gridAlloc(h, &C.b, C.d));
            keepBuffer(D);
            fillGridP(h, C, false, listLen, runCells, 0, 0);
            C.g.lab = b.owned();
            C.g.ghostLo = lo ? 1 : 0;
            C.g.ghostHi = hi ? 1 : 0;
        }
        MGPS_HIP(h, hipDeviceSynchronize());
        return MGPS_OK;
    }();
    sclock.lap("slab: halo lists + collapse level");
    if (status == MGPS_OK) status = commonDeviceState(h, false);
    {
        int all = MGPS_OK;
        const int rc = agree(status, &all);
        if (rc != MGPS_OK) return bail(rc);
        if (all != MGPS_OK) {
            if (status == MGPS_OK) failH(h, all, "slab set-up failed on another rank");
            return bail(status != MGPS_OK ? status : all);
        }
    }
    // ---- the collapsed tail on rank 0: the collapse level's labels of all ranks, then the whole-grid builder
    int tailRc = MGPS_OK;
    {
        const DevLevel &C = h->lv[size_t(D)];
        const Dims cg{nx >> D, ny >> D, nzg >> D};
        const size_t planeC = size_t(cg.nx) * cg.ny, mine = C.d.cells();
        uint8_t *all = nullptr;
        if (rank == 0 && devAlloc(h, &all, cg.cells(), false) != MGPS_OK) tailRc = MGPS_ERR_ALLOC;
        bool uniform = true;
        for (int r = 1; r < P; ++r) uniform = uniform && (splits[r + 1] - splits[r]) == (splits[1] - splits[0]);
        int crc = 0;
        if (uniform) crc = h->comm.gather(h->comm.user, C.g.lab, rank == 0 ? all : nullptr, mine, 0, nullptr);
        else {
            std::vector<size_t> counts, displs;
            for (int r = 0; r < P; ++r) {
                displs.push_back(size_t(splits[r] >> D) * planeC);
                counts.push_back(size_t((splits[r + 1] - splits[r]) >> D) * planeC);
            }
            crc = h->comm.gatherv(h->comm.user, C.g.lab, mine, rank == 0 ? all : nullptr, counts.data(), displs.data(), 0, nullptr);
        }
        if (crc != 0) {
            (void)cacheFree(all);
            return bail(failH(h, MGPS_ERR_COMM, "collapse level labels: gather failed"));
        }
        if (rank == 0 && tailRc == MGPS_OK) {
            (void)hipStreamSynchronize(nullptr);
            tailRc = [&]() -> int {
                int trc = createWholeOnDevice(&h->tail, cg.nx, cg.ny, cg.nz, all, nullptr, nullptr, nullptr, hipMemcpyDeviceToDevice, levels - D, useGS, o, device, true);
                if (trc != MGPS_OK) return failH(h, trc, std::string("collapsed tail: ") + lastGlobalError());
                trc = gridAlloc(h->tail, &h->tailX, cg);
                if (trc == MGPS_OK) trc = gridAlloc(h->tail, &h->tailB, cg);
                if (trc != MGPS_OK) return trc;
                h->tail->userGrids.push_back(h->tailX - planeC);
                h->tail->userGrids.push_back(h->tailB - planeC);
                if (hipDeviceSynchronize() != hipSuccess) return failH(h, MGPS_ERR_HIP, "device synchronize failed");
                return MGPS_OK;
            }();
        }
        (void)hipDeviceSynchronize();
        (void)cacheFree(all);
    }
    sclock.lap("slab: tail");
    {  // every rank leaves with the same verdict, and with the number of levels the tail ended with (the level cap below the collapse level)
        double v[2] = {double(tailRc), rank == 0 && h->tail ? double(h->tail->totalLevels) : 0.0};
        if (h->comm.allreduce(h->comm.user, v, 2, 1) != 0) return bail(failH(h, MGPS_ERR_COMM, "all-reduce failed during set-up"));
        if (int(v[0]) != MGPS_OK) {
            if (tailRc == MGPS_OK) failH(h, int(v[0]), "the collapsed tail could not be built on rank 0 (status " + std::to_string(int(v[0])) + ")");
            return bail(tailRc != MGPS_OK ? tailRc : int(v[0]));
        }
        if (int(v[1]) >= 1) levels = D + int(v[1]);  // (a transport that moves nothing leaves the other ranks with what they asked for)
    }
    h->totalLevels = levels;
    {
        const int rcLight = hierarchyLight(&h->hier, nx, ny, nzg, levels, nullptr, o, false);
        if (rcLight != MGPS_OK) return bail(failH(h, rcLight, lastGlobalError()));
    }
    *out = h;
    return MGPS_OK;
}

// weightsOnDevice: wx_slab / wy_slab / wz_slab are DEVICE arrays (the field passes of mgps_fields.h leave them there): nothing of
// their 12 B per cell crosses PCIe -- at 1024^3 / 8 ranks the 1.6 GB of a slab's weights coming from pageable host memory were
// 300 of the 470 ms a rank's set-up took -- and the operator rows of the slab's BOUNDARY cells are evaluated by
// launchBoundaryRows like mgps_create_device_weights does for a whole grid
int createSlabImpl(mgps_solver **out, int nx, int ny, int nz_global, const uint8_t *labels_global_host, const float *wx_slab,
                   const float *wy_slab, const float *wz_slab, int mg_levels, int use_gauss_seidel, const mgps_options *opt,
                   const mgps_comm *comm, const int *splits, bool weightsOnDevice)
{
    if (!out) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: out is NULL");
    *out = nullptr;
    if (!labels_global_host || !wx_slab || !wy_slab || !wz_slab || !comm || !splits)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: labels, the slab weights, a comm and the cuts are required");
    // (a transport built against the header before `allreduce_device` was appended is accepted: the missing tail reads as NULL)
    if (comm->struct_size < int(offsetof(mgps_comm, allreduce_device)) || comm->struct_size > int(sizeof(mgps_comm)) || !comm->exchange || !comm->allreduce ||
        !comm->gather || !comm->scatter || comm->size < 1 || comm->rank < 0 || comm->rank >= comm->size)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: incomplete mgps_comm");
    const int P = comm->size, rank = comm->rank;
    bool cutsOk = splits[0] == 0 && splits[P] == nz_global, even = true;
    for (int r = 0; r < P && cutsOk; ++r) {
        cutsOk = splits[r + 1] > splits[r];
        even = even && (splits[r + 1] - splits[r]) == (splits[1] - splits[0]);
    }
    if (!cutsOk) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: the cuts must run from 0 to nz, increasing");
    if (!even && (!comm->gatherv || !comm->scatterv))
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: slabs of different sizes need a transport with gatherv / scatterv");
    mgps_options o;
    MGPS_TRY(readOptions(opt, &o));
    if (o.precision != 0) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: options.precision = 1 is for single-device solvers");
    int device = 0;
    MGPS_TRY(pickDevice(o, &device));
    // the default since round 5: everything on the device, from the rank's window of the labels (createSlabOnDevice).  What follows is
    // the host builder (options.host_setup = 1): the checker of the device arrays, with the band stage pass by pass on cut levels
    if (!hostSetup(o))
        return createSlabOnDevice(out, nx, ny, nz_global, labels_global_host, wx_slab, wy_slab, wz_slab, weightsOnDevice, mg_levels, use_gauss_seidel != 0, o, comm, splits, device);
    StageClock sclock(setupTimingOn());
    mgps_hierarchy *hier = nullptr;
    {  // the rank's window of the hierarchy: labels of every level, band lists around its slab only
        const int window[2] = {splits[rank], splits[rank + 1]};
        MGPS_TRY(hierarchyCreate(&hier, nx, ny, nz_global, labels_global_host, mg_levels, &o, false, true, P > 1 ? window : nullptr));
    }
    sclock.lap("slab: host hierarchy");
    // distributed levels (distributedLevelsFor); the last level is always collapsed
    const int D = distributedLevelsFor(splits, P, nx, ny, hier->levels, use_gauss_seidel != 0, o);
    if (D < 1) {
        mgps_hierarchy_destroy(hier);
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT,
                     "mgps_create_slab: every rank needs at least 16 planes cut on even planes (multiples of 16 with Gauss-Seidel) and the "
                     "hierarchy at least 2 levels");
    }
    auto *h = new mgps_solver();
    h->hier = hier;
    h->opt = o;
    h->useGS = use_gauss_seidel != 0;
    h->device = device;
    h->dist = true;
    h->comm = mgps_comm{};
    std::memcpy(&h->comm, comm, size_t(comm->struct_size));  // (struct_size bytes are the caller's; the rest stays NULL)
    h->comm.struct_size = int(sizeof(mgps_comm));
    h->splits.assign(splits, splits + P + 1);
    h->distLevels = D;
    h->totalLevels = hier->levels;
    auto bail = [&](int code) {
        setLastGlobalError(h->lastError);
        freeAll(h);
        return code;
    };
    const int z0 = splits[rank], nzl = splits[rank + 1] - z0;
    const size_t wn[3] = {size_t(nx + 1) * ny * nzl, size_t(nx) * (ny + 1) * nzl, size_t(nx) * ny * (nzl + 1)};
    const float *wh[3] = {wx_slab, wy_slab, wz_slab};
    if (weightsOnDevice && o.borrow_device_weights) {
        for (int a = 0; a < 3; ++a) h->w[a] = const_cast<float *>(wh[a]);
        h->weightsBorrowed = true;
    } else
        for (int a = 0; a < 3; ++a) {
            int rc = devAlloc(h, &h->w[a], wn[a], false);
            if (rc != MGPS_OK) return bail(rc);
            if (hipMemcpy(h->w[a], wh[a], wn[a] * sizeof(float), weightsOnDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) != hipSuccess)
                return bail(failH(h, MGPS_ERR_HIP, "weight upload failed"));
        }
    (void)hipDeviceSynchronize();
    sclock.lap("slab: weights");
    // device weights: the rows of the slab's BOUNDARY cells (band order) from a kernel over the slab's labels + ghost planes
    std::vector<float> rows0;
    int rowsRc = MGPS_OK;
    if (weightsOnDevice) {
        const HostLevel &G = hier->lv[0];
        const size_t plane = size_t(nx) * ny;
        const int gLo = z0 > 0 ? 1 : 0, gHi = z0 + nzl < nz_global ? 1 : 0;
        std::vector<int32_t> cells;  // slab-local linear indices, in band order
        const int64_t lo = int64_t(size_t(z0) * plane), hi = int64_t(size_t(z0 + nzl) * plane);
        for (int32_t c : G.band)
            if (c >= lo && c < hi && G.labels[size_t(c)] == MGPS_BOUNDARY_CELL) cells.push_back(int32_t(c - lo));
        rows0.resize(8 * cells.size());
        uint8_t *labDev = nullptr;
        int32_t *cellsDev = nullptr;
        float *rowsDev = nullptr;
        int *violDev = nullptr;
        int violations = 0;
        hipError_t e = cacheMalloc(reinterpret_cast<void **>(&labDev), size_t(nzl + 2) * plane);
        if (e == hipSuccess) e = cacheMalloc(reinterpret_cast<void **>(&cellsDev), std::max<size_t>(1, cells.size()) * sizeof(int32_t));
        if (e == hipSuccess) e = cacheMalloc(reinterpret_cast<void **>(&rowsDev), std::max<size_t>(1, rows0.size()) * sizeof(float));
        if (e == hipSuccess) e = cacheMalloc(reinterpret_cast<void **>(&violDev), sizeof(int));
        if (e == hipSuccess) e = hipMemset(labDev, MGPS_EXTERIOR_CELL, size_t(nzl + 2) * plane);  // (the planes outside the grid)
        if (e == hipSuccess)
            e = hipMemcpy(labDev + size_t(1 - gLo) * plane, G.labels.data() + size_t(z0 - gLo) * plane, size_t(nzl + gLo + gHi) * plane, hipMemcpyHostToDevice);
        if (e == hipSuccess && !cells.empty()) e = hipMemcpy(cellsDev, cells.data(), cells.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(violDev, 0, sizeof(int));
        Dims sd;
        sd.nx = nx, sd.ny = ny, sd.nz = nzl;
        if (e == hipSuccess) e = hipError_t(launchBoundaryRows(nullptr, sd, labDev + plane, wx_slab, wy_slab, wz_slab, cellsDev, int(cells.size()), rowsDev, violDev));
        if (e == hipSuccess && !rows0.empty()) e = hipMemcpy(rows0.data(), rowsDev, rows0.size() * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&violations, violDev, sizeof(int), hipMemcpyDeviceToHost);
        (void)cacheFree(labDev);
        (void)cacheFree(cellsDev);
        (void)cacheFree(rowsDev);
        (void)cacheFree(violDev);
        if (e != hipSuccess) rowsRc = failH(h, MGPS_ERR_HIP, std::string("mgps_create_slab (device weights): ") + hipGetErrorString(e));
        else if (violations != 0)
            rowsRc = failH(h, MGPS_ERR_HIERARCHY, "labels/weights violate the BOUNDARY-cell rules (unitTestBoundaryCells): run mgps_fields_set_boundary_labels");
    }
    {  // every rank leaves together: one with bad labels or weights must not leave the others waiting in the collectives below
        double v = double(rowsRc);
        if (h->comm.allreduce(h->comm.user, &v, 1, 1) != 0) return bail(failH(h, MGPS_ERR_COMM, "all-reduce failed during set-up"));
        if (int(v) != MGPS_OK) {
            if (rowsRc == MGPS_OK) failH(h, int(v), "slab set-up failed on another rank");
            return bail(rowsRc != MGPS_OK ? rowsRc : int(v));
        }
    }
    sclock.lap("slab: fine rows");
    h->lv.resize(D + 1);
    for (int l = 0; l <= D; ++l) {
        HostLevel HL;
        const int gz = hier->lv[l].d.nz, lz0 = z0 >> l, lz1 = (z0 + nzl) >> l;
        const bool hostW = l == 0 && !weightsOnDevice;
        buildSlabLevel(hier->lv[l], lz0, lz1, hostW ? wx_slab : nullptr, hostW ? wy_slab : nullptr, hostW ? wz_slab : nullptr, HL,
                       (l == 0 && weightsOnDevice) ? (rows0.empty() ? &kNoRows : rows0.data()) : nullptr);
        sclock.lap("slab: host lists", l);
        int rc = uploadLevel(h, h->lv[l], HL, lz0, lz1, gz, l == 0, l < D, l > 0);
        if (rc != MGPS_OK) return bail(rc);
        (void)hipDeviceSynchronize();
        sclock.lap("slab: upload level", l);
    }
    int rc = commonDeviceState(h, false);
    if (rc != MGPS_OK) return bail(rc);
    int tailRc = MGPS_OK;
    if (rank == 0) {  // the collapsed tail: levels D .. L-1 on the whole grid, unit weights
        tailRc = [&]() -> int {
            const HostLevel &C = hier->lv[D];
            mgps_hierarchy *tailHier = nullptr;
            int trc = hierarchyCreate(&tailHier, C.d.nx, C.d.ny, C.d.nz, C.labels.data(), hier->levels - D, &o, true, false);
            if (trc != MGPS_OK) return failH(h, trc, std::string("collapsed tail hierarchy: ") + lastGlobalError());
            trc = createWhole(&h->tail, tailHier, nullptr, nullptr, nullptr, h->useGS, o, device, true);
            if (trc != MGPS_OK) return failH(h, trc, std::string("collapsed tail: ") + lastGlobalError());
            trc = gridAlloc(h->tail, &h->tailX, C.d);
            if (trc == MGPS_OK) trc = gridAlloc(h->tail, &h->tailB, C.d);
            if (trc != MGPS_OK) return trc;
            h->tail->userGrids.push_back(h->tailX - size_t(C.d.nx) * C.d.ny);
            h->tail->userGrids.push_back(h->tailB - size_t(C.d.nx) * C.d.ny);
            if (hipDeviceSynchronize() != hipSuccess) return failH(h, MGPS_ERR_HIP, "device synchronize failed");
            return MGPS_OK;
        }();
    }
    sclock.lap("slab: common state + tail");
    // every rank leaves with the same verdict: rank 0 failing alone would leave the others waiting in the first gather
    double failed = tailRc != MGPS_OK ? double(tailRc) : 0.0;
    if (h->comm.allreduce(h->comm.user, &failed, 1, 1) != 0) return bail(failH(h, MGPS_ERR_COMM, "all-reduce failed during set-up"));
    if (failed != 0.0) {
        if (tailRc == MGPS_OK) failH(h, int(failed), "the collapsed tail could not be built on rank 0 (status " + std::to_string(int(failed)) + ")");
        return bail(tailRc != MGPS_OK ? tailRc : int(failed));
    }
    *out = h;
    return MGPS_OK;
}
}  // namespace
extern "C" {
int mgps_create_slab_ranges(mgps_solver **out, int nx, int ny, int nz_global, const uint8_t *labels_global_host, const float *wx_slab,
                            const float *wy_slab, const float *wz_slab, int mg_levels, int use_gauss_seidel, const mgps_options *opt,
                            const mgps_comm *comm, const int *splits)
try {
    return createSlabImpl(out, nx, ny, nz_global, labels_global_host, wx_slab, wy_slab, wz_slab, mg_levels, use_gauss_seidel, opt, comm, splits, false);
}
MGPS_API_CATCH(nullptr)
int mgps_create_slab_device_weights(mgps_solver **out, int nx, int ny, int nz_global, const uint8_t *labels_global_host, const float *wx_slab_dev,
                                    const float *wy_slab_dev, const float *wz_slab_dev, int mg_levels, int use_gauss_seidel, const mgps_options *opt,
                                    const mgps_comm *comm, const int *splits)
try {
    return createSlabImpl(out, nx, ny, nz_global, labels_global_host, wx_slab_dev, wy_slab_dev, wz_slab_dev, mg_levels, use_gauss_seidel, opt, comm, splits, true);
}
MGPS_API_CATCH(nullptr)

int mgps_create_slab(mgps_solver **out, int nx, int ny, int nz_global, const uint8_t *labels_global_host, const float *wx_slab,
                     const float *wy_slab, const float *wz_slab, int mg_levels, int use_gauss_seidel, const mgps_options *opt,
                     const mgps_comm *comm)
try {
    if (!comm || comm->size < 1) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: a comm is required");
    if (nz_global % comm->size != 0)
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_create_slab: nz must divide evenly over the ranks (or pass cuts to mgps_create_slab_ranges)");
    std::vector<int> cuts(size_t(comm->size) + 1);
    for (int r = 0; r <= comm->size; ++r) cuts[size_t(r)] = nz_global / comm->size * r;
    return mgps_create_slab_ranges(out, nx, ny, nz_global, labels_global_host, wx_slab, wy_slab, wz_slab, mg_levels, use_gauss_seidel, opt, comm,
                                   cuts.data());
}
MGPS_API_CATCH(nullptr)

void mgps_destroy(mgps_solver *h) { freeAll(h); }
int mgps_levels(const mgps_solver *h) { return h ? h->totalLevels : 0; }
// A solver set up on the device holds only the extents of its levels on the host; the first caller who asks for the
// hierarchy gets the host builder's, made from the fine labels (the device codes with the simple cells folded back).
const mgps_hierarchy *mgps_get_hierarchy(const mgps_solver *hc)
try {
    auto *h = const_cast<mgps_solver *>(hc);
    if (h && h->hier && h->hier->windowed) {  // a slab rank's window: complete it from the global labels it holds
        mgps_hierarchy *full = nullptr;
        const Dims d0 = h->hier->lv[0].d;
        if (hierarchyCreate(&full, d0.nx, d0.ny, d0.nz, h->hier->lv[0].labels.data(), h->totalLevels, &h->opt, false, true) != MGPS_OK) {
            failH(h, MGPS_ERR_HIERARCHY, std::string("mgps_get_hierarchy: ") + lastGlobalError());
            return nullptr;
        }
        // (the device levels keep views into the old labels: the old hierarchy stays alive with the solver)
        h->retiredHier.push_back(h->hier);
        h->hier = full;
        return full;
    }
    if (!h || !h->hier || !h->hier->light) return h ? h->hier : nullptr;
    if (h->dist) {  // (a slab rank set up on the device holds its window of the labels only)
        failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_get_hierarchy: a slab rank set up on the device has no host hierarchy (options.host_setup = 1 builds one)");
        return nullptr;
    }
    (void)hipSetDevice(h->device);
    const Dims d = h->lv[0].d;
    RawVec<uint8_t> labels(d.cells());
    if (hipMemcpy(labels.data(), h->lv[0].g.lab, d.cells(), hipMemcpyDeviceToHost) != hipSuccess) {
        failH(h, MGPS_ERR_HIP, "mgps_get_hierarchy: copying the labels to the host failed");
        return nullptr;
    }
    for (uint8_t &l : labels)
        if (l > MGPS_BOUNDARY_CELL) l = MGPS_BOUNDARY_CELL;
    mgps_hierarchy *full = nullptr;
    if (hierarchyCreate(&full, d.nx, d.ny, d.nz, labels.data(), h->requestedLevels, &h->opt, false, true) != MGPS_OK) {
        failH(h, MGPS_ERR_HIERARCHY, std::string("mgps_get_hierarchy: ") + lastGlobalError());
        return nullptr;
    }
    mgps_hierarchy_destroy(h->hier);
    h->hier = full;
    return full;
} catch (...) {
    return nullptr;
}

int mgps_level_array(mgps_solver *h, int level, int which, void *out, int64_t *count)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_level_array"));
    if (!count) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_level_array: count is NULL");
    const DevLevel &L = h->lv[level];
    const int nt = ((L.d.nx + kTile - 1) / kTile) * ((L.d.ny + kTile - 1) / kTile) * ((L.d.nz + kTile - 1) / kTile);
    const void *src = nullptr;
    size_t n = 0, elem = 4;
    switch (which) {
    case 0: src = L.g.lab, n = L.d.cells(), elem = 1; break;
    case 1: src = L.band, n = size_t(L.nband); break;
    case 2: src = L.bandDiag, n = size_t(L.nband), elem = 1; break;
    case 3: src = L.rows, n = size_t(7) * size_t(L.nbndGeneral); break;
    case 4: src = L.chunks, n = size_t(L.g.nchunks); break;
    case 5: src = L.planeBlocks, n = size_t(L.g.nplaneBlocks); break;
    case 6: src = L.pure[0], n = size_t(L.npure[0]); break;
    case 7: src = L.pure[1], n = size_t(L.npure[1]); break;
    case 8: src = L.mixed[0], n = size_t(L.nmixed[0]); break;
    case 9: src = L.mixed[1], n = size_t(L.nmixed[1]); break;
    case 10: src = L.tileBndStart, n = size_t(nt) + 1; break;
    case 11: src = L.bandBoxes.info, n = size_t(kBoxInfoInts) * size_t(L.bandBoxes.ngroups); break;
    case 12: src = L.bandBoxes.list, n = L.bandBoxes.listCount; break;
    case 13: src = L.bandBoxes.general, n = L.bandBoxes.generalInts; break;
    default: return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_level_array: unknown array");
    }
    *count = int64_t(n);
    if (out && n) {
        MGPS_HIP(h, hipStreamSynchronize(h->stream));
        MGPS_HIP(h, hipMemcpy(out, src, n * elem, hipMemcpyDeviceToHost));
    }
    return MGPS_OK;
}
MGPS_API_CATCH(h)
int mgps_distributed_levels(const mgps_solver *h) { return h ? h->distLevels : 0; }
int mgps_ghost_planes(const mgps_solver *h) { return h ? h->ghost : 1; }
int mgps_residual_restrict_fused(const mgps_solver *h, int level, int *fused)
try {
    if (!h || !fused || level < 0 || level >= int(h->lv.size())) return MGPS_ERR_INVALID_ARGUMENT;
    *fused = (h->opt.precision == 0 || level > 0) && residualRestrictFuses(h, level) ? 1 : 0;
    return MGPS_OK;
}
MGPS_API_CATCH(h)
int mgps_band_stage_form(const mgps_solver *h, int level, int *form)
try {
    if (!h || !form || level < 0 || level >= int(h->lv.size())) return MGPS_ERR_INVALID_ARGUMENT;
    *form = levelHasBoxes(h, level) ? 1 : 0;
    return MGPS_OK;
}
MGPS_API_CATCH(h)
int64_t mgps_overlapped_exchanges(const mgps_solver *) { return 0; }  // (round 5: every exchange runs on the solver's stream until a first run on real links)
int64_t mgps_exchange_count(const mgps_solver *h) { return h ? h->exchanges : 0; }

int mgps_level_dims(const mgps_solver *h, int level, int out_dims[3])
try {
    if (!h || !out_dims || level < 0) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_level_dims: bad arguments");
    if (level < int(h->lv.size())) {  // the grids this object works on (the slab in a slab run)
        out_dims[0] = h->lv[level].d.nx;
        out_dims[1] = h->lv[level].d.ny;
        out_dims[2] = h->lv[level].d.nz;
        return MGPS_OK;
    }
    return mgps_hierarchy_level_dims(h->hier, level, out_dims);
}
MGPS_API_CATCH(h)

int mgps_slab_range(const mgps_solver *h, int level, int *z0, int *z1)
try {
    if (!h || !z0 || !z1 || level < 0 || level >= int(h->lv.size()))
        return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_slab_range: bad arguments");
    *z0 = h->lv[level].z0;
    *z1 = h->lv[level].z1;
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_set_stream(mgps_solver *h, void *hip_stream)
try {
    if (!h) return failH(nullptr, MGPS_ERR_INVALID_ARGUMENT, "mgps_set_stream: NULL handle");
    h->stream = static_cast<hipStream_t>(hip_stream);
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_synchronize(mgps_solver *h)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_synchronize"));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_grid_alloc(mgps_solver *h, int level, float **out_dev)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_grid_alloc"));
    if (!out_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_grid_alloc: out is NULL");
    MGPS_TRY(gridAlloc(h, out_dev, h->lv[level].d));
    h->userGrids.push_back(*out_dev - size_t(h->ghost) * size_t(h->lv[level].d.nx) * h->lv[level].d.ny);
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_grid_free(mgps_solver *h, float *dev)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_grid_free"));
    for (auto it = h->userGrids.begin(); it != h->userGrids.end(); ++it)
        for (const DevLevel &L : h->lv)
            if (static_cast<float *>(*it) + size_t(h->ghost) * size_t(L.d.nx) * L.d.ny == dev) {
                void *base = *it;
                h->userGrids.erase(it);
                MGPS_HIP(h, hipStreamSynchronize(h->stream));
                MGPS_HIP(h, cacheFree(base));
                return MGPS_OK;
            }
    return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_grid_free: not a grid of this solver");
}
MGPS_API_CATCH(h)

int mgps_grid_upload(mgps_solver *h, int level, float *dst_dev, const float *src_host)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_grid_upload"));
    if (!dst_dev || !src_host) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_grid_upload: NULL pointer");
    MGPS_HIP(h, hipMemcpyAsync(dst_dev, src_host, h->lv[level].d.cells() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_grid_download(mgps_solver *h, int level, float *dst_host, const float *src_dev)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_grid_download"));
    if (!dst_host || !src_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_grid_download: NULL pointer");
    MGPS_HIP(h, hipMemcpyAsync(dst_host, src_dev, h->lv[level].d.cells() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_copy_to_host(mgps_solver *h, void *dst_host, const void *src_dev, size_t bytes)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_copy_to_host"));
    MGPS_HIP(h, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_copy_to_device(mgps_solver *h, void *dst_dev, const void *src_host, size_t bytes)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_copy_to_device"));
    MGPS_HIP(h, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, h->stream));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_apply_vcycle(mgps_solver *h, float *x_dev, const float *b_dev, int use_initial_guess)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_apply_vcycle"));
    if (!x_dev || !b_dev || x_dev == b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_apply_vcycle: bad grid pointers");
    if (h->opt.precision == 1) return vcycleMixed(h, x_dev, b_dev, use_initial_guess != 0);
    return vcycle(h, x_dev, b_dev, use_initial_guess != 0);
}
MGPS_API_CATCH(h)

int mgps_jacobi_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_jacobi_smooth"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_jacobi_smooth: NULL grid");
    DevLevel &L = h->lv[level];
    if (!L.tmp) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_jacobi_smooth: level has no work grids");
    MGPS_TRY(applyOp(h, OP_JACOBI, level, L.tmp, x_dev, b_dev));
    MGPS_HIP(h, hipMemcpyAsync(x_dev, L.tmp, L.d.cells() * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_tiled_gs_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev, int smooth_odd_tiles,
                         int smooth_forward)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_tiled_gs_smooth"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_tiled_gs_smooth: NULL grid");
    return gsHalfSweep(h, level, x_dev, b_dev, smooth_odd_tiles ? 1 : 0, smooth_forward != 0);
}
MGPS_API_CATCH(h)

int mgps_boundary_jacobi_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_boundary_jacobi_smooth"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_boundary_jacobi_smooth: NULL grid");
    DevLevel &L = h->lv[level];
    MGPS_TRY(exchangeGhosts(h, level, x_dev));
    MGPS_LAUNCH(h, launchBandJacobi(h->stream, L.g, x_dev, b_dev, L.band, L.nband, L.bandTmp, h->opt.jacobi_weight));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_boundary_jacobi_stage(mgps_solver *h, int level, float *x_dev, const float *b_dev)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_boundary_jacobi_stage"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_boundary_jacobi_stage: NULL grid");
    return bandPasses(h, level, x_dev, b_dev, GHOST_FULL);
}
MGPS_API_CATCH(h)

int mgps_apply_poisson(mgps_solver *h, int level, float *y_dev, const float *x_dev)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_apply_poisson"));
    if (!y_dev || !x_dev || y_dev == x_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_apply_poisson: bad grid pointers");
    return applyOp(h, OP_APPLY, level, y_dev, const_cast<float *>(x_dev), nullptr);
}
MGPS_API_CATCH(h)

int mgps_residual(mgps_solver *h, int level, float *r_dev, const float *x_dev, const float *b_dev)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_residual"));
    if (!r_dev || !x_dev || !b_dev || r_dev == x_dev)
        return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_residual: bad grid pointers");
    return applyOp(h, OP_RESIDUAL, level, r_dev, const_cast<float *>(x_dev), b_dev);
}
MGPS_API_CATCH(h)

int mgps_downsample(mgps_solver *h, int fine_level, float *coarse_dev, const float *fine_dev)
try {
    MGPS_TRY(checkLevel(h, fine_level + 1, "mgps_downsample"));
    if (fine_level < 0 || !coarse_dev || !fine_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_downsample: bad arguments");
    MGPS_TRY(exchangeGhosts(h, fine_level, const_cast<float *>(fine_dev)));
    // "destination cleared first" (Ops.h:756): the kernel itself only visits chunks with active cells
    MGPS_LAUNCH(h, launchZero(h->stream, coarse_dev, h->lv[fine_level + 1].d.cells()));
    MGPS_LAUNCH(h, launchRestrict(h->stream, h->lv[fine_level + 1].g, coarse_dev, fine_dev));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_residual_downsample(mgps_solver *h, int fine_level, float *coarse_dev, const float *x_dev, const float *b_dev)
try {
    MGPS_TRY(checkLevel(h, fine_level + 1, "mgps_residual_downsample"));
    if (fine_level < 0 || !coarse_dev || !x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_residual_downsample: bad arguments");
    if (h->opt.precision == 1 && fine_level == 0) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_residual_downsample: fp32 levels only");
    DevLevel &F = h->lv[fine_level];
    MGPS_LAUNCH(h, launchZero(h->stream, coarse_dev, h->lv[fine_level + 1].d.cells()));  // "destination cleared first" (Ops.h:756)
    if (residualRestrictFuses(h, fine_level)) return residualRestrict(h, fine_level, x_dev, b_dev, coarse_dev);
    if (!F.r) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_residual_downsample: level has no work grids");
    MGPS_TRY(applyOp(h, OP_RESIDUAL, fine_level, F.r, const_cast<float *>(x_dev), b_dev, true));
    MGPS_TRY(exchangeGhosts(h, fine_level, F.r));
    MGPS_LAUNCH(h, launchRestrict(h->stream, h->lv[fine_level + 1].g, coarse_dev, F.r));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_upsample_add(mgps_solver *h, int fine_level, float *fine_dev, const float *coarse_dev)
try {
    MGPS_TRY(checkLevel(h, fine_level + 1, "mgps_upsample_add"));
    if (fine_level < 0 || !coarse_dev || !fine_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_upsample_add: bad arguments");
    MGPS_TRY(exchangeGhosts(h, fine_level + 1, const_cast<float *>(coarse_dev)));
    MGPS_LAUNCH(h, launchProlongAdd(h->stream, h->lv[fine_level].g, fine_dev, coarse_dev));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_coarse_solve(mgps_solver *h, float *x_dev, const float *b_dev)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_coarse_solve"));
    if (!x_dev || !b_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_coarse_solve: NULL grid");
    if (h->dist) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_coarse_solve: not available on a slab solver");
    MGPS_LAUNCH(h, launchCoarseSolve(h->stream, h->cn, h->cinv, h->ccells, x_dev, b_dev, h->cvec));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_dot(mgps_solver *h, int level, const float *a_dev, const float *b_dev, double *out)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_dot"));
    if (!a_dev || !b_dev || !out) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_dot: NULL pointer");
    return reduceToHost(h, 0, level, a_dev, b_dev, out);
}
MGPS_API_CATCH(h)

int mgps_squared_l2_norm(mgps_solver *h, int level, const float *a_dev, double *out)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_squared_l2_norm"));
    if (!a_dev || !out) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_squared_l2_norm: NULL pointer");
    return reduceToHost(h, 1, level, a_dev, nullptr, out);
}
MGPS_API_CATCH(h)

int mgps_l2_norm(mgps_solver *h, int level, const float *a_dev, double *out)
try {
    MGPS_TRY(mgps_squared_l2_norm(h, level, a_dev, out));
    *out = std::sqrt(*out);  // Ops.h:1202
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_inf_norm(mgps_solver *h, int level, const float *a_dev, int reference_signed_max, double *out)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_inf_norm"));
    if (!a_dev || !out) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_inf_norm: NULL pointer");
    return reduceToHost(h, reference_signed_max ? 2 : 3, level, a_dev, nullptr, out);
}
MGPS_API_CATCH(h)

int mgps_add_to_vector(mgps_solver *h, int level, float *dst_dev, const float *src_dev, double scale)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_add_to_vector"));
    if (!dst_dev || !src_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_add_to_vector: NULL grid");
    MGPS_LAUNCH(h, launchAxpy(h->stream, h->lv[level].g, dst_dev, src_dev, nullptr, float(scale), 1.f));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_add_vectors(mgps_solver *h, int level, float *dst_dev, const float *a_dev, const float *scaled_dev, double scale)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_add_vectors"));
    if (!dst_dev || !a_dev || !scaled_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_add_vectors: NULL grid");
    MGPS_LAUNCH(h, launchXpay(h->stream, h->lv[level].g, dst_dev, a_dev, scaled_dev, nullptr, float(scale)));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_scale_vector(mgps_solver *h, int level, float *v_dev, double scale)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_scale_vector"));
    if (!v_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_scale_vector: NULL grid");
    MGPS_LAUNCH(h, launchScale(h->stream, h->lv[level].g, v_dev, float(scale)));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_zero_inactive(mgps_solver *h, int level, float *grid_dev)
try {
    MGPS_TRY(checkLevel(h, level, "mgps_zero_inactive"));
    if (!grid_dev) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_zero_inactive: NULL grid");
    MGPS_LAUNCH(h, launchZeroInactive(h->stream, h->lv[level].g, grid_dev));
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_solve_pcg(mgps_solver *h, float *x_dev, const float *b_dev, double tolerance, int max_iterations,
                   int use_mg_preconditioner, mgps_pcg_stats *stats)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_solve_pcg"));
    if (!x_dev || !b_dev || x_dev == b_dev || !(tolerance >= 0) || max_iterations < 0)
        return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_solve_pcg: bad arguments");
    return pcg(h, x_dev, const_cast<float *>(b_dev), tolerance, max_iterations, use_mg_preconditioner != 0, stats);
}
MGPS_API_CATCH(h)

int mgps_profile_enable(mgps_solver *h, int enable)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_profile_enable"));
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    h->profiling = enable != 0;
    h->stageProfiling = enable >= 2;
    h->profUsed = 0;
    h->profSweeps = 0;
    h->stageMarks.clear();
    for (double &ms : h->stageMs) ms = 0;
    h->stageCycles = 0;
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_profile_read(mgps_solver *h, double *fine_smoother_ms, int *fine_smoother_launches)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_profile_read"));
    if (!fine_smoother_ms || !fine_smoother_launches)
        return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_profile_read: NULL pointer");
    MGPS_HIP(h, hipStreamSynchronize(h->stream));
    double total = 0;
    for (size_t q = 0; q + 1 < h->profUsed; q += 2) {
        float ms = 0.f;
        MGPS_HIP(h, hipEventElapsedTime(&ms, h->profEvents[q], h->profEvents[q + 1]));
        total += ms;
    }
    *fine_smoother_ms = total;
    *fine_smoother_launches = h->profSweeps;
    h->profUsed = 0;
    h->profSweeps = 0;
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_swept_cells(const mgps_solver *h, int level, long long *stencil_cells, long long *gs_cells)
try {
    if (!h || level < 0 || level >= int(h->lv.size()) || !stencil_cells || !gs_cells) return MGPS_ERR_INVALID_ARGUMENT;
    const DevLevel &L = h->lv[level];
    *stencil_cells = (long long)stencilSweptCells(L.g);
    *gs_cells = (long long)(L.npure[0] + L.npure[1] + L.nmixed[0] + L.nmixed[1]) * 4096;
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_stage_times(mgps_solver *h, double out_ms[6], int *cycles)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_stage_times"));
    if (!out_ms) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_stage_times: NULL pointer");
    stageFlush(h);
    if (h->tail) {  // (rank 0 of a slab run: the collapsed tail is inside the coarse-solve stage)
        h->tail->stageMarks.clear();
    }
    for (int q = 0; q < 6; ++q) {
        out_ms[q] = h->stageMs[q];
        h->stageMs[q] = 0;
        h->stageMsFineLast[q] = h->stageMsFine[q];
        h->stageMsFine[q] = 0;
    }
    if (cycles) *cycles = h->stageCycles;
    h->stageCycles = 0;
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_stage_times_fine(mgps_solver *h, double out_ms[6])
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_stage_times_fine"));
    if (!out_ms) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "mgps_stage_times_fine: NULL pointer");
    for (int q = 0; q < 6; ++q) out_ms[q] = h->stageMsFineLast[q];
    return MGPS_OK;
}
MGPS_API_CATCH(h)

int mgps_stencil_kernel(const mgps_solver *h, int level, int *kernel)
try {
    if (!h || level < 0 || level >= int(h->lv.size()) || !kernel) return MGPS_ERR_INVALID_ARGUMENT;
    *kernel = stencilKernelOf(h->lv[level].g);
    return MGPS_OK;
}
MGPS_API_CATCH(h)

static int withHostGrids(mgps_solver *h, float *x_host, const float *b_host, bool uploadX,
                         int (*body)(mgps_solver *, float *, const float *, void *), void *ctx)
{
    if (!x_host || !b_host) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "host form: NULL pointer");
    float *xd = nullptr, *bd = nullptr;
    MGPS_TRY(mgps_grid_alloc(h, 0, &xd));
    int rc = mgps_grid_alloc(h, 0, &bd);
    if (rc == MGPS_OK) rc = mgps_grid_upload(h, 0, bd, b_host);
    if (rc == MGPS_OK && uploadX) rc = mgps_grid_upload(h, 0, xd, x_host);
    if (rc == MGPS_OK) rc = body(h, xd, bd, ctx);
    if (rc == MGPS_OK) rc = mgps_grid_download(h, 0, x_host, xd);
    const std::string keep = h->lastError;
    if (bd) mgps_grid_free(h, bd);
    mgps_grid_free(h, xd);
    if (rc != MGPS_OK) h->lastError = keep;
    return rc;
}

int mgps_apply_vcycle_host(mgps_solver *h, float *x_host, const float *b_host, int use_initial_guess)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_apply_vcycle_host"));
    int guess = use_initial_guess;
    return withHostGrids(
        h, x_host, b_host, use_initial_guess != 0,
        [](mgps_solver *hh, float *xd, const float *bd, void *c) { return mgps_apply_vcycle(hh, xd, bd, *static_cast<int *>(c)); },
        &guess);
}
MGPS_API_CATCH(h)

struct PcgHostCtx {
    double tol;
    int maxIt, useMG;
    mgps_pcg_stats *stats;
};

int mgps_solve_pcg_host(mgps_solver *h, float *x_host, const float *b_host, double tolerance, int max_iterations,
                        int use_mg_preconditioner, mgps_pcg_stats *stats)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_solve_pcg_host"));
    PcgHostCtx ctx{tolerance, max_iterations, use_mg_preconditioner, stats};
    return withHostGrids(
        h, x_host, b_host, true,
        [](mgps_solver *hh, float *xd, const float *bd, void *c) {
            auto *p = static_cast<PcgHostCtx *>(c);
            return mgps_solve_pcg(hh, xd, bd, p->tol, p->maxIt, p->useMG, p->stats);
        },
        &ctx);
}
MGPS_API_CATCH(h)

}  // extern "C"

namespace {
// double host grids: upload, narrow on the device, run `body` on fp32 device grids, widen, download
template <class Body>
int withHostGrids64(mgps_solver *h, double *x_host, const double *b_host, bool uploadX, Body body)
{
    if (!x_host || !b_host) return failH(h, MGPS_ERR_INVALID_ARGUMENT, "host form: NULL pointer");
    const size_t n = h->lv[0].d.cells();
    float *xd = nullptr, *bd = nullptr;
    double *stage = nullptr;
    MGPS_TRY(mgps_grid_alloc(h, 0, &xd));
    int rc = mgps_grid_alloc(h, 0, &bd);
    if (rc == MGPS_OK) rc = devAlloc(h, &stage, n, false);
    auto up = [&](float *dst, const double *src) -> int {
        MGPS_HIP(h, hipMemcpyAsync(stage, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        MGPS_LAUNCH(h, launchNarrow(h->stream, dst, stage, n));
        MGPS_HIP(h, hipStreamSynchronize(h->stream));
        return MGPS_OK;
    };
    if (rc == MGPS_OK) rc = up(bd, b_host);
    if (rc == MGPS_OK && uploadX) rc = up(xd, x_host);
    if (rc == MGPS_OK) rc = body(xd, bd);
    if (rc == MGPS_OK) {
        rc = [&]() -> int {
            MGPS_LAUNCH(h, launchWiden(h->stream, stage, xd, n));
            MGPS_HIP(h, hipMemcpyAsync(x_host, stage, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            MGPS_HIP(h, hipStreamSynchronize(h->stream));
            return MGPS_OK;
        }();
    }
    const std::string keep = h->lastError;
    (void)cacheFree(stage);
    if (bd) mgps_grid_free(h, bd);
    mgps_grid_free(h, xd);
    if (rc != MGPS_OK) h->lastError = keep;
    return rc;
}
}  // namespace

extern "C" {

int mgps_apply_vcycle_host_f64(mgps_solver *h, double *x_host, const double *b_host, int use_initial_guess)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_apply_vcycle_host_f64"));
    return withHostGrids64(h, x_host, b_host, use_initial_guess != 0,
                           [&](float *xd, const float *bd) { return mgps_apply_vcycle(h, xd, bd, use_initial_guess); });
}
MGPS_API_CATCH(h)

int mgps_solve_pcg_host_f64(mgps_solver *h, double *x_host, const double *b_host, double tolerance, int max_iterations,
                            int use_mg_preconditioner, mgps_pcg_stats *stats)
try {
    MGPS_TRY(checkLevel(h, 0, "mgps_solve_pcg_host_f64"));
    return withHostGrids64(h, x_host, b_host, true, [&](float *xd, const float *bd) {
        return mgps_solve_pcg(h, xd, bd, tolerance, max_iterations, use_mg_preconditioner, stats);
    });
}
MGPS_API_CATCH(h)

}  // extern "C"
