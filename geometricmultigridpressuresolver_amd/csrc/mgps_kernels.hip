// HIP kernels of the multigrid hot path, written for gfx950 (MI355X, CDNA4): 64-wide wavefronts,
// 16-byte vector loads along x (the contiguous axis), HBM-bound streaming structure, no MFMA (there
// is no dense contraction in a 7-point stencil).  Reference semantics for every kernel are cited as
// "Ops.h" = Source/HDK_GeometricMultigridOperators.h, "MG.cpp" =
// Source/HDK_GeometricMultigridPoissonSolver.cpp.
//
// Shared conventions
//  * grids are flat, x fastest; a cell is active iff its label is INTERIOR (0) or BOUNDARY (3);
//  * INTERIOR cells have six active neighbours, unit face weights and diagonal 6
//    (Ops.h:191-207) -- the fast path needs no label or weight look-ups for the neighbours;
//  * BOUNDARY cells take the general path of computeLaplacian (Ops.h:208-256), evaluated once at
//    set-up.  "Simple" ones (all relevant face weights exactly 1: every BOUNDARY cell of a coarse
//    level, and of a fine level away from cut cells / the free surface) carry their diagonal in the
//    cell code and are swept in line; the rows of the "general" ones (six off-diagonal weights +
//    diagonal) live in a compact SoA list per level and a short list kernel patches those cells
//    after each full-domain sweep.  The sweeps themselves stay free of per-lane weight look-ups (on
//    a box-shaped liquid every x-row holds two BOUNDARY cells: a look-up path would diverge in every
//    single wavefront);
//  * arithmetic is fp32; reductions accumulate in fp64.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "mgps_internal.h"

namespace mgps {

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned char v4b __attribute__((ext_vector_type(4)));
// streamed inputs (rhs, cell codes): read once per sweep, kept from displacing the x rows the caches re-serve
__device__ __forceinline__ float4 streamLoad4(const float *p)
{
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uchar4 streamLoad4(const uint8_t *p)
{
    const v4b v = __builtin_nontemporal_load(reinterpret_cast<const v4b *>(p));
    return make_uchar4(v.x, v.y, v.z, v.w);
}
// Storage type of a grid: float, or binary16 for the fine-level iterate / residual of the mixed-precision V-cycle
// (options.precision = 1; arithmetic stays fp32).  Four consecutive cells move as one 16-byte (8-byte) access.
template <class T>
struct Cell;
template <>
struct Cell<float> {
    static __device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ float4 load4nt(const float *p) { return streamLoad4(p); }
    static __device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    static __device__ __forceinline__ void store4nt(float *p, float4 v) { __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f *>(p)); }
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
    static __device__ __forceinline__ void store1(float *p, float v) { *p = v; }
};
// binary16 conversions saturate (a value past 65504 must not become inf: the next sweep would turn it into NaNs)
__device__ __forceinline__ __half toHalfSat(float v) { return __float2half_rn(fminf(fmaxf(v, -65504.f), 65504.f)); }
template <>
struct Cell<__half> {
    static __device__ __forceinline__ float4 load4(const __half *p)
    {
        const uint2 raw = *reinterpret_cast<const uint2 *>(p);
        const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&raw.x)), b = __half22float2(*reinterpret_cast<const __half2 *>(&raw.y));
        return make_float4(a.x, a.y, b.x, b.y);
    }
    static __device__ __forceinline__ float4 load4nt(const __half *p) { return load4(p); }
    static __device__ __forceinline__ void store4(__half *p, float4 v)
    {
        const __half2 a = __halves2half2(toHalfSat(v.x), toHalfSat(v.y)), b = __halves2half2(toHalfSat(v.z), toHalfSat(v.w));
        uint2 raw;
        raw.x = *reinterpret_cast<const unsigned *>(&a);
        raw.y = *reinterpret_cast<const unsigned *>(&b);
        *reinterpret_cast<uint2 *>(p) = raw;
    }
    static __device__ __forceinline__ void store4nt(__half *p, float4 v) { store4(p, v); }
    static __device__ __forceinline__ float load1(const __half *p) { return __half2float(*p); }
    static __device__ __forceinline__ void store1(__half *p, float v) { *p = toHalfSat(v); }
};
// Scales of the mixed-precision cycle (MixScale in mgps_internal.h): the rhs enters as (*sigma * c1) * b, the operator
// term as c2 * (A x).  fp32 grids never use them (bit-identical arithmetic to the unscaled kernels).
__device__ __forceinline__ float mixRhsScale(const MixScale &ms) { return (ms.sigma ? *ms.sigma : 1.f) * ms.c1; }

constexpr int kWave = 64;
constexpr int kXcds = 8;  // MI355X: 8 XCDs, blocks are dealt round-robin over them

// device cell codes (mgps_internal.h): 0 INTERIOR, 1 EXTERIOR, 2 DIRICHLET, 3 general BOUNDARY,
// 4 + d simple BOUNDARY with diagonal d
__device__ __forceinline__ bool activeLabel(unsigned l) { return l == MGPS_INTERIOR_CELL || l >= kCodeGeneral; }
__device__ __forceinline__ bool anyActive(uchar4 l) { return activeLabel(l.x) || activeLabel(l.y) || activeLabel(l.z) || activeLabel(l.w); }
__device__ __forceinline__ bool simpleCell(unsigned l) { return l == MGPS_INTERIOR_CELL || l > kCodeSimple; }
__device__ __forceinline__ float simpleDiag(unsigned l) { return l == MGPS_INTERIOR_CELL ? 6.f : float(int(l) - int(kCodeSimple)); }
// 1/diag of a simple cell: diag is a small integer, one v_rcp_f32 (1 ulp) instead of the ~10-instruction
// IEEE division sequence -- the sweeps issue four of these per thread per plane
__device__ __forceinline__ float simpleRcp(float diag) { return __builtin_amdgcn_rcpf(diag); }

// Row t of the BOUNDARY-cell list applied to x (any callable size_t -> float): lap = diag x_c -
// sum_q w_q x_(c+off_q), the value computeLaplacian returns at Ops.h:258.
template <class X>
__device__ __forceinline__ void boundaryRow(const GridP &g, const X &xAt, int t, size_t c, float &lap, float &diag)
{
    const size_t sy = size_t(g.nx), sz = size_t(g.nx) * g.ny;
    const size_t nb = size_t(g.nbnd);
    const float *r = g.rows + t;
    float acc = 0.f;
    acc -= r[0] * xAt(c - 1);
    acc -= r[nb] * xAt(c + 1);
    acc -= r[2 * nb] * xAt(c - sy);
    acc -= r[3 * nb] * xAt(c + sy);
    acc -= r[4 * nb] * xAt(c - sz);
    acc -= r[5 * nb] * xAt(c + sz);
    diag = r[6 * nb];
    lap = acc + diag * xAt(c);
}

template <int OP>
__device__ __forceinline__ float epilogue(float xc, float bc, float lap, float diag, float omega)
{
    if (OP == OP_JACOBI) return xc + omega * ((bc - lap) / diag);  // Ops.h:356-361
    if (OP == OP_RESIDUAL) return bc - lap;                        // Ops.h:728-731
    return lap;                                                    // Ops.h:708
}
// the same with the reciprocal of the diagonal supplied (simple cells)
template <int OP>
__device__ __forceinline__ float epilogueRcp(float xc, float bc, float lap, float rdiag, float omega)
{
    if (OP == OP_JACOBI) return xc + omega * ((bc - lap) * rdiag);
    if (OP == OP_RESIDUAL) return bc - lap;
    return lap;
}
// mixed precision: rhs scaled by bm, the operator term by c2 (residual only: Jacobi works in the iterate's own units)
template <int OP>
__device__ __forceinline__ float epilogueMix(float xc, float bc, float lap, float rdiag, float omega, float bm, float c2)
{
    if (OP == OP_JACOBI) return xc + omega * ((bm * bc - lap) * rdiag);
    if (OP == OP_RESIDUAL) return bm * bc - c2 * lap;
    return lap;
}
template <int OP>
__device__ __forceinline__ float inactiveValue(float xc)
{
    return OP == OP_JACOBI ? xc : 0.f;  // Jacobi leaves inactive cells alone; r and y are 0 there
}
// what a sweep with the DOT flag sums over the active cells besides its own work: <x, A x> for A.x (the CG loop's
// <p, A p>), <x', b> for the Jacobi sweep (the last sweep of a preconditioning V-cycle delivers <z, r>, CG.h:86, 180)
template <int OP>
__device__ __forceinline__ double dotTerm(float xc, float bc, float res)
{
    return OP == OP_JACOBI ? double(res) * double(bc) : double(xc) * double(res);
}

// Sum of `acc` over the workgroup (up to 1024 threads), left in partials[slot] by thread 0: the A.p launches of the
// CG loop also deliver their share of <p, A p> (CG.h:110-121) instead of a second pass over p and A p.
__device__ __forceinline__ void blockDotStore(double acc, double *__restrict__ partials, unsigned slot)
{
    __shared__ double part[16];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = part[0];
        for (int w = 1; w < int((blockDim.x * blockDim.y + kWave - 1) / kWave); ++w) total += part[w];
        partials[slot] = total;
    }
}

// XCD-aware block remap: the hardware deals consecutive block ids round-robin over the 8 XCDs, each
// with a private 4 MiB L2.  Giving every XCD one contiguous run of logical blocks keeps the y/z
// neighbour rows a block re-reads inside the L2 that already holds them.  Pure speed: any mapping
// is correct.
__device__ __forceinline__ unsigned remapBlock(unsigned bid, unsigned nblocks)
{
    const unsigned per = nblocks / kXcds;
    if (per == 0 || bid >= per * kXcds) return bid;  // ragged tail keeps its id
    return (bid % kXcds) * per + bid / kXcds;
}

// ---------------------------------------------------------------------------------------------
// 7-point stencil sweep: Jacobi (out of place), residual, A.x.  One thread = 4 consecutive cells
// along x (one 16-byte load / store per array), a wave = 256 contiguous cells of a row (or several
// shorter rows).  x-1 / x+1 come from the neighbouring lanes (ds_bpermute), row ends from memory.
// Requires nx % 4 == 0.
// ---------------------------------------------------------------------------------------------
// The quad (4 cells) this thread owns in share `block` of a level's activity list -- the list entries are runs of
// chunkCells consecutive cells that hold an active cell: 1024 = one entry per workgroup, 256 = one per wavefront, 64 / 32 =
// one per 16 / 8 lanes (free surfaces that cut the x-rows: most of a 256-cell run would be air).  false: list padding (-1).
// (tid: the thread's index in its 256-thread share -- threadIdx.x, except in kernels whose workgroups take several shares)
__device__ __forceinline__ bool listQuad(const int32_t *__restrict__ chunks, int chunkCells, size_t block, size_t &t, unsigned tid)
{
    if (chunkCells == kChunkCells) {
        t = size_t(chunks[block]) * 256 + tid;
        return true;
    }
    int ch;
    if (chunkCells == kWaveChunkCells) {  // wave-uniform index: a scalar load
        ch = chunks[block * 4 + __builtin_amdgcn_readfirstlane(tid >> 6)];
        t = size_t(max(ch, 0)) * kWave + (tid & (kWave - 1));
    } else {
        const int lanes = chunkCells >> 2, shift = __ffs(lanes) - 1;  // 16 or 8 lanes per run
        ch = chunks[(block << (8 - shift)) + (tid >> shift)];
        t = (size_t(max(ch, 0)) << shift) + (tid & (lanes - 1));
    }
    return ch >= 0;
}
__device__ __forceinline__ bool listQuad(const int32_t *__restrict__ chunks, int chunkCells, size_t block, size_t &t) { return listQuad(chunks, chunkCells, block, t, threadIdx.x); }
// lanes whose x-neighbour quad lives in another lane's registers: all but the ends of a run
__device__ __forceinline__ int listRunMask(const int32_t *chunks, int chunkCells) { return (chunks && chunkCells < kWaveChunkCells) ? (chunkCells >> 2) - 1 : kWave - 1; }

// XZERO: the iterate is known to be zero everywhere (the first sweep of a stroke that starts from the cleared grid, MG.cpp:439-440 /
// 566): nothing of x is loaded and nobody had to clear it
// The body serves two kernels: stencilQuadKernel (a 256-thread workgroup = one share `vblock` of the launch) and strokeFrontKernel
// (1024-thread workgroups, four shares each).  KEEP: `keep` holds one bit per cell, set = this launch must not write the cell
// (strokeFrontKernel: the band closure, which the band boxes own)
template <int OP, bool DOT, class TX, bool XZERO, bool KEEP>
__device__ __forceinline__ void stencilQuadBody(const GridP &g, TX *__restrict__ out, const TX *__restrict__ x, const float *__restrict__ b, float omega,
                                                unsigned nblocks, const int32_t *__restrict__ chunks, double *__restrict__ dotPartials, const MixScale &ms,
                                                unsigned vblock, unsigned tid, const uint32_t *__restrict__ keep)
{
    constexpr bool kMixed = !std::is_same<TX, float>::value;
    const unsigned nq = unsigned(g.nx) >> 2;  // quads per row
    const size_t rows = size_t(g.ny) * g.nz;
    const size_t totalQuads = size_t(nq) * rows;
    const unsigned block = remapBlock(vblock, nblocks);
    size_t t = size_t(block) * 256 + tid;
    bool valid = true;
    if (chunks) valid = listQuad(chunks, g.chunkCells, block, t, tid);  // only the runs that hold active cells
    valid = valid && t < totalQuads;
    const size_t tt = valid ? t : totalQuads - 1;
    // Round 5: no division in the index arithmetic.  The quad's first cell is 4 tt (a row holds nq quads of 4 cells); its x index --
    // needed for the active x range alone -- is a mask where nq is a power of two, else one 32-bit modulo; the row and the plane are
    // not needed at all: a neighbour row / plane that would leave the array (the first and last x-row and plane of the grid) is
    // replaced by the quad's own, as before, by comparing the flat index -- elsewhere at a grid face the neighbour "row" is the
    // last row of the plane before: EXTERIOR shell cells either way, whose results are never formed from their neighbours.
    // (the three divisions by nq and ny and the 64-bit products behind them were 100 of the kernel's 207 vector instructions per
    // wave -- 59 % of the launch's SIMD cycles at 1024^3, r05_sq_1024.txt)
    const size_t c = tt << 2;
    const size_t sy = size_t(g.nx), sz = size_t(g.nx) * g.ny, ncells = totalQuads << 2;
    const bool nqPow2 = (nq & (nq - 1u)) == 0u;
    const int i = int((nqPow2 ? (unsigned(tt) & (nq - 1u)) : (totalQuads <= 0xffffffffull ? unsigned(tt) % nq : unsigned(tt % nq))) << 2);

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // quads outside the level's active x range (GridP::xlo): EXTERIOR padding, zero in every grid -- nothing of theirs is loaded or
    // stored.  Their lanes issue the same (unconditional) loads as everybody, aimed at the nearest quad of the range in their row --
    // lines the wave fetches anyway -- and drop what arrives: a branch or a select per load put every binary16 load + conversion
    // behind its own wait (mixed-precision sweep 0.19 -> 0.26 ms at 512^3)
    const bool live = i >= g.xlo && i < g.xhi;
    valid = valid && live;
    const size_t cl = live ? c : c - size_t(i) + size_t(min(max(i, g.xlo), max(g.xhi - 4, 0)));
    // neighbour rows and planes: the loads only have to stay inside the array (or its ghost planes)
    const size_t cym = cl >= sy ? cl - sy : cl, cyp = cl + sy < ncells ? cl + sy : cl;
    const size_t czm = (cl >= sz || g.ghostLo) ? cl - sz : cl, czp = (cl + sz < ncells || g.ghostHi) ? cl + sz : cl;
    float4 xc = XZERO ? zero4 : Cell<TX>::load4(x + cl);
    const float4 ym = XZERO ? zero4 : Cell<TX>::load4(x + cym);
    const float4 yp = XZERO ? zero4 : Cell<TX>::load4(x + cyp);
    const float4 zm = XZERO ? zero4 : Cell<TX>::load4(x + czm);
    const float4 zp = XZERO ? zero4 : Cell<TX>::load4(x + czp);
    // (the four codes as one word, taken apart only where they are used: unpacking them here made the load's wait precede the rhs load)
    const unsigned labw = g.streaming ? __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(g.lab + cl)) : *reinterpret_cast<const unsigned *>(g.lab + cl);
    float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (OP != OP_APPLY) bc = g.streaming ? streamLoad4(b + cl) : *reinterpret_cast<const float4 *>(b + cl);
    if (!live) xc = zero4;  // (what the neighbour lanes see of this quad; its own results are never stored, `valid`)
    const bool ld = live && !XZERO;

    // x neighbours across the quad boundary
    const int runMask = listRunMask(chunks, g.chunkCells), lane = tid & runMask;
    float left = __shfl_up(xc.w, 1);
    float right = __shfl_down(xc.x, 1);
    if (ld) {  // (a run that starts / ends at a row end reads the EXTERIOR cell of the row before / after, or nothing at the array's ends)
        if (lane == 0 || i == 0) left = (i > 0) ? Cell<TX>::load1(x + c - 1) : 0.f;
        if (lane == runMask || i + 4 >= g.nx || t + 1 >= totalQuads) right = (i + 4 < g.nx) ? Cell<TX>::load1(x + c + 4) : 0.f;
    }

    const float xs[6] = {left, xc.x, xc.y, xc.z, xc.w, right};
    const float yms[4] = {ym.x, ym.y, ym.z, ym.w}, yps[4] = {yp.x, yp.y, yp.z, yp.w};
    const float zms[4] = {zm.x, zm.y, zm.z, zm.w}, zps[4] = {zp.x, zp.y, zp.z, zp.w};
    const float bs[4] = {bc.x, bc.y, bc.z, bc.w};
    const unsigned ls[4] = {labw & 255u, (labw >> 8) & 255u, (labw >> 16) & 255u, labw >> 24};
    const float bm = kMixed ? mixRhsScale(ms) : 1.f;
    float res[4];
    // INTERIOR and simple BOUNDARY cells; general BOUNDARY cells are patched by boundaryOpKernel
    // right after this launch
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float diag = simpleDiag(ls[e]);
        const float lap = diag * xs[e + 1] - (xs[e] + xs[e + 2] + yms[e] + yps[e] + zms[e] + zps[e]);
        if (kMixed) res[e] = simpleCell(ls[e]) ? epilogueMix<OP>(xs[e + 1], bs[e], lap, simpleRcp(diag), omega, bm, ms.c2) : inactiveValue<OP>(xs[e + 1]);
        else res[e] = simpleCell(ls[e]) ? epilogueRcp<OP>(xs[e + 1], bs[e], lap, simpleRcp(diag), omega) : inactiveValue<OP>(xs[e + 1]);
    }
    unsigned skip = 0;  // cells of this quad the launch leaves alone
    if (KEEP && valid) skip = (keep[c >> 5] >> (c & 31)) & 15u;
    if (valid && !skip) {
        if (g.streaming) Cell<TX>::store4nt(out + c, make_float4(res[0], res[1], res[2], res[3]));
        else Cell<TX>::store4(out + c, make_float4(res[0], res[1], res[2], res[3]));
    } else if (KEEP && valid) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!(skip & (1u << e))) Cell<TX>::store1(out + c + e, res[e]);
    }
    if (DOT) {  // general BOUNDARY cells add theirs in boundaryOpKernel
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (valid && simpleCell(ls[e])) acc += dotTerm<OP>(xs[e + 1], bs[e], kMixed ? __half2float(toHalfSat(res[e])) : res[e]);  // (the value as stored)
        blockDotStore(acc, dotPartials, vblock);
    }
}
template <int OP, bool DOT = false, class TX = float, bool XZERO = false>
__global__ __launch_bounds__(256) void stencilQuadKernel(GridP g, TX *__restrict__ out, const TX *__restrict__ x,
                                                          const float *__restrict__ b, float omega, unsigned nblocks,
                                                          const int32_t *__restrict__ chunks, double *__restrict__ dotPartials = nullptr,
                                                          MixScale ms = MixScale{})
{
    stencilQuadBody<OP, DOT, TX, XZERO, false>(g, out, x, b, omega, nblocks, chunks, dotPartials, ms, blockIdx.x, threadIdx.x, nullptr);
}

// ---------------------------------------------------------------------------------------------
// The same sweep for the large levels (nx >= 256): a workgroup of 64 x kPlaneRows threads owns a
// 256 x kPlaneRows tile in (x, y) and marches through `zc` planes.  z-1 / z / z+1 of a thread's own
// quad live in registers (each x value is loaded from memory once per tile instead of three times),
// the current plane is staged in an LDS tile with a one-cell halo (double buffered, one barrier per
// plane) for the x+-1 / y+-1 neighbours, and the next plane's loads are issued before the current
// plane is computed.  History of the comparison with the cache-only kernel above on 4 MiB planes (1024^3): round 1 2.63 vs
// 2.87 ms (HBM-side reads 9.2 vs 14.8 B per cell, rocprofv3 FETCH_SIZE: the quad kernel walked its runs plane by plane and
// the z neighbours missed the L2); rounds 2-3 the quad kernel walks 32-row strips through all planes (1.03 x the algorithmic
// traffic) and skips the row-end padding: 1.82 ms against 1.85 ms here -- planes up to 4 MiB (kPlaneSweepMinPlaneBytes) take
// the quad kernel since the end of round 3, larger planes this one.
// ---------------------------------------------------------------------------------------------
constexpr int kPlanePitch = 256 + 8;  // 4 floats of halo on each side keep the rows 16-byte aligned
// lerp of Ops.h:841-871: (1 - f) a + f b, this exact form (HDK's SYSlerp breaks the R / P symmetry, Ops.h:837-839)
__device__ __forceinline__ float lerpRef(float a, float b, float f) { return (1.f - f) * a + f * b; }

// A wave-uniform pointer pinned to a scalar register pair, its derivation hidden from the optimiser (an empty asm): in the
// plane-marching kernels below the loop optimiser otherwise folds "plane base + lane offset" into one 64-bit vector induction
// variable per array -- two vector registers each, and the spills that follow.  With the base in scalar registers an access takes
// the form global_load v, v_offset32, s[base:base+1].  The asm also hides that the pointer came from a kernel argument, and a
// pointer of unknown origin is accessed with flat_load / flat_store: the accessors below go through the global address space
// explicitly (base + a 32-bit offset in cells).
template <class T>
__device__ __forceinline__ T *scalarBase(T *p)
{
    asm volatile("" : "+s"(p));
    return p;
}
#define MGPS_GLOBAL_AS __attribute__((address_space(1)))
__device__ __forceinline__ float4 gLoad4(const float *base, unsigned cell)
{
    const v4f v = *(const MGPS_GLOBAL_AS v4f *)((const MGPS_GLOBAL_AS char *)base + cell * 4u);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 gLoad4nt(const float *base, unsigned cell)
{
    const v4f v = __builtin_nontemporal_load((const MGPS_GLOBAL_AS v4f *)((const MGPS_GLOBAL_AS char *)base + cell * 4u));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float gLoad1(const float *base, unsigned cell) { return *(const MGPS_GLOBAL_AS float *)((const MGPS_GLOBAL_AS char *)base + cell * 4u); }
__device__ __forceinline__ uchar4 gLoadCodes4nt(const uint8_t *base, unsigned cell)
{
    const v4b v = __builtin_nontemporal_load((const MGPS_GLOBAL_AS v4b *)((const MGPS_GLOBAL_AS char *)base + cell));
    return make_uchar4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned gLoadCode1(const uint8_t *base, unsigned cell) { return *((const MGPS_GLOBAL_AS uint8_t *)base + cell); }
__device__ __forceinline__ void gStore4(float *base, unsigned cell, float4 v) { *(MGPS_GLOBAL_AS v4f *)((MGPS_GLOBAL_AS char *)base + cell * 4u) = v4f{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void gStore4nt(float *base, unsigned cell, float4 v)
{
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, (MGPS_GLOBAL_AS v4f *)((MGPS_GLOBAL_AS char *)base + cell * 4u));
}

// v where `keep`, else 0 -- component by component with the literal (a float4 of zeros as the other operand of the selects was a
// register quad the compiler spilled)
__device__ __forceinline__ float4 keepIf(float4 v, bool keep) { return make_float4(keep ? v.x : 0.f, keep ? v.y : 0.f, keep ? v.z : 0.f, keep ? v.w : 0.f); }

template <int OP, bool DOT = false, bool XZERO = false>  // XZERO: see stencilQuadKernel
__global__ __launch_bounds__(64 * kPlaneRows, 8) void stencilPlaneKernel(  // (8 waves per SIMD = two workgroups per CU: at most 64 registers)
GridP g, float *__restrict__ out,
                                                                      const float *__restrict__ x,
                                                                      const float *__restrict__ b, float omega,
                                                                      unsigned nbx, unsigned nby, unsigned nbz, int zc,
                                                                      const int32_t *__restrict__ blocks,
                                                                      double *__restrict__ dotPartials = nullptr)
{
    double dotAcc = 0.0;
    __shared__ float plane[2][(kPlaneRows + 2) * kPlanePitch];
    unsigned bid = remapBlock(blockIdx.x, gridDim.x);
    if (blocks) bid = unsigned(blocks[bid]);  // only blocks that hold active cells
    bid = __builtin_amdgcn_readfirstlane(bid);
    const unsigned bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int lane = threadIdx.x & (kWave - 1), ty = threadIdx.x / kWave;
    const int i = int(bx) * 256 + lane * 4, j = int(by) * kPlaneRows + ty;
    // threads past the grid edge shadow the last quad / row: their loads stay in bounds, they take
    // part in the barriers, they do not store
    const int ic = min(i, g.nx - 4), jc = min(j, g.ny - 1);
    // quad columns outside the level's active x range (GridP::xlo): zero in every grid, staged as zeros, nothing loaded or stored
    const bool live = ic >= g.xlo && ic < g.xhi;
    const bool valid = i < g.nx && j < g.ny && live;
    const bool ld = live && !XZERO;
    const ptrdiff_t sz = ptrdiff_t(g.nx) * g.ny;
    const int k0 = int(bz) * zc, k1 = min(k0 + zc, g.nz);
    // addresses: the plane's base (scalarBase) + one 32-bit offset inside the plane per thread.  Planes are clamped to what exists:
    // the ghost planes of a slab, else the first / last plane (EXTERIOR shell there: results 0 whatever the neighbours hold)
    const int kLo = g.ghostLo ? -1 : 0, kHi = g.ghostHi ? g.nz : g.nz - 1;
    const unsigned off = unsigned(jc) * unsigned(g.nx) + unsigned(ic);
    const unsigned offYm = jc > 0 ? off - unsigned(g.nx) : off, offYp = jc < g.ny - 1 ? off + unsigned(g.nx) : off;
    auto planeOf = [&](const float *p, int k) { return scalarBase(p + ptrdiff_t(min(max(k, kLo), kHi)) * sz); };
    const bool rowTop = ty == 0, rowBot = ty == kPlaneRows - 1, colL = lane == 0, colR = lane == kWave - 1;
    // the x-halo cell of the first / last lane: one unconditional load per wave (the other lanes re-read their own cell and drop
    // it) -- a branch per side made every wave wait for all its loads in flight before each of the two
    const bool useHx = ld && ((colL && ic > 0) || (colR && ic + 4 < g.nx));
    const unsigned offHx = !useHx ? off : (colL ? off - 1u : off + 4u);

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const uchar4 ext4 = make_uchar4(MGPS_EXTERIOR_CELL, MGPS_EXTERIOR_CELL, MGPS_EXTERIOR_CELL, MGPS_EXTERIOR_CELL);
    // Two planes of a thread's own x quad are in flight at any time: plane k + 2 is requested while plane k is computed, so the
    // plane k + 1 a step needs (its z + 1 neighbours) was requested a whole step earlier; the halo is requested one plane ahead,
    // the rhs and the codes of a plane at the top of its own step (in flight across the barrier).  The z - 1 values of a step are
    // the thread's own store of the step before, read back from the other LDS buffer.
    // (Tried on top and dropped: the four planes in a ring of named registers with the march unrolled four times, every load
    // unconditional and ordered by first use: 1024^3 cycle 10.32 -> 12.22 ms.  Round 4: plane bases in scalar registers, the
    // z - 1 quad out of LDS, no rhs / code look-ahead, the branch-free x-halo load -- no spills left: see residualZKernel.)
    float *const mine0 = plane[0] + (ty + 1) * kPlanePitch + 4 + lane * 4;
    constexpr int kBufFloats = (kPlaneRows + 2) * kPlanePitch;
    *reinterpret_cast<float4 *>(mine0 + kBufFloats) = ld ? gLoad4(planeOf(x, k0 - 1), off) : zero4;
    const float *xk = XZERO ? x : planeOf(x, k0);
    float4 xc = ld ? gLoad4(xk, off) : zero4;
    float4 xp = ld ? gLoad4(planeOf(x, k0 + 1), off) : zero4;
    float4 hy = zero4;  // y-halo row this thread stages (top / bottom rows only)
    if (ld && rowTop) hy = gLoad4(xk, offYm);
    if (ld && rowBot) hy = gLoad4(xk, offYp);
    float hx = ld ? gLoad1(xk, offHx) : 0.f;  // x-halo cell this thread stages (first / last lane only; the others drop it where it is staged)

    int buf = 0;
    for (int k = k0; k < k1; ++k) {
        float *me = mine0 + buf * kBufFloats;
        *reinterpret_cast<float4 *>(me) = xc;
        if (rowTop) *reinterpret_cast<float4 *>(me - kPlanePitch) = hy;
        if (rowBot) *reinterpret_cast<float4 *>(me + kPlanePitch) = hy;
        if (colL) me[-1] = useHx ? hx : 0.f;
        if (colR) me[4] = useHx ? hx : 0.f;
        float4 bc = zero4;
        uchar4 lc = ext4;
        if (live) {
            if (OP != OP_APPLY) bc = gLoad4nt(scalarBase(b + ptrdiff_t(k) * sz), off);
            lc = gLoadCodes4nt(scalarBase(g.lab + ptrdiff_t(k) * sz), off);
        }
        // the plane after the next one (own quad); the next plane's halo
        float4 xq = zero4, hyn = hy;
        float hxn = hx;
        if (k + 1 < k1 && ld) {  // (the last step's z + 1 plane is already here: xp)
            const float *xn = planeOf(x, k + 1);
            xq = gLoad4(planeOf(x, k + 2), off);
            if (rowTop) hyn = gLoad4(xn, offYm);
            if (rowBot) hyn = gLoad4(xn, offYp);
            hxn = gLoad1(xn, offHx);
        }
        __syncthreads();
        const float4 ym = *reinterpret_cast<const float4 *>(me - kPlanePitch);
        const float4 yp = *reinterpret_cast<const float4 *>(me + kPlanePitch);
        const float4 xm = *reinterpret_cast<const float4 *>(mine0 + (buf ^ 1) * kBufFloats);
        const float xs[6] = {me[-1], xc.x, xc.y, xc.z, xc.w, me[4]};
        const float yms[4] = {ym.x, ym.y, ym.z, ym.w}, yps[4] = {yp.x, yp.y, yp.z, yp.w};
        const float zms[4] = {xm.x, xm.y, xm.z, xm.w}, zps[4] = {xp.x, xp.y, xp.z, xp.w};
        const float bs[4] = {bc.x, bc.y, bc.z, bc.w};
        const unsigned ls[4] = {lc.x, lc.y, lc.z, lc.w};
        float res[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float diag = simpleDiag(ls[e]);
            const float lap = diag * xs[e + 1] - (xs[e] + xs[e + 2] + yms[e] + yps[e] + zms[e] + zps[e]);
            res[e] = simpleCell(ls[e]) ? epilogueRcp<OP>(xs[e + 1], bs[e], lap, simpleRcp(diag), omega) : inactiveValue<OP>(xs[e + 1]);
        }
        if (valid)  // streamed out: nothing re-reads the sweep's output before it has left the caches (+5 % at 1024^3)
            gStore4nt(scalarBase(out + ptrdiff_t(k) * sz), off, make_float4(res[0], res[1], res[2], res[3]));
        if (DOT) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (valid && simpleCell(ls[e])) dotAcc += dotTerm<OP>(xs[e + 1], bs[e], res[e]);
        }
        xc = xp;
        xp = xq;
        hy = hyn;
        hx = hxn;
        buf ^= 1;
    }
    if (DOT) {
        __syncthreads();  // (the LDS planes are done with)
        blockDotStore(dotAcc, dotPartials, blockIdx.x);
    }
}

// ---------------------------------------------------------------------------------------------
// The residual of a down-stroke, restricted along z as it is formed (round 4; Ops.h:716-732 into the z part of Ops.h:734-835).
// Full weighting is separable: coarse(I, J, K) = sum_c w_c sum_b w_b sum_a w_a r(2I-1+a, 2J-1+b, 2K-1+c).  The plane-marching
// form above walks the planes of a column in order, so a thread can fold the four planes of every coarse K itself and store
// rz(i, j, K) = w0 r(2K-1) + w1 r(2K) + w2 r(2K+1) + w3 r(2K+2) -- half the planes of r, which is never written: 11 B per
// cell instead of 13, and the x-y restriction that follows (restrictXYKernel) reads 2 B per fine cell with no overlap along
// z instead of 4 B with the 4-plane footprint.  A block of planes [k0, k1) also forms r on planes k0 - 1 and k1 (two more
// planes per zc).  Levels without ghost planes or binary16 grids; general BOUNDARY cells (not a simple code: r = 0 here) join through
// residualZGeneralKernel (launchResidualZ); `rz` is a grid
// of nx x ny x nz/2 that nobody else writes: blocks with no active cell in them, below them or above them are never written and stay
// zero (residualZEdgeKernel serves the blocks next to active ones).
// ---------------------------------------------------------------------------------------------
// A cut level of a slab run (g.ghostLo / g.ghostHi, round 5): the planes just outside the slab are the neighbours' -- x there is the
// ghost plane, and r there comes complete from the neighbour in the ghost planes of `rEdge` (the level's residual grid: the ranks
// exchange r on their boundary planes, as the separate restriction does): w0 r(-1) opens coarse plane 0, w3 r(nz) closes the last
// one, in the order of the planes like everywhere else.
__global__ __launch_bounds__(64 * kPlaneRows, 8) void residualZKernel(GridP g, float *__restrict__ rz, const float *__restrict__ x,
                                                                     const float *__restrict__ b, unsigned nbx, unsigned nby, int zc,
                                                                     const int32_t *__restrict__ blocks, const float *__restrict__ rEdge)
{
    __shared__ float plane[2][(kPlaneRows + 2) * kPlanePitch];
    unsigned bid = remapBlock(blockIdx.x, gridDim.x);
    if (blocks) bid = unsigned(blocks[bid]);  // (blocks without active cells: residualZEdgeKernel writes what they owe rz)
    bid = __builtin_amdgcn_readfirstlane(bid);  // (the plane bases below: scalar registers)
    const unsigned bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int lane = threadIdx.x & (kWave - 1), ty = threadIdx.x / kWave;
    const int i = int(bx) * 256 + lane * 4, j = int(by) * kPlaneRows + ty;
    const int ic = min(i, g.nx - 4), jc = min(j, g.ny - 1);
    const bool live = ic >= g.xlo && ic < g.xhi;
    const bool valid = i < g.nx && j < g.ny && live;
    const size_t sz = size_t(g.nx) * g.ny;
    const int k0 = int(bz) * zc, k1 = min(k0 + zc, g.nz);  // both even
    const int ks = max(k0 - 1, 0), ke = min(k1, g.nz - 1);   // planes outside the grid: r = 0, nothing to add
    // addresses: the plane's base (wave-uniform: scalar registers) + one 32-bit offset inside the plane per thread
    // Round 5: every load of the march is UNCONDITIONAL.  A lane outside the active x range (`live`) aims at the nearest quad of
    // the range in its row -- a line the wave fetches anyway -- and what arrives is dropped where it is used; the wave of a middle
    // row re-reads its own quad as its "y halo".  With no branch around a load the compiler can count them: the wait behind the
    // barrier becomes vmcnt(3) -- this plane's rhs and codes -- and the three loads for the next planes stay in flight across the
    // arithmetic (round 4: a branch per load left it no choice but vmcnt(0) there, a memory round trip per plane).
    // (a live lane's own quad IS that quad: offL serves its stores as well)
    const unsigned offL = unsigned(jc) * unsigned(g.nx) + unsigned(min(max(ic, g.xlo), max(g.xhi - 4, g.xlo)));
    const bool rowTop = ty == 0, rowBot = ty == kPlaneRows - 1, colL = lane == 0, colR = lane == kWave - 1;
    const bool hasL = ic > 0, hasR = ic + 4 < g.nx;  // (the grid continues on that side)
    // (planes clamped to what exists -- the ghost planes of a slab, else the grid: the first and the last plane of a whole-grid level
    // are EXTERIOR shell, whose results are 0 whatever their neighbours hold -- the assumption stencilPlaneKernel makes at the faces)
    const int kLo = g.ghostLo ? -1 : 0, kHi = g.ghostHi ? g.nz : g.nz - 1;
    auto planeOf = [&](const float *p, int k) { return scalarBase(p + ptrdiff_t(min(max(k, kLo), kHi)) * ptrdiff_t(sz)); };
    float *const mine0 = plane[0] + (ty + 1) * kPlanePitch + 4 + lane * 4;
    constexpr int kBufFloats = (kPlaneRows + 2) * kPlanePitch;
    // the y-halo row a wave stages: the row above the tile (its first wave), below it (its last), else its own quad again
    const unsigned offHy = (rowTop && jc > 0) ? offL - unsigned(g.nx) : (rowBot && jc < g.ny - 1) ? offL + unsigned(g.nx) : offL;
    {  // plane ks - 1 of the thread's own quad: the z - 1 values of a step are read back from the LDS buffer of the step before
        const float4 xm = gLoad4(planeOf(x, ks - 1), offL);
        *reinterpret_cast<float4 *>(mine0 + kBufFloats) = keepIf(xm, live);
    }
    const float *xk = planeOf(x, ks);
    float4 xc = gLoad4(xk, offL);
    float4 xp = gLoad4(planeOf(x, ks + 1), offL);
    float4 hy = gLoad4(xk, offHy);
    // the x-halo cell of the first / last lane: one unconditional load per wave (the other lanes re-read their own cell and drop it)
    // -- a branch per side made every wave wait for all its loads in flight before each of the two
    const bool useHx = live && ((colL && hasL) || (colR && hasR));
    const unsigned offHx = !useHx ? offL : colL ? offL - 1u : offL + 4u;
    float hx = gLoad1(xk, offHx);  // (lanes without such a cell: whatever arrives is dropped where hx is staged -- a select here
                                                // would wait for the load, the last one requested, and with it for every load in flight)
    constexpr float w0 = 0.125f, w1 = 0.375f, w2 = 0.375f, w3 = 0.125f;
    // coarse plane (k - 1) / 2 with its first terms (accPrev) and the one after it (accCur), see the fold below
    float accPrev[4] = {0.f, 0.f, 0.f, 0.f}, accCur[4] = {0.f, 0.f, 0.f, 0.f};
    if (rEdge && k0 == 0 && g.ghostLo) {  // the neighbour's r on the plane below the slab: the first term of coarse plane 0 (wave-uniform branch)
        const float4 e = gLoad4(scalarBase(rEdge - ptrdiff_t(sz)), offL);
        accCur[0] = w0 * e.x;
        accCur[1] = w0 * e.y;
        accCur[2] = w0 * e.z;
        accCur[3] = w0 * e.w;
    }
    int buf = 0;
    for (int k = ks; k <= ke; ++k) {
        float *me = mine0 + buf * kBufFloats;
        xc = keepIf(xc, live);  // (what the neighbours read of a quad outside the range is 0; its own results are never stored)
        *reinterpret_cast<float4 *>(me) = xc;
        if (rowTop) *reinterpret_cast<float4 *>(me - kPlanePitch) = keepIf(hy, live);
        if (rowBot) *reinterpret_cast<float4 *>(me + kPlanePitch) = keepIf(hy, live);
        if (colL) me[-1] = useHx ? hx : 0.f;
        if (colR) me[4] = useHx ? hx : 0.f;
        // this plane's rhs and codes (in flight across the barrier), the own quad two planes ahead, the next plane's halo
        const float *bk = planeOf(b, k);
        const uint8_t *lk = scalarBase(g.lab + size_t(k) * sz);
        const float *xn = planeOf(x, k + 1), *xq2 = planeOf(x, k + 2);
        // (unconditional, this plane's first: see offL; past the last plane the bases are clamped, what arrives is never used)
        const float4 bc = gLoad4nt(bk, offL);
        const uchar4 lc = gLoadCodes4nt(lk, offL);
        // (in the order of their use: the halo of the next plane is staged at the top of the next step, the quad two planes ahead is
        // needed behind the next barrier -- requested last, it is the one left in flight across the top of the loop)
        const float hxn = gLoad1(xn, offHx);
        float4 hyn = hy;
        if (rowTop || rowBot) hyn = gLoad4(xn, offHy);  // (wave-uniform: the tile's first and last wave only)
        const float4 xq = gLoad4(xq2, offL);
        __syncthreads();
        const float4 ym = *reinterpret_cast<const float4 *>(me - kPlanePitch);
        const float4 yp = *reinterpret_cast<const float4 *>(me + kPlanePitch);
        const float4 xm = *reinterpret_cast<const float4 *>(mine0 + (buf ^ 1) * kBufFloats);  // (this thread's own store of the last step)
        const float xs[6] = {me[-1], xc.x, xc.y, xc.z, xc.w, me[4]};
        const float yms[4] = {ym.x, ym.y, ym.z, ym.w}, yps[4] = {yp.x, yp.y, yp.z, yp.w};
        const float zms[4] = {xm.x, xm.y, xm.z, xm.w}, zps[4] = {xp.x, xp.y, xp.z, xp.w};
        const float bs[4] = {bc.x, bc.y, bc.z, bc.w};
        const unsigned ls[4] = {lc.x, lc.y, lc.z, lc.w};
        float res[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // the arithmetic of stencilPlaneKernel<OP_RESIDUAL>
            const float diag = simpleDiag(ls[e]);
            const float lap = diag * xs[e + 1] - (xs[e] + xs[e + 2] + yms[e] + yps[e] + zms[e] + zps[e]);
            res[e] = simpleCell(ls[e]) ? epilogueRcp<OP_RESIDUAL>(xs[e + 1], bs[e], lap, simpleRcp(diag), 0.f) : inactiveValue<OP_RESIDUAL>(xs[e + 1]);
        }
        // the fold along z, terms in the order of the planes: plane 2 m + 1 is the third term of coarse plane m and the first of
        // m + 1, plane 2 m the second term of m and the last of m - 1
        if (k & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                accPrev[e] = accCur[e] + w2 * res[e];
                accCur[e] = w0 * res[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                accPrev[e] += w3 * res[e];
                accCur[e] += w1 * res[e];
            }
            if (valid && k >= k0 + 2) gStore4nt(scalarBase(rz + size_t((k >> 1) - 1) * sz), offL, make_float4(accPrev[0], accPrev[1], accPrev[2], accPrev[3]));
        }
        xc = xp;
        xp = xq;
        hy = hyn;
        hx = hxn;
        buf ^= 1;
    }
    // the top block of the grid: plane nz does not exist, the last coarse plane is complete with three terms -- or, on a cut, its
    // fourth term is the neighbour's r on the plane above the slab
    if (valid && k1 == g.nz) {
        if (rEdge && g.ghostHi) {
            const float4 e = gLoad4(scalarBase(rEdge + ptrdiff_t(g.nz) * ptrdiff_t(sz)), offL);
            accPrev[0] += w3 * e.x;
            accPrev[1] += w3 * e.y;
            accPrev[2] += w3 * e.z;
            accPrev[3] += w3 * e.w;
        }
        gStore4(rz + size_t((g.nz >> 1) - 1) * sz, offL, make_float4(accPrev[0], accPrev[1], accPrev[2], accPrev[3]));
    }
}

// A block without active cells still owes rz the terms of the planes next to it when the block below or above holds active cells:
// w0 r on the last plane of the block below is the first (and only non-zero) term of its first coarse plane, w3 r on the first plane
// of the block above the last term of its last one -- coarse cells of a NEIGHBOURING column of blocks read those entries.  One
// workgroup per such block and side (a list made once, planeBlockEdges); every other entry of a block off the activity list is
// 0, which is what rz holds there since it was made.
// One plane, no march: the quad kernel's way of gathering the neighbours (x by lane shuffle, y and z from the caches).
__global__ __launch_bounds__(64 * kPlaneRows) void residualZEdgeKernel(GridP g, float *__restrict__ rz, const float *__restrict__ x, const float *__restrict__ b,
                                                                      unsigned nbx, unsigned nby, int zc, const int32_t *__restrict__ edges)
{
    // edges[w] = 2 * block + side: the blocks without active cells whose neighbour below (side 0) / above (side 1) has some
    const unsigned bid = unsigned(edges[blockIdx.x]) >> 1, side = unsigned(edges[blockIdx.x]) & 1u;
    const unsigned bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int k0 = int(bz) * zc, k1 = min(k0 + zc, g.nz);
    const int k = side == 0 ? k0 - 1 : k1, K = side == 0 ? (k0 >> 1) : (k1 >> 1) - 1;
    const float wz = 0.125f;  // w0 = w3
    if (k < 0 || k >= g.nz) return;
    const int lane = threadIdx.x & (kWave - 1), ty = threadIdx.x / kWave;
    const int i = int(bx) * 256 + lane * 4, j = int(by) * kPlaneRows + ty;
    if (i >= g.nx || j >= g.ny) return;  // (whole wavefront rows leave together only when j is past the grid; lanes past nx: no shuffle partner needs them)
    const bool live = i >= g.xlo && i < g.xhi;
    const size_t sy = size_t(g.nx), sz = size_t(g.nx) * g.ny;
    const size_t c = size_t(k) * sz + size_t(j) * sy + i;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // (faces of the grid: EXTERIOR shell, results 0 whatever the clamped neighbours hold)
    const size_t cym = j > 0 ? c - sy : c, cyp = j < g.ny - 1 ? c + sy : c, czm = k > 0 ? c - sz : c, czp = k < g.nz - 1 ? c + sz : c;
    const float4 xc = live ? *reinterpret_cast<const float4 *>(x + c) : zero4;
    const float4 ym = live ? *reinterpret_cast<const float4 *>(x + cym) : zero4, yp = live ? *reinterpret_cast<const float4 *>(x + cyp) : zero4;
    const float4 zm = live ? *reinterpret_cast<const float4 *>(x + czm) : zero4, zp = live ? *reinterpret_cast<const float4 *>(x + czp) : zero4;
    const float4 bc = live ? *reinterpret_cast<const float4 *>(b + c) : zero4;
    const unsigned labw = live ? *reinterpret_cast<const unsigned *>(g.lab + c) : 0x01010101u * unsigned(MGPS_EXTERIOR_CELL);
    float left = __shfl_up(xc.w, 1), right = __shfl_down(xc.x, 1);
    if (lane == 0) left = (live && i > 0) ? x[c - 1] : 0.f;
    if (lane == kWave - 1 || i + 4 >= g.nx) right = (live && i + 4 < g.nx) ? x[c + 4] : 0.f;
    if (!live) return;
    const float xs[6] = {left, xc.x, xc.y, xc.z, xc.w, right};
    const float yms[4] = {ym.x, ym.y, ym.z, ym.w}, yps[4] = {yp.x, yp.y, yp.z, yp.w};
    const float zms[4] = {zm.x, zm.y, zm.z, zm.w}, zps[4] = {zp.x, zp.y, zp.z, zp.w};
    const float bs[4] = {bc.x, bc.y, bc.z, bc.w};
    const unsigned ls[4] = {labw & 255u, (labw >> 8) & 255u, (labw >> 16) & 255u, labw >> 24};
    float res[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // the arithmetic of stencil...Kernel<OP_RESIDUAL>, then the one z weight
        const float diag = simpleDiag(ls[e]);
        const float lap = diag * xs[e + 1] - (xs[e] + xs[e + 2] + yms[e] + yps[e] + zms[e] + zps[e]);
        const float r = simpleCell(ls[e]) ? epilogueRcp<OP_RESIDUAL>(xs[e + 1], bs[e], lap, simpleRcp(diag), 0.f) : inactiveValue<OP_RESIDUAL>(xs[e + 1]);
        res[e] = wz * r;
    }
    *reinterpret_cast<float4 *>(rz + size_t(K) * sz + size_t(j) * sy + i) = make_float4(res[0], res[1], res[2], res[3]);
}

// Scalar fallback for levels whose nx is not a multiple of 4 (only the tiniest coarse levels).
template <int OP, bool DOT = false>
__global__ void stencilScalarKernel(GridP g, float *__restrict__ out, const float *__restrict__ x,
                                    const float *__restrict__ b, float omega, double *__restrict__ dotPartials = nullptr)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const size_t c = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (c < n) {
        const unsigned l = g.lab[c];
        const float xc = x[c];
        if (!simpleCell(l)) out[c] = inactiveValue<OP>(xc);
        else {
            const size_t sy = size_t(g.nx), sz = size_t(g.nx) * g.ny;
            const float diag = simpleDiag(l);
            const float lap = diag * xc - (x[c - 1] + x[c + 1] + x[c - sy] + x[c + sy] + x[c - sz] + x[c + sz]);
            const float bc = OP == OP_APPLY ? 0.f : b[c];
            const float res = epilogueRcp<OP>(xc, bc, lap, simpleRcp(diag), omega);
            out[c] = res;
            acc = dotTerm<OP>(xc, bc, res);
        }
    }
    if (DOT) blockDotStore(acc, dotPartials, blockIdx.x);
}

// BOUNDARY cells of a full-domain sweep: one thread per list entry, out of place like the sweep.
template <int OP, bool DOT = false, class TX = float>
__global__ void boundaryOpKernel(GridP g, TX *__restrict__ out, const TX *__restrict__ x,
                                 const float *__restrict__ b, float omega, unsigned nblocks, double *__restrict__ dotPartials = nullptr,
                                 MixScale ms = MixScale{})
{
    constexpr bool kMixed = !std::is_same<TX, float>::value;
    const unsigned block = remapBlock(blockIdx.x, nblocks);
    const int t = int(block * blockDim.x + threadIdx.x);
    double acc = 0.0;
    if (t < g.nbnd) {
        const size_t c = size_t(g.bnd[t]);
        float lap, diag;
        boundaryRow(g, [&](size_t p) { return Cell<TX>::load1(x + p); }, t, c, lap, diag);
        const float xc = Cell<TX>::load1(x + c), bc = OP == OP_APPLY ? 0.f : b[c];
        float res;
        if (kMixed) {
            const float bm = mixRhsScale(ms);
            res = OP == OP_JACOBI ? xc + omega * ((bm * bc - lap) / diag) : OP == OP_RESIDUAL ? bm * bc - ms.c2 * lap : lap;
        } else
            res = epilogue<OP>(xc, bc, lap, diag, omega);
        Cell<TX>::store1(out + c, res);
        acc = dotTerm<OP>(xc, bc, kMixed ? __half2float(toHalfSat(res)) : res);
    }
    if (DOT) blockDotStore(acc, dotPartials, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Band Jacobi (Ops.h:524-619): phase 1 computes the damped update of every band cell into a list,
// phase 2 scatters it -- exactly the reference's two phases, so no band cell sees a neighbour
// updated in the same pass.
// ---------------------------------------------------------------------------------------------
__global__ void bandComputeKernel(GridP g, const float *__restrict__ x, const float *__restrict__ b,
                                  const int32_t *__restrict__ band, int nband, float *__restrict__ tmp, float omega,
                                  unsigned nblocks)
{
    const unsigned block = remapBlock(blockIdx.x, nblocks);
    const int t = int(block * blockDim.x + threadIdx.x);
    if (t >= nband) return;
    const size_t c = size_t(band[t]);
    const size_t sy = size_t(g.nx), sz = size_t(g.nx) * g.ny;
    const float xc = x[c];
    float lap, diag;
    if (t >= g.nbnd) {  // INTERIOR or simple BOUNDARY cell (the list holds the general cells first)
        diag = float(g.bandDiag[t]);
        lap = diag * xc - (x[c - 1] + x[c + 1] + x[c - sy] + x[c + sy] + x[c - sz] + x[c + sz]);
        tmp[t] = xc + omega * ((b[c] - lap) * simpleRcp(diag));  // Ops.h:596-599
    } else {
        boundaryRow(g, [&](size_t p) { return x[p]; }, t, c, lap, diag);
        tmp[t] = xc + omega * ((b[c] - lap) / diag);
    }
}
// DOT: the workgroup also leaves sum (new - old) * b over its cells: the correction that turns <x, b> taken before the
// band passes into <x, b> after them
template <bool DOT = false, class TX = float>
__global__ void bandScatterKernel(TX *__restrict__ x, const int32_t *__restrict__ band, int nband,
                                  const float *__restrict__ tmp, unsigned nblocks, const float *__restrict__ b = nullptr,
                                  double *__restrict__ dotPartials = nullptr)
{
    const int t = int(remapBlock(blockIdx.x, nblocks) * blockDim.x + threadIdx.x);
    double acc = 0.0;
    if (t < nband) {
        const int32_t c = band[t];
        const float v = tmp[t];
        if (DOT) {
            const float stored = std::is_same<TX, float>::value ? v : __half2float(toHalfSat(v));
            acc = (double(stored) - double(Cell<TX>::load1(x + c))) * double(b[c]);
        }
        Cell<TX>::store1(x + c, v);  // Ops.h:604-618
    }
    if (DOT) blockDotStore(acc, dotPartials, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// The fused band stage, box form (BandBoxes in mgps_internal.h; whole-grid levels).  A workgroup takes one group: the
// values of its region R (a box of the grid, <= kBoxMaxNodes cells) sit in a dense LDS block -- the six neighbours of
// region cell n are n +- 1, n +- rx, n +- rx*ry, no ids -- and the threads walk the group's list of the region cells that
// matter (4 B each, region order).  Pass p recomputes the band cells with ring <= H - p (redundant work near the rim
// instead of a round trip through HBM per pass), with the arithmetic of bandComputeKernel:
//   CLOSURE = false  H = depth band passes (Ops.h:524-619 x depth); the owned box's band cells go to `dst`.  Nothing else
//                    may read `dst` cells in this launch: either src is another grid (a snapshot, or the stage runs out
//                    of place into a scratch grid that bandBoxCopyKernel copies back), never src == dst;
//   CLOSURE = true   the same passes and then ONE damped Jacobi step (Ops.h:262-367, same omega, same formula) on the
//                    band cells of the owned box and their active face neighbours (class 11), H = depth + 1: what the
//                    full-domain sweep would write there had the band passes gone before it.  The sweep itself runs
//                    over the un-smoothed grid first; this launch overwrites its output on the band closure and leaves
//                    the same values in `snap`, from which the stage after the sweep reads (then dst = the sweep's
//                    output itself: written in place, no scatter anywhere).
// General band cells (operator rows): at most kBoxMaxGeneral per group, rows and rhs staged in LDS, one thread each
// (GEN: the level has any; levels without them keep 18 KB of LDS free).
// DOT: the workgroup leaves sum (new - old) * b over the cells it writes (old = dotOld at that cell).
// What was tried on the way (1024^3, us per closure / plain stage): one code byte per region cell with every thread walking
// the dense block 312 / 259; lists sorted by ring (a pass = a prefix) with separate read lists 684 / 386 (longer chain of
// dependent loads), merged 812 / 688 at 512 threads x 8 entries (31 registers spilled), 470 / 337 at 1024 x 4 (the staging
// loads of a wave hop between rows); this form 369 / 294 with a third of the first form's HBM traffic (rocprofv3 PMC at
// 512^3: 79 / 65 B per band cell against 178 / 92) -- what bounds it is the chain info -> list -> values -> passes with two
// workgroups per CU in flight, not bytes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool boxBand(unsigned cls) { return cls >= kBoxGeneral && cls <= kBoxSimple + 6; }

// Round 4, what was tried on this kernel and what stayed (LABNOTES.md R4 has the numbers): persistent workgroups with the next
// group's loads in flight (no change: not the load chain); a second, ring-sorted list of the cells that are ever updated so
// that a pass is a dense prefix (LDS instructions halved, +10 % fabric traffic for the second list, 7 % SLOWER per launch at
// 1024^3: dropped -- the stage follows its bytes, not its instructions); 24-bit address multiplies and the Morton launch
// order of the groups (orderBandBoxes: overlapping regions of neighbouring groups meet in the chiplet's L2): kept, 0.300 ->
// 0.277 ms per launch at 1024^3.
// XZERO: src is zero everywhere (see stencilQuadKernel): no value of it is loaded -- closure mode, and the plain mode of a
// Gauss-Seidel down-stroke, which then writes the iterate in place
// (the body: bandBoxKernel runs group remapBlock(blockIdx.x), strokeFrontKernel its first workgroups)
template <class TX, bool CLOSURE, bool DOT, bool GEN, bool XZERO>
// (dst and dotOld carry no __restrict__: the gathered dot of a Jacobi stroke reads the sweep's value of a cell through dotOld == dst
// right before the store that replaces it)
__device__ __forceinline__ void bandBoxBody(const GridP &g, const TX *__restrict__ src, const float *__restrict__ b, TX *dst,
                                            TX *__restrict__ snap, const int32_t *__restrict__ info, const uint32_t *__restrict__ list,
                                            const int32_t *__restrict__ general, float omega, int depth, const MixScale &ms,
                                            double *__restrict__ dotPartials, const TX *dotOld, int outClosure, unsigned group, unsigned slot)
{
    constexpr bool kMixed = !std::is_same<TX, float>::value;
    constexpr int kGenRows = GEN ? kBoxMaxGeneral : 1;
    __shared__ __attribute__((aligned(16))) float val[2][kBoxMaxNodes];
    __shared__ float grow[7][kGenRows];
    __shared__ float gbv[kGenRows];
    __shared__ uint16_t gnode[kGenRows], gring[kGenRows];  // region cell and ring
    // the group's description is wave-uniform: scalar registers (addresses below: scalar base + one 32-bit vector offset)
    const int32_t *gip = info + kBoxInfoInts * size_t(group);
    int gi[kBoxInfoInts];
#pragma unroll
    for (int q = 0; q < kBoxInfoInts; ++q) gi[q] = __builtin_amdgcn_readfirstlane(gip[q]);
    const int rx = gi[1] & 255, ry = (gi[1] >> 8) & 255, sxy = rx * ry;
    // a region cell's address = the region's origin + a 32-bit offset inside the region (it stays below 32 planes)
    const unsigned sy = unsigned(g.nx), sz = unsigned(g.nx) * unsigned(g.ny);
    const ptrdiff_t origin = gi[0];
    if (!XZERO) src += origin;
    b += origin;
    if (dst) dst += origin;
    if (CLOSURE && snap) snap += origin;
    if (DOT) dotOld += origin;
    const uint32_t *U = list + gi[2];
    // (lists as compactBandBoxLists leaves them: no entries for the inactive cells -- the whole block is cleared instead -- and
    // the cells only the closure mode reads at the end, past the info[3] entries the plain mode walks)
    const int ngen = GEN ? gi[5] : 0, nList = CLOSURE ? gi[7] : gi[3];
    const int H = depth + (CLOSURE ? 1 : 0);
    const float bm = kMixed ? mixRhsScale(ms) : 1.f;
    const int tid = threadIdx.x;
    // (24-bit multiplies: v_mul_u32_u24 / v_mad_u32_u24 run at the full rate, the 32-bit v_mul_lo_u32 / v_mad_u64_u32 the compiler
    // emits otherwise at a quarter of it; launchBandBox refuses levels whose planes do not fit 24 bits: boxPlaneFits)
    auto nodeOf = [&](uint32_t e) { return int(__umul24((e >> 10) & 31u, unsigned(sxy)) + __umul24((e >> 5) & 31u, unsigned(rx)) + (e & 31u)); };
    auto cellOf = [&](uint32_t e) { return (e & 31u) + __umul24((e >> 5) & 31u, sy) + __umul24((e >> 10) & 31u, sz); };
    // element c of a grid whose base is the region's origin: scalar base + 32-bit BYTE offset (the form the global_load /
    // global_store instructions take with one vector register)
    auto rd = [&](const TX *base, unsigned c) { return Cell<TX>::load1(reinterpret_cast<const TX *>(reinterpret_cast<const char *>(base) + c * unsigned(sizeof(TX)))); };
    auto rdf = [&](const float *base, unsigned c) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + c * 4u); };
    auto wr = [&](TX *base, unsigned c, float v) { Cell<TX>::store1(reinterpret_cast<TX *>(reinterpret_cast<char *>(base) + c * unsigned(sizeof(TX))), v); };
    uint32_t ue[kBoxSlots];
    float bv[kBoxSlots];
    {
        // first batch of loads: every list entry of this thread (and its general entry); second batch: every value
#pragma unroll
        for (int m = 0; m < kBoxSlots; ++m) {
            const int k = tid + m * kBoxThreads;
            ue[m] = k < nList ? U[k] : (uint32_t(kBoxSkip) << 16);
        }
        uint32_t ge = 0;  // this thread's general cell: its list entry and its row, in the first batch like the rest
        int32_t grw = 0;
        if (GEN && tid < ngen) {
            ge = uint32_t(general[2 * size_t(gi[4] + tid)]);
            grw = general[2 * size_t(gi[4] + tid) + 1];
        }
        // (unconditional loads -- a cell that needs none reads the region's origin cell and drops the value -- so that all of
        // them leave in one batch: a branch per slot made the compiler wait for each load on its own)
        float xv[kBoxSlots];
#pragma unroll
        for (int m = 0; m < kBoxSlots; ++m) {
            const unsigned cls = (ue[m] >> 16) & 15u;
            const bool need = cls != kBoxSkip && cls != kBoxZero && (CLOSURE || cls != kBoxFrozenFar);
            const bool bneed = (cls > kBoxSimple && cls <= kBoxSimple + 6 && int(ue[m] >> 20) <= H - 1) || (CLOSURE && cls == kBoxFrozenOut);
            const unsigned c = cellOf(ue[m]);
            xv[m] = XZERO ? 0.f : rd(src, need ? c : 0u);
            bv[m] = rdf(b, bneed ? c : 0u);
        }
#pragma unroll
        for (int m = 0; m < kBoxSlots; ++m) {
            const unsigned cls = (ue[m] >> 16) & 15u;
            if (cls == kBoxSkip || cls == kBoxZero) xv[m] = 0.f;  // (class 12 in the plain mode: whatever was loaded, nobody reads it)
            if (kMixed) bv[m] *= bm;
        }
        {  // inactive neighbours read 0: the region's block is cleared while the loads above are in flight
            const int rz = (gi[1] >> 16) & 255, ncell4 = (sxy * rz + 3) >> 2;
            float4 *z0 = reinterpret_cast<float4 *>(val[0]), *z1 = reinterpret_cast<float4 *>(val[1]);
            for (int q = tid; q < ncell4; q += kBoxThreads) {
                z0[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                z1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        if (GEN && tid < ngen) {
            const uint32_t e = ge;
            gring[tid] = uint16_t(e >> 20);
            gnode[tid] = uint16_t(nodeOf(e));
            const size_t nb = size_t(g.nbnd);
#pragma unroll
            for (int q = 0; q < 7; ++q) grow[q][tid] = g.rows[size_t(q) * nb + grw];
            const unsigned c = cellOf(e);
            gbv[tid] = kMixed ? bm * rdf(b, c) : rdf(b, c);
        }
#pragma unroll
        for (int m = 0; m < kBoxSlots; ++m)
            if (tid + m * kBoxThreads < nList) {
                const int n = nodeOf(ue[m]);
                val[0][n] = xv[m];
                val[1][n] = xv[m];
            }
    }
    __syncthreads();
    for (int p = 1; p <= H; ++p) {
        const float *from = val[(p - 1) & 1];
        float *to = val[p & 1];
        const int lim = H - p;  // the band cells with ring <= H - p
        const bool last = CLOSURE && p == H;
#pragma unroll
        for (int m = 0; m < kBoxSlots; ++m) {
            const unsigned cls = (ue[m] >> 16) & 15u;
            const bool simple = cls > kBoxSimple && cls <= kBoxSimple + 6;
            if ((simple && int(ue[m] >> 20) <= lim) || (last && cls == kBoxFrozenOut)) {
                const int n = nodeOf(ue[m]);
                const float xc = from[n];
                const float diag = cls == kBoxFrozenOut ? 6.f : float(int(cls) - int(kBoxSimple));
                const float lap = diag * xc - (from[n - 1] + from[n + 1] + from[n - rx] + from[n + rx] + from[n - sxy] + from[n + sxy]);
                to[n] = xc + omega * ((bv[m] - lap) * simpleRcp(diag));  // Ops.h:596-599 / 356-361
            }
        }
        if (GEN && tid < ngen && int(gring[tid]) <= lim) {
            const int nd = gnode[tid];
            const float xc = from[nd];
            float acc = 0.f;
            acc -= grow[0][tid] * from[nd - 1];
            acc -= grow[1][tid] * from[nd + 1];
            acc -= grow[2][tid] * from[nd - rx];
            acc -= grow[3][tid] * from[nd + rx];
            acc -= grow[4][tid] * from[nd - sxy];
            acc -= grow[5][tid] * from[nd + sxy];
            const float diag = grow[6][tid];
            const float lap = acc + diag * xc;
            to[nd] = xc + omega * ((gbv[tid] - lap) / diag);
        }
        __syncthreads();
    }
    const float *fin = val[H & 1];
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < kBoxSlots; ++m) {
        const unsigned cls = (ue[m] >> 16) & 15u;
        // plain mode with outClosure: the closure-output cells too -- their staged value (what the closure launch left in the
        // snapshot) goes to dst, which then needs nothing from the closure launch itself
        if ((ue[m] >> 20) == 0u && (boxBand(cls) || ((CLOSURE || outClosure) && cls == kBoxFrozenOut))) {
            const unsigned c = cellOf(ue[m]);
            const float v = fin[nodeOf(ue[m])];
            if (DOT) {
                const float stored = kMixed ? __half2float(toHalfSat(v)) : v;
                acc += (double(stored) - double(rd(dotOld, c))) * double(rdf(b, c));
            }
            if (dst) wr(dst, c, v);
            if (CLOSURE && snap) wr(snap, c, v);
        }
    }
    if (DOT) blockDotStore(acc, dotPartials, slot);
}
template <class TX, bool CLOSURE, bool DOT, bool GEN, bool XZERO = false>
__global__ __launch_bounds__(kBoxThreads, 8) void bandBoxKernel(GridP g, const TX *__restrict__ src, const float *__restrict__ b, TX *dst,
                                                              TX *__restrict__ snap, const int32_t *__restrict__ info, const uint32_t *__restrict__ list,
                                                              const int32_t *__restrict__ general, float omega, int depth, MixScale ms,
                                                              double *__restrict__ dotPartials, const TX *dotOld, int outClosure)
{
    bandBoxBody<TX, CLOSURE, DOT, GEN, XZERO>(g, src, b, dst, snap, info, list, general, omega, depth, ms, dotPartials, dotOld, outClosure,
                                              remapBlock(blockIdx.x, gridDim.x), blockIdx.x);
}

// The front of a smoothing stroke in ONE launch: the closure launch of the band boxes (workgroups [0, ngroups)) and the sweep
// (the rest; a workgroup = four 256-thread shares of stencilQuadKernel).  The two are independent once the sweep leaves the band
// closure alone -- the closure launch writes only the snapshot, and the plain launch that follows writes band and closure cells
// of the sweep's output whatever the sweep put there: `keep` (one bit per cell: owned band / closure-output cell of some box)
// masks the sweep's stores, so the plain launch no longer has to come after the sweep's stores to those cells.  For levels
// whose launches are latency chains (a closure launch of a 64^3 level is ~6 us whatever it does): one chain instead of two.
template <bool GEN, bool XZERO>
__global__ __launch_bounds__(kBoxThreads, 8) void strokeFrontKernel(GridP g, float *__restrict__ out, const float *__restrict__ x, const float *__restrict__ b,
                                                                  float *__restrict__ snap, const int32_t *__restrict__ info, const uint32_t *__restrict__ list,
                                                                  const int32_t *__restrict__ general, float omega, int depth, unsigned ngroups,
                                                                  unsigned nshares, const int32_t *__restrict__ chunks, const uint32_t *__restrict__ keep)
{
    if (blockIdx.x < ngroups) {
        bandBoxBody<float, true, false, GEN, XZERO>(g, x, b, nullptr, snap, info, list, general, omega, depth, MixScale{}, nullptr, nullptr, 0,
                                                    remapBlock(blockIdx.x, ngroups), 0u);
        return;
    }
    const unsigned share = (blockIdx.x - ngroups) * (kBoxThreads / 256) + (threadIdx.x >> 8);
    if (share < nshares) stencilQuadBody<OP_JACOBI, false, float, XZERO, true>(g, out, x, b, omega, nshares, chunks, nullptr, MixScale{}, share, threadIdx.x & 255u, keep);
}
// keep bits of a level: the owned band and closure-output cells of its boxes
__global__ __launch_bounds__(256) void markClosureKernel(GridP g, const int32_t *__restrict__ info, const uint32_t *__restrict__ list, uint32_t *__restrict__ bits)
{
    const int32_t *gi = info + kBoxInfoInts * size_t(blockIdx.x);
    const size_t sy = size_t(g.nx), sz = size_t(g.nx) * g.ny, origin = size_t(gi[0]);
    const uint32_t *U = list + gi[2];
    for (int k = threadIdx.x; k < gi[7]; k += 256) {
        const uint32_t e = U[k];
        const unsigned cls = (e >> 16) & 15u;
        if ((e >> 20) != 0u || !(boxBand(cls) || cls == kBoxFrozenOut)) continue;
        const size_t c = origin + (e & 31u) + ((e >> 5) & 31u) * sy + ((e >> 10) & 31u) * sz;
        atomicOr(bits + (c >> 5), 1u << (c & 31));
    }
}

// snapshot tiles of a level: the 16^3 tiles that hold a cell some group reads in the plain mode (launchMarkSnapTiles)
__global__ __launch_bounds__(256) void markSnapTilesKernel(GridP g, const int32_t *__restrict__ info, const uint32_t *__restrict__ list, uint8_t *__restrict__ tiles)
{
    const int32_t *gi = info + kBoxInfoInts * size_t(blockIdx.x);
    const int tx = (g.nx + kTile - 1) / kTile, ty = (g.ny + kTile - 1) / kTile;
    const size_t origin = size_t(gi[0]);
    const int ox = int(origin % size_t(g.nx)), oy = int((origin / size_t(g.nx)) % size_t(g.ny)), oz = int(origin / (size_t(g.nx) * g.ny));
    const uint32_t *U = list + gi[2];
    for (int k = threadIdx.x; k < gi[7]; k += 256) {
        const uint32_t e = U[k];
        const unsigned cls = (e >> 16) & 15u;
        if (cls == kBoxSkip || cls == kBoxZero) continue;  // (class 12 cells are not read by the plain mode; their tiles may be marked for nothing)
        const int i = ox + int(e & 31u), j = oy + int((e >> 5) & 31u), k3 = oz + int((e >> 10) & 31u);
        tiles[(size_t(k3 / kTile) * ty + j / kTile) * tx + i / kTile] = 1;
    }
}

// dst = src on the band cells of every owned box
template <class TX>
__global__ __launch_bounds__(kBoxThreads) void bandBoxCopyKernel(GridP g, const TX *__restrict__ src, TX *__restrict__ dst, const int32_t *__restrict__ info,
                                                                const uint32_t *__restrict__ list)
{
    const int32_t *gi = info + kBoxInfoInts * size_t(remapBlock(blockIdx.x, gridDim.x));
    const ptrdiff_t sy = g.nx, sz = ptrdiff_t(g.nx) * g.ny, origin = gi[0];
    const uint32_t *U = list + gi[2];
    for (int k = threadIdx.x; k < gi[7]; k += kBoxThreads) {
        const uint32_t e = U[k];
        if ((e >> 20) != 0u || !boxBand((e >> 16) & 15u)) continue;
        const ptrdiff_t c = origin + ptrdiff_t(e & 31u) + ptrdiff_t((e >> 5) & 31u) * sy + ptrdiff_t((e >> 10) & 31u) * sz;
        dst[c] = src[c];
    }
}

// ---------------------------------------------------------------------------------------------
// Tile-coloured Gauss-Seidel (Ops.h:369-520).  One 256-thread workgroup owns one 16^3 tile of the
// requested colour: the 18^3 halo cube of x and the 16^3 rhs are staged in LDS, then the tile is
// swept along anti-diagonal planes i+j+k = s.  For a 7-point stencil every cell of plane s depends
// only on planes s-1 (already updated) and s+1 (still old), so marching s upwards reproduces the
// reference's lexicographic forward sweep exactly and marching downwards its reversed sweep
// (Ops.h:497-516).  Face neighbours outside the tile belong to tiles of the other colour and are
// constant during the pass (Ops.h:436-448).
//   * "pure" tiles (all 4096 cells INTERIOR) need no labels and no weights: 39 KB of LDS;
//   * "mixed" tiles also stage the 18^3 labels and the operator rows of their BOUNDARY cells
//     (tile-major slice of the level's row list) in LDS; a BOUNDARY cell finds its row through a
//     per-x-row mask + prefix count built at load time.
// ---------------------------------------------------------------------------------------------
constexpr int kHalo = kTile + 2;
constexpr int kHalo3 = kHalo * kHalo * kHalo;
constexpr int kTile3 = kTile * kTile * kTile;
constexpr int kPlanes = 3 * (kTile - 1) + 1;
constexpr int kRowPool = 240;   // BOUNDARY rows of one tile kept in LDS (the rest is read from memory): 53.3 KB in all, three tiles per CU

__device__ __forceinline__ int haloIdx(int li, int lj, int lk) { return ((lk + 1) * kHalo + (lj + 1)) * kHalo + (li + 1); }

// stage the halo cube of x (and optionally the labels) and the rhs of tile (i0,j0,k0)
// TX = __half (options.precision = 1): the iterate lives in binary16 and is staged / swept in fp32; bm scales the rhs into the
// iterate's units (mixRhsScale; 1 for fp32 grids)
template <bool LABELS, class TX = float>
__device__ __forceinline__ void gsLoadTile(const GridP &g, const TX *__restrict__ x, const float *__restrict__ b,
                                           int i0, int j0, int k0, float *sx, float *sb, unsigned char *sl, float bm = 1.f)
{
    constexpr bool kMixed = !std::is_same<TX, float>::value;
    const bool full = i0 + kTile <= g.nx && j0 + kTile <= g.ny && k0 + kTile <= g.nz;
    if (full) {
        // 18 x 18 rows of the cube: the 16 interior floats of a row as 4 aligned float4, the two x-halo cells as
        // scalars; 16 x 16 rows of the rhs.  Every global load of the tile is issued before the first LDS write
        // (13 in flight per thread): with one load per thread in flight the pass ran at 2.2 TB/s of useful bytes, the
        // memory side of a tile was latency- rather than bandwidth-bound (tools/gsbench: 528 -> 451 us per 512^3 sweep)
        const int tid = threadIdx.x;  // blockDim.x == 256
        float4 xv[6], bv[4];
        uchar4 lv[6];
        float hv[3];
        unsigned char hl[3];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int r = tid + m * 256;
            const int q = r & 3, lj = (r >> 2) % kHalo, lk = (r >> 2) / kHalo;
            const int gj = j0 + lj - 1, gk = k0 + lk - 1;
            const bool in = r < kHalo * kHalo * 4 && gj >= 0 && gk >= -g.ghostLo && gj < g.ny && gk < g.nz + g.ghostHi;
            const ptrdiff_t rowOff = (ptrdiff_t(gk) * g.ny + gj) * g.nx + i0 + 4 * q;
            xv[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            lv[m] = make_uchar4(MGPS_EXTERIOR_CELL, MGPS_EXTERIOR_CELL, MGPS_EXTERIOR_CELL, MGPS_EXTERIOR_CELL);
            if (in) {
                xv[m] = Cell<TX>::load4(x + rowOff);
                // labels of the tile's own rows only: the sweep never looks at a halo cell's label (a 16-byte label row is
                // an eighth of a line: 68 halo rows would be 68 more lines per tile)
                if (LABELS && lj >= 1 && lj <= kTile && lk >= 1 && lk <= kTile) lv[m] = *reinterpret_cast<const uchar4 *>(g.lab + rowOff);
            }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int r = tid + m * 256;
            const int side = r & 1, lj = (r >> 1) % kHalo, lk = (r >> 1) / kHalo;
            const int gi = side ? i0 + kTile : i0 - 1, gj = j0 + lj - 1, gk = k0 + lk - 1;
            const bool in = r < kHalo * kHalo * 2 && gi >= 0 && gi < g.nx && gj >= 0 && gk >= -g.ghostLo && gj < g.ny && gk < g.nz + g.ghostHi;
            const ptrdiff_t c = (ptrdiff_t(gk) * g.ny + gj) * g.nx + gi;
            hv[m] = 0.f;
            hl[m] = (unsigned char)MGPS_EXTERIOR_CELL;  // (x-halo labels are never read either)
            if (in) hv[m] = Cell<TX>::load1(x + c);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = tid + m * 256;
            const int q = r & 3, lj = (r >> 2) % kTile, lk = (r >> 2) / kTile;
            bv[m] = g.streaming ? streamLoad4(b + (size_t(k0 + lk) * g.ny + j0 + lj) * g.nx + i0 + 4 * q)
                                : *reinterpret_cast<const float4 *>(b + (size_t(k0 + lk) * g.ny + j0 + lj) * g.nx + i0 + 4 * q);
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const int r = tid + m * 256;
            if (r < kHalo * kHalo * 4) {
                const int at = (r >> 2) * kHalo + 1 + 4 * (r & 3);
                sx[at] = xv[m].x;
                sx[at + 1] = xv[m].y;
                sx[at + 2] = xv[m].z;
                sx[at + 3] = xv[m].w;
                if (LABELS) {
                    sl[at] = lv[m].x;
                    sl[at + 1] = lv[m].y;
                    sl[at + 2] = lv[m].z;
                    sl[at + 3] = lv[m].w;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int r = tid + m * 256;
            if (r < kHalo * kHalo * 2) {
                const int h = (r >> 1) * kHalo + ((r & 1) ? kHalo - 1 : 0);
                sx[h] = hv[m];
                if (LABELS) sl[h] = hl[m];
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float *dst = sb + (tid + m * 256) * 4;
            dst[0] = kMixed ? bm * bv[m].x : bv[m].x;
            dst[1] = kMixed ? bm * bv[m].y : bv[m].y;
            dst[2] = kMixed ? bm * bv[m].z : bv[m].z;
            dst[3] = kMixed ? bm * bv[m].w : bv[m].w;
        }
    } else {  // ragged tile at the end of a grid whose extent is not a multiple of 16
        for (int h = threadIdx.x; h < kHalo3; h += blockDim.x) {
            const int li = h % kHalo, lj = (h / kHalo) % kHalo, lk = h / (kHalo * kHalo);
            const int gi = i0 + li - 1, gj = j0 + lj - 1, gk = k0 + lk - 1;
            const bool in = gi >= 0 && gj >= 0 && gk >= -g.ghostLo && gi < g.nx && gj < g.ny && gk < g.nz + g.ghostHi;
            const ptrdiff_t c = in ? (ptrdiff_t(gk) * g.ny + gj) * g.nx + gi : 0;
            sx[h] = in ? Cell<TX>::load1(x + c) : 0.f;
            if (LABELS) sl[h] = in ? g.lab[c] : (unsigned char)MGPS_EXTERIOR_CELL;
        }
        for (int h = threadIdx.x; h < kTile3; h += blockDim.x) {
            const int li = h % kTile, lj = (h / kTile) % kTile, lk = h / (kTile * kTile);
            const int gi = i0 + li, gj = j0 + lj, gk = k0 + lk;
            const bool in = gi < g.nx && gj < g.ny && gk < g.nz;
            sb[h] = in ? (kMixed ? bm * b[(size_t(gk) * g.ny + gj) * g.nx + gi] : b[(size_t(gk) * g.ny + gj) * g.nx + gi]) : 0.f;
        }
    }
}

// one pure tile by one workgroup; slot: where its <x, b> goes (DOT)
// snap / snapTile (optional, here and in the mixed tile): a tile whose byte is set in snapTile (launchMarkSnapTiles: some box group
// of the fused band stage reads a cell of it) leaves a second copy of its result in `snap` -- the snapshot from which the band
// stage after the sweep reads while it writes x in place (smoothStroke, Gauss-Seidel strokes)
template <bool DOT, class TX = float>
__device__ __forceinline__ void gsPureTile(const GridP &g, TX *__restrict__ x, const float *__restrict__ b, int tile, int forward,
                                           double *__restrict__ dotPartials, unsigned slot, float *sx, float *sb, float bm = 1.f,
                                           TX *__restrict__ snap = nullptr, const uint8_t *__restrict__ snapTile = nullptr)
{
    const int tilesX = (g.nx + kTile - 1) / kTile, tilesY = (g.ny + kTile - 1) / kTile;
    const int i0 = (tile % tilesX) * kTile, j0 = ((tile / tilesX) % tilesY) * kTile, k0 = (tile / (tilesX * tilesY)) * kTile;
    gsLoadTile<false, TX>(g, x, b, i0, j0, k0, sx, sb, nullptr, bm);
    __syncthreads();
    const int li = threadIdx.x % kTile, lj = threadIdx.x / kTile;  // this thread's (i, j) column
    for (int step = 0; step < kPlanes; ++step) {
        const int s = forward ? step : kPlanes - 1 - step;
        const int lk = s - li - lj;
        if (lk >= 0 && lk < kTile) {
            const int h = haloIdx(li, lj, lk);
            const float xc = sx[h];
            const float lap = 6.f * xc - (sx[h - 1] + sx[h + 1] + sx[h - kHalo] + sx[h + kHalo] + sx[h - kHalo * kHalo] +
                                          sx[h + kHalo * kHalo]);
            sx[h] = xc + (sb[(lk * kTile + lj) * kTile + li] - lap) * (1.f / 6.f);  // undamped, Ops.h:493
        }
        __syncthreads();
    }
    double acc = 0.0;
    const bool snapOn = snap && snapTile[tile];
    for (int r = threadIdx.x; r < kTile * kTile * 4; r += blockDim.x) {
        const int q = r & 3, cj = (r >> 2) % kTile, ck = (r >> 2) / kTile;
        const float *src = sx + haloIdx(4 * q, cj, ck);
        const size_t at = (size_t(k0 + ck) * g.ny + j0 + cj) * g.nx + i0 + 4 * q;
        Cell<TX>::store4(x + at, make_float4(src[0], src[1], src[2], src[3]));
        if (snapOn) Cell<TX>::store4(snap + at, make_float4(src[0], src[1], src[2], src[3]));
        if (DOT) {
            const float *bq = sb + (ck * kTile + cj) * kTile + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += double(src[e]) * double(bq[e]);
        }
    }
    if (DOT) blockDotStore(acc, dotPartials, slot);
}
template <bool DOT = false, class TX = float>  // DOT: the workgroup also leaves <x, b> over its tile (see dotTerm)
__global__ __launch_bounds__(256) void tiledGSPureKernel(GridP g, TX *__restrict__ x, const float *__restrict__ b,
                                                         const int32_t *__restrict__ tiles, int forward,
                                                         double *__restrict__ dotPartials = nullptr, MixScale ms = MixScale{},
                                                         TX *__restrict__ snap = nullptr, const uint8_t *__restrict__ snapTile = nullptr)
{
    __shared__ float sx[kHalo3];
    __shared__ float sb[kTile3];
    // (XCD-aware: a chiplet takes a contiguous run of the colour's tile list, so the lines a tile shares with the next tile of its
    // colour -- its x-halo cell sits in that tile's own 128-byte line, its y / z halo rows in lines the tiles around it read too --
    // meet in one L2 instead of being fetched by two)
    const unsigned bid = remapBlock(blockIdx.x, gridDim.x);
    gsPureTile<DOT, TX>(g, x, b, tiles[bid], forward, dotPartials, bid, sx, sb, std::is_same<TX, float>::value ? 1.f : mixRhsScale(ms), snap, snapTile);
}

// one mixed tile by one workgroup (rowMask: bit i set = cell (i, j, k) of x-row (j, k) is BOUNDARY; rowStart: BOUNDARY cells of
// the tile before that x-row)
template <bool DOT, class TX = float>
__device__ __forceinline__ void gsMixedTile(const GridP &g, TX *__restrict__ x, const float *__restrict__ b, int tile,
                                            const int32_t *__restrict__ tileBndStart, int forward, double *__restrict__ dotPartials,
                                            unsigned slot, float *sx, float *sb, unsigned char *sl, float *srow, unsigned short *rowMask,
                                            unsigned short *rowStart, int *scanTmp, float bm = 1.f, TX *__restrict__ snap = nullptr,
                                            const uint8_t *__restrict__ snapTile = nullptr)
{
    const int tilesX = (g.nx + kTile - 1) / kTile, tilesY = (g.ny + kTile - 1) / kTile;
    const int i0 = (tile % tilesX) * kTile, j0 = ((tile / tilesX) % tilesY) * kTile, k0 = (tile / (tilesX * tilesY)) * kTile;
    const int bndBase = tileBndStart[tile], bndCount = tileBndStart[tile + 1] - bndBase;
    const size_t nb = size_t(g.nbnd);
    gsLoadTile<true, TX>(g, x, b, i0, j0, k0, sx, sb, sl, bm);
    for (int r = threadIdx.x; r < 7 * min(bndCount, kRowPool); r += blockDim.x) {
        const int q = r / min(bndCount, kRowPool), t = r % min(bndCount, kRowPool);
        srow[q * kRowPool + t] = g.rows[q * nb + bndBase + t];
    }
    __syncthreads();
    if (bndCount > 0) {  // per x-row BOUNDARY mask and exclusive prefix over the 256 rows in (k, j) order = list order (uniform branch)
        const int rj = threadIdx.x % kTile, rk = threadIdx.x / kTile;
        unsigned m = 0;
        for (int i = 0; i < kTile; ++i) m |= (sl[haloIdx(i, rj, rk)] == kCodeGeneral) ? (1u << i) : 0u;
        rowMask[threadIdx.x] = (unsigned short)m;
        int cnt = __popc(m), incl = cnt;
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == kWave - 1) scanTmp[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += scanTmp[w];
        rowStart[threadIdx.x] = (unsigned short)(base + incl - cnt);
    }
    __syncthreads();

    const int li = threadIdx.x % kTile, lj = threadIdx.x / kTile;
    for (int step = 0; step < kPlanes; ++step) {
        const int s = forward ? step : kPlanes - 1 - step;
        const int lk = s - li - lj;
        if (lk >= 0 && lk < kTile) {
            const int h = haloIdx(li, lj, lk);
            // label, rhs and the seven values are read together (one LDS round trip per step, not label -> branch -> values:
            // a step is a barrier-to-barrier latency chain); inactive cells read them for nothing
            const unsigned l = sl[h];
            const float xc = sx[h];
            const float bc = sb[(lk * kTile + lj) * kTile + li];
            const float xn[6] = {sx[h - 1], sx[h + 1], sx[h - kHalo], sx[h + kHalo], sx[h - kHalo * kHalo], sx[h + kHalo * kHalo]};
            if (simpleCell(l)) {
                const float diag = simpleDiag(l);
                const float lap = diag * xc - (xn[0] + xn[1] + xn[2] + xn[3] + xn[4] + xn[5]);
                sx[h] = xc + (bc - lap) * simpleRcp(diag);  // undamped, Ops.h:493 (reciprocal as in the Jacobi kernels)
            } else if (l == kCodeGeneral) {
                const int row = lk * kTile + lj;
                const int t = int(rowStart[row]) + __popc(unsigned(rowMask[row]) & ((1u << li) - 1u));
                float w[7];
                if (t < kRowPool) {
#pragma unroll
                    for (int q = 0; q < 7; ++q) w[q] = srow[q * kRowPool + t];
                } else {
#pragma unroll
                    for (int q = 0; q < 7; ++q) w[q] = g.rows[q * nb + bndBase + t];
                }
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < 6; ++q) acc -= w[q] * xn[q];
                const float diag = w[6];
                const float lap = acc + diag * xc;
                sx[h] = xc + (bc - lap) / diag;
            }
        }
        __syncthreads();
    }
    // whole quads where the grid has them (inactive cells still hold the value they were loaded with: exactly 0)
    double acc = 0.0;
    const bool snapOn = snap && snapTile[tile];
    for (int r = threadIdx.x; r < kTile * kTile * 4; r += blockDim.x) {
        const int q = r & 3, cj = (r >> 2) % kTile, ck = (r >> 2) / kTile;
        const int gi = i0 + 4 * q, gj = j0 + cj, gk = k0 + ck;
        if (gj >= g.ny || gk >= g.nz || gi >= g.nx) continue;
        const float *src = sx + haloIdx(4 * q, cj, ck);
        const size_t at = (size_t(gk) * g.ny + gj) * g.nx + gi;
        TX *dst = x + at;
        if (gi + 3 < g.nx && (g.nx & 3) == 0) {
            Cell<TX>::store4(dst, make_float4(src[0], src[1], src[2], src[3]));
            if (snapOn) Cell<TX>::store4(snap + at, make_float4(src[0], src[1], src[2], src[3]));
        } else
            for (int e = 0; e < 4 && gi + e < g.nx; ++e) {
                Cell<TX>::store1(dst + e, src[e]);
                if (snapOn) Cell<TX>::store1(snap + at + e, src[e]);
            }
        if (DOT) {
            const float *bq = sb + (ck * kTile + cj) * kTile + 4 * q;
            const unsigned char *lq = sl + haloIdx(4 * q, cj, ck);
            for (int e = 0; e < 4 && gi + e < g.nx; ++e)
                if (activeLabel(lq[e])) acc += double(src[e]) * double(bq[e]);
        }
    }
    if (DOT) blockDotStore(acc, dotPartials, slot);
}
template <bool DOT = false, class TX = float>
__global__ __launch_bounds__(256) void tiledGSMixedKernel(GridP g, TX *__restrict__ x, const float *__restrict__ b,
                                                          const int32_t *__restrict__ tiles,
                                                          const int32_t *__restrict__ tileBndStart, int forward,
                                                          double *__restrict__ dotPartials = nullptr, MixScale ms = MixScale{},
                                                          TX *__restrict__ snap = nullptr, const uint8_t *__restrict__ snapTile = nullptr)
{
    __shared__ float sx[kHalo3];
    __shared__ float sb[kTile3];
    __shared__ unsigned char sl[kHalo3];
    __shared__ float srow[7 * kRowPool];
    __shared__ unsigned short rowMask[kTile * kTile];
    __shared__ unsigned short rowStart[kTile * kTile];
    __shared__ int scanTmp[4];
    const unsigned bid = remapBlock(blockIdx.x, gridDim.x);  // (see tiledGSPureKernel)
    gsMixedTile<DOT, TX>(g, x, b, tiles[bid], tileBndStart, forward, dotPartials, bid, sx, sb, sl, srow, rowMask, rowStart, scanTmp,
                         std::is_same<TX, float>::value ? 1.f : mixRhsScale(ms), snap, snapTile);
}
// Both lists of a colour in one launch: workgroups [0, nmixed) take the mixed tiles, the rest the pure ones.  For the small
// levels, where a launch is a handful of workgroups and each tile a chain of 46 barrier steps (~10 us): the two launches of
// a colour pass then cost two such chains back to back, this one costs one.  (Every workgroup reserves the mixed tile's 53 KB
// of LDS, so large levels -- where three instead of four pure tiles per CU would cost 7 % -- keep the two launches.)
template <bool DOT = false>
__global__ __launch_bounds__(256) void tiledGSBothKernel(GridP g, float *__restrict__ x, const float *__restrict__ b,
                                                         const int32_t *__restrict__ mixedTiles, int nmixed,
                                                         const int32_t *__restrict__ pureTiles, const int32_t *__restrict__ tileBndStart,
                                                         int forward, double *__restrict__ dotPartials = nullptr, float *__restrict__ snap = nullptr,
                                                         const uint8_t *__restrict__ snapTile = nullptr)
{
    __shared__ float sx[kHalo3];
    __shared__ float sb[kTile3];
    __shared__ unsigned char sl[kHalo3];
    __shared__ float srow[7 * kRowPool];
    __shared__ unsigned short rowMask[kTile * kTile];
    __shared__ unsigned short rowStart[kTile * kTile];
    __shared__ int scanTmp[4];
    if (int(blockIdx.x) < nmixed)
        gsMixedTile<DOT>(g, x, b, mixedTiles[blockIdx.x], tileBndStart, forward, dotPartials, blockIdx.x, sx, sb, sl, srow, rowMask, rowStart, scanTmp, 1.f, snap, snapTile);
    else
        gsPureTile<DOT>(g, x, b, pureTiles[int(blockIdx.x) - nmixed], forward, dotPartials, blockIdx.x, sx, sb, 1.f, snap, snapTile);
}

// ---------------------------------------------------------------------------------------------
// Full-weighting restriction (Ops.h:734-835): coarse C = sum over the 4x4x4 fine block starting at
// 2C-1 with weights {1/8,3/8,3/8,1/8}^3; inactive coarse cells are 0 (destination cleared first,
// Ops.h:756).  One thread per coarse cell.
// ---------------------------------------------------------------------------------------------
template <class TF = float>
__global__ void restrictKernel(GridP cg, float *__restrict__ coarse, const TF *__restrict__ fine, float fm = 1.f)
{
    constexpr bool kMixed = !std::is_same<TF, float>::value;
    const size_t n = size_t(cg.nx) * cg.ny * cg.nz;
    // with a chunk list: workgroups of 256 over the active chunks (four per 1024-cell chunk); the rest of `coarse` stays 0
    const unsigned bid = remapBlock(blockIdx.x, gridDim.x);  // a chiplet's L2 serves a contiguous run of the list
    size_t c = size_t(bid) * blockDim.x + threadIdx.x;
    if (cg.chunks) {  // 256 threads = a quarter of a 1024-cell run, a 256-cell run, or four 64-cell / eight 32-cell runs
        int ch;
        if (cg.chunkCells >= 256) {
            const int per = cg.chunkCells / 256;
            ch = cg.chunks[bid / per];
            c = size_t(max(ch, 0)) * cg.chunkCells + (bid % per) * 256 + threadIdx.x;
        } else {
            const int shift = __ffs(cg.chunkCells) - 1;
            ch = cg.chunks[(size_t(bid) << (8 - shift)) + (threadIdx.x >> shift)];
            c = size_t(max(ch, 0)) * cg.chunkCells + (threadIdx.x & (cg.chunkCells - 1));
        }
        if (ch < 0) return;
    }
    if (c >= n) return;
    if (!activeLabel(cg.lab[c])) {
        coarse[c] = 0.f;
        return;
    }
    const int i = int(c % cg.nx), j = int((c / cg.nx) % cg.ny), k = int(c / (size_t(cg.nx) * cg.ny));
    const int fnx = 2 * cg.nx, fny = 2 * cg.ny;
    const float w[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    float acc = 0.f;
#pragma unroll
    for (int zo = 0; zo < 4; ++zo)
#pragma unroll
        for (int yo = 0; yo < 4; ++yo) {
            const TF *row = fine + (ptrdiff_t(2 * k - 1 + zo) * fny + (2 * j - 1 + yo)) * fnx + (2 * i - 1);
            const float wyz = w[yo] * w[zo];
#pragma unroll
            for (int xo = 0; xo < 4; ++xo) acc += (w[xo] * wyz) * Cell<TF>::load1(row + xo);
        }
    coarse[c] = kMixed ? fm * acc : acc;
}

// The same operator marching along z: one thread owns a coarse (I, J) column of kc coarse planes and keeps the
// in-plane 4 x 4 weighted sums of the last fine planes in registers -- every fine plane's sum is formed once and
// feeds the two coarse planes it belongs to, where the per-cell kernel above forms it twice (and its z-overlap
// reads miss the L2 on large planes: 1.6x the algorithmic HBM traffic).  x, then y, then z summation.
// (Round 3, tried and dropped: one 8-byte load per row and lane -- the fine pair (2 I, 2 I + 1) -- with 2 I - 1 / 2 I + 2 taken
// from the neighbour lanes instead of four 4-byte loads at an 8-byte lane stride: 78 registers and twelve ds_bpermute per fine
// plane; 1024^3 cycle 10.44 -> 11.25 ms, 512^3 1.65 -> 1.77 ms.)
template <class TF = float>
__global__ __launch_bounds__(256) void restrictMarchKernel(GridP cg, float *__restrict__ coarse, const TF *__restrict__ fine,
                                                           int kc, unsigned nbx, unsigned nby, float fm = 1.f)
{
    constexpr bool kMixed = !std::is_same<TF, float>::value;
    // a thread owns the coarse columns (I, J) and (I, J + 1), J even: their fine footprints share two of six rows
    const unsigned bid = remapBlock(blockIdx.x, gridDim.x);
    const unsigned bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int I = int(bx) * 64 + int(threadIdx.x & 63), J = 2 * (int(by) * 4 + int(threadIdx.x >> 6));
    const int K0 = int(bz) * kc, K1 = min(K0 + kc, cg.nz);
    if (I >= cg.nx || J >= cg.ny || I < cg.xlo || I >= cg.xhi) return;  // (columns outside the coarse level's active x range hold no active cell)
    const bool second = J + 1 < cg.ny;
    const size_t cplane = size_t(cg.nx) * cg.ny, col = size_t(J) * cg.nx + I;
    bool any = false;
    for (int K = K0; K < K1; ++K) {
        any = any || activeLabel(cg.lab[size_t(K) * cplane + col]);
        if (second) any = any || activeLabel(cg.lab[size_t(K) * cplane + col + cg.nx]);
    }
    if (!any) return;  // (columns made of EXTERIOR shell cells only leave here)
    const int fnx = 2 * cg.nx, fny = 2 * cg.ny;
    const int kLo = cg.ghostLo ? -1 : 0, kHi = cg.ghostHi ? 2 * cg.nz : 2 * cg.nz - 1;
    const float w[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    // rows / planes / the column left of the grid only feed EXTERIOR coarse cells: clamp, their results are masked
    const int xBase = max(2 * I - 1, 0);
    struct Pair {
        float a, b;
    };
    auto planeSums = [&](int fk) {
        const TF *p = fine + ptrdiff_t(min(max(fk, kLo), kHi)) * fny * fnx + xBase;
        float rs[6];
#pragma unroll
        for (int yo = 0; yo < 6; ++yo) {
            const TF *r = p + ptrdiff_t(min(max(2 * J - 1 + yo, 0), fny - 1)) * fnx;
            rs[yo] = w[0] * Cell<TF>::load1(r) + w[1] * Cell<TF>::load1(r + 1) + w[2] * Cell<TF>::load1(r + 2) + w[3] * Cell<TF>::load1(r + 3);
        }
        return Pair{w[0] * rs[0] + w[1] * rs[1] + w[2] * rs[2] + w[3] * rs[3], w[0] * rs[2] + w[1] * rs[3] + w[2] * rs[4] + w[3] * rs[5]};
    };
    Pair p0 = planeSums(2 * K0 - 1), p1 = planeSums(2 * K0);
    for (int K = K0; K < K1; ++K) {
        const Pair p2 = planeSums(2 * K + 1), p3 = planeSums(2 * K + 2);
        const size_t c = size_t(K) * cplane + col;
        const float va = w[0] * p0.a + w[1] * p1.a + w[2] * p2.a + w[3] * p3.a, vb = w[0] * p0.b + w[1] * p1.b + w[2] * p2.b + w[3] * p3.b;
        coarse[c] = activeLabel(cg.lab[c]) ? (kMixed ? fm * va : va) : 0.f;
        if (second) coarse[c + cg.nx] = activeLabel(cg.lab[c + cg.nx]) ? (kMixed ? fm * vb : vb) : 0.f;
        p0 = p2;
        p1 = p3;
    }
}

// The marching restriction with every fine row staged ONCE per workgroup (round 4).  rocprofv3 on the kernel above, 1024^3 -> 512^3:
// 5.2e7 L1 -> L2 line requests and 3.5e7 L2 -> fabric requests for 2.25e7 lines of data (FETCH 4.4 GB = 1.5 x) -- a wave's four
// 4-byte loads at an 8-byte lane stride touch six lines for four lines of data (2 I - 1 and 2 I + 2 sit in the neighbours' lines),
// the four waves of a workgroup re-request the two rows they share, and the L2 catches a third of it.  Here a workgroup of
// 64 x 4 threads (coarse I x coarse row pairs, as above) loads the 18 fine rows x 136 columns of a fine plane as aligned 16-byte
// quads (three per thread), leaves them in LDS, and every thread forms its two in-plane 4 x 4 sums from there; the next plane's
// quads are in flight while the current plane is summed (two LDS buffers, one barrier per fine plane).  Same summation order
// as above: x, then y, then z.  Coarse nx even (fine nx % 4 == 0).
constexpr int kRtI = 64, kRtJ = 8;                       // coarse tile
constexpr int kRtRows = 2 * kRtJ + 2, kRtQuads = 2 * kRtI / 4 + 2;  // 18 fine rows, 34 quads: fine x from 2 I0 - 4 to 2 I0 + 131
constexpr int kRtStride = 4 * kRtQuads;
constexpr int kRtLoads = (kRtRows * kRtQuads + 255) / 256;  // 3
template <class TF = float>
__global__ __launch_bounds__(256) void restrictTileKernel(GridP cg, float *__restrict__ coarse, const TF *__restrict__ fine,
                                                          int kc, unsigned nbx, unsigned nby, float fm = 1.f)
{
    constexpr bool kMixed = !std::is_same<TF, float>::value;
    __shared__ float tile[2][kRtRows * kRtStride];
    __shared__ int anyActiveCol;
    const unsigned bid = remapBlock(blockIdx.x, gridDim.x);
    const unsigned bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int I0 = int(bx) * kRtI, J0 = int(by) * kRtJ;
    const int li = int(threadIdx.x & 63), lp = int(threadIdx.x >> 6);
    const int I = I0 + li, J = J0 + 2 * lp;
    const int K0 = int(bz) * kc, K1 = min(K0 + kc, cg.nz);
    const size_t cplane = size_t(cg.nx) * cg.ny, col = size_t(J) * cg.nx + I;
    const bool inGrid = I < cg.nx && J < cg.ny && I >= cg.xlo && I < cg.xhi;
    const bool second = J + 1 < cg.ny;
    bool any = false;
    if (inGrid)
        for (int K = K0; K < K1; ++K) {
            any = any || activeLabel(cg.lab[size_t(K) * cplane + col]);
            if (second) any = any || activeLabel(cg.lab[size_t(K) * cplane + col + cg.nx]);
        }
    if (threadIdx.x == 0) anyActiveCol = 0;
    __syncthreads();
    if (any) anyActiveCol = 1;
    __syncthreads();
    if (!anyActiveCol) return;  // (a tile of EXTERIOR / DIRICHLET columns: the destination holds 0 there already)
    const int fnx = 2 * cg.nx, fny = 2 * cg.ny;
    const int kLo = cg.ghostLo ? -1 : 0, kHi = cg.ghostHi ? 2 * cg.nz : 2 * cg.nz - 1;
    // the quads this thread stages of every fine plane: (row r, quad q) of the tile, fine x = 2 I0 - 4 + 4 q, fine y = 2 J0 - 1 + r.
    // Quads outside the grid or outside the fine image of the active x range are not loaded (they feed EXTERIOR cells only)
    const int fxlo = max(2 * cg.xlo - 4, 0), fxhi = min(2 * cg.xhi + 4, fnx);
    ptrdiff_t off[kRtLoads];
    bool ok[kRtLoads];
#pragma unroll
    for (int m = 0; m < kRtLoads; ++m) {
        const int t = int(threadIdx.x) + m * 256;
        const int r = t / kRtQuads, q = t % kRtQuads;
        const int fx = 2 * I0 - 4 + 4 * q, fy = min(max(2 * J0 - 1 + r, 0), fny - 1);  // (rows clamped: results masked)
        ok[m] = t < kRtRows * kRtQuads && fx >= fxlo && fx + 3 < fxhi;
        off[m] = ptrdiff_t(fy) * fnx + fx;
    }
    auto loadPlane = [&](int fk, float4 (&v)[kRtLoads]) {
        const TF *p = fine + ptrdiff_t(min(max(fk, kLo), kHi)) * fny * fnx;
#pragma unroll
        for (int m = 0; m < kRtLoads; ++m) v[m] = ok[m] ? Cell<TF>::load4(p + off[m]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto storePlane = [&](float *buf, const float4 (&v)[kRtLoads]) {
#pragma unroll
        for (int m = 0; m < kRtLoads; ++m) {
            const int t = int(threadIdx.x) + m * 256;
            if (t < kRtRows * kRtQuads) *reinterpret_cast<float4 *>(buf + (t / kRtQuads) * kRtStride + 4 * (t % kRtQuads)) = v[m];
        }
    };
    const float w[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    struct Pair {
        float a, b;
    };
    // in-plane sums of this thread's two coarse columns from the staged plane: fine x 2 I - 1 .. 2 I + 2 = tile column 2 li + 3 ..
    auto planeSums = [&](const float *buf) {
        const float *p = buf + (4 * lp) * kRtStride + 2 * li + 3;
        float rs[6];
#pragma unroll
        for (int yo = 0; yo < 6; ++yo) {
            const float *r = p + yo * kRtStride;
            const float2 mid = *reinterpret_cast<const float2 *>(r + 1);  // (column 2 li + 4: 8-byte aligned)
            rs[yo] = w[0] * r[0] + w[1] * mid.x + w[2] * mid.y + w[3] * r[3];
        }
        return Pair{w[0] * rs[0] + w[1] * rs[1] + w[2] * rs[2] + w[3] * rs[3], w[0] * rs[2] + w[1] * rs[3] + w[2] * rs[4] + w[3] * rs[5]};
    };
    // fine planes 2 K0 - 1 .. 2 K1: plane f sits in buffer f & 1 while it is summed, plane f + 1 is in flight meanwhile
    float4 nextv[kRtLoads];
    const int f0 = 2 * K0 - 1, f1 = 2 * (K1 - 1) + 2;
    loadPlane(f0, nextv);
    storePlane(tile[f0 & 1], nextv);
    loadPlane(f0 + 1, nextv);
    __syncthreads();
    Pair p0 = planeSums(tile[f0 & 1]), p1{0.f, 0.f}, p2{0.f, 0.f};
    int have = 1;  // sums held: p0 (and p1, p2 as they come)
    for (int f = f0 + 1; f <= f1; ++f) {
        storePlane(tile[f & 1], nextv);  // (the buffer of plane f - 2: everybody finished with it before the last barrier)
        if (f < f1) loadPlane(f + 1, nextv);
        __syncthreads();
        const Pair s = planeSums(tile[f & 1]);
        if (have == 1) p1 = s, have = 2;
        else if (have == 2) p2 = s, have = 3;
        else {  // s is the fourth plane of coarse plane K = (f - 2) / 2
            const int K = (f - 2) >> 1;
            if (inGrid && any) {
                const size_t c = size_t(K) * cplane + col;
                const float va = w[0] * p0.a + w[1] * p1.a + w[2] * p2.a + w[3] * s.a, vb = w[0] * p0.b + w[1] * p1.b + w[2] * p2.b + w[3] * s.b;
                coarse[c] = activeLabel(cg.lab[c]) ? (kMixed ? fm * va : va) : 0.f;
                if (second) coarse[c + cg.nx] = activeLabel(cg.lab[c + cg.nx]) ? (kMixed ? fm * vb : vb) : 0.f;
            }
            p0 = p2;
            p1 = s;
            have = 2;
        }
    }
}

// The x-y half of the restriction for a residual that residualZKernel folded along z already: `rz` has the fine level's x-y
// extents and the coarse level's planes; coarse(I, J, K) = sum_b w_b sum_a w_a rz(2I-1+a, 2J-1+b, K), x first, then y.  The
// staging of restrictTileKernel (18 rows x 34 aligned quads per plane, two LDS buffers, one barrier per plane), one plane in,
// one plane out: no overlap along z.
__global__ __launch_bounds__(256) void restrictXYKernel(GridP cg, float *__restrict__ coarse, const float *__restrict__ rz, int kc, unsigned nbx, unsigned nby)
{
    __shared__ float tile[2][kRtRows * kRtStride];
    __shared__ int anyActiveCol;
    const unsigned bid = remapBlock(blockIdx.x, gridDim.x);
    const unsigned bx = bid % nbx, by = (bid / nbx) % nby, bz = bid / (nbx * nby);
    const int I0 = int(bx) * kRtI, J0 = int(by) * kRtJ;
    const int li = int(threadIdx.x & 63), lp = int(threadIdx.x >> 6);
    const int I = I0 + li, J = J0 + 2 * lp;
    const int K0 = int(bz) * kc, K1 = min(K0 + kc, cg.nz);
    const size_t cplane = size_t(cg.nx) * cg.ny, col = size_t(J) * cg.nx + I;
    const bool inGrid = I < cg.nx && J < cg.ny && I >= cg.xlo && I < cg.xhi;
    const bool second = J + 1 < cg.ny;
    bool any = false;
    if (inGrid)
        for (int K = K0; K < K1; ++K) {
            any = any || activeLabel(cg.lab[size_t(K) * cplane + col]);
            if (second) any = any || activeLabel(cg.lab[size_t(K) * cplane + col + cg.nx]);
        }
    if (threadIdx.x == 0) anyActiveCol = 0;
    __syncthreads();
    if (any) anyActiveCol = 1;
    __syncthreads();
    if (!anyActiveCol) return;  // (the destination holds 0 there already)
    const int fnx = 2 * cg.nx, fny = 2 * cg.ny;
    const int fxlo = max(2 * cg.xlo - 4, 0), fxhi = min(2 * cg.xhi + 4, fnx);
    ptrdiff_t off[kRtLoads];
    bool ok[kRtLoads];
#pragma unroll
    for (int m = 0; m < kRtLoads; ++m) {
        const int t = int(threadIdx.x) + m * 256;
        const int r = t / kRtQuads, q = t % kRtQuads;
        const int fx = 2 * I0 - 4 + 4 * q, fy = min(max(2 * J0 - 1 + r, 0), fny - 1);  // (rows clamped: results masked)
        ok[m] = t < kRtRows * kRtQuads && fx >= fxlo && fx + 3 < fxhi;
        off[m] = ptrdiff_t(fy) * fnx + fx;
    }
    auto loadPlane = [&](int K, float4 (&v)[kRtLoads]) {
        const float *p = rz + ptrdiff_t(K) * fny * fnx;
#pragma unroll
        for (int m = 0; m < kRtLoads; ++m) v[m] = ok[m] ? *reinterpret_cast<const float4 *>(p + off[m]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const float w[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    float4 nextv[kRtLoads];
    loadPlane(K0, nextv);
    for (int K = K0; K < K1; ++K) {
        float *buf = tile[K & 1];  // (the buffer of plane K - 2: everybody finished with it before the last barrier)
#pragma unroll
        for (int m = 0; m < kRtLoads; ++m) {
            const int t = int(threadIdx.x) + m * 256;
            if (t < kRtRows * kRtQuads) *reinterpret_cast<float4 *>(buf + (t / kRtQuads) * kRtStride + 4 * (t % kRtQuads)) = nextv[m];
        }
        if (K + 1 < K1) loadPlane(K + 1, nextv);
        __syncthreads();
        // this thread's two coarse columns: fine x 2 I - 1 .. 2 I + 2 = tile column 2 li + 3 .., fine rows 4 lp .. 4 lp + 5 of the tile
        const float *p = buf + (4 * lp) * kRtStride + 2 * li + 3;
        float rs[6];
#pragma unroll
        for (int yo = 0; yo < 6; ++yo) {
            const float *r = p + yo * kRtStride;
            const float2 mid = *reinterpret_cast<const float2 *>(r + 1);
            rs[yo] = w[0] * r[0] + w[1] * mid.x + w[2] * mid.y + w[3] * r[3];
        }
        if (inGrid && any) {
            const size_t c = size_t(K) * cplane + col;
            const float va = w[0] * rs[0] + w[1] * rs[1] + w[2] * rs[2] + w[3] * rs[3], vb = w[0] * rs[2] + w[1] * rs[3] + w[2] * rs[4] + w[3] * rs[5];
            coarse[c] = activeLabel(cg.lab[c]) ? va : 0.f;
            if (second) coarse[c + cg.nx] = activeLabel(cg.lab[c + cg.nx]) ? vb : 0.f;
        }
    }
}

// General BOUNDARY cells of a level that takes the pair above: residualZKernel leaves them out (their code is not a simple one: r = 0),
// this kernel adds their part to rz -- r = b - (row . x) from the cell's operator row (boundaryOpKernel<OP_RESIDUAL>), times the z
// weight, into the two coarse planes K with 2 K - 1 <= k <= 2 K + 2.  One launch per k mod 4 (`phase`): two general cells of one
// column that add into the same rz entry are less than four planes apart, so within a launch no two threads write the same entry
// and the adds need no atomics -- every entry gets its terms in the order of the launches, the same on every run.
__global__ __launch_bounds__(256) void residualZGeneralKernel(GridP g, float *__restrict__ rz, const float *__restrict__ x, const float *__restrict__ b, int phase)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= g.nbnd) return;
    const size_t c = size_t(g.bnd[t]);
    const size_t sz = size_t(g.nx) * g.ny;
    const int k = int(c / sz);
    if ((k & 3) != phase) return;
    float lap, diag;
    boundaryRow(g, [&](size_t p) { return x[p]; }, t, c, lap, diag);
    const float r = epilogue<OP_RESIDUAL>(x[c], b[c], lap, diag, 0.f);
    const size_t col = c - size_t(k) * sz;
    // plane 2 m is the second term of coarse plane m and the last of m - 1, plane 2 m + 1 the third of m and the first of m + 1
    const int m = k >> 1, Ka = (k & 1) ? m : m - 1, Kb = (k & 1) ? m + 1 : m;
    const float wa = (k & 1) ? 0.375f : 0.125f, wb = (k & 1) ? 0.125f : 0.375f;
    const int cnz = g.nz >> 1;
    if (Ka >= 0 && Ka < cnz) rz[size_t(Ka) * sz + col] += wa * r;
    if (Kb >= 0 && Kb < cnz) rz[size_t(Kb) * sz + col] += wb * r;
}

// ---------------------------------------------------------------------------------------------
// Prolongation + add (Ops.h:873-972): fine c += 4 * trilerp(coarse) at sample point c/2 - 1/4:
// even c = 2m reads coarse m-1, m with f = 3/4; odd c = 2m+1 reads m, m+1 with f = 1/4.  lerp is
// (1-f) a + f b, x first, then y, then z (Ops.h:841-871).  One thread per fine cell.
// ---------------------------------------------------------------------------------------------
// snap / snapTile (optional, all three prolongation kernels): cells of the tiles flagged in snapTile (launchMarkSnapTiles) also
// leave their new value in `snap` -- the snapshot from which the band stage in front of a Gauss-Seidel sweep reads while it
// writes the iterate in place (smoothStroke)
__device__ __forceinline__ bool snapTileOf(const GridP &g, const uint8_t *__restrict__ snapTile, int i, int j, int k)
{
    const int tx = (g.nx + kTile - 1) / kTile, ty = (g.ny + kTile - 1) / kTile;
    return snapTile[(size_t(k / kTile) * ty + j / kTile) * tx + i / kTile] != 0;
}
__global__ void prolongAddKernel(GridP fg, float *__restrict__ fine, const float *__restrict__ coarse, float *__restrict__ snap = nullptr,
                                 const uint8_t *__restrict__ snapTile = nullptr)
{
    const size_t n = size_t(fg.nx) * fg.ny * fg.nz;
    const size_t c = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n) return;
    if (!activeLabel(fg.lab[c])) return;
    const int i = int(c % fg.nx), j = int((c / fg.nx) % fg.ny), k = int(c / (size_t(fg.nx) * fg.ny));
    const int cnx = fg.nx >> 1, cny = fg.ny >> 1;
    const int bi = (i - 1) >> 1, bj = (j - 1) >> 1, bk = (k - 1) >> 1;  // active cells have i,j,k >= 1
    const float fx = (i & 1) ? 0.25f : 0.75f, fy = (j & 1) ? 0.25f : 0.75f, fz = (k & 1) ? 0.25f : 0.75f;
    const float *p = coarse + (ptrdiff_t(bk) * cny + bj) * cnx + bi;  // bk = -1 addresses the lower ghost plane
    const size_t sy = size_t(cnx), sz = size_t(cnx) * cny;
    const float v00 = lerpRef(p[0], p[1], fx), v10 = lerpRef(p[sy], p[sy + 1], fx);
    const float v01 = lerpRef(p[sz], p[sz + 1], fx), v11 = lerpRef(p[sz + sy], p[sz + sy + 1], fx);
    const float t = lerpRef(lerpRef(v00, v10, fy), lerpRef(v01, v11, fy), fz);
    const float v = fine[c] + 4.f * t;  // Ops.h:964
    fine[c] = v;
    if (snap && snapTileOf(fg, snapTile, i, j, k)) snap[c] = v;
}

// Same operator, one thread per 4 consecutive fine cells (16-byte read-modify-write of the fine
// grid).  Fine cells 4m..4m+3 interpolate from coarse x-indices 2m-1..2m+2:
//   4m: (2m-1, 2m; f=3/4)   4m+1: (2m, 2m+1; 1/4)   4m+2: (2m, 2m+1; 3/4)   4m+3: (2m+1, 2m+2; 1/4)
// Requires fine nx % 4 == 0.
__global__ __launch_bounds__(256) void prolongAddQuadKernel(GridP fg, float *__restrict__ fine,
                                                            const float *__restrict__ coarse, unsigned nblocks, float *__restrict__ snap = nullptr,
                                                            const uint8_t *__restrict__ snapTile = nullptr)
{
    const unsigned nq = unsigned(fg.nx) >> 2;
    const size_t total = size_t(nq) * fg.ny * fg.nz;
    const unsigned block = remapBlock(blockIdx.x, nblocks);
    size_t t = size_t(block) * blockDim.x + threadIdx.x;
    if (fg.chunks && !listQuad(fg.chunks, fg.chunkCells, block, t)) return;
    if (t >= total) return;
    const unsigned m = unsigned(t % nq);
    const size_t row = t / nq;
    const int j = int(row % fg.ny), k = int(row / fg.ny);
    const size_t c = row * size_t(fg.nx) + (size_t(m) << 2);
    const uchar4 l = *reinterpret_cast<const uchar4 *>(fg.lab + c);
    const bool a0 = activeLabel(l.x), a1 = activeLabel(l.y), a2 = activeLabel(l.z), a3 = activeLabel(l.w);
    if (!(a0 | a1 | a2 | a3)) return;
    const int cnx = fg.nx >> 1, cny = fg.ny >> 1, cnz = fg.nz >> 1;
    // clamped coarse indices: the clamps only bite for cells on the EXTERIOR shell, whose values are dropped
    const int x0 = max(2 * int(m) - 1, 0), x1 = 2 * int(m), x2 = 2 * int(m) + 1, x3 = min(2 * int(m) + 2, cnx - 1);
    const int bj = max((j - 1) >> 1, 0), bk = max((k - 1) >> 1, fg.ghostLo ? -1 : 0);
    const int bj1 = min(bj + 1, cny - 1), bk1 = min(bk + 1, fg.ghostHi ? cnz : cnz - 1);
    const float fy = (j & 1) ? 0.25f : 0.75f, fz = (k & 1) ? 0.25f : 0.75f;
    float v[2][2][4];  // [z][y][fine x]
#pragma unroll
    for (int zz = 0; zz < 2; ++zz)
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const float *r = coarse + (ptrdiff_t(zz ? bk1 : bk) * cny + (yy ? bj1 : bj)) * cnx;
            const float c0 = r[x0], c3 = r[x3];
            const float2 c12 = *reinterpret_cast<const float2 *>(r + x1);
            (void)x2;
            v[zz][yy][0] = lerpRef(c0, c12.x, 0.75f);
            v[zz][yy][1] = lerpRef(c12.x, c12.y, 0.25f);
            v[zz][yy][2] = lerpRef(c12.x, c12.y, 0.75f);
            v[zz][yy][3] = lerpRef(c12.y, c3, 0.25f);
        }
    float4 f = *reinterpret_cast<const float4 *>(fine + c);
    float add[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        add[e] = 4.f * lerpRef(lerpRef(v[0][0][e], v[0][1][e], fy), lerpRef(v[1][0][e], v[1][1][e], fy), fz);
    if (a0) f.x += add[0];
    if (a1) f.y += add[1];
    if (a2) f.z += add[2];
    if (a3) f.w += add[3];
    *reinterpret_cast<float4 *>(fine + c) = f;
    if (snap && snapTileOf(fg, snapTile, int(m) << 2, j, k)) *reinterpret_cast<float4 *>(snap + c) = f;
}

// The same operator with the coarse reads shared: one thread owns the fine quad 4m..4m+3 of the two rows
// j = 2jp+1, 2jp+2 in the two planes k = 2kp+1, 2kp+2 -- the 16 fine cells that interpolate from the same
// 2 x 2 coarse rows (bj = jp, jp+1; bk = kp, kp+1) -- so the twelve coarse loads serve 16 cells instead
// of 4.  Rows 0 / ny-1 are EXTERIOR shell and are never touched; planes -1 / nz exist only as the other
// rank's cells of a slab run (ghostLo / ghostHi) and are masked.  Same arithmetic per cell as above.
template <class TX = float>
__global__ __launch_bounds__(256) void prolongAddBlockKernel(GridP fg, TX *__restrict__ fine,
                                                             const float *__restrict__ coarse, unsigned nblocks,
                                                             int npj, int kp0, size_t total, float pm = 1.f, TX *__restrict__ snap = nullptr,
                                                             const uint8_t *__restrict__ snapTile = nullptr)
{
    constexpr bool kMixed = !std::is_same<TX, float>::value;
    const unsigned nq = unsigned(fg.nx) >> 2;
    const size_t t = size_t(remapBlock(blockIdx.x, nblocks)) * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const unsigned m = unsigned(t % nq);
    if (int(4 * m) < fg.xlo || int(4 * m) >= fg.xhi) return;  // (outside the level's active x range: nothing to add to, not even codes to read)
    const size_t rest = t / nq;
    const int jp = int(rest % unsigned(npj)), kp = kp0 + int(rest / unsigned(npj));
    const int js[2] = {2 * jp + 1, 2 * jp + 2}, ks[2] = {2 * kp + 1, 2 * kp + 2};
    const bool kv[2] = {ks[0] >= 0, ks[1] < fg.nz};
    const size_t sy = size_t(fg.nx), sz = size_t(fg.nx) * fg.ny;
    size_t c[2][2];
    uchar4 l[2][2];
    bool any = false;
#pragma unroll
    for (int zz = 0; zz < 2; ++zz)
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            c[zz][yy] = size_t(kv[zz] ? ks[zz] : 0) * sz + size_t(js[yy]) * sy + (size_t(m) << 2);
            l[zz][yy] = fg.streaming ? streamLoad4(fg.lab + c[zz][yy]) : *reinterpret_cast<const uchar4 *>(fg.lab + c[zz][yy]);
            any = any || (kv[zz] && anyActive(l[zz][yy]));
        }
    if (!any) return;
    const int cnx = fg.nx >> 1, cny = fg.ny >> 1;
    const int x0 = max(2 * int(m) - 1, 0), x1 = 2 * int(m), x3 = min(2 * int(m) + 2, cnx - 1);
    float v[2][2][4];  // [coarse z][coarse y][fine x]
#pragma unroll
    for (int zz = 0; zz < 2; ++zz)
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const float *r = coarse + (ptrdiff_t(kp + zz) * cny + (jp + yy)) * cnx;  // kp = -1: the lower ghost plane
            const float c0 = r[x0], c3 = r[x3];
            const float2 c12 = *reinterpret_cast<const float2 *>(r + x1);
            v[zz][yy][0] = lerpRef(c0, c12.x, 0.75f);
            v[zz][yy][1] = lerpRef(c12.x, c12.y, 0.25f);
            v[zz][yy][2] = lerpRef(c12.x, c12.y, 0.75f);
            v[zz][yy][3] = lerpRef(c12.y, c3, 0.25f);
        }
    const float fs[2] = {0.25f, 0.75f};  // odd index first
#pragma unroll
    for (int zz = 0; zz < 2; ++zz)
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
            const uchar4 lab = l[zz][yy];
            const bool a0 = activeLabel(lab.x), a1 = activeLabel(lab.y), a2 = activeLabel(lab.z), a3 = activeLabel(lab.w);
            if (!kv[zz] || !anyActive(lab)) continue;
            // read-modify-write of a cell nobody else touches in this launch
            float4 f = fg.streaming ? Cell<TX>::load4nt(fine + c[zz][yy]) : Cell<TX>::load4(fine + c[zz][yy]);
            float add[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                add[e] = 4.f * lerpRef(lerpRef(v[0][0][e], v[0][1][e], fs[yy]), lerpRef(v[1][0][e], v[1][1][e], fs[yy]), fs[zz]);
                if (kMixed) add[e] *= pm;
            }
            if (a0) f.x += add[0];
            if (a1) f.y += add[1];
            if (a2) f.z += add[2];
            if (a3) f.w += add[3];
            if (fg.streaming) Cell<TX>::store4nt(fine + c[zz][yy], f);
            else Cell<TX>::store4(fine + c[zz][yy], f);
            if (snap && snapTileOf(fg, snapTile, int(m) << 2, js[yy], ks[zz])) Cell<TX>::store4(snap + c[zz][yy], f);
        }
}

// ---------------------------------------------------------------------------------------------
// Coarsest-level direct solve (MG.cpp:669-692) as x = A^-1 b with the dense inverse built at
// set-up.  gather b -> v, one wave per row of the mat-vec, scatter into x.
// ---------------------------------------------------------------------------------------------
__global__ void coarseGatherKernel(int n, const int32_t *__restrict__ cells, const float *__restrict__ b,
                                   float *__restrict__ v)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) v[t] = b[cells[t]];
}
__global__ __launch_bounds__(256) void coarseMatVecKernel(int n, const float *__restrict__ inv,
                                                          const int32_t *__restrict__ cells,
                                                          const float *__restrict__ v, float *__restrict__ x)
{
    const int row = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (row >= n) return;
    const float *r = inv + size_t(row) * n;
    double acc = 0.0;
    for (int c = lane; c < n; c += kWave) acc += double(r[c]) * double(v[c]);
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) x[cells[row]] = float(acc);
}

// ---------------------------------------------------------------------------------------------
// Vector updates on active cells (Ops.h:974-1195).  Quad version (16-byte accesses) + scalar tail.
// ---------------------------------------------------------------------------------------------
enum VecOp { V_AXPY = 0, V_XPAY = 1, V_SCALE = 2, V_MUL = 3 };

template <int VOP>
__device__ __forceinline__ float vecOp(float d, float a, float s, float scale)
{
    if (VOP == V_AXPY) return d + scale * a;   // addToVector: d += scale * a
    if (VOP == V_XPAY) return a + scale * s;   // addVectors: d = a + scale * s
    if (VOP == V_SCALE) return scale * d;      // scaleVector
    return a * s;                              // d = a .* s (diagonal preconditioner)
}

// quad index of this thread's `it`-th piece of work: plain grid stride, or (chunk list) the thread's quad
// inside the it-th active chunk of this workgroup; returns false when the thread is done
// (bid: the workgroup's place in the launch -- blockIdx.x, or its XCD-aware remap for kernels that read neighbour rows)
__device__ __forceinline__ bool nextQuad(const int32_t *chunks, int nchunks, int chunkCells, size_t nq, size_t it, size_t &q, unsigned bid)
{
    if (chunks && chunkCells == kChunkCells) {  // one list entry per workgroup
        const size_t ci = size_t(bid) + it * gridDim.x;
        if (ci >= size_t(nchunks)) return false;
        q = size_t(chunks[ci]) * (kChunkCells / 4) + threadIdx.x;
        return true;  // (a ragged last chunk is guarded by q < nq at the use)
    }
    if (chunks) {  // one entry per wavefront / per 16 lanes: nchunks is a multiple of 4 / 16
        const size_t share = size_t(bid) + it * gridDim.x;
        if (share * size_t(kChunkCells / chunkCells) >= size_t(nchunks)) return false;
        if (!listQuad(chunks, chunkCells, share, q)) q = nq;
        return true;  // (list padding and a ragged last chunk are guarded by q < nq at the use)
    }
    q = size_t(bid) * blockDim.x + threadIdx.x + it * size_t(gridDim.x) * blockDim.x;
    return q < nq;
}
__device__ __forceinline__ bool nextQuad(const int32_t *chunks, int nchunks, int chunkCells, size_t nq, size_t it, size_t &q)
{
    return nextQuad(chunks, nchunks, chunkCells, nq, it, q, blockIdx.x);
}

template <int VOP>
__global__ __launch_bounds__(256) void vecKernel(size_t n, const uint8_t *__restrict__ lab, float *dst, const float *a,
                                                 const float *s, const float *scaleDev, float scaleHost, float sign,
                                                 const int32_t *__restrict__ chunks, int nchunks, int chunkCells)
{
    const float scale = sign * (scaleDev ? *scaleDev : scaleHost);
    const size_t nq = n >> 2;
    size_t q;
    for (size_t it = 0; nextQuad(chunks, nchunks, chunkCells, nq, it, q); ++it) {
        if (q >= nq) continue;
        const uchar4 l = reinterpret_cast<const uchar4 *>(lab)[q];
        float4 d = streamLoad4(dst + (q << 2));
        float4 av = d, sv = d;
        if (VOP != V_SCALE) av = streamLoad4(a + (q << 2));
        if (VOP == V_XPAY || VOP == V_MUL) sv = streamLoad4(s + (q << 2));
        if (activeLabel(l.x)) d.x = vecOp<VOP>(d.x, av.x, sv.x, scale);
        if (activeLabel(l.y)) d.y = vecOp<VOP>(d.y, av.y, sv.y, scale);
        if (activeLabel(l.z)) d.z = vecOp<VOP>(d.z, av.z, sv.z, scale);
        if (activeLabel(l.w)) d.w = vecOp<VOP>(d.w, av.w, sv.w, scale);
        reinterpret_cast<float4 *>(dst)[q] = d;
    }
    // tail (n % 4 cells)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t c = (nq << 2) + threadIdx.x;
        if (activeLabel(lab[c])) dst[c] = vecOp<VOP>(dst[c], VOP != V_SCALE ? a[c] : 0.f, (VOP == V_XPAY || VOP == V_MUL) ? s[c] : 0.f, scale);
    }
}

// 1/diag of the fine operator for the diagonal preconditioner (Plug.cpp:500-553): 1/6 on INTERIOR
// cells, 1/(sum of the six face weights) on BOUNDARY cells, 0 elsewhere.
__global__ void diagInverseKernel(GridP g, float *__restrict__ dinv)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const size_t c = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const unsigned l = g.lab[c];
    float v = 0.f;
    if (l == MGPS_INTERIOR_CELL) v = 1.f / 6.f;
    else if (l >= kCodeGeneral) {
        const int i = int(c % g.nx), j = int((c / g.nx) % g.ny), k = int(c / (size_t(g.nx) * g.ny));
        float d = 6.f;
        if (g.wx) {
            const size_t fx = (size_t(k) * g.ny + j) * (g.nx + 1) + i;
            const size_t fy = (size_t(k) * (g.ny + 1) + j) * g.nx + i;
            d = g.wx[fx] + g.wx[fx + 1] + g.wy[fy] + g.wy[fy + g.nx] + g.wz[c] + g.wz[c + size_t(g.nx) * g.ny];
        }
        v = 1.f / d;
    }
    dinv[c] = v;
}

// ---------------------------------------------------------------------------------------------
// Reductions over active cells (Ops.h:1020-1085, 1205-1326): per-thread fp64 accumulation,
// wavefront shuffle reduction, one partial per workgroup, then a single-workgroup second stage
// that adds the partials in a fixed order (bitwise reproducible from run to run).
// ---------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ double redTerm(float a, float b)
{
    if (KIND == 0) return double(a) * double(b);
    if (KIND == 1) return double(a) * double(a);
    if (KIND == 2) return double(a);
    return double(fabsf(a));
}
template <int KIND>
__device__ __forceinline__ double redCombine(double u, double v)
{
    return KIND <= 1 ? u + v : (u > v ? u : v);
}
template <int KIND>
__device__ __forceinline__ double blockReduce(double acc)
{
    __shared__ double part[4];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) acc = redCombine<KIND>(acc, __shfl_down(acc, off));
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    double total = part[0];
    for (int w = 1; w < int(blockDim.x / kWave); ++w) total = redCombine<KIND>(total, part[w]);
    return total;
}

template <int KIND>
__global__ __launch_bounds__(256) void reduceKernel(size_t n, const uint8_t *__restrict__ lab, const float *__restrict__ a,
                                                    const float *__restrict__ b, double *__restrict__ partials,
                                                    const int32_t *__restrict__ chunks, int nchunks, int chunkCells)
{
    double acc = 0.0;  // identity for sums and for max(0, .) / max|.|
    const size_t nq = n >> 2;
    size_t q;
    for (size_t it = 0; nextQuad(chunks, nchunks, chunkCells, nq, it, q); ++it) {
        if (q >= nq) continue;
        const uchar4 l = reinterpret_cast<const uchar4 *>(lab)[q];
        const float4 av = streamLoad4(a + (q << 2));  // (read once: nontemporal, like every pure stream of the CG loop -- see cgUpdateKernel)
        float4 bv = av;
        if (KIND == 0) bv = streamLoad4(b + (q << 2));
        if (activeLabel(l.x)) acc = redCombine<KIND>(acc, redTerm<KIND>(av.x, bv.x));
        if (activeLabel(l.y)) acc = redCombine<KIND>(acc, redTerm<KIND>(av.y, bv.y));
        if (activeLabel(l.z)) acc = redCombine<KIND>(acc, redTerm<KIND>(av.z, bv.z));
        if (activeLabel(l.w)) acc = redCombine<KIND>(acc, redTerm<KIND>(av.w, bv.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t c = (nq << 2) + threadIdx.x;
        if (activeLabel(lab[c])) acc = redCombine<KIND>(acc, redTerm<KIND>(a[c], KIND == 0 ? b[c] : 0.f));
    }
    const double total = blockReduce<KIND>(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

template <int KIND>
__global__ __launch_bounds__(256) void reduceFinalKernel(int nparts, const double *__restrict__ partials,
                                                         double *__restrict__ result)
{
    double acc = 0.0;
    for (int p = threadIdx.x; p < nparts; p += blockDim.x) acc = redCombine<KIND>(acc, partials[p]);
    const double total = blockReduce<KIND>(acc);
    if (threadIdx.x == 0) *result = total;
}

// plain fill with zeros, one 16-byte store per thread (tools/zerobench.hip: 6.8 TB/s on 512 MiB against 4.4 TB/s for a
// grid-stride loop over 2048 workgroups); `a` only needs 4-byte alignment
__global__ __launch_bounds__(256) void zeroKernel(float *__restrict__ a, size_t n)
{
    const size_t head = min(n, (size_t(16) - (reinterpret_cast<size_t>(a) & 15)) / 4 & 3);
    float4 *v = reinterpret_cast<float4 *>(a + head);
    const size_t nq = (n - head) >> 2;
    const size_t q = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (q < nq) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) a[threadIdx.x] = 0.f;
        const size_t tail = head + (nq << 2);
        if (tail + threadIdx.x < n) a[tail + threadIdx.x] = 0.f;
    }
}

// band-only ghost exchange of a slab run: gather / scatter the band cells of one x-y plane
__global__ void packKernel(float *__restrict__ buf, const float *__restrict__ a, const int32_t *__restrict__ idx, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) buf[t] = a[ptrdiff_t(idx[t])];
}
__global__ void unpackKernel(float *__restrict__ a, const float *__restrict__ buf, const int32_t *__restrict__ idx, int n)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) a[ptrdiff_t(idx[t])] = buf[t];
}

inline unsigned blocksFor(size_t work, unsigned per) { return unsigned((work + per - 1) / per); }

}  // namespace

// ---- launchers -------------------------------------------------------------------------------

int forcedStencilPath()
{
    static const int forced = [] {  // MGPS_STENCIL=quad|plane: A/B switch for tuning runs
        const char *e = getenv("MGPS_STENCIL");
        return !e ? 0 : (e[0] == 'q' ? 1 : 2);
    }();
    return forced;
}
static int forcedStencil(const GridP &g) { return g.sweepPath ? g.sweepPath : forcedStencilPath(); }  // options.stencil_path wins over the environment

// Cells one activity-skipping full-domain sweep visits (the denominator of the measured bytes per cell): the cells of the listed
// runs / blocks that lie inside the level's active x range (GridP::xlo) -- counted on the device from the list the sweep walks
// (a diagnostic: one small launch and a synchronisation per call).
namespace {
__global__ __launch_bounds__(256) void sweptCountKernel(GridP g, int plane, unsigned nbx, const int32_t *__restrict__ list, size_t nentries,
                                                        unsigned long long *__restrict__ total)
{
    unsigned long long acc = 0;
    for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < nentries; e += size_t(gridDim.x) * blockDim.x) {
        if (plane) {  // a block: 256 cells of kPlaneRows rows of planeZc planes
            const unsigned bid = list ? unsigned(list[e]) : unsigned(e);
            const int x0 = int(bid % nbx) * 256;
            const int w = max(0, min(min(x0 + 256, g.nx), g.xhi) - max(x0, g.xlo));
            acc += (unsigned long long)(w) * kPlaneRows * g.planeZc;
        } else {  // a run of chunkCells consecutive cells (-1: list padding), or the whole grid as one run per row
            const int32_t ch = list ? list[e] : int32_t(e);
            if (ch < 0) continue;
            const size_t cells = list ? size_t(g.chunkCells) : size_t(g.nx);
            size_t c = size_t(ch) * cells, end = c + cells;
            while (c < end) {  // row by row
                const int i = int(c % size_t(g.nx));
                const size_t rowEnd = std::min(end, c + size_t(g.nx - i));
                const int i1 = i + int(rowEnd - c);
                acc += (unsigned long long)(max(0, min(i1, g.xhi) - max(i, g.xlo)));
                c = rowEnd;
            }
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & (kWave - 1)) == 0 && acc) atomicAdd(total, acc);
}
}  // namespace
size_t stencilSweptCells(const GridP &g)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const int kind = stencilKernelOf(g);
    if (kind == 3) return n;
    const bool plane = kind == 2;
    const unsigned nbx = (g.nx + 255) / 256, nby = (g.ny + kPlaneRows - 1) / kPlaneRows, nbz = plane ? (g.nz + g.planeZc - 1) / g.planeZc : 0;
    const int32_t *list = plane ? g.planeBlocks : g.chunks;
    const size_t entries = plane ? (list ? size_t(g.nplaneBlocks) : size_t(nbx) * nby * nbz) : (list ? size_t(g.nchunks) : size_t(g.ny) * g.nz);
    unsigned long long *dev = nullptr, host = 0;
    if (entries == 0) return 0;
    if (hipMalloc(reinterpret_cast<void **>(&dev), sizeof(host)) != hipSuccess || hipMemset(dev, 0, sizeof(host)) != hipSuccess) {
        (void)hipGetLastError();
        if (dev) (void)hipFree(dev);
        return n;
    }
    sweptCountKernel<<<unsigned(std::min<size_t>((entries + 255) / 256, 4096)), 256>>>(g, plane ? 1 : 0, nbx, list, entries, dev);
    const bool ok = hipMemcpy(&host, dev, sizeof(host), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(dev);
    if (!ok) {
        (void)hipGetLastError();
        return n;
    }
    return std::min(n, size_t(host));
}

int launchStencil(void *stream, StencilOp op, const GridP &g, float *out, const float *x, const float *b, float omega,
                  bool skipInactive)
{
    // x == nullptr: the iterate is zero everywhere (Jacobi on a level that takes the quad or the plane sweep and has no general cells to patch)
    if (!x && (op != OP_JACOBI || stencilKernelOf(g) == 3 || g.nbnd > 0)) return int(hipErrorInvalidValue);
    if (!skipInactive && (g.xlo > 0 || g.xhi < g.nx)) {  // every cell of the grid is written
        GridP whole = g;
        whole.xlo = 0;
        whole.xhi = g.nx;
        return launchStencil(stream, op, whole, out, x, b, omega, false);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const int forced = forcedStencil(g);
    const int zc = g.planeZc;  // 0: the plane-marching sweep does not apply to this shape
    // measured on MI355X (fine Jacobi sweep, plane vs quad kernel), round 1: 256^3 42.8 vs 40.8 us, 512^3 367 vs
    // 345 us, 1024^3 2.68 vs 2.95 ms -- the cache-only kernel wins while three x-y planes of x stay in an
    // XCD's L2 share, the register/LDS-marching one beyond that.  Round 3, after the quad kernel's loads were regrouped
    // and both kernels leave the row-end padding alone: 1024^3 1.854 (plane) vs 1.823 ms (quad), cycle 95.9 vs 98.1
    // per second -- the 256 MiB Infinity Cache holds the z neighbours of a 4 MiB plane.  The plane kernel keeps the
    // levels whose x-y planes are LARGER than 4 MiB (kPlaneSweepMinPlaneBytes), and whatever options.stencil_path = 2 /
    // MGPS_STENCIL=plane send to it
    const bool planeWins = size_t(g.nx) * g.ny * sizeof(float) > kPlaneSweepMinPlaneBytes;
    if (zc && (forced == 2 || (forced == 0 && planeWins))) {
        const unsigned nbx = (g.nx + 255) / 256, nby = (g.ny + kPlaneRows - 1) / kPlaneRows;
        const unsigned nbz = (g.nz + zc - 1) / zc;
        const bool list = skipInactive && g.planeBlocks != nullptr;
        const unsigned nb = list ? unsigned(g.nplaneBlocks) : nbx * nby * nbz;
        const int32_t *blocks = list ? g.planeBlocks : nullptr;
        if (nb > 0) switch (op) {
                case OP_JACOBI:
                    if (x) stencilPlaneKernel<OP_JACOBI><<<nb, 64 * kPlaneRows, 0, s>>>(g, out, x, b, omega, nbx, nby, nbz, zc, blocks);
                    else stencilPlaneKernel<OP_JACOBI, false, true><<<nb, 64 * kPlaneRows, 0, s>>>(g, out, x, b, omega, nbx, nby, nbz, zc, blocks);  // x == 0 everywhere
                    break;
                case OP_RESIDUAL: stencilPlaneKernel<OP_RESIDUAL><<<nb, 64 * kPlaneRows, 0, s>>>(g, out, x, b, omega, nbx, nby, nbz, zc, blocks); break;
                default: stencilPlaneKernel<OP_APPLY><<<nb, 64 * kPlaneRows, 0, s>>>(g, out, x, b, omega, nbx, nby, nbz, zc, blocks); break;
            }
    } else if ((g.nx & 3) == 0) {
        const bool list = skipInactive && g.chunks != nullptr;
        const unsigned nb = list ? unsigned(g.nchunks) / unsigned(kChunkCells / g.chunkCells) : blocksFor(n >> 2, 256);
        const int32_t *chunks = list ? g.chunks : nullptr;
        if (nb > 0) switch (op) {
                case OP_JACOBI:
                    if (x) stencilQuadKernel<OP_JACOBI><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, chunks);
                    else stencilQuadKernel<OP_JACOBI, false, float, true><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, chunks);  // x == 0 everywhere
                    break;
                case OP_RESIDUAL: stencilQuadKernel<OP_RESIDUAL><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, chunks); break;
                default: stencilQuadKernel<OP_APPLY><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, chunks); break;
            }
    } else {
        const unsigned nb = blocksFor(n, 256);
        switch (op) {
            case OP_JACOBI: stencilScalarKernel<OP_JACOBI><<<nb, 256, 0, s>>>(g, out, x, b, omega); break;
            case OP_RESIDUAL: stencilScalarKernel<OP_RESIDUAL><<<nb, 256, 0, s>>>(g, out, x, b, omega); break;
            default: stencilScalarKernel<OP_APPLY><<<nb, 256, 0, s>>>(g, out, x, b, omega); break;
        }
    }
    if (g.nbnd > 0) {
        const unsigned nb = blocksFor(size_t(g.nbnd), 256);
        switch (op) {
            case OP_JACOBI: boundaryOpKernel<OP_JACOBI><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb); break;
            case OP_RESIDUAL: boundaryOpKernel<OP_RESIDUAL><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb); break;
            default: boundaryOpKernel<OP_APPLY><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb); break;
        }
    }
    return int(hipGetLastError());
}

// min / max x over the active cells of a level (GridP::xlo / xhi); nx % 4 == 0: a thread takes 16 codes (four words) at a time
__global__ __launch_bounds__(256) void activeXRangeKernel(const uint8_t *__restrict__ lab, int nx, size_t cells, int *__restrict__ range)
{
    int lo = nx, hi = -1;
    const unsigned nq = unsigned(nx) >> 2;  // quads per row
    const size_t quads = cells >> 2;
    const unsigned *w = reinterpret_cast<const unsigned *>(lab);
    for (size_t q0 = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) * 4; q0 < quads; q0 += size_t(gridDim.x) * blockDim.x * 4) {
        unsigned v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = q0 + m < quads ? w[q0 + m] : 0x01010101u;  // (EXTERIOR)
        const unsigned iq = quads <= 0xffffffffull ? unsigned(q0) % nq : unsigned(q0 % nq);  // (a 64-bit remainder costs a hundred instructions)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (v[m] == 0x01010101u) continue;  // (four EXTERIOR cells: most of what lies outside the range)
            unsigned any = 0xFu;                // (four INTERIOR cells: most of what lies inside)
            if (v[m] != 0u) {
                any = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) any |= activeLabel((v[m] >> (8 * e)) & 255u) ? (1u << e) : 0u;
            }
            if (any) {
                unsigned qi = iq + m;
                if (qi >= nq) qi -= nq;  // (at most one row wrap in four quads: nq >= 4 where this kernel runs)
                lo = min(lo, int(4 * qi) + __ffs(any) - 1);
                hi = max(hi, int(4 * qi) + 31 - __clz(any));
            }
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        lo = min(lo, __shfl_down(lo, off));
        hi = max(hi, __shfl_down(hi, off));
    }
    if ((threadIdx.x & (kWave - 1)) == 0 && hi >= 0) {
        atomicMin(range, lo);
        atomicMax(range + 1, hi);
    }
}
int launchActiveXRange(void *stream, const uint8_t *lab, int nx, size_t cells, int *range)
{
    if (cells == 0 || (nx & 3) != 0 || nx < 16) return 0;
    const unsigned nb = unsigned(std::min<size_t>((cells / 16 + 255) / 256, 256 * 32));
    activeXRangeKernel<<<std::max(nb, 1u), 256, 0, static_cast<hipStream_t>(stream)>>>(lab, nx, cells, range);
    return int(hipGetLastError());
}

// blocks of the main launch of a sweep over level g (same choice of kernel as launchStencil)
static unsigned sweepBlocks(const GridP &g, bool skipInactive, int *path)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const int forced = forcedStencil(g);
    const int zc = g.planeZc;
    const bool planeWins = size_t(g.nx) * g.ny * sizeof(float) > kPlaneSweepMinPlaneBytes;
    if (zc && (forced == 2 || (forced == 0 && planeWins))) {
        *path = 0;
        const unsigned nbx = (g.nx + 255) / 256, nby = (g.ny + kPlaneRows - 1) / kPlaneRows, nbz = (g.nz + zc - 1) / zc;
        return (skipInactive && g.planeBlocks) ? unsigned(g.nplaneBlocks) : nbx * nby * nbz;
    }
    if ((g.nx & 3) == 0) {
        *path = 1;
        return (skipInactive && g.chunks) ? unsigned(g.nchunks) / unsigned(kChunkCells / g.chunkCells) : blocksFor(n >> 2, 256);
    }
    *path = 2;
    return blocksFor(n, 256);
}

int stencilKernelOf(const GridP &g)
{
    int path = 0;
    (void)sweepBlocks(g, true, &path);
    return path == 0 ? 2 : path == 1 ? 1 : 3;
}

size_t applyDotPartialCount(const GridP &g)
{
    int path;
    const unsigned a = sweepBlocks(g, true, &path), b = sweepBlocks(g, false, &path);
    return size_t(std::max(a, b)) + blocksFor(size_t(std::max(g.nbnd, 0)), 256) + 64 + 1;
}

// stage 1 of the final sum: `nout` workgroups fold contiguous runs of the partials, in a fixed order
__global__ __launch_bounds__(256) void foldPartialsKernel(int nparts, const double *__restrict__ partials, double *__restrict__ out)
{
    const int per = (nparts + int(gridDim.x) - 1) / int(gridDim.x);
    const int lo = int(blockIdx.x) * per, hi = min(nparts, lo + per);
    double acc = 0.0;
    for (int p = lo + int(threadIdx.x); p < hi; p += int(blockDim.x)) acc += partials[p];
    blockDotStore(acc, out, blockIdx.x);
}

// A sweep (A.x, or the out-of-place Jacobi sweep) that also leaves its per-workgroup shares of the dot product of
// dotTerm in partials[0 .. *nparts); launchFoldDot sums them (with whatever other launches appended) in a fixed order
int launchStencilDot(void *stream, StencilOp op, const GridP &g, float *out, const float *x, const float *b, float omega,
                     double *partials, unsigned *nparts)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    int path = 0;
    const unsigned nb = sweepBlocks(g, true, &path);
    const bool jac = op == OP_JACOBI;
    if (path == 0) {
        const int zc = g.planeZc;
        const unsigned nbx = (g.nx + 255) / 256, nby = (g.ny + kPlaneRows - 1) / kPlaneRows, nbz = (g.nz + zc - 1) / zc;
        if (nb > 0) {
            if (jac) stencilPlaneKernel<OP_JACOBI, true><<<nb, 64 * kPlaneRows, 0, s>>>(g, out, x, b, omega, nbx, nby, nbz, zc, g.planeBlocks, partials);
            else stencilPlaneKernel<OP_APPLY, true><<<nb, 64 * kPlaneRows, 0, s>>>(g, out, x, nullptr, 0.f, nbx, nby, nbz, zc, g.planeBlocks, partials);
        }
    } else if (path == 1) {
        if (nb > 0) {
            if (jac) stencilQuadKernel<OP_JACOBI, true><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, g.chunks, partials);
            else stencilQuadKernel<OP_APPLY, true><<<nb, 256, 0, s>>>(g, out, x, nullptr, 0.f, nb, g.chunks, partials);
        }
    } else {
        if (jac) stencilScalarKernel<OP_JACOBI, true><<<nb, 256, 0, s>>>(g, out, x, b, omega, partials);
        else stencilScalarKernel<OP_APPLY, true><<<nb, 256, 0, s>>>(g, out, x, nullptr, 0.f, partials);
    }
    unsigned used = nb;
    if (g.nbnd > 0) {
        const unsigned nbb = blocksFor(size_t(g.nbnd), 256);
        if (jac) boundaryOpKernel<OP_JACOBI, true><<<nbb, 256, 0, s>>>(g, out, x, b, omega, nbb, partials + used);
        else boundaryOpKernel<OP_APPLY, true><<<nbb, 256, 0, s>>>(g, out, x, nullptr, 0.f, nbb, partials + used);
        used += nbb;
    }
    *nparts = used;
    return int(hipGetLastError());
}

// *resultDev = sum of partials[0 .. nparts); the 64 slots behind them are scratch
int launchFoldDot(void *stream, double *partials, unsigned nparts, double *resultDev)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (nparts > 4096) {
        double *folded = partials + nparts;
        foldPartialsKernel<<<64, 256, 0, s>>>(int(nparts), partials, folded);
        reduceFinalKernel<0><<<1, 256, 0, s>>>(64, folded, resultDev);
    } else
        reduceFinalKernel<0><<<1, 256, 0, s>>>(int(nparts), partials, resultDev);
    return int(hipGetLastError());
}

// out = A x on level g and *resultDev = <x, A x> over the active cells, in one pass over x (CG.h:110-121)
int launchApplyDot(void *stream, const GridP &g, float *out, const float *x, double *partials, double *resultDev)
{
    unsigned nparts = 0;
    const int e = launchStencilDot(stream, OP_APPLY, g, out, x, nullptr, 0.f, partials, &nparts);
    return e ? e : launchFoldDot(stream, partials, nparts, resultDev);
}

int launchBandJacobi(void *stream, const GridP &g, float *x, const float *b, const int32_t *band, int nband,
                     float *bandTmp, float omega, double *dotPartials)
{
    if (nband <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned nb = blocksFor(size_t(nband), 256);
    bandComputeKernel<<<nb, 256, 0, s>>>(g, x, b, band, nband, bandTmp, omega, nb);
    if (dotPartials) bandScatterKernel<true><<<nb, 256, 0, s>>>(x, band, nband, bandTmp, nb, b, dotPartials);
    else bandScatterKernel<<<nb, 256, 0, s>>>(x, band, nband, bandTmp, nb);
    return int(hipGetLastError());
}

// The list part of a cut level's band-stage messages (round 5, box form): buf = [a0 at base + idx | a1 at base + idx] for the cells the
// neighbour's boxes read (pack), and the same back into the deep ghost planes of the grids (unpack).  Both sides in one launch.
__global__ __launch_bounds__(256) void haloListPackKernel(HaloList lo, HaloList hi, unsigned blocksLo, const float *__restrict__ a0, const float *__restrict__ a1)
{
    const bool upper = blockIdx.x >= blocksLo;
    const HaloList &s = upper ? hi : lo;
    const size_t t = size_t(blockIdx.x - (upper ? blocksLo : 0)) * blockDim.x + threadIdx.x, n = size_t(s.n);
    if (t < n) s.buf[t] = a0[s.base + s.idx[t]];
    else if (a1 && t < 2 * n) s.buf[t] = a1[s.base + s.idx[t - n]];
}
__global__ __launch_bounds__(256) void haloListUnpackKernel(HaloList lo, HaloList hi, unsigned blocksLo, float *__restrict__ a0, float *__restrict__ a1)
{
    const bool upper = blockIdx.x >= blocksLo;
    const HaloList &s = upper ? hi : lo;
    const size_t t = size_t(blockIdx.x - (upper ? blocksLo : 0)) * blockDim.x + threadIdx.x, n = size_t(s.n);
    if (t < n) a0[s.base + s.idx[t]] = s.buf[t];
    else if (a1 && t < 2 * n) a1[s.base + s.idx[t - n]] = s.buf[t];
}
int launchHaloListPack(void *stream, const HaloList &lo, const HaloList &hi, const float *a0, const float *a1)
{
    const unsigned per = a1 ? 2u : 1u, bl = lo.buf ? blocksFor(size_t(lo.n) * per, 256) : 0, bh = hi.buf ? blocksFor(size_t(hi.n) * per, 256) : 0;
    if (bl + bh == 0) return 0;
    haloListPackKernel<<<bl + bh, 256, 0, static_cast<hipStream_t>(stream)>>>(lo, hi, bl, a0, a1);
    return int(hipGetLastError());
}
int launchHaloListUnpack(void *stream, const HaloList &lo, const HaloList &hi, float *a0, float *a1)
{
    const unsigned per = a1 ? 2u : 1u, bl = lo.buf ? blocksFor(size_t(lo.n) * per, 256) : 0, bh = hi.buf ? blocksFor(size_t(hi.n) * per, 256) : 0;
    if (bl + bh == 0) return 0;
    haloListUnpackKernel<<<bl + bh, 256, 0, static_cast<hipStream_t>(stream)>>>(lo, hi, bl, a0, a1);
    return int(hipGetLastError());
}

unsigned bandScatterBlocks(int nband) { return nband > 0 ? blocksFor(size_t(nband), 256) : 0; }
namespace {
template <class TX>
int launchBandBoxT(hipStream_t s, const GridP &g, const BandBoxesDev &bx, bool closure, const TX *src, const float *b, TX *dst, TX *snap, float omega,
                   const MixScale &ms, double *dotPartials, const TX *dotOld, int outClosure)
{
    const unsigned ng = unsigned(bx.ngroups);
    const bool dot = dotPartials != nullptr;
#define MGPS_BOX_LAUNCH3(C, D, G, Z) \
    bandBoxKernel<TX, C, D, G, Z><<<ng, kBoxThreads, 0, s>>>(g, src, b, dst, snap, bx.info, bx.list, bx.general, omega, bx.depth, ms, dotPartials, dotOld, outClosure)
#define MGPS_BOX_LAUNCH(C, D, Z)                               \
    do {                                                       \
        if (bx.anyGeneral) MGPS_BOX_LAUNCH3(C, D, true, Z);    \
        else MGPS_BOX_LAUNCH3(C, D, false, Z);                 \
    } while (0)
    if (!src) {  // the iterate is zero everywhere
        if (dot) return int(hipErrorInvalidValue);
        if (closure) MGPS_BOX_LAUNCH(true, false, true);
        else MGPS_BOX_LAUNCH(false, false, true);
    } else if (closure) {
        if (dot) MGPS_BOX_LAUNCH(true, true, false);
        else MGPS_BOX_LAUNCH(true, false, false);
    } else {
        if (dot) MGPS_BOX_LAUNCH(false, true, false);
        else MGPS_BOX_LAUNCH(false, false, false);
    }
#undef MGPS_BOX_LAUNCH
#undef MGPS_BOX_LAUNCH3
    return int(hipGetLastError());
}
}  // namespace
int launchBandBox(void *stream, const GridP &g, const BandBoxesDev &bx, bool closure, const void *src, const float *b, void *dst, void *snap, float omega,
                  bool half, const MixScale &ms, double *dotPartials, const void *dotOld, bool outClosure)
{
    if (bx.ngroups <= 0) return 0;
    if (!boxPlaneFits(Dims{g.nx, g.ny, g.nz})) return int(hipErrorInvalidValue);  // (24-bit plane stride)
    if (!dst && !(closure && snap)) return int(hipErrorInvalidValue);  // (dst == nullptr: the closure launch fills the snapshot only)
    if ((src && src == dst) || (dotPartials && !dotOld)) return int(hipErrorInvalidValue);  // (a group reads what its neighbours own; src == nullptr: the iterate is zero everywhere, nothing is read and dst may be the iterate itself)
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (half)
        return launchBandBoxT<__half>(s, g, bx, closure, static_cast<const __half *>(src), b, static_cast<__half *>(dst), static_cast<__half *>(snap), omega, ms,
                                      dotPartials, static_cast<const __half *>(dotOld), outClosure ? 1 : 0);
    return launchBandBoxT<float>(s, g, bx, closure, static_cast<const float *>(src), b, static_cast<float *>(dst), static_cast<float *>(snap), omega, ms,
                                 dotPartials, static_cast<const float *>(dotOld), outClosure ? 1 : 0);
}
// closure launch + sweep of a stroke in one launch (strokeFrontKernel): levels that take the quad sweep, fp32, no gathered dot.
// x == nullptr: the zero iterate.  keep: launchMarkClosure's bits
int launchStrokeFront(void *stream, const GridP &g, const BandBoxesDev &bx, float *out, const float *x, const float *b, float *snap, float omega, const uint32_t *keep)
{
    if (bx.ngroups <= 0 || stencilKernelOf(g) != 1 || !keep || !snap || !out || out == snap) return int(hipErrorInvalidValue);  // (g with its general rows: the boxes read them; the sweep's general cells are band cells, left to the boxes)
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const bool list = g.chunks != nullptr;
    const unsigned nshares = list ? unsigned(g.nchunks) / unsigned(kChunkCells / g.chunkCells) : blocksFor(n >> 2, 256);
    const unsigned ng = unsigned(bx.ngroups), nb = ng + (nshares + kBoxThreads / 256 - 1) / (kBoxThreads / 256);
    const int32_t *chunks = list ? g.chunks : nullptr;
#define MGPS_FRONT(G, Z) strokeFrontKernel<G, Z><<<nb, kBoxThreads, 0, s>>>(g, out, x, b, snap, bx.info, bx.list, bx.general, omega, bx.depth, ng, nshares, chunks, keep)
    if (bx.anyGeneral) {
        if (x) MGPS_FRONT(true, false);
        else MGPS_FRONT(true, true);
    } else {
        if (x) MGPS_FRONT(false, false);
        else MGPS_FRONT(false, true);
    }
#undef MGPS_FRONT
    return int(hipGetLastError());
}
int launchMarkClosure(void *stream, const GridP &g, const BandBoxesDev &bx, uint32_t *bits)
{
    if (bx.ngroups <= 0) return 0;
    markClosureKernel<<<unsigned(bx.ngroups), 256, 0, static_cast<hipStream_t>(stream)>>>(g, bx.info, bx.list, bits);
    return int(hipGetLastError());
}
int launchMarkSnapTiles(void *stream, const GridP &g, const BandBoxesDev &bx, uint8_t *tiles)
{
    if (bx.ngroups <= 0) return 0;
    markSnapTilesKernel<<<unsigned(bx.ngroups), 256, 0, static_cast<hipStream_t>(stream)>>>(g, bx.info, bx.list, tiles);
    return int(hipGetLastError());
}
int launchBandBoxCopy(void *stream, const GridP &g, const BandBoxesDev &bx, const void *src, void *dst, bool half)
{
    if (bx.ngroups <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (half) bandBoxCopyKernel<__half><<<unsigned(bx.ngroups), kBoxThreads, 0, s>>>(g, static_cast<const __half *>(src), static_cast<__half *>(dst), bx.info, bx.list);
    else bandBoxCopyKernel<float><<<unsigned(bx.ngroups), kBoxThreads, 0, s>>>(g, static_cast<const float *>(src), static_cast<float *>(dst), bx.info, bx.list);
    return int(hipGetLastError());
}


// ---- mixed precision (options.precision = 1): the fine level's iterate and residual live in binary16 ----------------
// The launchers take the binary16 grids as void* (the solver layer does not see __half).  Solver-owned grids only:
// chunks without active cells hold 0 and are never visited.
// dotPartials / nparts (Jacobi only, optional): the sweep also leaves its per-workgroup shares of sum x~' b (the stored,
// rounded x~'; the unscaled rhs) in dotPartials[0 .. *nparts)
int launchStencilMixed(void *stream, StencilOp op, const GridP &g, void *outH, const void *xH, const float *b, float omega, const MixScale &ms,
                       double *dotPartials, unsigned *nparts)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    __half *out = static_cast<__half *>(outH);
    const __half *x = static_cast<const __half *>(xH);
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const bool list = g.chunks != nullptr;
    const unsigned nb = list ? unsigned(g.nchunks) / unsigned(kChunkCells / g.chunkCells) : blocksFor(n >> 2, 256);
    const bool dot = dotPartials != nullptr && op == OP_JACOBI;
    if (!x && (op != OP_JACOBI || dot || g.nbnd > 0)) return int(hipErrorInvalidValue);  // x == nullptr: the zero iterate (see launchStencil)
    if (nb > 0) {
        if (!x) stencilQuadKernel<OP_JACOBI, false, __half, true><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, g.chunks, nullptr, ms);
        else if (dot) stencilQuadKernel<OP_JACOBI, true, __half><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, g.chunks, dotPartials, ms);
        else if (op == OP_JACOBI) stencilQuadKernel<OP_JACOBI, false, __half><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, g.chunks, nullptr, ms);
        else stencilQuadKernel<OP_RESIDUAL, false, __half><<<nb, 256, 0, s>>>(g, out, x, b, omega, nb, g.chunks, nullptr, ms);
    }
    unsigned used = dot ? nb : 0;
    if (g.nbnd > 0) {
        const unsigned nbb = blocksFor(size_t(g.nbnd), 256);
        if (dot) boundaryOpKernel<OP_JACOBI, true, __half><<<nbb, 256, 0, s>>>(g, out, x, b, omega, nbb, dotPartials + used, ms);
        else if (op == OP_JACOBI) boundaryOpKernel<OP_JACOBI, false, __half><<<nbb, 256, 0, s>>>(g, out, x, b, omega, nbb, nullptr, ms);
        else boundaryOpKernel<OP_RESIDUAL, false, __half><<<nbb, 256, 0, s>>>(g, out, x, b, omega, nbb, nullptr, ms);
        if (dot) used += nbb;
    }
    if (nparts) *nparts = used;
    return int(hipGetLastError());
}

// *result *= mul / *sigma: turns the gathered sum x~ b of the mixed cycle into <z, b>
__global__ void scaleResultKernel(double *__restrict__ result, const float *__restrict__ sigma, float mul) { *result *= double(mul) / double(*sigma); }
int launchScaleResult(void *stream, double *resultDev, const float *sigmaDev, float mul)
{
    scaleResultKernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(resultDev, sigmaDev, mul);
    return int(hipGetLastError());
}

// dst = x~ * (mul / *sigma): the cycle's result back in the caller's units, whole grid
__global__ __launch_bounds__(256) void fromHalfKernel(float *__restrict__ dst, const __half *__restrict__ src, const float *__restrict__ sigma, float mul,
                                                      size_t nq)
{
    const size_t q = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (q >= nq) return;
    const float m = mul / *sigma;
    const float4 v = Cell<__half>::load4(src + 4 * q);
    reinterpret_cast<float4 *>(dst)[q] = make_float4(m * v.x, m * v.y, m * v.z, m * v.w);
}
int launchFromHalf(void *stream, float *dst, const void *srcH, const float *sigmaDev, float mul, size_t cells)
{
    fromHalfKernel<<<blocksFor(cells >> 2, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(dst, static_cast<const __half *>(srcH), sigmaDev, mul, cells >> 2);
    return int(hipGetLastError());
}
// *sigma = 2^-ceil(log2(*maxAbs)): the power of two that brings max |b| into (1/2, 1]; 1 for a zero rhs
__global__ void mixSigmaKernel(const double *__restrict__ maxAbs, float *__restrict__ sigma)
{
    const double m = *maxAbs;
    int e = 0;
    if (m > 0.0 && m < 1e300) (void)frexp(m, &e);  // m = f 2^e, f in [1/2, 1)
    *sigma = ldexpf(1.f, -e);
}
int launchMixSigma(void *stream, const double *maxAbsDev, float *sigmaDev)
{
    mixSigmaKernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(maxAbsDev, sigmaDev);
    return int(hipGetLastError());
}
// zeros on the active chunks of a binary16 grid of level g
__global__ __launch_bounds__(256) void zeroChunksHalfKernel(__half *__restrict__ a, const int32_t *__restrict__ chunks, int chunkCells, size_t nq)
{
    size_t q;
    if (!listQuad(chunks, chunkCells, blockIdx.x, q)) return;
    if (q < nq) reinterpret_cast<uint2 *>(a)[q] = make_uint2(0u, 0u);
}
int launchZeroActiveHalf(void *stream, const GridP &g, void *aH)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    if (!g.chunks || (n & 3) != 0) return int(hipMemsetAsync(aH, 0, n * sizeof(__half), static_cast<hipStream_t>(stream)));
    const unsigned nb = unsigned(g.nchunks) / unsigned(kChunkCells / g.chunkCells);
    if (nb > 0) zeroChunksHalfKernel<<<nb, 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<__half *>(aH), g.chunks, g.chunkCells, n >> 2);
    return int(hipGetLastError());
}

// dotPartials (optional): nmixed + npure slots, one per tile, mixed tiles first
int launchTiledGS(void *stream, const GridP &g, float *x, const float *b, const int32_t *pureTiles, int npure,
                  const int32_t *mixedTiles, int nmixed, const int32_t *tileBndStart, int forward, double *dotPartials, float *snap, const uint8_t *snapTile)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (snap && !snapTile) return int(hipErrorInvalidValue);
    // same colour => no two tiles of either launch share a face: the two launches are independent.  Up to three workgroups
    // per CU they go as one launch (tiledGSBothKernel)
    if (nmixed > 0 && npure > 0 && nmixed + npure <= 768) {
        if (dotPartials) tiledGSBothKernel<true><<<unsigned(nmixed + npure), 256, 0, s>>>(g, x, b, mixedTiles, nmixed, pureTiles, tileBndStart, forward, dotPartials, snap, snapTile);
        else tiledGSBothKernel<<<unsigned(nmixed + npure), 256, 0, s>>>(g, x, b, mixedTiles, nmixed, pureTiles, tileBndStart, forward, nullptr, snap, snapTile);
        return int(hipGetLastError());
    }
    if (dotPartials) {
        if (nmixed > 0) tiledGSMixedKernel<true><<<unsigned(nmixed), 256, 0, s>>>(g, x, b, mixedTiles, tileBndStart, forward, dotPartials, MixScale{}, snap, snapTile);
        if (npure > 0) tiledGSPureKernel<true><<<unsigned(npure), 256, 0, s>>>(g, x, b, pureTiles, forward, dotPartials + nmixed, MixScale{}, snap, snapTile);
        return int(hipGetLastError());
    }
    if (nmixed > 0) tiledGSMixedKernel<<<unsigned(nmixed), 256, 0, s>>>(g, x, b, mixedTiles, tileBndStart, forward, nullptr, MixScale{}, snap, snapTile);
    if (npure > 0) tiledGSPureKernel<<<unsigned(npure), 256, 0, s>>>(g, x, b, pureTiles, forward, nullptr, MixScale{}, snap, snapTile);
    return int(hipGetLastError());
}

// one colour's tiles of the binary16 iterate (options.precision = 1 with the Gauss-Seidel smoother): staged, swept and summed in
// fp32 like the fp32 grids, rounded once when the tile is written back; the rhs in the iterate's units (ms)
int launchTiledGSMixed(void *stream, const GridP &g, void *xH, const float *b, const int32_t *pureTiles, int npure, const int32_t *mixedTiles, int nmixed,
                       const int32_t *tileBndStart, int forward, const MixScale &ms)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    __half *x = static_cast<__half *>(xH);
    if (nmixed > 0) tiledGSMixedKernel<false, __half><<<unsigned(nmixed), 256, 0, s>>>(g, x, b, mixedTiles, tileBndStart, forward, nullptr, ms);
    if (npure > 0) tiledGSPureKernel<false, __half><<<unsigned(npure), 256, 0, s>>>(g, x, b, pureTiles, forward, nullptr, ms);
    return int(hipGetLastError());
}

__global__ void patchSimpleCodesKernel(uint8_t *codes, const int32_t *band, const uint8_t *bandDiag, int nbnd, int nband)
{
    const int t = nbnd + int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= nband) return;
    const int32_t c = band[t];
    if (codes[c] == MGPS_BOUNDARY_CELL) codes[c] = uint8_t(kCodeSimple + bandDiag[t]);
}

int launchPatchSimpleCodes(void *stream, uint8_t *codes, const int32_t *band, const uint8_t *bandDiag, int nbnd, int nband)
{
    if (nband > nbnd)
        patchSimpleCodesKernel<<<blocksFor(size_t(nband - nbnd), 256), 256, 0, static_cast<hipStream_t>(stream)>>>(codes, band, bandDiag, nbnd, nband);
    return int(hipGetLastError());
}

int launchRestrict(void *stream, const GridP &coarse, float *coarseOut, const float *fine)
{
    const size_t n = size_t(coarse.nx) * coarse.ny * coarse.nz;
    // the march is a serial chain of kc coarse planes per thread: below ~2048 workgroups of columns the chip is not
    // filled and the thread-per-cell kernel wins (coarse 128^3: 19 us against 51 us)
    // (kc: coarse planes a workgroup marches.  16 where that still leaves >= 8192 workgroups; fewer planes -- more workgroups, a
    // little more plane overlap -- on smaller levels: a 256^3 coarse level has 2048 columns of tiles, two waves per SIMD at kc = 16)
    int kc = 16;
    const unsigned nbx = (coarse.nx + 63) / 64, nby = (coarse.ny + 7) / 8;
    while (kc > 4 && size_t(nbx) * nby * ((coarse.nz + kc - 1) / kc) < 8192) kc >>= 1;
    const unsigned nbz = (coarse.nz + kc - 1) / kc;
    // (the march walks every coarse column; the per-cell kernel walks the coarse level's activity runs at 1.3 x the cost per
    // cell -- 1.13 vs 0.88 ms at 1024^3 -> 512^3, 0.14 vs 0.118 ms one level down: it takes over where those runs hold a
    // clearly smaller part of the level.  On the cube they hold 76 %: the march stays)
    const bool runsWin = coarse.chunks && double(coarse.nchunks) * coarse.chunkCells * 1.3 * runCostFactor(coarse.chunkCells) < 0.8 * double(n);
    if (!runsWin && coarse.nx >= 64 && coarse.nz >= kc && nbx * nby * nbz >= 2048u) {
        // (odd coarse nx: the register-only march, which the LDS-tiled one replaced elsewhere -- LABNOTES R4)
        if ((coarse.nx & 1) == 0) restrictTileKernel<<<nbx * nby * nbz, 256, 0, static_cast<hipStream_t>(stream)>>>(coarse, coarseOut, fine, kc, nbx, nby);
        else restrictMarchKernel<<<nbx * nby * nbz, 256, 0, static_cast<hipStream_t>(stream)>>>(coarse, coarseOut, fine, kc, nbx, nby);
        return int(hipGetLastError());
    }
    const unsigned nb = coarse.chunks ? unsigned(size_t(coarse.nchunks) * size_t(coarse.chunkCells) / 256) : blocksFor(n, 256);
    if (nb > 0) restrictKernel<<<nb, 256, 0, static_cast<hipStream_t>(stream)>>>(coarse, coarseOut, fine);
    return int(hipGetLastError());
}

// The residual of level `fine` restricted to level `coarse` without writing the residual: residualZKernel into `rz` (a grid of
// fine.nx x fine.ny x fine.nz / 2 floats, zero where no block with active cells ever writes), then restrictXYKernel.
// residualRestrictFits: the shapes both kernels take
bool residualRestrictFits(const GridP &fine, const GridP &coarse)
{
    return fine.planeZc >= 4 && (fine.planeZc & 1) == 0 && (fine.nz & 1) == 0 && coarse.nx >= 64 && (coarse.nx & 1) == 0 && 2 * coarse.nx == fine.nx &&
           2 * coarse.ny == fine.ny && 2 * coarse.nz == fine.nz;
}
// the (block, side) pairs residualZEdgeKernel serves, from a host copy of the block flags (planeBlockCount bytes): 2 * block + side
std::vector<int32_t> planeBlockEdges(const GridP &g, const std::vector<uint8_t> &flags)
{
    const int zc = g.planeZc > 0 ? g.planeZc : 1;
    const size_t layer = size_t((g.nx + 255) / 256) * size_t((g.ny + kPlaneRows - 1) / kPlaneRows), nbz = size_t((g.nz + zc - 1) / zc);
    std::vector<int32_t> edges;
    for (size_t bid = 0; bid < flags.size() && bid < layer * nbz; ++bid) {
        if (flags[bid]) continue;
        const size_t bz = bid / layer;
        if (bz > 0 && flags[bid - layer]) edges.push_back(int32_t(2 * bid));
        if (bz + 1 < nbz && flags[bid + layer]) edges.push_back(int32_t(2 * bid + 1));
    }
    return edges;
}
// The residual on the boundary planes of a cut level alone (plane 0 where the slab has a lower neighbour, plane nz - 1 where it has
// an upper one), general BOUNDARY cells included, into `r`: what the neighbours' marches fold in as their edge terms.
int launchResidualEdgePlanes(void *stream, const GridP &g, float *r, const float *x, const float *b)
{
    const size_t sz = size_t(g.nx) * g.ny;
    for (int side = 0; side < 2; ++side) {
        if (!(side == 0 ? g.ghostLo : g.ghostHi)) continue;
        const int k = side == 0 ? 0 : g.nz - 1;
        GridP p = g;  // the plane as a grid of its own: its neighbours below and above are its "ghost planes"
        p.nz = 1;
        p.lab = g.lab + size_t(k) * sz;
        p.ghostLo = (k > 0 || g.ghostLo) ? 1 : 0;
        p.ghostHi = (k < g.nz - 1 || g.ghostHi) ? 1 : 0;
        p.chunks = nullptr;
        p.nchunks = 0;
        p.planeBlocks = nullptr;
        p.nplaneBlocks = 0;
        p.planeZc = 0;
        p.sweepPath = 1;
        p.nbnd = 0;
        p.streaming = 0;
        const int e = launchStencil(stream, OP_RESIDUAL, p, r + size_t(k) * sz, x + size_t(k) * sz, b + size_t(k) * sz, 0.f, true);
        if (e) return e;
    }
    if (g.nbnd > 0) {  // (every general cell of the slab: the ones on the two planes are among them, the rest lands in scratch)
        const unsigned nb = blocksFor(size_t(g.nbnd), 256);
        boundaryOpKernel<OP_RESIDUAL><<<nb, 256, 0, static_cast<hipStream_t>(stream)>>>(g, r, x, b, 0.f, nb);
    }
    return int(hipGetLastError());
}
int launchResidualZ(void *stream, const GridP &fine, float *rz, const float *x, const float *b, const int32_t *edges, int nedges, const float *rEdge)
{
    const int zc = fine.planeZc;
    const unsigned nbx = (fine.nx + 255) / 256, nby = (fine.ny + kPlaneRows - 1) / kPlaneRows, nbz = (fine.nz + zc - 1) / zc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool list = fine.planeBlocks != nullptr;
    const unsigned nb = list ? unsigned(fine.nplaneBlocks) : nbx * nby * nbz;
    if (nb > 0) residualZKernel<<<nb, 64 * kPlaneRows, 0, s>>>(fine, rz, x, b, nbx, nby, zc, list ? fine.planeBlocks : nullptr, rEdge);
    if (list && nedges > 0) residualZEdgeKernel<<<unsigned(nedges), 64 * kPlaneRows, 0, s>>>(fine, rz, x, b, nbx, nby, zc, edges);
    if (fine.nbnd > 0)  // the general BOUNDARY cells' part (their entries lie in blocks the launches above have just written)
        for (int phase = 0; phase < 4; ++phase) residualZGeneralKernel<<<blocksFor(size_t(fine.nbnd), 256), 256, 0, s>>>(fine, rz, x, b, phase);
    return int(hipGetLastError());
}
int launchRestrictXY(void *stream, const GridP &coarse, float *coarseOut, const float *rz)
{
    int kc = 16;  // (no overlap along z: any kc costs the same bytes; enough workgroups to fill the chip several times over)
    const unsigned nbx = (coarse.nx + kRtI - 1) / kRtI, nby = (coarse.ny + kRtJ - 1) / kRtJ;
    while (kc > 1 && size_t(nbx) * nby * ((coarse.nz + kc - 1) / kc) < 8192) kc >>= 1;
    const unsigned nbz = (coarse.nz + kc - 1) / kc;
    restrictXYKernel<<<nbx * nby * nbz, 256, 0, static_cast<hipStream_t>(stream)>>>(coarse, coarseOut, rz, kc, nbx, nby);
    return int(hipGetLastError());
}

int launchRestrictMixed(void *stream, const GridP &coarse, float *coarseOut, const void *fineH, float fm)
{
    const __half *fine = static_cast<const __half *>(fineH);
    const size_t n = size_t(coarse.nx) * coarse.ny * coarse.nz;
    const int kc = 16;
    const unsigned nbx = (coarse.nx + 63) / 64, nby = (coarse.ny + 7) / 8, nbz = (coarse.nz + kc - 1) / kc;
    // (the same choice as launchRestrict: where the coarse level's activity runs hold a clearly smaller part of the level -- a
    // free surface -- the list-driven kernel wins: round 2 always marched here, 124 against 84 us on the 512^3 pool)
    const bool runsWin = coarse.chunks && double(coarse.nchunks) * coarse.chunkCells * 1.3 * runCostFactor(coarse.chunkCells) < 0.8 * double(n);
    if (!runsWin && coarse.nx >= 64 && coarse.nz >= kc && nbx * nby * nbz >= 2048u) {
        restrictMarchKernel<__half><<<nbx * nby * nbz, 256, 0, static_cast<hipStream_t>(stream)>>>(coarse, coarseOut, fine, kc, nbx, nby, fm);
        return int(hipGetLastError());
    }
    const unsigned nb = coarse.chunks ? unsigned(size_t(coarse.nchunks) * size_t(coarse.chunkCells) / 256) : blocksFor(n, 256);
    if (nb > 0) restrictKernel<__half><<<nb, 256, 0, static_cast<hipStream_t>(stream)>>>(coarse, coarseOut, fine, fm);
    return int(hipGetLastError());
}

// fine (binary16, stored scaled by pm) += pm * 4 * trilerp(coarse); requires the block kernel's shape rule (mixedPrecisionShapeOk)
int launchProlongAddMixed(void *stream, const GridP &fine, void *fineH, const float *coarse, float pm)
{
    const int npj = fine.ny / 2 - 1;
    const int kp0 = 0, kp1 = fine.nz / 2 - 2;
    const size_t total = size_t(fine.nx >> 2) * npj * size_t(std::max(kp1 - kp0 + 1, 0));
    const unsigned nb = blocksFor(total, 256);
    if (nb > 0)
        prolongAddBlockKernel<__half><<<nb, 256, 0, static_cast<hipStream_t>(stream)>>>(fine, static_cast<__half *>(fineH), coarse, nb, npj, kp0, total, pm);
    return int(hipGetLastError());
}
bool mixedPrecisionShapeOk(int nx, int ny, int nz) { return (nx & 3) == 0 && nx >= 8 && (ny & 1) == 0 && (nz & 1) == 0 && ny >= 4 && nz >= 2; }

int launchProlongAdd(void *stream, const GridP &fine, float *fineInOut, const float *coarse, float *snap, const uint8_t *snapTile)
{
    if (snap && !snapTile) return int(hipErrorInvalidValue);
    const size_t n = size_t(fine.nx) * fine.ny * fine.nz;
    // The block kernel (one thread = 16 fine cells that share their coarse rows) walks the whole grid; the quad kernel walks the
    // level's activity runs at 1.7 x the cost per cell (0.272 vs 0.162 ms on the full 512^3 cube) -- and at more than that where the
    // runs are short: on the 512^3 free-surface pool (47 M of 134 M cells in 32-cell runs) the block kernel is the faster one
    // (MG-PCG 58.2 -> 57.4 ms, round 5: the factor below was 1.7).  Where the liquid fills a small part of the grid -- a 480^3
    // simulation in the 1024^3 power-of-two expansion: 62 M of 1074 M cells in runs -- the runs win.
    const bool runsWin = fine.chunks && double(fine.nchunks) * fine.chunkCells * 3.4 * runCostFactor(fine.chunkCells) < 0.8 * double(n);  // (a clear win only)
    if (!runsWin && (fine.nx & 3) == 0 && fine.nx >= 8 && (fine.ny & 1) == 0 && (fine.nz & 1) == 0 && fine.ny >= 4 && fine.nz >= 2) {
        const int npj = fine.ny / 2 - 1;  // row pairs (1,2) .. (ny-3, ny-2)
        const int kp0 = fine.ghostLo ? -1 : 0, kp1 = fine.ghostHi ? fine.nz / 2 - 1 : fine.nz / 2 - 2;
        const size_t total = size_t(fine.nx >> 2) * npj * size_t(std::max(kp1 - kp0 + 1, 0));
        const unsigned nb = blocksFor(total, 256);
        if (nb > 0)
            prolongAddBlockKernel<<<nb, 256, 0, static_cast<hipStream_t>(stream)>>>(fine, fineInOut, coarse, nb, npj, kp0, total, 1.f, snap, snapTile);
    } else if ((fine.nx & 3) == 0 && fine.nx >= 8) {
        const unsigned nb = fine.chunks ? unsigned(fine.nchunks) / unsigned(kChunkCells / fine.chunkCells) : blocksFor(n >> 2, 256);
        if (nb > 0) prolongAddQuadKernel<<<nb, 256, 0, static_cast<hipStream_t>(stream)>>>(fine, fineInOut, coarse, nb, snap, snapTile);
    } else
        prolongAddKernel<<<blocksFor(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(fine, fineInOut, coarse, snap, snapTile);
    return int(hipGetLastError());
}

// flags[block] = 1 for the blocks of the plane-marching sweep's activity list (a byte per 256 x 16 x zc block, zeroed by the caller)
__global__ __launch_bounds__(256) void planeBlockFlagsKernel(const int32_t *__restrict__ blocks, int n, uint8_t *__restrict__ flags)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t < n) flags[blocks[t]] = 1;
}
size_t planeBlockCount(const GridP &g)
{
    const int zc = g.planeZc > 0 ? g.planeZc : 1;
    return size_t((g.nx + 255) / 256) * size_t((g.ny + kPlaneRows - 1) / kPlaneRows) * size_t((g.nz + zc - 1) / zc);
}
int launchPlaneBlockFlags(void *stream, const GridP &g, uint8_t *flags)
{
    if (!g.planeBlocks || g.nplaneBlocks <= 0) return 0;
    planeBlockFlagsKernel<<<blocksFor(size_t(g.nplaneBlocks), 256), 256, 0, static_cast<hipStream_t>(stream)>>>(g.planeBlocks, g.nplaneBlocks, flags);
    return int(hipGetLastError());
}

// Coarsest levels past kHostCoarseMax unknowns: the dense matrix of MG.cpp:359-382 assembled on the device in fp64 (row r =
// unknown r: -1 per active neighbour, diagonal = active + DIRICHLET neighbours), factorised and inverted by hipSOLVER
// (potrf / potri leave one triangle), then folded into the fp32 inverse that coarseMatVecKernel multiplies with.
__global__ __launch_bounds__(256) void coarseAssembleKernel(int n, int nx, int ny, const int32_t *__restrict__ cells, const int32_t *__restrict__ index,
                                                            const uint8_t *__restrict__ lab, double *__restrict__ A)
{
    const int r = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (r >= n) return;
    const ptrdiff_t c = cells[r], off[6] = {-1, 1, -ptrdiff_t(nx), ptrdiff_t(nx), -ptrdiff_t(nx) * ny, ptrdiff_t(nx) * ny};
    double diag = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const unsigned l = lab[c + off[q]];
        if (activeLabel(l)) {
            A[size_t(r) * n + index[c + off[q]]] = -1.0;
            diag += 1.0;
        } else if (l == MGPS_DIRICHLET_CELL)
            diag += 1.0;
    }
    A[size_t(r) * n + r] = diag;
}
// inv[r][c] = float(A[min(r,c)][max(r,c)]): the triangle potri(LOWER) fills in column-major storage is the upper one of the
// row-major view
__global__ __launch_bounds__(256) void coarseNarrowKernel(int n, const double *__restrict__ A, float *__restrict__ inv)
{
    const size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (t >= size_t(n) * n) return;
    const size_t r = t / n, c = t - r * n;
    inv[t] = float(r <= c ? A[t] : A[c * n + r]);
}
int launchCoarseAssemble(void *stream, int n, int nx, int ny, const int32_t *cells, const int32_t *index, const uint8_t *lab, double *A)
{
    coarseAssembleKernel<<<blocksFor(size_t(n), 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, nx, ny, cells, index, lab, A);
    return int(hipGetLastError());
}
int launchCoarseNarrow(void *stream, int n, const double *A, float *inv)
{
    coarseNarrowKernel<<<blocksFor(size_t(n) * n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, A, inv);
    return int(hipGetLastError());
}

// the same with the gather inside: every workgroup (four rows) stages the gathered rhs in LDS itself -- one launch instead of
// two on the cycle's critical path (the rhs is a few KB: re-gathered from the L2 by each workgroup).  n <= kCoarseLdsMax
constexpr int kCoarseLdsMax = 8192;
__global__ __launch_bounds__(256) void coarseSolveFusedKernel(int n, const float *__restrict__ inv, const int32_t *__restrict__ cells,
                                                              const float *__restrict__ b, float *__restrict__ x)
{
    __shared__ __attribute__((aligned(16))) float v[kCoarseLdsMax];
    for (int c = threadIdx.x; c < n; c += 256) v[c] = b[cells[c]];
    __syncthreads();
    const int row = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (row >= n) return;
    const float *r = inv + size_t(row) * n;
    double acc = 0.0;
    if ((n & 3) == 0) {  // rows start on 16-byte boundaries: a quarter of the load instructions (a row of the 14^3-unknown level of the
                         // BASELINE cubes is 11 KB: 43 dependent-issue scalar loads per lane against 11 -- 17 -> 12 us per cycle)
        const float4 *r4 = reinterpret_cast<const float4 *>(r), *v4 = reinterpret_cast<const float4 *>(v);
        for (int c = lane; c < (n >> 2); c += kWave) {
            const float4 a = r4[c], w = v4[c];
            acc += double(a.x) * double(w.x);
            acc += double(a.y) * double(w.y);
            acc += double(a.z) * double(w.z);
            acc += double(a.w) * double(w.w);
        }
    } else
        for (int c = lane; c < n; c += kWave) acc += double(r[c]) * double(v[c]);
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) x[cells[row]] = float(acc);
}

int launchCoarseSolve(void *stream, int n, const float *inverse, const int32_t *cells, float *x, const float *b,
                      float *gathered)
{
    if (n <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n <= kCoarseLdsMax && x != b) {
        coarseSolveFusedKernel<<<blocksFor(size_t(n), 4), 256, 0, s>>>(n, inverse, cells, b, x);
        return int(hipGetLastError());
    }
    coarseGatherKernel<<<blocksFor(size_t(n), 256), 256, 0, s>>>(n, cells, b, gathered);
    coarseMatVecKernel<<<blocksFor(size_t(n), 4), 256, 0, s>>>(n, inverse, cells, gathered, x);
    return int(hipGetLastError());
}

// grid of a grid-stride kernel: over all quads, or over the active chunks of g when it carries a list
static unsigned vecBlocks(const GridP &g, size_t n)
{
    if (g.chunks) return unsigned(std::min<size_t>(std::max(1, g.nchunks / (kChunkCells / g.chunkCells)), 2048));
    return unsigned(std::min<size_t>(std::max<size_t>(1, ((n >> 2) + 255) / 256), 2048));
}

static unsigned streamingBlocks(size_t quads)
{
    // memory-bound grid-stride kernels: enough workgroups to fill 256 CUs x 8, no more
    return unsigned(std::min<size_t>(std::max<size_t>(1, (quads + 255) / 256), 2048));
}

int launchAxpy(void *stream, const GridP &g, float *dst, const float *src, const float *scaleDev, float scaleHost,
               float sign)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    vecKernel<V_AXPY><<<vecBlocks(g, n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, g.lab, dst, src, nullptr,
                                                                                              scaleDev, scaleHost, sign, g.chunks, g.nchunks, g.chunkCells);
    return int(hipGetLastError());
}
int launchXpay(void *stream, const GridP &g, float *dst, const float *a, const float *sv, const float *scaleDev,
               float scaleHost)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    vecKernel<V_XPAY><<<vecBlocks(g, n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, g.lab, dst, a, sv, scaleDev,
                                                                                              scaleHost, 1.f, g.chunks, g.nchunks, g.chunkCells);
    return int(hipGetLastError());
}
int launchScale(void *stream, const GridP &g, float *v, float scale)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    vecKernel<V_SCALE><<<vecBlocks(g, n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, g.lab, v, nullptr, nullptr,
                                                                                               nullptr, scale, 1.f, g.chunks, g.nchunks, g.chunkCells);
    return int(hipGetLastError());
}
int launchMulMasked(void *stream, const GridP &g, float *dst, const float *a, const float *b)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    vecKernel<V_MUL><<<vecBlocks(g, n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, g.lab, dst, a, b, nullptr, 1.f,
                                                                                             1.f, g.chunks, g.nchunks, g.chunkCells);
    return int(hipGetLastError());
}
int launchDiagInverse(void *stream, const GridP &g, float *dinv)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    diagInverseKernel<<<blocksFor(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(g, dinv);
    return int(hipGetLastError());
}

// Operator rows of BOUNDARY cells (Ops.h:208-256) from device labels / weights, for the set-up path that never
// brings the weights to the host; the arithmetic of the host's rowOf (mgps_host.cpp) term by term.
__global__ void boundaryRowsKernel(Dims d, const uint8_t *__restrict__ lab, const float *__restrict__ wx,
                                   const float *__restrict__ wy, const float *__restrict__ wz, const int32_t *__restrict__ cells,
                                   int n, float *__restrict__ rows, int *__restrict__ violations)
{
    const int t = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n) return;
    const size_t c = size_t(cells[t]);
    const int i = int(c % d.nx), j = int((c / d.nx) % d.ny), k = int(c / (size_t(d.nx) * d.ny));
    const size_t plane = size_t(d.nx) * d.ny;
    const size_t fx = (size_t(k) * d.ny + j) * (d.nx + 1) + i, fy = (size_t(k) * (d.ny + 1) + j) * d.nx + i, fz = c;
    const float w[6] = {wx[fx], wx[fx + 1], wy[fy], wy[fy + d.nx], wz[fz], wz[fz + plane]};
    const ptrdiff_t off[6] = {-1, 1, -ptrdiff_t(d.nx), ptrdiff_t(d.nx), -ptrdiff_t(plane), ptrdiff_t(plane)};
    float r[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, diag = 0.f;
    bool simple = true, ruleOk = false;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const uint8_t nl = lab[ptrdiff_t(c) + off[q]];
        if (nl == MGPS_INTERIOR_CELL) {
            r[q] = 1.f;
            diag += 1.f;
        } else if (nl == MGPS_BOUNDARY_CELL) {
            r[q] = w[q];
            diag += w[q];
            simple = simple && w[q] == 1.f;
            ruleOk = ruleOk || w[q] != 1.f;
        } else if (nl == MGPS_DIRICHLET_CELL) {
            diag += w[q];
            simple = simple && w[q] == 1.f;
            ruleOk = true;
        } else
            ruleOk = true;
    }
    float *o = rows + 8 * size_t(t);
#pragma unroll
    for (int q = 0; q < 6; ++q) o[q] = r[q];
    o[6] = diag;
    o[7] = (simple && diag >= 1.f) ? 1.f : 0.f;  // (no open face: a general row, see evalRow in mgps_setup.hip)
    if (!ruleOk) atomicAdd(violations, 1);
}

int launchBoundaryRows(void *stream, const Dims &d, const uint8_t *labels, const float *wx, const float *wy, const float *wz,
                       const int32_t *cells, int n, float *rows, int *violations)
{
    if (n <= 0) return 0;
    boundaryRowsKernel<<<blocksFor(size_t(n), 256), 256, 0, static_cast<hipStream_t>(stream)>>>(d, labels, wx, wy, wz, cells, n, rows,
                                                                                          violations);
    return int(hipGetLastError());
}

// zeros on the chunks that hold active cells only: for grids whose other chunks are known to hold 0 already (the
// solver's own grids: nothing ever writes a chunk without active cells)
// blocks past `listBlocks` clear the two ghost planes (planeQuads quads each, just below and just above the grid)
__global__ __launch_bounds__(256) void zeroChunksKernel(float *__restrict__ a, const int32_t *__restrict__ chunks, int chunkCells, size_t nq,
                                                        unsigned listBlocks, size_t planeQuads)
{
    if (blockIdx.x >= listBlocks) {
        const size_t t = size_t(blockIdx.x - listBlocks) * blockDim.x + threadIdx.x;
        if (t < 2 * planeQuads) {
            float4 *dst = t < planeQuads ? reinterpret_cast<float4 *>(a) - planeQuads + t : reinterpret_cast<float4 *>(a) + nq + (t - planeQuads);
            *dst = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    size_t q;
    if (!listQuad(chunks, chunkCells, blockIdx.x, q)) return;
    if (q < nq) Cell<float>::store4nt(a + (q << 2), make_float4(0.f, 0.f, 0.f, 0.f));
}

// ---------------------------------------------------------------------------------------------
// CG vectors in fp64 around the fp32 V-cycle (options.pcg_fp64_vectors): the operator, the vector updates and the
// reductions of CG.h:18-207 on double grids, label-masked like their fp32 forms; face weights, rows and the rhs stay
// fp32.  One thread = 4 consecutive cells (nextQuad), neighbours from the caches: these passes are a fraction of an
// iteration next to the V-cycle.
// ---------------------------------------------------------------------------------------------
// MODE 0: out = A x (and <x, A x> shares in partials), MODE 1: out = b - A x (and |out|^2 shares), out32 = float(out)
// MODE 2 (round 5, the fp64-iterate CG loop): the iterate is x + dx -- the fp64 grid as of the last residual replacement and the
// fp32 sum of the updates alpha p since then (van der Vorst & Ye's group-wise update) -- read as such at all seven points;
// out32 = float(b - A (x + dx)) like MODE 1, and `out` receives the FLUSHED iterate x + dx (active cells; the others keep x):
// the loop swaps the two fp64 grids and starts the next group
template <int MODE>
__global__ __launch_bounds__(256) void stencil64Kernel(GridP g, double *__restrict__ out, const double *__restrict__ x,
                                                        const float *__restrict__ b, float *__restrict__ out32,
                                                        double *__restrict__ partials, const float *__restrict__ dx = nullptr)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz, nq = n >> 2;
    const ptrdiff_t sy = g.nx, sz = ptrdiff_t(g.nx) * g.ny;
    double acc = 0.0;
    size_t q;
    // (an XCD takes a contiguous run of the launch's shares: the y / z neighbour rows -- 8 B per cell here -- then meet in its own L2;
    // with the hardware's round-robin the pass moved 2.7 GB for 1.4 GB algorithmic on the 512^3 pool)
    const unsigned bid = remapBlock(blockIdx.x, gridDim.x);
    for (size_t it = 0; nextQuad(g.chunks, g.nchunks, g.chunkCells, nq, it, q, bid); ++it) {
        if (q >= nq) continue;
        const uchar4 l4 = reinterpret_cast<const uchar4 *>(g.lab)[q];
        const unsigned ls[4] = {l4.x, l4.y, l4.z, l4.w};
        const size_t c0 = q << 2;
        const bool any = simpleCell(ls[0]) || simpleCell(ls[1]) || simpleCell(ls[2]) || simpleCell(ls[3]);
        double res[4] = {0.0, 0.0, 0.0, 0.0};  // general BOUNDARY cells: boundary64Kernel right after
        double flushed[4] = {0.0, 0.0, 0.0, 0.0};  // MODE 2: x + dx on the quad's cells
        if (any && (g.nx & 3) == 0) {
            // the quad lies in one x-row with an active cell in it: the row is not on a grid face, so the four
            // neighbour rows and the cells left and right of the quad exist (EXTERIOR shell); 16-byte loads
            typedef double d2 __attribute__((ext_vector_type(2)));
            auto row = [&](ptrdiff_t at, double *v) {
                const d2 a = *reinterpret_cast<const d2 *>(x + at), bb = *reinterpret_cast<const d2 *>(x + at + 2);
                v[0] = a.x;
                v[1] = a.y;
                v[2] = bb.x;
                v[3] = bb.y;
                if (MODE == 2) {  // (dx is zero -- never written -- on inactive cells of solver-owned grids; the caller's x there is the centre's business below)
                    const float4 d = *reinterpret_cast<const float4 *>(dx + at);
                    v[0] += double(d.x);
                    v[1] += double(d.y);
                    v[2] += double(d.z);
                    v[3] += double(d.w);
                }
            };
            double xs[6], ym[4], yp[4], zm[4], zp[4];
            row(ptrdiff_t(c0), xs + 1);
            xs[0] = x[c0 - 1];
            xs[5] = x[c0 + 4];
            if (MODE == 2) {
                xs[0] += double(dx[c0 - 1]);
                xs[5] += double(dx[c0 + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) flushed[e] = xs[e + 1];
            }
            row(ptrdiff_t(c0) - sy, ym);
            row(ptrdiff_t(c0) + sy, yp);
            row(ptrdiff_t(c0) - sz, zm);
            row(ptrdiff_t(c0) + sz, zp);
            float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE != 0) bq = streamLoad4(b + (q << 2));  // (the rhs and the outputs are streams: nontemporal, like the CG loop's vectors)
            const float bs[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (simpleCell(ls[e])) {
                    const double lap = double(simpleDiag(ls[e])) * xs[e + 1] - (xs[e] + xs[e + 2] + ym[e] + yp[e] + zm[e] + zp[e]);
                    res[e] = MODE == 0 ? lap : double(bs[e]) - lap;
                    acc += MODE == 0 ? xs[e + 1] * res[e] : res[e] * res[e];
                }
        } else {
            auto at = [&](size_t c) { return MODE == 2 ? x[c] + double(dx[c]) : x[c]; };
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t c = c0 + e;
                if (MODE == 2) flushed[e] = at(c);
                if (any && simpleCell(ls[e])) {  // INTERIOR / simple BOUNDARY: all six neighbours exist (EXTERIOR shell)
                    const double xc = at(c);
                    const double lap = double(simpleDiag(ls[e])) * xc - (at(c - 1) + at(c + 1) + at(c - sy) + at(c + sy) + at(c - sz) + at(c + sz));
                    res[e] = MODE == 0 ? lap : double(b[c]) - lap;
                    acc += MODE == 0 ? xc * res[e] : res[e] * res[e];
                }
            }
        }
        if (MODE == 2) {  // the flushed iterate: x + dx on the active cells (general BOUNDARY cells included), x elsewhere
            typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!activeLabel(ls[e])) flushed[e] = x[c0 + e];
            __builtin_nontemporal_store(d2{flushed[0], flushed[1]}, reinterpret_cast<d2 *>(out + c0));
            __builtin_nontemporal_store(d2{flushed[2], flushed[3]}, reinterpret_cast<d2 *>(out + c0) + 1);
        } else if (out) {  // (out == nullptr: only float(out) is wanted)
            typedef double d2 __attribute__((ext_vector_type(2)));
            reinterpret_cast<d2 *>(out + c0)[0] = d2{res[0], res[1]};
            reinterpret_cast<d2 *>(out + c0)[1] = d2{res[2], res[3]};
        }
        if (MODE != 0) reinterpret_cast<float4 *>(out32)[q] = make_float4(float(res[0]), float(res[1]), float(res[2]), float(res[3]));
    }
    const double total = blockReduce<0>(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
template <int MODE>
__global__ __launch_bounds__(256) void boundary64Kernel(GridP g, double *__restrict__ out, const double *__restrict__ x,
                                                         const float *__restrict__ b, float *__restrict__ out32,
                                                         double *__restrict__ partials, const float *__restrict__ dx = nullptr)
{
    double acc = 0.0;
    for (int t = int(blockIdx.x * blockDim.x + threadIdx.x); t < g.nbnd; t += int(gridDim.x * blockDim.x)) {
        const size_t c = size_t(g.bnd[t]), nb = size_t(g.nbnd);
        const ptrdiff_t sy = g.nx, sz = ptrdiff_t(g.nx) * g.ny;
        const float *r = g.rows + t;
        auto at = [&](ptrdiff_t p) { return MODE == 2 ? x[p] + double(dx[p]) : x[p]; };  // (MODE 2: the iterate is x + dx; the stencil pass has stored the flushed cell itself)
        const ptrdiff_t cc = ptrdiff_t(c);
        double lap = double(r[6 * nb]) * at(cc);
        lap -= double(r[0]) * at(cc - 1);
        lap -= double(r[nb]) * at(cc + 1);
        lap -= double(r[2 * nb]) * at(cc - sy);
        lap -= double(r[3 * nb]) * at(cc + sy);
        lap -= double(r[4 * nb]) * at(cc - sz);
        lap -= double(r[5 * nb]) * at(cc + sz);
        const double res = MODE == 0 ? lap : double(b[c]) - lap;
        if (MODE != 2 && out) out[c] = res;
        if (MODE != 0) out32[c] = float(res);
        acc += MODE == 0 ? at(cc) * res : res * res;
    }
    const double total = blockReduce<0>(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
// x += alpha p, r -= alpha t, r32 = float(r), shares of |r|^2 (CG.h:132-153).  16-byte accesses: a thread's quad is two
// double2 per fp64 array (round 2 moved every double on its own: 846 us per pass at the 512^3 pool, 0.37 of the HBM peak);
// inactive cells keep their values (0 in r, whatever x holds)
typedef double d2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void cgUpdate64Kernel(GridP g, double *__restrict__ x, const double *__restrict__ p,
                                                         double *__restrict__ r, const double *__restrict__ t, double alpha,
                                                         float *__restrict__ r32, double *__restrict__ partials)
{
    const size_t nq = (size_t(g.nx) * g.ny * g.nz) >> 2;
    double acc = 0.0;
    size_t q;
    for (size_t it = 0; nextQuad(g.chunks, g.nchunks, g.chunkCells, nq, it, q); ++it) {
        if (q >= nq) continue;
        const uchar4 l4 = reinterpret_cast<const uchar4 *>(g.lab)[q];
        if (!anyActive(l4)) continue;
        const bool on[4] = {activeLabel(l4.x), activeLabel(l4.y), activeLabel(l4.z), activeLabel(l4.w)};
        d2v *x2 = reinterpret_cast<d2v *>(x) + 2 * q, *r2 = reinterpret_cast<d2v *>(r) + 2 * q;
        const d2v *p2 = reinterpret_cast<const d2v *>(p) + 2 * q, *t2 = reinterpret_cast<const d2v *>(t) + 2 * q;
        const d2v xa = x2[0], xb = x2[1], pa = p2[0], pb = p2[1], ra = r2[0], rb = r2[1], ta = t2[0], tb = t2[1];
        double xv[4] = {xa.x, xa.y, xb.x, xb.y}, rv[4] = {ra.x, ra.y, rb.x, rb.y};
        const double pv[4] = {pa.x, pa.y, pb.x, pb.y}, tv[4] = {ta.x, ta.y, tb.x, tb.y};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (on[e]) {
                xv[e] += alpha * pv[e];
                rv[e] -= alpha * tv[e];
                acc += rv[e] * rv[e];
            }
        x2[0] = d2v{xv[0], xv[1]};
        x2[1] = d2v{xv[2], xv[3]};
        r2[0] = d2v{rv[0], rv[1]};
        r2[1] = d2v{rv[2], rv[3]};
        reinterpret_cast<float4 *>(r32)[q] = make_float4(float(rv[0]), float(rv[1]), float(rv[2]), float(rv[3]));
    }
    const double total = blockReduce<0>(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
// p = z + beta p (CG.h:189-191); first: p = z
__global__ __launch_bounds__(256) void xpay64Kernel(GridP g, double *__restrict__ p, const float *__restrict__ z, double beta, int first)
{
    const size_t nq = (size_t(g.nx) * g.ny * g.nz) >> 2;
    size_t q;
    for (size_t it = 0; nextQuad(g.chunks, g.nchunks, g.chunkCells, nq, it, q); ++it) {
        if (q >= nq) continue;
        const uchar4 l4 = reinterpret_cast<const uchar4 *>(g.lab)[q];
        if (!anyActive(l4)) continue;
        const bool on[4] = {activeLabel(l4.x), activeLabel(l4.y), activeLabel(l4.z), activeLabel(l4.w)};
        d2v *p2 = reinterpret_cast<d2v *>(p) + 2 * q;
        const float4 zq = reinterpret_cast<const float4 *>(z)[q];
        const float zv[4] = {zq.x, zq.y, zq.z, zq.w};
        double pv[4] = {0.0, 0.0, 0.0, 0.0};
        if (!first || !(on[0] && on[1] && on[2] && on[3])) {
            const d2v pa = p2[0], pb = p2[1];
            pv[0] = pa.x;
            pv[1] = pa.y;
            pv[2] = pb.x;
            pv[3] = pb.y;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (on[e]) pv[e] = first ? double(zv[e]) : double(zv[e]) + beta * pv[e];
        p2[0] = d2v{pv[0], pv[1]};
        p2[1] = d2v{pv[2], pv[3]};
    }
}
// widen / narrow a whole grid (inactive cells hold 0 on both sides)
__global__ __launch_bounds__(256) void widenKernel(double *__restrict__ dst, const float *__restrict__ src, size_t n)
{
    const size_t q = size_t(blockIdx.x) * blockDim.x + threadIdx.x;  // n is a multiple of 4 (grids with nx % 4 == 0) or handled by the tail
    if (4 * q + 3 < n) {
        const float4 v = reinterpret_cast<const float4 *>(src)[q];
        reinterpret_cast<d2v *>(dst)[2 * q] = d2v{double(v.x), double(v.y)};
        reinterpret_cast<d2v *>(dst)[2 * q + 1] = d2v{double(v.z), double(v.w)};
    } else
        for (size_t c = 4 * q; c < n; ++c) dst[c] = double(src[c]);
}
__global__ __launch_bounds__(256) void narrowKernel(float *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    const size_t q = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (4 * q + 3 < n) {
        const d2v a = reinterpret_cast<const d2v *>(src)[2 * q], b = reinterpret_cast<const d2v *>(src)[2 * q + 1];
        reinterpret_cast<float4 *>(dst)[q] = make_float4(float(a.x), float(a.y), float(b.x), float(b.y));
    } else
        for (size_t c = 4 * q; c < n; ++c) dst[c] = float(src[c]);
}

// workgroups of an fp64 vector pass: one per 1024 cells of the active chunks, capped by the room for their partial sums
static unsigned cg64Blocks(const GridP &g, size_t capacity)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const unsigned want = g.chunks ? unsigned(g.nchunks) / unsigned(kChunkCells / g.chunkCells) : blocksFor(n >> 2, 256);
    const unsigned room = unsigned(std::min<size_t>(capacity > 1100 ? capacity - 1100 : 1, 1u << 20));  // 1024 boundary slots + 64 fold slots
    return std::max(1u, std::min(want, room));
}
// mode 0: out = A x, *resultDev = <x, A x>; mode 1: out = b - A x, out32 = float(out), *resultDev = |out|^2
int launchStencil64(void *stream, int mode, const GridP &g, double *out, const double *x, const float *b, float *out32,
                    double *partials, size_t capacity, double *resultDev, const float *dx)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned nb = cg64Blocks(g, capacity);
    if (mode == 2 && (!dx || !out || out == x)) return int(hipErrorInvalidValue);
    if (mode == 0) stencil64Kernel<0><<<nb, 256, 0, s>>>(g, out, x, b, out32, partials);
    else if (mode == 1) stencil64Kernel<1><<<nb, 256, 0, s>>>(g, out, x, b, out32, partials);
    else stencil64Kernel<2><<<nb, 256, 0, s>>>(g, out, x, b, out32, partials, dx);
    unsigned nparts = nb;
    if (g.nbnd > 0) {
        const unsigned nbb = std::min(blocksFor(size_t(g.nbnd), 256), 1024u);  // grid-stride beyond that
        if (mode == 0) boundary64Kernel<0><<<nbb, 256, 0, s>>>(g, out, x, b, out32, partials + nb);
        else if (mode == 1) boundary64Kernel<1><<<nbb, 256, 0, s>>>(g, out, x, b, out32, partials + nb);
        else boundary64Kernel<2><<<nbb, 256, 0, s>>>(g, out, x, b, out32, partials + nb, dx);
        nparts += nbb;
    }
    return launchFoldDot(stream, partials, nparts, resultDev);
}
// x32 = float(x64 (+ x32)) on the active cells of level g (the others keep what the caller put there): the fp64-iterate CG loop hands its result back
__global__ __launch_bounds__(256) void narrowSumKernel(size_t n, const uint8_t *__restrict__ lab, float *__restrict__ x32, const double *__restrict__ x64, int addDx)
{
    const size_t c = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= n) return;
    if (activeLabel(lab[c])) x32[c] = addDx ? float(x64[c] + double(x32[c])) : float(x64[c]);  // (inactive cells: the caller's values, never touched)
}
int launchNarrowSum(void *stream, const GridP &g, float *x32, const double *x64, bool addDx)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    narrowSumKernel<<<blocksFor(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, g.lab, x32, x64, addDx ? 1 : 0);
    return int(hipGetLastError());
}
int launchCgUpdate64(void *stream, const GridP &g, double *x, const double *p, double *r, const double *t, double alpha, float *r32,
                     double *partials, size_t capacity, double *resultDev)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned nb = cg64Blocks(g, capacity);
    cgUpdate64Kernel<<<nb, 256, 0, s>>>(g, x, p, r, t, alpha, r32, partials);
    return launchFoldDot(stream, partials, nb, resultDev);
}
int launchXpay64(void *stream, const GridP &g, double *p, const float *z, double beta, int first)
{
    xpay64Kernel<<<cg64Blocks(g, size_t(1) << 21), 256, 0, static_cast<hipStream_t>(stream)>>>(g, p, z, beta, first);
    return int(hipGetLastError());
}
int launchWiden(void *stream, double *dst, const float *src, size_t n)
{
    widenKernel<<<blocksFor((n + 3) / 4, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(dst, src, n);
    return int(hipGetLastError());
}
int launchNarrow(void *stream, float *dst, const double *src, size_t n)
{
    narrowKernel<<<blocksFor((n + 3) / 4, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(dst, src, n);
    return int(hipGetLastError());
}

// a[c] = 0 wherever the cell is not active (whole grid): restores the "exactly 0 outside active cells" invariant on a
// caller grid that carries values in air / solid cells
__global__ __launch_bounds__(256) void zeroInactiveKernel(size_t n, const uint8_t *__restrict__ lab, float *__restrict__ a)
{
    const size_t c = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c < n && !activeLabel(lab[c])) a[c] = 0.f;
}
int launchZeroInactive(void *stream, const GridP &g, float *a)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    zeroInactiveKernel<<<blocksFor(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(n, g.lab, a);
    return int(hipGetLastError());
}

int launchZero(void *stream, float *a, size_t count);
int launchZeroActive(void *stream, const GridP &g, float *a, bool ghostPlanes)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz, plane = size_t(g.nx) * g.ny;
    if (!g.chunks || (n & 3) != 0 || (plane & 3) != 0) {
        if (ghostPlanes) return launchZero(stream, a - plane, n + 2 * plane);
        return launchZero(stream, a, n);
    }
    const unsigned nb = unsigned(g.nchunks) / unsigned(kChunkCells / g.chunkCells);
    const unsigned extra = ghostPlanes ? blocksFor(2 * (plane >> 2), 256) : 0;
    if (nb + extra > 0)
        zeroChunksKernel<<<nb + extra, 256, 0, static_cast<hipStream_t>(stream)>>>(a, g.chunks, g.chunkCells, n >> 2, nb, plane >> 2);
    return int(hipGetLastError());
}

int launchZero(void *stream, float *a, size_t count)
{
    if (!count) return 0;
    zeroKernel<<<blocksFor((count >> 2) + 1, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(a, count);
    return int(hipGetLastError());
}
int launchPack(void *stream, float *buf, const float *a, const int32_t *idx, int n)
{
    if (n > 0) packKernel<<<blocksFor(size_t(n), 256), 256, 0, static_cast<hipStream_t>(stream)>>>(buf, a, idx, n);
    return int(hipGetLastError());
}
int launchUnpack(void *stream, float *a, const float *buf, const int32_t *idx, int n)
{
    if (n > 0) unpackKernel<<<blocksFor(size_t(n), 256), 256, 0, static_cast<hipStream_t>(stream)>>>(a, buf, idx, n);
    return int(hipGetLastError());
}

// One pass for the two vector updates of a CG iteration and the norm that follows them (CG.h:132-153):
// x += alpha p, r -= alpha t on active cells, partial sums of the new r^2 -- 25 B per cell instead of 13 + 13 + 5.
// XT = double (options.pcg_fp64_vectors = 2): the ITERATE alone lives in fp64 -- x += alpha p with p widened, 33 B per cell; r, p
// and A p stay fp32.  fp32 storage of x is what limits the recomputed residual b - A x (A fl(x) differs from A x by eps |A| |x|,
// ghost-fluid weights up to 100), not the recurrence: see mgps_options::pcg_fp64_vectors
template <class XT>
__global__ __launch_bounds__(256) void cgUpdateKernel(size_t n, const uint8_t *__restrict__ lab, XT *__restrict__ x,
                                                      const float *__restrict__ p, float *__restrict__ r,
                                                      const float *__restrict__ t, float alpha, double *__restrict__ partials,
                                                      const int32_t *__restrict__ chunks, int nchunks, int chunkCells,
                                                      const double *__restrict__ alphaDev, double *__restrict__ maxPartials, int xFirst = 0)
{
    // xFirst (fp32 x, the fp64-iterate loop of round 5): x is the sum of the updates since the last flush and this is the first
    // of a group -- x = alpha p on the active cells, nothing of the old x read
    constexpr bool kWide = std::is_same<XT, double>::value;
    double alphaD = double(alpha);
    if (alphaDev) {  // <z, r> / <p, A p> left on the device by the reductions (CG.h:121)
        alphaD = alphaDev[0] / alphaDev[1];
        alpha = float(alphaD);
    }
    double acc = 0.0;
    float big = 0.f;  // max |r| of the new residual: the mixed-precision V-cycle normalises its rhs by it
    const size_t nq = n >> 2;
    size_t q;
    for (size_t it = 0; nextQuad(chunks, nchunks, chunkCells, nq, it, q); ++it) {
        if (q >= nq) continue;
        const uchar4 l = reinterpret_cast<const uchar4 *>(lab)[q];
        // (nontemporal loads and stores: these vectors are streams -- 512 MB each at 512^3, nothing of them is in a cache when it is
        // wanted again -- and as streams they leave the L2 and the Infinity Cache to what does get re-read; 512^3 pool MG-PCG 60.4 ->
        // 58.4 ms (Jacobi), 75.4 -> 73.9 (GS) in a same-box A/B.  r alone is stored normally: the V-cycle reads it next as its rhs)
        float4 rv = streamLoad4(r + (q << 2));
        const float4 pv = streamLoad4(p + (q << 2)), tv = streamLoad4(t + (q << 2));
        if (kWide) {
            double2 *xq = reinterpret_cast<double2 *>(x) + 2 * q;
            double2 xa = xq[0], xb = xq[1];
            if (activeLabel(l.x)) xa.x += alphaD * double(pv.x);
            if (activeLabel(l.y)) xa.y += alphaD * double(pv.y);
            if (activeLabel(l.z)) xb.x += alphaD * double(pv.z);
            if (activeLabel(l.w)) xb.y += alphaD * double(pv.w);
            xq[0] = xa;
            xq[1] = xb;
        } else {
            const bool all = activeLabel(l.x) && activeLabel(l.y) && activeLabel(l.z) && activeLabel(l.w);
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!xFirst || !all) xv = streamLoad4(reinterpret_cast<const float *>(x) + (q << 2));  // (a quad with an inactive cell keeps that cell's value)
            if (activeLabel(l.x)) xv.x = (xFirst ? 0.f : xv.x) + alpha * pv.x;
            if (activeLabel(l.y)) xv.y = (xFirst ? 0.f : xv.y) + alpha * pv.y;
            if (activeLabel(l.z)) xv.z = (xFirst ? 0.f : xv.z) + alpha * pv.z;
            if (activeLabel(l.w)) xv.w = (xFirst ? 0.f : xv.w) + alpha * pv.w;
            Cell<float>::store4nt(reinterpret_cast<float *>(x) + (q << 2), xv);
        }
        if (activeLabel(l.x)) { rv.x = rv.x + (-alpha) * tv.x; acc += double(rv.x) * double(rv.x); big = fmaxf(big, fabsf(rv.x)); }
        if (activeLabel(l.y)) { rv.y = rv.y + (-alpha) * tv.y; acc += double(rv.y) * double(rv.y); big = fmaxf(big, fabsf(rv.y)); }
        if (activeLabel(l.z)) { rv.z = rv.z + (-alpha) * tv.z; acc += double(rv.z) * double(rv.z); big = fmaxf(big, fabsf(rv.z)); }
        if (activeLabel(l.w)) { rv.w = rv.w + (-alpha) * tv.w; acc += double(rv.w) * double(rv.w); big = fmaxf(big, fabsf(rv.w)); }
        reinterpret_cast<float4 *>(r)[q] = rv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t c = (nq << 2) + threadIdx.x;
        if (activeLabel(lab[c])) {
            x[c] = kWide ? XT(double(x[c]) + alphaD * double(p[c])) : XT((xFirst ? 0.f : float(x[c])) + alpha * p[c]);
            r[c] = r[c] + (-alpha) * t[c];
            acc += double(r[c]) * double(r[c]);
            big = fmaxf(big, fabsf(r[c]));
        }
    }
    const double total = blockReduce<1>(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
    if (maxPartials) {  // (uniform branch)
        __syncthreads();  // blockReduce's scratch is reused
        const double top = blockReduce<3>(double(big));
        if (threadIdx.x == 0) maxPartials[blockIdx.x] = top;
    }
}

int launchCgUpdate(void *stream, const GridP &g, float *x, const float *p, float *r, const float *t, float alpha, double *partials,
                   double *resultDev, const double *alphaDev, double *maxAbsDev, double *xWide, bool xFirst)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const unsigned nb = std::min<unsigned>(vecBlocks(g, n), unsigned(kReducePartials));
    double *maxPartials = maxAbsDev ? partials + kReducePartials : nullptr;  // (`partials` holds 2 x kReducePartials doubles)
    if (xWide) cgUpdateKernel<double><<<nb, 256, 0, s>>>(n, g.lab, xWide, p, r, t, alpha, partials, g.chunks, g.nchunks, g.chunkCells, alphaDev, maxPartials);
    else cgUpdateKernel<float><<<nb, 256, 0, s>>>(n, g.lab, x, p, r, t, alpha, partials, g.chunks, g.nchunks, g.chunkCells, alphaDev, maxPartials, xFirst ? 1 : 0);
    reduceFinalKernel<1><<<1, 256, 0, s>>>(int(nb), partials, resultDev);
    if (maxAbsDev) reduceFinalKernel<3><<<1, 256, 0, s>>>(int(nb), maxPartials, maxAbsDev);
    return int(hipGetLastError());
}

// ---- CG steps on the binary16 result of the mixed-precision V-cycle: z = (mul / *sigma) x~ is never written out ------
// <z, r> over the active cells
__global__ __launch_bounds__(256) void halfDotKernel(size_t n, const uint8_t *__restrict__ lab, const __half *__restrict__ xh, const float *__restrict__ r,
                                                     double *__restrict__ partials, const int32_t *__restrict__ chunks, int nchunks, int chunkCells)
{
    double acc = 0.0;
    const size_t nq = n >> 2;
    size_t q;
    for (size_t it = 0; nextQuad(chunks, nchunks, chunkCells, nq, it, q); ++it) {
        if (q >= nq) continue;
        const uchar4 l = reinterpret_cast<const uchar4 *>(lab)[q];
        const float4 xv = Cell<__half>::load4(xh + 4 * q), rv = reinterpret_cast<const float4 *>(r)[q];
        if (activeLabel(l.x)) acc += double(xv.x) * double(rv.x);
        if (activeLabel(l.y)) acc += double(xv.y) * double(rv.y);
        if (activeLabel(l.z)) acc += double(xv.z) * double(rv.z);
        if (activeLabel(l.w)) acc += double(xv.w) * double(rv.w);
    }
    const double total = blockReduce<0>(acc);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void reduceFinalScaledKernel(int nparts, const double *__restrict__ partials, double *__restrict__ result,
                                                               const float *__restrict__ sigma, float mul)
{
    double acc = 0.0;
    for (int p = threadIdx.x; p < nparts; p += blockDim.x) acc += partials[p];
    const double total = blockReduce<0>(acc);
    if (threadIdx.x == 0) *result = total * (double(mul) / double(*sigma));
}
int launchHalfDot(void *stream, const GridP &g, const void *xH, const float *r, const float *sigmaDev, float mul, double *partials, double *resultDev)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const unsigned nb = std::min<unsigned>(vecBlocks(g, n), unsigned(kReducePartials));
    halfDotKernel<<<nb, 256, 0, s>>>(n, g.lab, static_cast<const __half *>(xH), r, partials, g.chunks, g.nchunks, g.chunkCells);
    reduceFinalScaledKernel<<<1, 256, 0, s>>>(int(nb), partials, resultDev, sigmaDev, mul);
    return int(hipGetLastError());
}
// p = z + beta p on active cells (CG.h:191)
__global__ __launch_bounds__(256) void xpayHalfKernel(size_t n, const uint8_t *__restrict__ lab, float *__restrict__ p, const __half *__restrict__ xh,
                                                      const float *__restrict__ sigma, float mul, const float *__restrict__ betaDev, float betaHost,
                                                      const int32_t *__restrict__ chunks, int nchunks, int chunkCells)
{
    const float beta = betaDev ? *betaDev : betaHost, m = mul / *sigma;
    const size_t nq = n >> 2;
    size_t q;
    for (size_t it = 0; nextQuad(chunks, nchunks, chunkCells, nq, it, q); ++it) {
        if (q >= nq) continue;
        const uchar4 l = reinterpret_cast<const uchar4 *>(lab)[q];
        const float4 xv = Cell<__half>::load4(xh + 4 * q);
        float4 pv = reinterpret_cast<const float4 *>(p)[q];
        if (activeLabel(l.x)) pv.x = m * xv.x + beta * pv.x;
        if (activeLabel(l.y)) pv.y = m * xv.y + beta * pv.y;
        if (activeLabel(l.z)) pv.z = m * xv.z + beta * pv.z;
        if (activeLabel(l.w)) pv.w = m * xv.w + beta * pv.w;
        reinterpret_cast<float4 *>(p)[q] = pv;
    }
}
int launchXpayHalf(void *stream, const GridP &g, float *p, const void *xH, const float *sigmaDev, float mul, const float *betaDev, float betaHost)
{
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    xpayHalfKernel<<<vecBlocks(g, n), 256, 0, static_cast<hipStream_t>(stream)>>>(n, g.lab, p, static_cast<const __half *>(xH), sigmaDev, mul, betaDev, betaHost,
                                                                                 g.chunks, g.nchunks, g.chunkCells);
    return int(hipGetLastError());
}

// The scalars of the CG loop kept on the device so that only the convergence test meets the host:
// scal[0] = <z, r> of the current direction, scal[1] = <p, A p>, scal[3] = the <z, r> just reduced.
// init 1: scal[0] = scal[3]; init 2 (restart of the direction: p = z): the same and *beta = 0; else *beta = float(scal[3] / scal[0])
// (CG.h:180-191), then scal[0] = scal[3]
__global__ void cgScalarsKernel(double *__restrict__ scal, float *__restrict__ beta, int init)
{
    const double fresh = scal[3];
    if (!init) *beta = float(fresh / scal[0]);
    if (init == 2) *beta = 0.f;
    scal[0] = fresh;
}
int launchCgScalars(void *stream, double *scal, float *beta, int init)
{
    cgScalarsKernel<<<1, 1, 0, static_cast<hipStream_t>(stream)>>>(scal, beta, init);
    return int(hipGetLastError());
}

int launchReduce(void *stream, int kind, const GridP &g, const float *a, const float *b, double *partials,
                 double *resultDev)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n = size_t(g.nx) * g.ny * g.nz;
    const unsigned nb = std::min<unsigned>(vecBlocks(g, n), unsigned(kReducePartials));
    switch (kind) {
        case 0:
            reduceKernel<0><<<nb, 256, 0, s>>>(n, g.lab, a, b, partials, g.chunks, g.nchunks, g.chunkCells);
            reduceFinalKernel<0><<<1, 256, 0, s>>>(int(nb), partials, resultDev);
            break;
        case 1:
            reduceKernel<1><<<nb, 256, 0, s>>>(n, g.lab, a, nullptr, partials, g.chunks, g.nchunks, g.chunkCells);
            reduceFinalKernel<1><<<1, 256, 0, s>>>(int(nb), partials, resultDev);
            break;
        case 2:
            reduceKernel<2><<<nb, 256, 0, s>>>(n, g.lab, a, nullptr, partials, g.chunks, g.nchunks, g.chunkCells);
            reduceFinalKernel<2><<<1, 256, 0, s>>>(int(nb), partials, resultDev);
            break;
        default:
            reduceKernel<3><<<nb, 256, 0, s>>>(n, g.lab, a, nullptr, partials, g.chunks, g.nchunks, g.chunkCells);
            reduceFinalKernel<3><<<1, 256, 0, s>>>(int(nb), partials, resultDev);
            break;
    }
    return int(hipGetLastError());
}

}  // namespace mgps
