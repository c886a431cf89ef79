/*
 * mgps.h -- C ABI of the MI355X-native geometric multigrid pressure solver ("mgps").
 *
 * This is the drop-in boundary for the hot path of rgoldade/GeometricMultigridPressureSolver:
 * everything the Houdini plugin HDK_GeometricFreeSurfacePressureSolver::solveGasSubclass calls
 * between "build MG domain labels" and "apply solution to pressure"
 * (Source/HDK_GeometricFreeSurfacePressureSolver.cpp:344-362 and 426-629) is reachable through
 * the entry points below.  The reference has no C ABI of its own (it is C++ templates over HDK
 * UT_VoxelArray); each entry point cites the reference interface it replaces.  INTEGRATION.md
 * shows the binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - Grids are dense flat arrays, x fastest: index = (k*ny + j)*nx + i -- the logical order
 *    UT_VoxelArray exposes.  The face-weight grid of axis a has one more entry along a
 *    (MG.cpp:167-177): wx is (nx+1)*ny*nz, wy nx*(ny+1)*nz, wz nx*ny*(nz+1).
 *  - Labels are uint8 with the reference's values (HDK_GeometricMultigridOperators.h:11).
 *  - Storage is fp32 (the reference stores double; fp32/mixed precision is its README TO-DO).
 *    Scalars crossing the boundary (dot products, norms, tolerances) are double.
 *  - Pointers named *_dev are HIP device pointers on the solver's device; *_host are host
 *    pointers.  Every solution / rhs / residual grid must hold exactly 0 outside active
 *    (INTERIOR or BOUNDARY) cells, the invariant the reference asserts at
 *    HDK_GeometricMultigridOperators.h:821-823 and 950-953.
 *  - Every function returns an mgps_status; no exception crosses the boundary; nothing aborts.
 *    A handle is not thread-safe; distinct handles are independent (MG.h:35-52 has the same rule).
 *  - All device work is enqueued on the handle's stream (mgps_set_stream); functions that return
 *    a scalar to the host synchronise that stream, the others are asynchronous.
 */
#ifndef MGPS_H
#define MGPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGPS_VERSION 1

/* HDK::GeometricMultigridOperators::CellLabels (HDK_GeometricMultigridOperators.h:11) */
enum { MGPS_INTERIOR_CELL = 0, MGPS_EXTERIOR_CELL = 1, MGPS_DIRICHLET_CELL = 2, MGPS_BOUNDARY_CELL = 3 };

typedef enum mgps_status {
    MGPS_OK = 0,
    MGPS_ERR_INVALID_ARGUMENT = 1,
    MGPS_ERR_NO_DEVICE = 2,      /* no HIP device / HIP runtime failure at start-up */
    MGPS_ERR_HIP = 3,            /* a HIP call failed; see mgps_last_error */
    MGPS_ERR_ALLOC = 4,
    MGPS_ERR_HIERARCHY = 5,      /* no solvable cell / level cap left no level (MG.cpp:233-248) */
    MGPS_ERR_COARSE_TOO_LARGE = 6, /* coarsest level has more unknowns than the direct solver takes */
    MGPS_ERR_COARSE_FACTOR = 7,  /* coarsest matrix not positive definite (MG.cpp:411) */
    MGPS_ERR_COMM = 8,           /* multi-GPU exchange failed */
    MGPS_ERR_INTERRUPTED = 9,    /* the interrupt callback asked to stop */
    MGPS_ERR_INTERNAL = 10       /* a C++ exception other than std::bad_alloc (which maps to MGPS_ERR_ALLOC) was caught
                                    at the boundary; the call had no effect the caller can rely on */
} mgps_status;

/* outcome of mgps_solve_pcg, mirrors the early-outs of HDK_GeometricCGPoissonSolver.h:36-40, 60-64 */
enum { MGPS_PCG_CONVERGED = 0, MGPS_PCG_RHS_ZERO = 1, MGPS_PCG_ALREADY_CONVERGED = 2, MGPS_PCG_MAX_ITERATIONS = 3 };

typedef struct mgps_solver mgps_solver;       /* device-resident solver: the MG object + level storage */
typedef struct mgps_hierarchy mgps_hierarchy; /* host-only multigrid hierarchy (labels, bands, coarse factor) */

/* Compile-time constants of the reference exposed as options with the reference's values as
 * defaults: band width 3 / band iterations 3 (MG.cpp:141-142), omega = 2/3 (Ops.h:291, 554). */
typedef struct mgps_options {
    int struct_size;        /* sizeof(mgps_options), set by mgps_default_options */
    int band_width;         /* 3 */
    int band_iterations;    /* 3 */
    float jacobi_weight;    /* 2/3 */
    int device;             /* HIP device ordinal; -1 = current device */
    int print_stats;        /* doPrintStats of MG.h:24: per-stage timings on stdout */
    int max_coarse_unknowns;/* direct-solve cap, default 32768 (a 32^3 coarsest level); above 8192 unknowns -- or where the
                             * host's banded factor would cost more than 4e9 operations: a coarsest level larger than one
                             * 16^3 tile with a few thousand unknowns -- the solver factorises the dense matrix on the
                             * device (hipSOLVER) instead of on a host thread */
    int fuse_band_passes;   /* 1 (default) = run the band_iterations band-Jacobi passes of a level that is not cut
                               into slabs as one launch (same arithmetic per cell); 0 = one launch pair per pass */
    int deep_band_halo;     /* slab runs, 1 (default): the band stage of a cut level runs in the box form of the single-device
                               solver (one launch per stage) -- the cells of the neighbours' planes that the boxes next to
                               a cut read, up to band_iterations + 1 planes deep, travel packed (two list messages per
                               stroke, the first with the ghost plane) and live in the deep ghost planes of the grids
                               (mgps_ghost_planes) -- instead of a launch pair and an exchange per band pass (0).  The
                               ranks trade the index lists and the face weights of the planes next to each cut once, at
                               set-up */
    int min_cells_per_rank; /* slab runs: a level below the finest stays distributed only while every rank owns at least
                               this many cells of it (default 2097152 = 128^3); smaller levels are gathered to rank 0,
                               where one GPU finishes the cycle faster than 17 ghost exchanges per level cost */
    int pcg_fp64_vectors;   /* 0: the CG vectors x, r, p, A p are fp32 like every grid (the default of rounds 1-3).  1: mgps_solve_pcg keeps
                               them in fp64 (the reference's precision, MG.h:14-15) around the unchanged fp32 V-cycle;
                               x and b stay fp32 at the boundary.  Same iteration counts either way (measured at 512^3
                               and 1024^3); with fp64 vectors the residual recomputed at the end (CG.h:203-206) is a
                               true one instead of flooring at eps * cond (3e-3 at 512^3, 2e-2 at 1024^3 on the
                               free-surface case), for +15..21 % solve time at 512^3 (round 3: 16-byte accesses in the fp64 passes; +31..40 % before),
                               +30..36 % at 1024^3.  Slab runs exchange the ghost planes of
                               these vectors as doubles.  2 (round 4): the ITERATE alone in fp64 -- x += alpha p with p
                               widened, the residuals of CG.h:50-51 and 203-205 taken in fp64 from it, r, p, A p fp32 as in
                               mode 0: what holds the recomputed residual of mode 0 at eps * cond is the fp32 storage of x
                               (A fl(x) is eps |A| |x| away from A x whatever multiplies it) and, behind it, the drift of the
                               fp32 recurrence (5e-4 on the 512^3 pool), which the loop removes by REPLACING r with
                               float(b - A x) every 8 iterations and whenever the recurrence claims convergence
                               (residual replacement).  DEFAULT since round 4: same iteration counts (+-1), "Recomputed
                               relative L2 Error" (CG.h:203-206) equal to the recurrence's.  Round 5: between two replacements
                               the updates alpha p are summed in fp32 in the caller's x (group-wise update) and the fp64
                               grids are touched by the replacement passes only; a group ends after 8 updates, at a
                               convergence claim, or when the residual has dropped a hundredfold since it began, and a
                               replacement that finds the true residual more than twice the recurrence's restarts the
                               direction (p = z).  +1.5..3 % solve time at 512^3, 16 B per fine cell of memory.  Not with
                               precision = 1 (falls back to 0) */
    int (*interrupt)(void *user); /* non-zero stops the call with MGPS_ERR_INTERRUPTED (UT_Interrupt::opInterrupt, which the
                                     reference polls in every operator loop, e.g. Ops.h:319).  Polled before every PCG
                                     iteration and, on single-device solvers, before every level of both strokes of a
                                     V-cycle (host side: the device runs at most one cycle behind) */
    void *interrupt_user;
    /* full-domain smoother sweeps per stroke.  The reference hard-wires one (MG.cpp:466-486 down, 740-757 up): one
       damped-Jacobi sweep, or the two tile colours of Gauss-Seidel once each.  pre_sweeps applies to the down-stroke of
       every level, post_sweeps to the up-stroke; a stroke is band stage, the smoother repeated that many times
       (Gauss-Seidel keeps its colour / direction order inside each repetition), band stage.  Defaults 1 / 1.
       BASELINE config 1's "2+2 damped-Jacobi sweeps" is pre_sweeps = post_sweeps = 2 */
    int pre_sweeps, post_sweeps;
    /* which full-domain sweep kernel the Jacobi / residual / A.x passes use: 0 (default) = by size (the
       plane-marching kernel where an x-y plane exceeds 4 MiB, i.e. beyond 1024^2; the cache-served quad kernel up to there),
       1 = the quad kernel, 2 = the plane-marching kernel wherever its shape rule allows (nx >= 256, nx % 4 == 0,
       ny >= 16).  Same arithmetic per cell either way; for tuning and for parity tests of both kernels at small sizes */
    int stencil_path;
    /* 0 (default): every grid of the V-cycle is fp32.  1 = mixed precision (BASELINE config 5; the reference's README
       TO-DO, README.md:34-35): the fine level of the V-cycle -- 7/8 of its bytes -- keeps its iterate and its residual in
       binary16 (damped-Jacobi sweeps 9 instead of 13 B per cell, the restriction reads 2 instead of 4, the prolongation
       updates 4 instead of 8); the rhs, every coarser level, the CG vectors, A.p and all reductions stay fp32 and all
       arithmetic is fp32.  The cycle runs on the rhs normalised by a power of two (max |b| in (1/2, 1]) with fixed
       per-grid scales, so that binary16's range is used where the values are; results are returned in the caller's
       units.  A cycle from an initial guess runs as x + M (b - A x) with the residual in fp32 (iterative refinement:
       only corrections pass through binary16).  Single-device solvers whose fine nx is a multiple of 4; either smoother (the tiled
       Gauss-Seidel passes stage and sweep a tile in fp32 and round it once when it is written back).
       mgps_solve_pcg then preconditions with this cycle: tolerance and iteration counts against fp32 in DESIGN.md */
    int precision;
    /* 0 (default): single-device solvers build the hierarchy (coarse labels MG.cpp:238-253, band lists MG.cpp:279-281) and
       every list the kernels use on the device, from labels that never cross PCIe again -- the reference rebuilds its solver
       every sub-step (Plug.cpp:463), so set-up is on the critical path.  1 = the host builder (threads; the builder of slab
       runs, and the checker the tests compare the device arrays with, entry for entry).  Same solver either way */
    int host_setup;
    /* mgps_create_device / mgps_create_device_weights, 1: the solver reads the caller's three weight arrays in place --
       they must stay valid and unchanged until mgps_destroy -- instead of keeping its own copy (12 B per cell, 13 GB and
       5 ms at 1024^3).  0 (default): copy; the caller may release them right after the constructor returns */
    int borrow_device_weights;
} mgps_options;

typedef struct mgps_pcg_stats {
    int outcome;                   /* MGPS_PCG_* */
    int iterations;                /* value of `iteration` printed at CG.h:198 */
    double rel_residual;           /* "Drifted relative L2 Error", CG.h:199-200 */
    double rel_residual_recomputed;/* "Recomputed relative L2 Error", CG.h:203-206 */
    double rhs_norm2;
    double solve_ms;               /* device time of the solve, HIP events */
} mgps_pcg_stats;

void mgps_default_options(mgps_options *opt);
const char *mgps_status_string(int status);
/* Last error text of this handle (never NULL); mgps_last_error(NULL) returns the text of the last
 * failed call that had no handle (mgps_create, mgps_hierarchy_create, domain helpers). */
const char *mgps_last_error(const mgps_solver *h);
int mgps_device_count(int *count);
/* Introspection for tests and tools: the set-up arrays of level `level` as they sit on the device.  `which`:
 * 0 cell codes (u8, nx*ny*nz), 1 band list in device order (i32), 2 band diagonals (u8), 3 operator rows of the general
 * BOUNDARY cells (f32, 7 x count SoA), 4 activity chunks (i32), 5 plane blocks (i32), 6 / 7 pure tiles even / odd (i32),
 * 8 / 9 mixed tiles even / odd (i32), 10 per-tile start of the general BOUNDARY cells (i32), 11..13 the boxes of the fused
 * band stage: info (i32, 16 per group, in launch order), list entries (u32), general entries (i32, 2 per entry).
 * *count = number of elements; out == NULL asks for the count only. */
int mgps_level_array(mgps_solver *h, int level, int which, void *out, int64_t *count);

/* ---- domain expansion: host arrays --------------------------------------------------------
 * buildExpandedCellLabels sizing rule (Ops.h:1340-1362).  levels_in = 0 applies the reference rule
 * levels = ceil(log2(min res)) - 1; a positive levels_in keeps the caller's count.  power_of_two
 * != 0 rounds every extent up to a power of two as the reference does (Ops.h:1353-1360); 0 gives
 * the tight extents (multiples of 2^levels) that the operators actually need. */
int mgps_expanded_layout(int bnx, int bny, int bnz, int levels_in, int power_of_two,
                         int out_dims[3], int *out_offset, int *out_levels);
/* buildExpandedCellLabels copy step (Ops.h:1364-1453): EXTERIOR everywhere, non-EXTERIOR base
 * labels at +offset as INTERIOR or DIRICHLET. */
int mgps_expand_labels(uint8_t *expanded, const uint8_t *base, int bnx, int bny, int bnz,
                       int enx, int eny, int enz, int offset);
/* buildExpandedBoundaryWeights (Ops.h:1458-1572): zero, positive base weights at +offset. */
int mgps_expand_weights(float *expanded, const float *base, int axis, int bnx, int bny, int bnz,
                        int enx, int eny, int enz, int offset);
/* setBoundaryCellLabels (Ops.h:1574-1644), in place. */
int mgps_set_boundary_labels(uint8_t *labels, const float *wx, const float *wy, const float *wz,
                             int nx, int ny, int nz);
/* unitTestBoundaryCells / unitTestExteriorCells / unitTestCoarsening (Ops.h:1771-1870,
 * Ops.cpp:471-632): *pass = 1 when the invariant holds.  Weights may be NULL. */
int mgps_check_boundary_cells(const uint8_t *labels, const float *wx, const float *wy,
                              const float *wz, int nx, int ny, int nz, int *pass);
int mgps_check_exterior_cells(const uint8_t *labels, int nx, int ny, int nz, int *pass);
int mgps_check_coarsening(const uint8_t *coarse, const uint8_t *fine, int fnx, int fny, int fnz, int *pass);

/* ---- host-only hierarchy: the setup half of the GeometricMultigridPoissonSolver constructor
 * (MG.cpp:135-418) without touching a GPU: label coarsening with the early level cap
 * (Ops.cpp:23-163, MG.cpp:238-253), band lists (Ops.cpp:165-469), coarsest matrix + factor
 * (MG.cpp:288-411).  mgps_create builds one of these and uploads it. */
int mgps_hierarchy_create(mgps_hierarchy **out, int nx, int ny, int nz, const uint8_t *labels,
                          int mg_levels, const mgps_options *opt);
void mgps_hierarchy_destroy(mgps_hierarchy *hier);
int mgps_hierarchy_levels(const mgps_hierarchy *hier);
int mgps_hierarchy_level_dims(const mgps_hierarchy *hier, int level, int out_dims[3]);
int mgps_hierarchy_level_labels(const mgps_hierarchy *hier, int level, uint8_t *out);
int64_t mgps_hierarchy_band_count(const mgps_hierarchy *hier, int level);
/* band cells as (i,j,k) int32 triples, in the reference's order (tile id, k, j, i) */
int mgps_hierarchy_band_cells(const mgps_hierarchy *hier, int level, int32_t *out_ijk);
/* host self-check of the box form of the fused band stage (round 3; what single-device solvers run): builds the level's
 * boxes for `depth` passes, verifies their structure and replays the three uses of launchBandBox group by group against
 * pass-by-pass band smoothing and a full Jacobi sweep on a seeded grid, bit for bit: the plain stage, the closure stage
 * (band passes + the sweep's values on the band closure) and the plain stage fed from the closure stage's snapshot.
 * wx / wy / wz: optional face weights of level 0 (host arrays, the layout mgps_create takes) so that general BOUNDARY
 * cells take part; NULL = unit weights.  Returns the group count, the sum of the region sizes and the general entries. */
int mgps_hierarchy_check_band_boxes(const mgps_hierarchy *hier, int level, int depth, const float *wx, const float *wy, const float *wz,
                                    int64_t *out_groups, int64_t *out_region_cells, int64_t *out_general);
int mgps_hierarchy_coarse_unknowns(const mgps_hierarchy *hier);
/* x = A_coarsest^{-1} b on the host, grids of the coarsest level's size (MG.cpp:669-692) */
int mgps_hierarchy_coarse_solve(const mgps_hierarchy *hier, float *x, const float *b);

/* ---- the solver object -------------------------------------------------------------------
 * GeometricMultigridPoissonSolver(labels, weights[3], mgLevels, useGaussSeidel, doPrintStats)
 * (MG.h:20-24, MG.cpp:135-418).  labels / weights are HOST arrays and are copied (MG.cpp:164,
 * 179-180).  Padding contract: every level must keep a shell of EXTERIOR cells on all six sides (the reference
 * asserts unitTestExteriorCells per level, MG.cpp:235, 252), i.e. the solver grid needs at least 2^(mg_levels-1)
 * EXTERIOR cells on every side -- what buildExpandedCellLabels / mgps_expanded_layout pad (Ops.h:1347-1351); labels
 * that break it are refused with MGPS_ERR_HIERARCHY.  use_gauss_seidel selects the tile-coloured Gauss-Seidel smoother (the plugin
 * hard-wires it on, Plug.cpp:466); 0 selects damped Jacobi.
 * Weights contract: every liquid cell needs an open face (a diagonal > 0; the reference asserts it, Ops.h:354).  A cell without one
 * is accepted -- it becomes an operator row with diagonal 0 and divides by zero in every smoother, as in the reference's release
 * build -- and poisons the result with inf / NaN; nothing else depends on it (no index, no list). */
int mgps_create(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels_host,
                const float *wx_host, const float *wy_host, const float *wz_host, int mg_levels,
                int use_gauss_seidel, const mgps_options *opt);
/* The same solver from face weights that already live on the DEVICE (e.g. written by mgps_fields_boundary_weights):
 * labels stay a host array (the hierarchy is built on the host from them, 1 byte per cell), the weights never
 * cross to the host -- the operator rows of the BOUNDARY cells and the weight half of unitTestBoundaryCells are
 * evaluated by a kernel, the solver keeps a device-to-device copy.  Same results as mgps_create. */
int mgps_create_device_weights(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels_host,
                               const float *wx_dev, const float *wy_dev, const float *wz_dev, int mg_levels,
                               int use_gauss_seidel, const mgps_options *opt);
/* As above with the labels on the device too (mgps_fields_domain_labels + mgps_fields_set_boundary_labels write them
 * there): nothing of the size of the grid crosses PCIe -- the hierarchy and every list are built on the device
 * (options.host_setup = 1: the library fetches a host copy of the 1 byte per cell for the host-side builder). */
int mgps_create_device(mgps_solver **out, int nx, int ny, int nz, const uint8_t *labels_dev, const float *wx_dev,
                       const float *wy_dev, const float *wz_dev, int mg_levels, int use_gauss_seidel,
                       const mgps_options *opt);
void mgps_destroy(mgps_solver *h);
/* Set-up stages its lists through page-locked host blocks and keeps released ones for the next solver (a new one
 * every sub-step in the reference's use, Plug.cpp:463; cap: MGPS_PINNED_CACHE_MB, default 4096).  This returns
 * them to the system. */
void mgps_trim_host_cache(void);
/* Device memory released by a solver (or by mgps_project_free_surface) is kept for the next one: once a process holds
 * tens of GiB, hipMalloc costs 60-130 ms per 4 GiB block and every hipFree 0.2 ms on this platform, more than the solve
 * (cap: MGPS_DEVICE_CACHE_MB, default a quarter of the device memory, oldest blocks leave first; an allocation that fails trims the cache and tries again).  This returns
 * the cached blocks to the system, e.g. before another library needs the memory -- including the dense inverses of large
 * coarsest levels that the library keeps past their solvers (the two most recent label patterns per device; a live solver
 * keeps its own). */
void mgps_trim_device_cache(void);
/* Page-locked host memory for the caller's staging buffers (the flattened fields a Houdini shim uploads every
 * sub-step).  On this platform a hipMemcpy out of a fresh pageable array runs at about 3 GB/s, out of a page-locked one
 * at about 55 GB/s; blocks come from (and return to) the same cache as the library's own set-up arrays, so a
 * time-stepping caller pays the page-locking once.  NULL when no HIP device is usable or the allocation fails; the
 * library never requires its inputs to come from here. */
void *mgps_host_alloc(size_t bytes);
void mgps_host_free(void *p);
int mgps_levels(const mgps_solver *h);                       /* getMGLevels(), MG.h:31 */
int mgps_level_dims(const mgps_solver *h, int level, int out_dims[3]);
const mgps_hierarchy *mgps_get_hierarchy(const mgps_solver *h);
int mgps_set_stream(mgps_solver *h, void *hip_stream);       /* hipStream_t; NULL = default stream */
int mgps_synchronize(mgps_solver *h);

/* device grids of a level's size (level 0 = the solver grid) */
int mgps_grid_alloc(mgps_solver *h, int level, float **out_dev);  /* zero-filled */
int mgps_grid_free(mgps_solver *h, float *dev);
int mgps_grid_upload(mgps_solver *h, int level, float *dst_dev, const float *src_host);
int mgps_grid_download(mgps_solver *h, int level, float *dst_host, const float *src_dev);

/* applyVCycle(solution, rhs, useInitialGuess) (MG.h:26-29, MG.cpp:420-881) */
int mgps_apply_vcycle(mgps_solver *h, float *x_dev, const float *b_dev, int use_initial_guess);

/* ---- HDK::GeometricMultigridOperators on level `level` of the hierarchy (level 0 carries the
 * face weights, coarser levels use unit weights exactly as MG.cpp:447-451 vs 572-575) -------- */
/* jacobiPoissonSmoother (Ops.h:262-367), in place */
int mgps_jacobi_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev);
/* tiledGaussSeidelPoissonSmoother (Ops.h:369-520), in place */
int mgps_tiled_gs_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev,
                         int smooth_odd_tiles, int smooth_forward);
/* boundaryJacobiPoissonSmoother over the level's band list (Ops.h:524-619), in place */
int mgps_boundary_jacobi_smooth(mgps_solver *h, int level, float *x_dev, const float *b_dev);
/* the band stage of a smoothing stroke: options.band_iterations of those passes back to back
 * (MG.cpp:452-458, 483-489), as mgps_apply_vcycle runs them (one fused launch when
 * options.fuse_band_passes and the level is not cut into slabs) */
int mgps_boundary_jacobi_stage(mgps_solver *h, int level, float *x_dev, const float *b_dev);
/* applyPoissonMatrix (Ops.h:621-714).  Inactive cells of y are set to 0. */
int mgps_apply_poisson(mgps_solver *h, int level, float *y_dev, const float *x_dev);
/* computePoissonResidual (Ops.h:716-732): r = b - A x, 0 on inactive cells */
int mgps_residual(mgps_solver *h, int level, float *r_dev, const float *x_dev, const float *b_dev);
/* downsample (Ops.h:734-835): coarse (level+1) = Restrict(fine (level)) */
int mgps_downsample(mgps_solver *h, int fine_level, float *coarse_dev, const float *fine_dev);
/* coarse = downsample(b - A x) of level fine_level the way a down-stroke of the V-cycle forms it (MG.cpp:519-553: computePoissonResidual,
 * Ops.h:716-732, then downsample, Ops.h:734-835): as two passes over the level's residual grid, or -- where
 * mgps_residual_restrict_fused says so -- as the pair that folds the residual along z while it is formed and restricts in x-y
 * from there, without writing the residual.  coarse_dev: a grid of level fine_level + 1 (cleared first, Ops.h:756). */
int mgps_residual_downsample(mgps_solver *h, int fine_level, float *coarse_dev, const float *x_dev, const float *b_dev);

/* upsampleAndAdd (Ops.h:873-972): fine (level) += 4 * Trilerp(coarse (level+1)) */
int mgps_upsample_add(mgps_solver *h, int fine_level, float *fine_dev, const float *coarse_dev);
/* coarsest-level direct solve (MG.cpp:669-692) on device grids of the coarsest level */
int mgps_coarse_solve(mgps_solver *h, float *x_dev, const float *b_dev);
/* dotProduct (Ops.h:1020-1085), squaredL2Norm / l2Norm (1197-1265); fp64 accumulation */
int mgps_dot(mgps_solver *h, int level, const float *a_dev, const float *b_dev, double *out);
int mgps_squared_l2_norm(mgps_solver *h, int level, const float *a_dev, double *out);
int mgps_l2_norm(mgps_solver *h, int level, const float *a_dev, double *out);
/* infNorm (Ops.h:1267-1326).  reference_signed_max != 0 reproduces the reference exactly:
 * max(0, max_active v) with no absolute value; 0 gives the true max |v|. */
int mgps_inf_norm(mgps_solver *h, int level, const float *a_dev, int reference_signed_max, double *out);
/* addToVector: dst += scale*src (Ops.h:1087-1137); addVectors: dst = a + scale*s, dst may alias s
 * (Ops.h:1139-1195); scaleVector (Ops.h:974-1018).  Active cells only. */
int mgps_add_to_vector(mgps_solver *h, int level, float *dst_dev, const float *src_dev, double scale);
int mgps_add_vectors(mgps_solver *h, int level, float *dst_dev, const float *a_dev,
                     const float *scaled_dev, double scale);
int mgps_scale_vector(mgps_solver *h, int level, float *v_dev, double scale);

/* The operators sum neighbour values without looking at the neighbours' labels wherever the weights allow it (INTERIOR
 * and unit-weight BOUNDARY cells): they rely on the invariant stated at the top -- every grid handed in holds exactly 0
 * outside active cells (the reference asserts the same of its grids, Ops.h:821-823, 950-953, and builds them that way,
 * Plug.cpp:380, 406).  A caller whose field carries values in air or solid cells (a pressure field used as the initial
 * guess, say) restores it with this call first: grid = 0 on every cell of `level` that is not INTERIOR / BOUNDARY. */
int mgps_zero_inactive(mgps_solver *h, int level, float *grid_dev);

/* solveGeometricConjugateGradient (CG.h:18-207) with A = applyPoissonMatrix and
 * M^-1 = applyVCycle (use_mg_preconditioner != 0; Plug.cpp:461-484) or the diagonal
 * preconditioner (0; Plug.cpp:485-618).  x_dev holds the initial guess and receives the solution. */
int mgps_solve_pcg(mgps_solver *h, float *x_dev, const float *b_dev, double tolerance,
                   int max_iterations, int use_mg_preconditioner, mgps_pcg_stats *stats);

/* ---- multi-GPU: Z-slab partition of the fine grid -------------------------------------------
 * The reference is single-process shared memory (TBB over 16^3 tiles); it has no counterpart for
 * anything in this section.  One process per GPU.  The solver grid is cut along z (the slowest
 * axis, so a slab and a ghost plane are contiguous) into `size` slabs of nz/size planes; rank r owns
 * planes [r*nz/size, (r+1)*nz/size).  Level l stays distributed while its per-rank plane count is a
 * multiple of 16 (keeps the 16^3 Gauss-Seidel tile colouring identical to the single-GPU run and
 * restriction/prolongation rank-local up to one ghost plane) and every rank owns at least
 * options.min_cells_per_rank cells of it; the first level that fails either test, and everything
 * coarser, is gathered to rank 0 and solved there ("collapse").  A rank is set up on the device from
 * its window of the labels (round 5; options.host_setup = 1: the host builder).  A Jacobi stroke of a
 * cut level (options.deep_band_halo) costs two messages per neighbour: the ghost plane of the iterate
 * with, packed, iterate and rhs at the neighbour's cells the band boxes read, and the snapshot at
 * the same cells between the sweep and the second band stage; the whole-grid operators that read
 * across a cut (the residual, the restriction's residual planes, the prolongation's coarse
 * correction, a Gauss-Seidel colour pass) refresh the ghost plane first.  With deep_band_halo = 0
 * every band pass is preceded by an exchange.
 *
 * The transport is a small vtable so that the same orchestration runs over RCCL (production,
 * mgps_comm_create_rccl: ncclSend/ncclRecv pairs in one group on the solver's stream over xGMI) or
 * over anything else (the test-suite plugs torch.distributed/gloo in through it).
 * All pointers handed to exchange / gather / scatter are device pointers; `hip_stream` is the
 * solver's stream: an implementation either enqueues on it or synchronises it and blocks. */
struct mgps_xfer2;
typedef struct mgps_comm {
    int struct_size; /* sizeof(mgps_comm) as the caller was compiled: members appended later (allreduce_device) may be missing and read as NULL */
    int rank, size;
    void *user;
    /* send `send_lo` (send_lo_bytes) to rank-1 and receive recv_lo_bytes from it into `recv_lo`; the
     * same with rank+1 for *_hi.  The lo pair is NULL on rank 0, the hi pair on the last rank.  The
     * two directions of a pair may differ in size (packed band cells of two different planes). */
    int (*exchange)(void *user, const void *send_lo, size_t send_lo_bytes, void *recv_lo, size_t recv_lo_bytes,
                    const void *send_hi, size_t send_hi_bytes, void *recv_hi, size_t recv_hi_bytes,
                    void *hip_stream);
    /* in-place all-reduce of `count` host doubles; op 0 = sum, 1 = max */
    int (*allreduce)(void *user, double *values, int count, int op);
    /* root receives size*bytes (rank order) into recv_dev; the others pass recv_dev = NULL */
    int (*gather)(void *user, const void *send_dev, void *recv_dev, size_t bytes, int root, void *hip_stream);
    /* root sends chunk r of send_dev to rank r; everybody receives `bytes` into recv_dev */
    int (*scatter)(void *user, const void *send_dev, void *recv_dev, size_t bytes, int root, void *hip_stream);
    void (*destroy)(void *user);
    /* The same two with a share per rank (slabs balanced by active planes differ in size; NULL is allowed as long as all
     * slabs are equal).  gatherv: every rank sends send_bytes; root receives counts[r] bytes from rank r at byte offset
     * displs[r] of recv_dev (counts / displs are host arrays, read on root only).  scatterv: root sends counts[r] bytes from
     * offset displs[r] of send_dev to rank r; every rank receives recv_bytes into recv_dev. */
    int (*gatherv)(void *user, const void *send_dev, size_t send_bytes, void *recv_dev, const size_t *counts, const size_t *displs,
                   int root, void *hip_stream);
    int (*scatterv)(void *user, const void *send_dev, const size_t *counts, const size_t *displs, void *recv_dev, size_t recv_bytes,
                    int root, void *hip_stream);
    /* in-place all-reduce of `count` DEVICE doubles, ordered on hip_stream with the kernels around it (op as above).  With it
     * the slab PCG keeps alpha and beta on the device like the single-device loop: the scalars <p, A p> and <z, r> never meet
     * the host and |r|^2 -- the convergence test -- is the one host round trip of an iteration.  NULL is allowed: the loop then
     * sums every scalar through `allreduce` (three round trips per iteration). */
    int (*allreduce_device)(void *user, double *values_dev, int count, int op, void *hip_stream);
    /* `exchange` with two segments per message (round 4; NULL is allowed: the solver then packs everything into one buffer).
     * The band-stage message of a cut level is the boundary plane of x -- 4 MiB of its ~4.5 MiB at 1024^3, contiguous in the
     * grid -- and the packed closure lists: with this entry the plane travels straight from / into the grids and only the lists
     * are packed.  Each argument describes one message: segment 0, then segment 1, sent / received back to back in that order
     * (a segment of 0 bytes is skipped); NULL = no neighbour on that side.  All four in ONE group, like `exchange`. */
    int (*exchange2)(void *user, const struct mgps_xfer2 *send_lo, const struct mgps_xfer2 *recv_lo, const struct mgps_xfer2 *send_hi,
                     const struct mgps_xfer2 *recv_hi, void *hip_stream);
} mgps_comm;
typedef struct mgps_xfer2 {
    void *ptr[2]; /* device pointers (read-only for a send) */
    size_t bytes[2];
} mgps_xfer2;

/* RCCL transport.  Rank 0 calls mgps_rccl_unique_id and ships the 128 bytes to the other ranks by
 * any means (the bench uses torch.distributed.broadcast); then every rank calls
 * mgps_comm_create_rccl, which runs ncclCommInitRank on `device`. */
int mgps_rccl_unique_id(unsigned char out_id[128]);
int mgps_comm_create_rccl(mgps_comm *out, int rank, int size, const unsigned char id[128], int device);
/* diagnostic: one ncclSend + ncclRecv of `floats` floats from this rank to itself in a single group (the call
 * shape of the ghost exchange) and a comparison of what arrived -- librccl's point-to-point path checked on any
 * box, one GPU is enough */
int mgps_comm_rccl_selftest(mgps_comm *comm, size_t floats);
/* diagnostic: device time (microseconds per group, HIP events) of `reps` back-to-back self send + receive
 * groups of `floats` floats -- the floor of one ghost exchange on this box */
int mgps_comm_rccl_selfbench(mgps_comm *comm, size_t floats, int reps, double *us_per_exchange);
/* A check of any transport before a run trusts it (bench.py --gpus N calls it before its timed region): every rank pushes
 * rank-stamped data through exchange, exchange2, allreduce_device, allreduce, gather and scatter and verifies what arrives from
 * its neighbours; the ranks leave with one verdict (MGPS_OK, or MGPS_ERR_COMM with the first finding in mgps_last_error(NULL)).
 * *ranks_seen: the number of ranks as the (device) all-reduce counted them.  Collective: every rank calls it. */
int mgps_comm_preflight(const mgps_comm *comm, size_t floats, int *ranks_seen);
void mgps_comm_destroy(mgps_comm *comm);

/* The slab form of the constructor.  labels_global_host: the WHOLE solver grid's labels
 * (nx*ny*nz_global bytes, they are small); the face weights only for this rank's slab: wx / wy hold
 * the owned planes, wz the owned planes plus the closing face plane (nz_slab + 1 planes).
 * Grids passed to the operators of a slab solver hold the owned planes and must come from
 * mgps_grid_alloc (which surrounds them with mgps_ghost_planes(h) ghost planes on either side).  The vtable is copied; the
 * transport state behind `comm->user` stays owned by the caller, who destroys it (mgps_comm_destroy)
 * after the solver. */
int mgps_create_slab(mgps_solver **out, int nx, int ny, int nz_global, const uint8_t *labels_global_host,
                     const float *wx_slab, const float *wy_slab, const float *wz_slab, int mg_levels,
                     int use_gauss_seidel, const mgps_options *opt, const mgps_comm *comm);
/* The cuts of a slab run.  out_splits[0 .. size]: rank r owns the fine planes [out_splits[r], out_splits[r + 1]).  With the
 * Jacobi smoother the cuts minimise the largest per-rank load (whole planes of the collapse level, at least 16 fine planes
 * apiece), load = active cells + 30 x BOUNDARY cells (the three-deep band costs ten times a cell of the full sweeps) + the
 * collapsed tail on rank 0: the EXTERIOR padding of the expanded grid (2^(levels-1) planes at either end) would otherwise
 * leave the first and the last rank half empty and the middle ranks with 1 / (size - 1) of the work each.  With Gauss-Seidel every cut must be a multiple of 16 planes of EVERY distributed level (the tile colouring), which
 * leaves the even cut.  Every rank must pass the same arguments and gets the same cuts. */
int mgps_slab_partition(int nx, int ny, int nz, const uint8_t *labels_global_host, int mg_levels, int size, int use_gauss_seidel,
                        const mgps_options *opt, int *out_splits);
/* mgps_create_slab with explicit cuts (from mgps_slab_partition, or the caller's own: every cut an even plane count, a
 * multiple of 16 with Gauss-Seidel, at least 16 planes per rank).  The slab weights cover this rank's planes
 * [splits[rank], splits[rank + 1]).  Slabs of different sizes need comm->gatherv / scatterv. */
int mgps_create_slab_ranges(mgps_solver **out, int nx, int ny, int nz_global, const uint8_t *labels_global_host,
                            const float *wx_slab, const float *wy_slab, const float *wz_slab, int mg_levels,
                            int use_gauss_seidel, const mgps_options *opt, const mgps_comm *comm, const int *splits);
/* The same with the slab's face weights already on the rank's DEVICE (what mgps_fields_build_* leave there): no weight
 * crosses PCIe -- at 1024^3 on 8 ranks the 1.6 GB of a slab's weights arriving from pageable host memory were 300 of the
 * 470 ms a rank's set-up took -- and the rows of the slab's BOUNDARY cells are evaluated on the device (Ops.h:208-256, as
 * mgps_create_device_weights does for a whole grid).  The arrays are copied unless options.borrow_device_weights is set
 * (then they must outlive the solver).  Same result as mgps_create_slab_ranges, bit for bit. */
int mgps_create_slab_device_weights(mgps_solver **out, int nx, int ny, int nz_global, const uint8_t *labels_global_host,
                                    const float *wx_slab_dev, const float *wy_slab_dev, const float *wz_slab_dev, int mg_levels,
                                    int use_gauss_seidel, const mgps_options *opt, const mgps_comm *comm, const int *splits);
/* owned plane range [z0, z1) of `level` on this rank (levels past the distributed ones: the range
 * of the collapse level) and the number of distributed levels */
int mgps_slab_range(const mgps_solver *h, int level, int *z0, int *z1);
int mgps_distributed_levels(const mgps_solver *h);
/* Planes every grid handed to this solver must carry below and above its owned planes (mgps_grid_alloc provides them): 1 for a
 * single-device solver (unused) and for a slab rank built on the host (the ghost plane); 5 for a slab rank set up on the device
 * (the default since round 5): besides the ghost plane the neighbours' cells that the band boxes next to a cut read live there,
 * up to band_iterations + 1 planes deep.  A caller that allocates slab grids itself (the Python front end does) pads by this. */
int mgps_ghost_planes(const mgps_solver *h);
/* how the band stage (the `band_iterations` boundary-Jacobi passes of Ops.h:524-619 around a full-domain sweep) of level `level`
 * runs: *form = 1: the box form (one launch per stage; on a cut level of a slab run with two list messages per stroke), 0: pass by
 * pass (a launch pair per pass, on cut levels an exchange per pass).  The same arithmetic per cell either way. */
int mgps_band_stage_form(const mgps_solver *h, int level, int *form);
/* *fused = 1 when a down-stroke of level `level` takes residual and restriction as the pair "residual folded along z as it is
 * formed + x-y restriction" (no residual grid written; Ops.h:716-732 into Ops.h:734-835), 0 when they run as two passes over
 * the residual grid.  By size (x-y planes >= 4 MiB) on levels whose shape the pair takes; MGPS_FUSE_RR=0 / 1 forces it off /
 * onto every level that fits (tests).  The same products either way, added along z first instead of last. */
int mgps_residual_restrict_fused(const mgps_solver *h, int level, int *fused);
/* kept for callers of round 4's ABI: always 0.  Round 4 queued the ghost exchanges that follow a sweep on a second stream beside
 * the sweep's interior part; round 5 runs every exchange on the solver's stream until a first run on real links has shown
 * where the time goes. */
int64_t mgps_overlapped_exchanges(const mgps_solver *h);
/* slab runs: ghost / band-stage exchanges this rank issued so far (each one message pair per neighbour); 0 on single-device solvers */
int64_t mgps_exchange_count(const mgps_solver *h);
/* raw copies between host memory and device memory of the solver's device (used by transports
 * that stage through the host) */
int mgps_copy_to_host(mgps_solver *h, void *dst_host, const void *src_dev, size_t bytes);
int mgps_copy_to_device(mgps_solver *h, void *dst_dev, const void *src_host, size_t bytes);

/* ---- measurement hooks (the reference's UT_StopWatch scopes, MG.cpp:461-492 "Smoother time") ----
 * While enabled, HIP events bracket the kernels of every fine-level full-domain smoother sweep inside
 * mgps_apply_vcycle (the Jacobi sweep; each of the two tile-coloured Gauss-Seidel half sweeps) -- the kernels only:
 * on slab runs the ghost exchange in front of a sweep stays outside the bracket.
 * mgps_profile_read synchronises, returns the accumulated device time and the number of full sweeps
 * (a Gauss-Seidel sweep = its two colours) since the last read, and resets both. */
int mgps_profile_enable(mgps_solver *h, int enable);  /* 0 off, 1 fine-smoother events, 2 = 1 + per-stage events */
/* Per-stage device time of the V-cycles run since the last call (or since mgps_profile_enable), summed over levels, the
 * reference's stopwatch scopes (MG.cpp:436-878): [0] boundary smoother, [1] smoother, [2] compute residual, [3] downsample,
 * [4] direct solve (slab runs: gather + collapsed tail + scatter), [5] upsample-and-add; *cycles = cycles they cover.  Recorded
 * after mgps_profile_enable(h, 2) or while options.print_stats is set (print_stats also prints every stage of every level
 * per cycle, as doPrintStats does); about 160 event records per cycle, which cost a cycle a few percent -- keep it out of
 * timed regions.  Synchronises the stream.  Each stage is bracketed by an event pair, so on small levels the
 * figures include the few microseconds between dependent launches. */
int mgps_stage_times(mgps_solver *h, double out_ms[6], int *cycles);
/* the level-0 share of what the last mgps_stage_times call returned (same six stages, same cycles) */
int mgps_stage_times_fine(mgps_solver *h, double out_ms[6]);
int mgps_profile_read(mgps_solver *h, double *fine_smoother_ms, int *fine_smoother_launches);
/* Cells one full-domain sweep of `level` visits: the kernels skip 1024-cell chunks / 256x16xzc blocks /
 * 16^3 tiles without active cells (the reference skips constant tiles the same way, Ops.h:300-312) and, inside the
 * runs / blocks they do visit, the quads outside the level's active x range (the EXTERIOR padding at the row ends);
 * counted on the device from the list the sweep walks (one small launch and a synchronisation per call);
 * stencil_cells is for the Jacobi / residual / apply sweep, gs_cells for the two tiled-GS half sweeps.
 * This is the cell count behind bench.py's algorithmic bytes per launch. */
int mgps_swept_cells(const mgps_solver *h, int level, long long *stencil_cells, long long *gs_cells);
/* Which kernel the Jacobi / residual / A.x sweeps of `level` launch (options.stencil_path and the level's shape
 * decide): 1 = the cache-served quad kernel, 2 = the plane-marching kernel, 3 = the scalar kernel (nx % 4 != 0). */
int mgps_stencil_kernel(const mgps_solver *h, int level, int *kernel);

/* Host-buffer convenience forms: what the Houdini shim calls (upload, run, download). */
int mgps_apply_vcycle_host(mgps_solver *h, float *x_host, const float *b_host, int use_initial_guess);
int mgps_solve_pcg_host(mgps_solver *h, float *x_host, const float *b_host, double tolerance,
                        int max_iterations, int use_mg_preconditioner, mgps_pcg_stats *stats);
/* The same with the reference's storage type on the host side (StoreReal = double, MG.h:14-15, Plug.h:18-19): the
 * doubles cross PCIe as they are and are narrowed / widened on the device, so a Houdini caller hands over its
 * UT_VoxelArray<double> contents without a host-side conversion pass. */
int mgps_apply_vcycle_host_f64(mgps_solver *h, double *x_host, const double *b_host, int use_initial_guess);
int mgps_solve_pcg_host_f64(mgps_solver *h, double *x_host, const double *b_host, double tolerance,
                            int max_iterations, int use_mg_preconditioner, mgps_pcg_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* MGPS_H */
