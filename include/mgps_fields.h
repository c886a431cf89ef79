/* mgps_fields.h -- C ABI of the plugin-side field pre/post-processing around the multigrid solve, on the
 * device.  These are the per-cell / per-face passes that
 * HDK_GeometricFreeSurfacePressureSolver::solveGasSubclass runs on the host before and after the solve
 * (reference: Source/HDK_GeometricFreeSurfacePressureSolver.cpp = "Plug.cpp", Source/HDK_Utilities.* =
 * "Util.h/.cpp"); with them the label / weight / right-hand-side grids are produced where the solver
 * consumes them and the projected velocity is produced from the pressure without a round trip.
 *
 * All pointers are DEVICE pointers; grids are dense, x fastest.  Base grid g = (gx, gy, gz): the simulation
 * grid.  Expanded grid e = (ex, ey, ez): the solver grid of mgps_expanded_layout, base cell c lives at
 * c + offset.  The face grid of axis a has one more entry along a; face f of axis a lies between cells
 * f - e_a (backward) and f (forward).  Material labels (int32): 0 SOLID, 1 LIQUID, 2 AIR (Util.h:17).
 * HDK samples the solid SDF and the solid velocity by interpolation at a position (Util.cpp:25,
 * Plug.cpp:925); here the caller passes them sampled at cell centres / face centres.
 * `stream` is a hipStream_t (NULL = the null stream).  Return values: the status codes of mgps.h. */
#ifndef MGPS_FIELDS_H
#define MGPS_FIELDS_H

#include <stdint.h>

#include "mgps.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HDK::Utilities::buildMaterialCellLabels + isCellLiquid (Util.cpp:5-45, 87-148) */
int mgps_fields_material_labels(int32_t *material, const float *liquid_phi, const float *solid_phi, const float *cwx,
                                const float *cwy, const float *cwz, int gx, int gy, int gz, void *stream);
/* buildValidFaces / classifyValidFaces (Plug.cpp:716-744, Util.h:140-195): valid[f] = 1 where the cut-cell
 * weight is positive, both cells exist and one of them is LIQUID */
int mgps_fields_valid_faces(int axis, uint8_t *valid, const int32_t *material, const float *cut_weights, int gx, int gy,
                            int gz, void *stream);
/* buildMGDomainLabels (Plug.cpp:746-793) written straight into the expanded label grid
 * (buildExpandedCellLabels copy step, Ops.h:1364-1453): LIQUID -> INTERIOR, AIR -> DIRICHLET, else EXTERIOR */
int mgps_fields_domain_labels(uint8_t *expanded_labels, const int32_t *material, int gx, int gy, int gz, int ex, int ey,
                              int ez, int offset, void *stream);
/* buildMGBoundaryWeights (Plug.cpp:795-865) written straight into the expanded face grid
 * (buildExpandedBoundaryWeights, Ops.h:1524-1571): cut-cell weight on valid faces, divided by
 * clamp(theta, 0.01, 1) (ghost fluid, Util.h:25-42) on liquid/air faces, 0 elsewhere */
int mgps_fields_boundary_weights(int axis, float *expanded_weights, const float *cut_weights, const float *liquid_phi,
                                 const uint8_t *valid, const int32_t *material, int gx, int gy, int gz, int ex, int ey,
                                 int ez, int offset, void *stream);
/* setBoundaryCellLabels (Ops.h:1574-1644) on the device: same rule as mgps_set_boundary_labels */
int mgps_fields_set_boundary_labels(uint8_t *expanded_labels, const float *wx, const float *wy, const float *wz, int ex,
                                    int ey, int ez, void *stream);
/* buildRHS (Plug.cpp:867-943): weighted velocity divergence (plus the solid-velocity flux through the closed
 * part of each face when solid velocities are given; svx/svy/svz may all be NULL) at the expanded position
 * of every LIQUID cell, 0 elsewhere */
int mgps_fields_rhs(float *expanded_rhs, const int32_t *material, const float *vx, const float *vy, const float *vz,
                    const float *svx, const float *svy, const float *svz, const float *cwx, const float *cwy,
                    const float *cwz, int gx, int gy, int gz, int ex, int ey, int ez, int offset, void *stream);
/* applyOldPressure (Plug.cpp:945-997): expanded solution grid = old pressure at LIQUID cells, 0 elsewhere */
int mgps_fields_pressure_to_solution(float *expanded_x, const float *pressure, const int32_t *material, int gx, int gy,
                                     int gz, int ex, int ey, int ez, int offset, void *stream);
/* applySolutionToPressure (Plug.cpp:999-1047): pressure at LIQUID cells = solution; other cells untouched */
int mgps_fields_solution_to_pressure(float *pressure, const float *expanded_x, const int32_t *material, int gx, int gy,
                                     int gz, int ex, int ey, int ez, int offset, void *stream);
/* applyPressureGradient (Plug.cpp:1049-1131): velocity -= grad p on valid faces, ghost-fluid scaled on
 * liquid/air faces */
int mgps_fields_pressure_gradient(int axis, float *velocity, const float *liquid_phi, const float *pressure,
                                  const uint8_t *valid, const int32_t *material, int gx, int gy, int gz, void *stream);
/* computeResultingDivergence (Plug.cpp:1133-1207): out_host[3] = {sum, max (starting from 0), LIQUID cell
 * count} of the weighted divergence over LIQUID cells; synchronises the stream */
int mgps_fields_divergence(double out_host[3], const int32_t *material, const float *vx, const float *vy, const float *vz,
                           const float *svx, const float *svy, const float *svz, const float *cwx, const float *cwy,
                           const float *cwz, int gx, int gy, int gz, void *stream);

/* ---- the whole of solveGasSubclass between "fields fetched" and "fields written back" in one call ------------------
 * HDK_GeometricFreeSurfacePressureSolver::solveGasSubclass (Plug.cpp:252-707) on HOST arrays: what the Houdini shim
 * (host/HDK_GeometricFreeSurfacePressureSolver.cpp) calls after flattening the SIM fields.  Uploads the inputs, runs
 * the device passes above in the reference's order -- material labels (Plug.cpp:270), valid faces (286), MG domain
 * labels + boundary weights + expansion + boundary labels (316-362), rhs (386), warm start (413) -- builds the
 * multigrid solver from the device grids, runs solveGeometricConjugateGradient (426-629), then pressure write-back,
 * pressure gradient and the divergence report (637-707), and downloads pressure, velocity and the valid-face flags.
 * Arrays are dense, x fastest, of `real_bytes` = 4 (float) or 8 (double: converted on the device); face grids have one
 * more entry along their axis.  The solid SDF and the solid velocity are passed sampled at cell / face centres. */
typedef struct mgps_projection {
    int struct_size;               /* sizeof(mgps_projection) */
    int gx, gy, gz;                /* simulation grid */
    int real_bytes;                /* 4 or 8: the type behind every `void *` real array below */
    const void *liquid_phi;        /* cell grid: liquid SDF ("surface") */
    const void *solid_phi;         /* cell grid: solid SDF ("collision") */
    const void *cut_weights[3];    /* face grids ("cutCellWeights") */
    void *velocity[3];             /* face grids, in: velocity, out: projected velocity */
    const void *solid_velocity[3]; /* face grids ("collisionvelocity"), or all NULL */
    void *pressure;                /* cell grid, in: previous pressure (read when use_old_pressure), out: pressure */
    uint8_t *valid_faces[3];       /* out, face grids: 1 = valid face (each may be NULL: not wanted) */
    int use_old_pressure;          /* "useOldPressure" */
    int use_mg_preconditioner;     /* "useMGPreconditioner" */
    int use_gauss_seidel;          /* 1 = the plugin's choice (Plug.cpp:466) */
    double tolerance;              /* SIM_NAME_TOLERANCE */
    int max_iterations;            /* "maxIterations" */
    int power_of_two;              /* 1 = the reference's expansion (Ops.h:1353-1360), 0 = tight extents */
    /* results */
    mgps_pcg_stats stats;
    int mg_levels, offset, expanded[3];
    double liquid_cells;
    double residual_inf, residual_l2;       /* of the computed solution (Plug.cpp:625-628; infNorm is the signed max) */
    double divergence_sum, divergence_max;  /* after the projection (Plug.cpp:704-706); divergence_count = liquid_cells */
    double setup_ms, solve_ms, total_ms;    /* host wall clock: everything before the solve / the solve / the whole call */
} mgps_projection;
/* status MGPS_ERR_HIERARCHY with outcome MGPS_PCG_RHS_ZERO-like early outs are reported through stats.outcome; a domain
 * without liquid returns MGPS_OK with liquid_cells = 0 and leaves velocity and pressure untouched */
int mgps_project_free_surface(mgps_projection *p, const mgps_options *opt);

#ifdef __cplusplus
}
#endif
#endif
