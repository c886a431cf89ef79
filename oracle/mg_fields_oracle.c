/*
 * mg_fields_oracle.c -- CPU restatement (fp64, flat arrays) of the plugin-side field pre/post-processing
 * around the multigrid solve: material labels, valid faces, multigrid domain labels
 * and boundary weights, right-hand side, pressure copy in / out, pressure-gradient update and the
 * post-projection divergence report.  TEST INFRASTRUCTURE ONLY, like mg_oracle.c: nothing in the product
 * path may include, link, import or call this file.  PARITY UNPINNED for the same reason (no fixtures in the
 * reference, HDK not buildable here); pinned by properties in tests/test_fields.py (the projected velocity
 * is divergence-free to solver tolerance -- the reference's own check, Plug.cpp:704-706).
 *
 * "Plug.cpp" = Source/HDK_GeometricFreeSurfacePressureSolver.cpp, "Util.h/.cpp" = Source/HDK_Utilities.*.
 * Grids are dense, x fastest; the face grid of axis a has one more entry along a; face f of axis a lies
 * between cells f - e_a (backward) and f (forward).  Material labels: 0 SOLID, 1 LIQUID, 2 AIR (Util.h:17).
 * HDK samples the solid SDF / solid velocity by interpolation at a position (Util.cpp:25, Plug.cpp:925); here
 * the caller passes them already sampled at cell centres / face centres.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

enum { MAT_SOLID = 0, MAT_LIQUID = 1, MAT_AIR = 2 };
enum { LAB_INTERIOR = 0, LAB_EXTERIOR = 1, LAB_DIRICHLET = 2, LAB_BOUNDARY = 3 };

static size_t cell(int gx, int gy, int i, int j, int k) { return ((size_t)k * gy + j) * gx + i; }
/* face (i,j,k) of axis a: grid extents g + e_a */
static size_t face(int gx, int gy, int axis, int i, int j, int k)
{
    const int fx = gx + (axis == 0), fy = gy + (axis == 1);
    return ((size_t)k * fy + j) * fx + i;
}
static size_t cell_face(int gx, int gy, int axis, int dir, int i, int j, int k) /* cellToFaceMap */
{
    return face(gx, gy, axis, i + (axis == 0 && dir), j + (axis == 1 && dir), k + (axis == 2 && dir));
}
static double ghost_fluid_theta(double phi0, double phi1) /* Util.h:25-42, clamp of Plug.cpp:850-851 */
{
    double theta = 0;
    if (phi0 < 0) {
        if (phi1 < 0) theta = 1;
        else theta = phi0 / (phi0 - phi1);
    } else if (phi1 < 0)
        theta = phi1 / (phi1 - phi0);
    return theta < 0.01 ? 0.01 : (theta > 1 ? 1 : theta);
}

/* buildMaterialCellLabels + isCellLiquid (Util.cpp:5-45, 87-148) */
void mgf_material_labels(int32_t *material, const double *liquid_phi, const double *solid_phi, const double *cwx,
                         const double *cwy, const double *cwz, int gx, int gy, int gz)
{
    const double *cw[3] = {cwx, cwy, cwz};
    const int g[3] = {gx, gy, gz};
    for (int k = 0; k < gz; ++k)
        for (int j = 0; j < gy; ++j)
            for (int i = 0; i < gx; ++i) {
                const size_t c = cell(gx, gy, i, j, k);
                int in_fluid = 0;
                for (int a = 0; a < 3; ++a)
                    for (int d = 0; d < 2; ++d)
                        if (cw[a][cell_face(gx, gy, a, d, i, j, k)] > 0) in_fluid = 1;
                material[c] = MAT_SOLID;
                if (!in_fluid) continue;
                int liquid = liquid_phi[c] <= 0;
                if (!liquid && solid_phi[c] >= 0)
                    for (int a = 0; a < 3 && !liquid; ++a)
                        for (int d = 0; d < 2; ++d) {
                            if (!(cw[a][cell_face(gx, gy, a, d, i, j, k)] > 0)) continue;
                            int n[3] = {i, j, k};
                            n[a] += d ? 1 : -1;
                            if (n[a] < 0 || n[a] >= g[a]) continue;
                            if (liquid_phi[cell(gx, gy, n[0], n[1], n[2])] <= 0) {
                                liquid = 1;
                                break;
                            }
                        }
                material[c] = liquid ? MAT_LIQUID : MAT_AIR;
            }
}

/* buildValidFaces / classifyValidFaces (Plug.cpp:716-744, Util.h:140-195) */
void mgf_valid_faces(int axis, uint8_t *valid, const int32_t *material, const double *cw, int gx, int gy, int gz)
{
    const int f[3] = {gx + (axis == 0), gy + (axis == 1), gz + (axis == 2)}, g[3] = {gx, gy, gz};
    for (int k = 0; k < f[2]; ++k)
        for (int j = 0; j < f[1]; ++j)
            for (int i = 0; i < f[0]; ++i) {
                const size_t fc = face(gx, gy, axis, i, j, k);
                int b[3] = {i, j, k}, fw[3] = {i, j, k};
                b[axis] -= 1;
                valid[fc] = 0;
                if (!(cw[fc] > 0) || b[axis] < 0 || fw[axis] >= g[axis]) continue;
                if (material[cell(gx, gy, b[0], b[1], b[2])] == MAT_LIQUID || material[cell(gx, gy, fw[0], fw[1], fw[2])] == MAT_LIQUID)
                    valid[fc] = 1;
            }
}

/* buildMGDomainLabels (Plug.cpp:746-793) written straight into the expanded grid (Ops.h:1364-1453) */
void mgf_domain_labels(int32_t *expanded, const int32_t *material, int gx, int gy, int gz, int ex, int ey, int ez, int offset)
{
    for (size_t c = 0; c < (size_t)ex * ey * ez; ++c) expanded[c] = LAB_EXTERIOR;
    for (int k = 0; k < gz; ++k)
        for (int j = 0; j < gy; ++j)
            for (int i = 0; i < gx; ++i) {
                const int m = material[cell(gx, gy, i, j, k)];
                if (m == MAT_SOLID) continue;
                expanded[cell(ex, ey, i + offset, j + offset, k + offset)] = m == MAT_LIQUID ? LAB_INTERIOR : LAB_DIRICHLET;
            }
}

/* buildMGBoundaryWeights (Plug.cpp:795-865) written straight into the expanded face grid (Ops.h:1524-1571) */
void mgf_boundary_weights(int axis, double *expanded_w, const double *cw, const double *liquid_phi, const uint8_t *valid,
                          const int32_t *material, int gx, int gy, int gz, int ex, int ey, int ez, int offset)
{
    const int f[3] = {gx + (axis == 0), gy + (axis == 1), gz + (axis == 2)};
    const size_t en = (size_t)(ex + (axis == 0)) * (ey + (axis == 1)) * (ez + (axis == 2));
    for (size_t c = 0; c < en; ++c) expanded_w[c] = 0;
    for (int k = 0; k < f[2]; ++k)
        for (int j = 0; j < f[1]; ++j)
            for (int i = 0; i < f[0]; ++i) {
                const size_t fc = face(gx, gy, axis, i, j, k);
                if (!valid[fc]) continue;
                int b[3] = {i, j, k};
                b[axis] -= 1;
                const size_t cb = cell(gx, gy, b[0], b[1], b[2]), cf = cell(gx, gy, i, j, k);
                double w = cw[fc];
                if ((material[cb] == MAT_LIQUID && material[cf] == MAT_AIR) || (material[cb] == MAT_AIR && material[cf] == MAT_LIQUID))
                    w /= ghost_fluid_theta(liquid_phi[cb], liquid_phi[cf]);
                expanded_w[face(ex, ey, axis, i + offset, j + offset, k + offset)] = w;
            }
}

/* buildRHS (Plug.cpp:867-943); solid velocities may be NULL */
void mgf_rhs(double *expanded_rhs, const int32_t *material, const double *vx, const double *vy, const double *vz,
             const double *svx, const double *svy, const double *svz, const double *cwx, const double *cwy, const double *cwz,
             int gx, int gy, int gz, int ex, int ey, int ez, int offset)
{
    const double *v[3] = {vx, vy, vz}, *sv[3] = {svx, svy, svz}, *cw[3] = {cwx, cwy, cwz};
    for (size_t c = 0; c < (size_t)ex * ey * ez; ++c) expanded_rhs[c] = 0;
    for (int k = 0; k < gz; ++k)
        for (int j = 0; j < gy; ++j)
            for (int i = 0; i < gx; ++i) {
                if (material[cell(gx, gy, i, j, k)] != MAT_LIQUID) continue;
                double div = 0;
                for (int a = 0; a < 3; ++a)
                    for (int d = 0; d < 2; ++d) {
                        const size_t fc = cell_face(gx, gy, a, d, i, j, k);
                        const double sign = d == 0 ? 1. : -1., w = cw[a][fc];
                        if (w > 0) div += sign * w * v[a][fc];
                        if (sv[a] && w < 1) div += sign * (1. - w) * sv[a][fc];
                    }
                expanded_rhs[cell(ex, ey, i + offset, j + offset, k + offset)] = div;
            }
}

/* applyOldPressure (Plug.cpp:945-997): warm start of the expanded solution grid */
void mgf_pressure_to_solution(double *expanded_x, const double *pressure, const int32_t *material, int gx, int gy, int gz,
                              int ex, int ey, int ez, int offset)
{
    for (size_t c = 0; c < (size_t)ex * ey * ez; ++c) expanded_x[c] = 0;
    for (int k = 0; k < gz; ++k)
        for (int j = 0; j < gy; ++j)
            for (int i = 0; i < gx; ++i)
                if (material[cell(gx, gy, i, j, k)] == MAT_LIQUID)
                    expanded_x[cell(ex, ey, i + offset, j + offset, k + offset)] = pressure[cell(gx, gy, i, j, k)];
}

/* applySolutionToPressure (Plug.cpp:999-1047); non-liquid cells of `pressure` keep their value */
void mgf_solution_to_pressure(double *pressure, const double *expanded_x, const int32_t *material, int gx, int gy, int gz,
                              int ex, int ey, int ez, int offset)
{
    (void)ez;
    for (int k = 0; k < gz; ++k)
        for (int j = 0; j < gy; ++j)
            for (int i = 0; i < gx; ++i)
                if (material[cell(gx, gy, i, j, k)] == MAT_LIQUID)
                    pressure[cell(gx, gy, i, j, k)] = expanded_x[cell(ex, ey, i + offset, j + offset, k + offset)];
}

/* applyPressureGradient (Plug.cpp:1049-1131) */
void mgf_pressure_gradient(int axis, double *velocity, const double *cw, const double *liquid_phi, const double *pressure,
                           const uint8_t *valid, const int32_t *material, int gx, int gy, int gz)
{
    const int f[3] = {gx + (axis == 0), gy + (axis == 1), gz + (axis == 2)}, g[3] = {gx, gy, gz};
    (void)cw;
    for (int k = 0; k < f[2]; ++k)
        for (int j = 0; j < f[1]; ++j)
            for (int i = 0; i < f[0]; ++i) {
                const size_t fc = face(gx, gy, axis, i, j, k);
                if (!valid[fc]) continue;
                int b[3] = {i, j, k}, fw[3] = {i, j, k};
                b[axis] -= 1;
                if (b[axis] < 0 || fw[axis] >= g[axis]) continue;
                const size_t cb = cell(gx, gy, b[0], b[1], b[2]), cf = cell(gx, gy, fw[0], fw[1], fw[2]);
                double grad = pressure[cf] - pressure[cb];
                if (material[cb] != MAT_LIQUID || material[cf] != MAT_LIQUID) grad /= ghost_fluid_theta(liquid_phi[cb], liquid_phi[cf]);
                velocity[fc] -= grad;
            }
}

/* computeResultingDivergence (Plug.cpp:1133-1207): out = {sum, max (from 0), liquid cell count} */
void mgf_divergence(double *out3, const int32_t *material, const double *vx, const double *vy, const double *vz,
                    const double *svx, const double *svy, const double *svz, const double *cwx, const double *cwy,
                    const double *cwz, int gx, int gy, int gz)
{
    const double *v[3] = {vx, vy, vz}, *sv[3] = {svx, svy, svz}, *cw[3] = {cwx, cwy, cwz};
    double sum = 0, mx = 0, count = 0;
    for (int k = 0; k < gz; ++k)
        for (int j = 0; j < gy; ++j)
            for (int i = 0; i < gx; ++i) {
                if (material[cell(gx, gy, i, j, k)] != MAT_LIQUID) continue;
                double div = 0;
                for (int a = 0; a < 3; ++a)
                    for (int d = 0; d < 2; ++d) {
                        const size_t fc = cell_face(gx, gy, a, d, i, j, k);
                        const double sign = d == 0 ? -1. : 1., w = cw[a][fc];
                        if (w > 0) div += sign * w * v[a][fc];
                        if (sv[a] && w < 1) div += sign * (1. - w) * sv[a][fc];
                    }
                sum += div;
                if (div > mx) mx = div;
                count += 1;
            }
    out3[0] = sum;
    out3[1] = mx;
    out3[2] = count;
}
