// See the header.  What stays on the host: looking the SIM fields up, the checks the reference makes on them
// (Plug.cpp:119-250: same conditions, same message texts and severities), flattening them, and writing pressure,
// velocity and the valid-face flags back.  Everything between (Plug.cpp:252-707) is mgps_project_free_surface.
#include "HDK_GeometricFreeSurfacePressureSolver.h"

#include <PRM/PRM_Include.h>
#include <SIM/SIM_DopDescription.h>
#include <SIM/SIM_FieldUtils.h>
#include <SIM/SIM_Object.h>
#include <SIM/SIM_PRMShared.h>
#include <SIM/SIM_ScalarField.h>
#include <SIM/SIM_VectorField.h>
#include <UT/UT_DSOVersion.h>
#include <UT/UT_Interrupt.h>
#include <UT/UT_PerfMonAutoEvent.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <iostream>

#include "mgps_fields.h"
#include "mgps_voxel_flatten.hpp"

void initializeSIM(void *)
{
    IMPLEMENT_DATAFACTORY(HDK_GeometricFreeSurfacePressureSolver);
}

HDK_GeometricFreeSurfacePressureSolver::HDK_GeometricFreeSurfacePressureSolver(const SIM_DataFactory *factory) : BaseClass(factory) {}
HDK_GeometricFreeSurfacePressureSolver::~HDK_GeometricFreeSurfacePressureSolver() {}

// The node interface: twelve parameters with the reference's tokens, labels, types, defaults and order (Plug.cpp:39-99).
const SIM_DopDescription *HDK_GeometricFreeSurfacePressureSolver::getDopDescription()
{
    static PRM_Name surfaceName(GAS_NAME_SURFACE, "Surface Field");
    static PRM_Default surfaceDefault(0, "surface");
    static PRM_Name velocityName(GAS_NAME_VELOCITY, "Velocity Field");
    static PRM_Default velocityDefault(0, "vel");
    static PRM_Name solidName(GAS_NAME_COLLISION, "Solid Field");
    static PRM_Default solidDefault(0, "collision");
    static PRM_Name solidVelocityName(GAS_NAME_COLLISIONVELOCITY, "Solid Velocity Field");
    static PRM_Default solidVelocityDefault(0, "collisionvel");
    static PRM_Name cutCellName("cutCellWeights", "Cut-cell Weights Field");
    static PRM_Default cutCellDefault(0, "collisionweights");
    static PRM_Name pressureName(GAS_NAME_PRESSURE, "Pressure");
    static PRM_Default pressureDefault(0, "pressure");
    static PRM_Name useOldPressureName("useOldPressure", "Use old pressure as an initial guess");
    static PRM_Name densityName(GAS_NAME_DENSITY, "Liquid Density Field");
    static PRM_Default densityDefault(0, "massdensity");
    static PRM_Name validFacesName("validFaces", "Valid Faces Field");
    static PRM_Name toleranceName(SIM_NAME_TOLERANCE, "Solver Tolerance");
    static PRM_Default toleranceDefault(1e-5);
    static PRM_Name maxIterationsName("maxIterations", "Max Solver Iterations");
    static PRM_Default maxIterationsDefault(2500);
    static PRM_Name useMGName("useMGPreconditioner", "Use Multigrid Preconditioner");

    static PRM_Template templates[] = {PRM_Template(PRM_STRING, 1, &surfaceName, &surfaceDefault),
                                       PRM_Template(PRM_STRING, 1, &velocityName, &velocityDefault),
                                       PRM_Template(PRM_STRING, 1, &solidName, &solidDefault),
                                       PRM_Template(PRM_STRING, 1, &solidVelocityName, &solidVelocityDefault),
                                       PRM_Template(PRM_STRING, 1, &cutCellName, &cutCellDefault),
                                       PRM_Template(PRM_STRING, 1, &pressureName, &pressureDefault),
                                       PRM_Template(PRM_TOGGLE, 1, &useOldPressureName, PRMoneDefaults),
                                       PRM_Template(PRM_STRING, 1, &densityName, &densityDefault),
                                       PRM_Template(PRM_STRING, 1, &validFacesName),
                                       PRM_Template(PRM_FLT, 1, &toleranceName, &toleranceDefault),
                                       PRM_Template(PRM_INT, 1, &maxIterationsName, &maxIterationsDefault),
                                       PRM_Template(PRM_TOGGLE, 1, &useMGName, PRMoneDefaults),
                                       PRM_Template()};

    static SIM_DopDescription description(true, "HDK_GeometricFreeSurfacePressureSolver", "HDK Geometric Free Surface Pressure Solver", "$OS",
                                          classname(), templates);
    setGasDescription(description);
    return &description;
}

namespace {

// A flattened field in page-locked memory (mgps_host_alloc): the upload of the ten field arrays is what a sub-step spends
// most of its PCIe time on, and a pageable source runs at a twentieth of the link's rate.  Falls back to plain memory.
template <class T>
struct Staging {
    T *p = nullptr;
    size_t n = 0;
    bool pinned = false;
    Staging() = default;
    Staging(const Staging &) = delete;
    Staging &operator=(const Staging &) = delete;
    ~Staging() { release(); }
    void release()
    {
        if (p && pinned) mgps_host_free(p);
        else if (p) std::free(p);
        p = nullptr;
        n = 0;
    }
    void resize(size_t count)
    {
        release();
        n = count;
        p = static_cast<T *>(mgps_host_alloc(count * sizeof(T)));
        pinned = p != nullptr;
        if (!p) p = static_cast<T *>(std::malloc(count * sizeof(T) + 1));
    }
    void assign(size_t count, T value)
    {
        resize(count);
        std::fill(p, p + count, value);
    }
    T *data() { return p; }
    size_t size() const { return n; }
};
template <class T, class VoxelArray>
void flattenInto(Staging<T> &dst, const VoxelArray &grid)
{
    dst.resize(size_t(grid.getXRes()) * grid.getYRes() * grid.getZRes());
    mgps::flattenGrid(dst.data(), grid);
}

// a SIM_RawField sampled at the sample positions of `like` (the solver wants the solid SDF at cell centres and the solid
// velocity at the liquid's face centres; the reference interpolates them at those positions, Util.cpp:25, Plug.cpp:925)
void sampleAt(Staging<float> &flat, const SIM_RawField &source, const SIM_RawField &like)
{
    int nx, ny, nz;
    like.getVoxelRes(nx, ny, nz);
    flat.resize(size_t(nx) * ny * nz);
    float *out = flat.data();
    mgps::forEachPlaneRange(nz, [&source, &like, out, nx, ny](int k0, int k1) {
        UT_Vector3 pos;
        for (int k = k0; k < k1; ++k)
            for (int j = 0; j < ny; ++j)
                for (int i = 0; i < nx; ++i) {
                    like.indexToPos(i, j, k, pos);
                    out[(size_t(k) * ny + j) * nx + i] = float(source.getValue(pos));
                }
    });
}

int pollInterrupt(void *) { return UTgetInterrupt()->opInterrupt() ? 1 : 0; }  // the reference polls in every loop (Ops.h:319)

}  // namespace

bool HDK_GeometricFreeSurfacePressureSolver::solveGasSubclass(SIM_Engine &, SIM_Object *obj, SIM_Time, SIM_Time)
{
    // ---- the fields and the reference's checks on them (Plug.cpp:119-250) -------------------------------------------
    const SIM_VectorField *solidVelocity = getConstVectorField(obj, GAS_NAME_COLLISIONVELOCITY);
    SIM_VectorField *velocity = getVectorField(obj, GAS_NAME_VELOCITY);
    if (!velocity) {
        addError(obj, SIM_MESSAGE, "Velocity field missing", UT_ERROR_WARNING);
        return false;
    }
    if (!velocity->isFaceSampled()) {
        addError(obj, SIM_MESSAGE, "Velocity field must be a staggered grid", UT_ERROR_WARNING);
        return false;
    }
    const SIM_VectorField *cutCellWeights = getConstVectorField(obj, "cutCellWeights");
    if (!cutCellWeights) {
        addError(obj, SIM_MESSAGE, "Cut-cell weights field missing", UT_ERROR_WARNING);
        return false;
    }
    if (!cutCellWeights->isAligned(velocity)) {
        addError(obj, SIM_MESSAGE, "Cut-cell weights must align with velocity samples", UT_ERROR_WARNING);
        return false;
    }
    SIM_VectorField *validFaces = getVectorField(obj, "validFaces");
    if (!validFaces) {
        addError(obj, SIM_MESSAGE, "No 'valid' field found", UT_ERROR_ABORT);
        return false;
    }
    if (!validFaces->isAligned(velocity)) {
        addError(obj, SIM_MESSAGE, "Valid field sampling needs to match velocity field", UT_ERROR_ABORT);
        return false;
    }
    const SIM_ScalarField *surfaceField = getConstScalarField(obj, GAS_NAME_SURFACE);
    if (!surfaceField) {
        addError(obj, SIM_MESSAGE, "Surface field is missing. There is nothing to represent the liquid", UT_ERROR_WARNING);
        return false;
    }
    const SIM_RawField &liquidSurface = *surfaceField->getField();
    const UT_Vector3I res = velocity->getTotalVoxelRes();
    SIM_ScalarField *pressureField = getScalarField(obj, GAS_NAME_PRESSURE, true);
    SIM_RawField localPressure, *pressure = &localPressure;
    if (pressureField) {
        pressureField->matchField(surfaceField);
        pressure = pressureField->getField();
    } else {
        localPressure.init(SIM_SAMPLE_CENTER, velocity->getOrig(), velocity->getSize(), res[0], res[1], res[2]);
        localPressure.makeConstant(0);
    }
    const SIM_ScalarField *solidField = getConstScalarField(obj, GAS_NAME_COLLISION);
    const SIM_ScalarField *densityField = getConstScalarField(obj, GAS_NAME_DENSITY);
    if (!densityField) {
        addError(obj, SIM_MESSAGE, "There is no liquid density to simulate with", UT_ERROR_WARNING);
        return false;
    }
    if (!densityField->getField()->isAligned(&liquidSurface)) {
        addError(obj, SIM_MESSAGE, "Density must align with the surface volume", UT_ERROR_WARNING);
        return false;
    }
    fpreal32 constantDensity;
    if (!densityField->getField()->field()->isConstant(&constantDensity)) {
        addError(obj, SIM_MESSAGE, "Variable density is not currently supported", UT_ERROR_WARNING);
        return false;
    }

    std::cout << "//\n//\n// Starting free surface pressure solver (mgps, MI355X)\n//\n//" << std::endl;

    // ---- flatten (SIM fields are fpreal32 voxel arrays: float is their own precision) ----------------------------------
    Staging<float> phi, solidPhi, p, cw[3], vel[3], solidVel[3];
    Staging<uint8_t> valid[3];
    {
        UT_PerfMonAutoSolveEvent event(this, "Flatten fields");
        flattenInto(phi, *liquidSurface.field());
        flattenInto(p, *pressure->field());
        if (solidField) sampleAt(solidPhi, *solidField->getField(), liquidSurface);
        else {  // no collision field: all fluid.  Houdini's solid SDF is positive inside (Plug.cpp:214-225)
            const fpreal dx = velocity->getVoxelSize().maxComponent();
            solidPhi.assign(phi.size(), float(-10. * dx));
        }
        for (int axis : {0, 1, 2}) {
            flattenInto(cw[axis], *cutCellWeights->getField(axis)->field());
            flattenInto(vel[axis], *velocity->getField(axis)->field());
            if (solidVelocity) sampleAt(solidVel[axis], *solidVelocity->getField(axis), *velocity->getField(axis));
            valid[axis].resize(vel[axis].size());
        }
    }

    // ---- the projection (Plug.cpp:252-707) on the GPU ---------------------------------------------------------------------
    mgps_projection job{};
    job.struct_size = int(sizeof(job));
    liquidSurface.getVoxelRes(job.gx, job.gy, job.gz);
    job.real_bytes = 4;
    job.liquid_phi = phi.data();
    job.solid_phi = solidPhi.data();
    job.pressure = p.data();
    for (int axis : {0, 1, 2}) {
        job.cut_weights[axis] = cw[axis].data();
        job.velocity[axis] = vel[axis].data();
        job.solid_velocity[axis] = solidVelocity ? solidVel[axis].data() : nullptr;
        job.valid_faces[axis] = valid[axis].data();
    }
    job.use_old_pressure = getUseOldPressure();
    job.use_mg_preconditioner = getUseMGPreconditioner();
    // The reference hard-wires the tiled Gauss-Seidel smoother here (Plug.cpp:466) and so does this node: the twelve parameters stay
    // as they are.  MGPS_DOP_SMOOTHER=jacobi in the environment of the Houdini session selects the damped-Jacobi smoother of
    // MG.cpp:480-486 instead -- on MI355X the faster preconditioner (512^3 free-surface pool, MG-PCG to 1e-5: 57 ms against 72 ms
    // with Gauss-Seidel, one iteration more): same solver tolerance, same pressure to that tolerance.
    const char *smoother = getenv("MGPS_DOP_SMOOTHER");
    job.use_gauss_seidel = (smoother && (smoother[0] == 'j' || smoother[0] == 'J')) ? 0 : 1;
    job.tolerance = getSolverTolerance();
    job.max_iterations = getMaxSolverIterations();
    job.power_of_two = 0;  // tight extents: same (offset, levels) contract, fewer padded cells than Ops.h:1353-1360
    mgps_options opt;
    mgps_default_options(&opt);
    opt.interrupt = pollInterrupt;
    int rc;
    {
        UT_PerfMonAutoSolveEvent event(this, "Solve linear system");
        rc = mgps_project_free_surface(&job, &opt);
    }
    if (rc != MGPS_OK) {
        addError(obj, SIM_MESSAGE, mgps_last_error(nullptr), rc == MGPS_ERR_INTERRUPTED ? UT_ERROR_WARNING : UT_ERROR_ABORT);
        return false;
    }
    // (no liquid cells: nothing was solved; like the reference, the valid faces and an all-zero pressure are still published below)
    if (job.liquid_cells == 0) addError(obj, SIM_MESSAGE, "No liquid cells found", UT_ERROR_WARNING);
    // the reference's printouts (CG.h:198-206, Plug.cpp:625-628, 704-706)
    else std::cout << "  MG levels: " << job.mg_levels << ", solver grid " << job.expanded[0] << " x " << job.expanded[1] << " x " << job.expanded[2]
              << "\n  Iterations: " << job.stats.iterations << "\n  Drifted relative L2 Error: " << job.stats.rel_residual
              << "\n  Recomputed relative L2 Error: " << job.stats.rel_residual_recomputed << "\n  L-infinity error: " << job.residual_inf
              << "\n  L-2 error: " << job.residual_l2 << "\n  Max divergence: " << job.divergence_max
              << "\n  Accumulated divergence: " << job.divergence_sum << "\n  Average divergence: " << job.divergence_sum / job.liquid_cells
              << "\n  set-up " << job.setup_ms << " ms, solve " << job.solve_ms << " ms, total " << job.total_ms << " ms" << std::endl;

    // ---- write back (Plug.cpp:637-713) ------------------------------------------------------------------------------------------
    {
        UT_PerfMonAutoSolveEvent event(this, "Write fields back");
        mgps::unflattenGrid(*pressure->fieldNC(), p.data(), p.size());
        for (int axis : {0, 1, 2}) {
            mgps::unflattenGrid(*velocity->getField(axis)->fieldNC(), vel[axis].data(), vel[axis].size());
            // 1 = valid face, 0 = invalid (HDK::Utilities VALID_FACE / INVALID_FACE); the voxel array converts uint8 -> fpreal32
            mgps::unflattenGrid(*validFaces->getField(axis)->fieldNC(), valid[axis].data(), valid[axis].size());
        }
    }
    if (pressureField) pressureField->pubHandleModification();
    velocity->pubHandleModification();
    validFaces->pubHandleModification();
    return true;
}
