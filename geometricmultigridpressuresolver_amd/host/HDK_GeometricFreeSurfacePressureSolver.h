// Houdini DOP shim of the MI355X-native multigrid pressure solver: the SAME node surface as the reference plugin
// (class name, data factory description, node type, the twelve parameters of
// /root/reference/Source/HDK_GeometricFreeSurfacePressureSolver.h:14-55 and .cpp:36-111) so that
// Scenes/flipSplash.hip loads unchanged, with the work behind it replaced by ONE call into libmgps.so
// (mgps_project_free_surface, include/mgps_fields.h): field pre-processing, multigrid set-up, MG-preconditioned CG and
// post-processing all run on the GPU.
//
// Needs the Houdini HDK (GAS/SIM/PRM/UT headers, `Houdini` CMake target): built only when CMake finds it
// (CMakeLists.txt at the repo root).  It cannot be compiled in the development image (no HDK); the HDK-free parts it is
// made of -- the flatten / unflatten templates (mgps_voxel_flatten.hpp) and the projection call -- are unit-tested there.
#ifndef MGPS_HDK_GEOMETRIC_FREE_SURFACE_PRESSURE_SOLVER_H
#define MGPS_HDK_GEOMETRIC_FREE_SURFACE_PRESSURE_SOLVER_H

#include <GAS/GAS_SubSolver.h>
#include <GAS/GAS_Utils.h>

class SIM_VectorField;
class SIM_ScalarField;
class SIM_RawField;

class GAS_API HDK_GeometricFreeSurfacePressureSolver : public GAS_SubSolver
{
public:
    // parameter accessors: same names and types as the reference (Plug.h:23-28)
    GET_DATA_FUNC_F(SIM_NAME_TOLERANCE, SolverTolerance);
    GET_DATA_FUNC_I("maxIterations", MaxSolverIterations);
    GET_DATA_FUNC_B("useMGPreconditioner", UseMGPreconditioner);
    GET_DATA_FUNC_B("useOldPressure", UseOldPressure);

protected:
    explicit HDK_GeometricFreeSurfacePressureSolver(const SIM_DataFactory *factory);
    ~HDK_GeometricFreeSurfacePressureSolver() override;

    // one object per call, on the cook thread (Plug.h:38-41)
    bool solveGasSubclass(SIM_Engine &engine, SIM_Object *obj, SIM_Time time, SIM_Time timestep) override;

private:
    static const SIM_DopDescription *getDopDescription();

    DECLARE_STANDARD_GETCASTTOTYPE();
    DECLARE_DATAFACTORY(HDK_GeometricFreeSurfacePressureSolver, GAS_SubSolver, "HDK Geometric Free Surface Pressure Solver",
                        getDopDescription());
};

#endif
