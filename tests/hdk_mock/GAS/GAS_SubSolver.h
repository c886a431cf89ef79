// HDK mock (tests/hdk_mock/README.md): GAS_SubSolver and the macros of a DOP data class, as the shim uses them.
#pragma once
#include "../SIM/SIM_Mock.h"
#define GAS_API
#define GAS_NAME_SURFACE "surface"
#define GAS_NAME_VELOCITY "velocity"
#define GAS_NAME_COLLISION "collision"
#define GAS_NAME_COLLISIONVELOCITY "collisionvel"
#define GAS_NAME_PRESSURE "pressure"
#define GAS_NAME_DENSITY "density"
class GAS_SubSolver
{
public:
    typedef GAS_SubSolver BaseClass;

protected:
    explicit GAS_SubSolver(const SIM_DataFactory *factory);
    virtual ~GAS_SubSolver();
    virtual bool solveGasSubclass(SIM_Engine &engine, SIM_Object *obj, SIM_Time time, SIM_Time timestep) = 0;
    SIM_ScalarField *getScalarField(SIM_Object *obj, const char *name, bool silent = false);
    SIM_VectorField *getVectorField(SIM_Object *obj, const char *name, bool silent = false);
    const SIM_ScalarField *getConstScalarField(SIM_Object *obj, const char *name);
    const SIM_VectorField *getConstVectorField(SIM_Object *obj, const char *name);
    void addError(const SIM_Object *obj, int code, const char *text, UT_ErrorSeverity severity) const;
    static void setGasDescription(SIM_DopDescription &description);
    fpreal getPropertyF(const char *name) const;
    int getPropertyI(const char *name) const;
    bool getPropertyB(const char *name) const;
};
#define GET_DATA_FUNC_F(name, Method) fpreal get##Method() const { return getPropertyF(name); }
#define GET_DATA_FUNC_I(name, Method) int get##Method() const { return getPropertyI(name); }
#define GET_DATA_FUNC_B(name, Method) bool get##Method() const { return getPropertyB(name); }
#define DECLARE_STANDARD_GETCASTTOTYPE() \
public:                                  \
    virtual void *getCastToType(const char *) const;
#define DECLARE_DATAFACTORY(Class, Super, Description, DopParms) \
public:                                                          \
    typedef Super BaseClass;                                     \
    static const char *classname() { return #Class; }            \
    static void createDataFactory();                             \
                                                                 \
private:
#define IMPLEMENT_DATAFACTORY(Class) Class::createDataFactory()
