// HDK mock (tests/hdk_mock/README.md): the SIM classes as the shim calls them.  Declarations only.
#pragma once
#include "../PRM/PRM_Include.h"
#include "../UT/UT_Mock.h"
typedef fpreal SIM_Time;
class SIM_Engine;
class SIM_Object;
class SIM_DataFactory;
enum SIM_FieldSample { SIM_SAMPLE_CENTER, SIM_SAMPLE_FACEX, SIM_SAMPLE_FACEY, SIM_SAMPLE_FACEZ, SIM_SAMPLE_CORNER };
#define SIM_MESSAGE 0
#define SIM_NAME_TOLERANCE "tolerance"
class SIM_RawField
{
public:
    void init(SIM_FieldSample sample, const UT_Vector3 &orig, const UT_Vector3 &size, int xres, int yres, int zres);
    void makeConstant(fpreal32 value);
    void getVoxelRes(int &xres, int &yres, int &zres) const;
    bool indexToPos(int x, int y, int z, UT_Vector3 &pos) const;
    fpreal getValue(UT_Vector3 pos) const;
    bool isAligned(const SIM_RawField *field) const;
    const UT_VoxelArrayF *field() const;
    UT_VoxelArrayF *fieldNC() const;
};
class SIM_ScalarField
{
public:
    const SIM_RawField *getField() const;
    SIM_RawField *getField();
    void matchField(const SIM_ScalarField *field);
    void pubHandleModification();
};
class SIM_VectorField
{
public:
    bool isFaceSampled() const;
    bool isAligned(const SIM_VectorField *field) const;
    UT_Vector3I getTotalVoxelRes() const;
    UT_Vector3 getOrig() const;
    UT_Vector3 getSize() const;
    UT_Vector3 getVoxelSize() const;
    const SIM_RawField *getField(int axis) const;
    SIM_RawField *getField(int axis);
    void pubHandleModification();
};
class SIM_DopDescription
{
public:
    SIM_DopDescription(bool createNode, const char *nodeName, const char *nodeLabel, const char *dataName, const char *dataType, const PRM_Template *templates);
};
