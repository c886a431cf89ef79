#pragma once
#include "../UT/UT_Mock.h"
enum PRM_Type { PRM_STRING, PRM_TOGGLE, PRM_FLT, PRM_INT };
class PRM_Name
{
public:
    PRM_Name(const char *token = nullptr, const char *label = nullptr);
};
class PRM_Default
{
public:
    PRM_Default(fpreal value = 0, const char *string = nullptr);
};
extern PRM_Default PRMoneDefaults[];
class PRM_Template
{
public:
    PRM_Template();
    PRM_Template(PRM_Type type, int vectorSize, PRM_Name *name, PRM_Default *defaults = nullptr);
};
