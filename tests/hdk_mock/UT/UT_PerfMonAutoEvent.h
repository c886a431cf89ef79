#pragma once
#include "UT_Mock.h"
class SIM_Solver;
class UT_PerfMonAutoSolveEvent
{
public:
    UT_PerfMonAutoSolveEvent(const void *solver, const char *label);
    ~UT_PerfMonAutoSolveEvent();
};
