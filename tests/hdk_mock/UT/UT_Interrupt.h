#pragma once
#include "UT_Mock.h"
class UT_Interrupt
{
public:
    bool opInterrupt(int percent = -1);
};
UT_Interrupt *UTgetInterrupt();
