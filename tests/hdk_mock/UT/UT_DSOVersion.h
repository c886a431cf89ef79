#pragma once
#include "UT_Mock.h"
