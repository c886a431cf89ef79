#pragma once
#include "SIM_Mock.h"
