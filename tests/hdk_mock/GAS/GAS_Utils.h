#pragma once
#include "GAS_SubSolver.h"
