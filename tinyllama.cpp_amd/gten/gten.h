// gten.h -- umbrella include of the HBM-backed gten API (gten/gten.h:3-8 of the
// reference).  Header-only: every function is inline, so unlike the reference
// this can be included from more than one translation unit.
#pragma once

// the reference's translation unit relies on these arriving through gten.h
// (its gten/tensor.cpp and gten/modules.cpp include them)
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "gten_types.h"
#include "log.h"
#include "modules.h"
#include "ops.h"
#include "quants.h"
#include "tensor.h"
