// gten.h -- umbrella include of the HBM-backed gten API (gten/gten.h:3-8 of the
// reference).  Header-only: every function is inline, so unlike the reference
// this can be included from more than one translation unit.
#pragma once

#include "gten_types.h"
#include "log.h"
#include "modules.h"
#include "ops.h"
#include "quants.h"
#include "tensor.h"
