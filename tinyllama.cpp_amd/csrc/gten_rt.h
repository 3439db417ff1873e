// gten_rt.h -- host-side internals shared by the .hip translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gten_hip.h"

#define GTEN_ROPE_MAX_POS 2048   // TinyLLamaParams::max_ctx, tinyllama.cpp:14

namespace gtr {

int fail(int code, const char* fmt, ...);
hipStream_t stream();
bool inited();
int rope_table(int d_head, const float2** out);

} // namespace gtr

#define GTR_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return gtr::fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                             __FILE__, __LINE__);                                              \
    } while (0)

#define GTR_NEED_INIT()                                                                        \
    do {                                                                                       \
        if (!gtr::inited()) return gtr::fail(-1, "gten_hip_init() has not been called");       \
    } while (0)

#define GTR_REQUIRE(cond, ...)                                                                 \
    do {                                                                                       \
        if (!(cond)) return gtr::fail(-4, __VA_ARGS__);                                        \
    } while (0)

// launch errors surface here (bad grid, missing code object, ...)
#define GTR_LAUNCHED() GTR_CHECK(hipGetLastError())
