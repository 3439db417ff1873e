// log.h -- error convention of the gten API (mirrors gten/log.h:6-23 of the
// reference): no return codes, no exceptions; a failed check prints a red
// "GTEN ERROR [File ... line ...]" line to stderr and exits with EXIT_FAILURE.
#pragma once

#include <cstdio>
#include <cstdlib>

namespace gten {
namespace detail {

[[noreturn]] inline void die_prefix_done() { std::fputc('\n', stderr); std::exit(EXIT_FAILURE); }

inline void die_prefix(const char* file, int line)
{
    std::fprintf(stderr, "\n\x1B[1;31mGTEN ERROR [File `%s` line %d]: ", file, line);
}

} // namespace detail
} // namespace gten

#define GTEN_ASSERT(condition)                                                  \
    do {                                                                        \
        if (!(condition)) {                                                     \
            ::gten::detail::die_prefix(__FILE__, __LINE__);                     \
            std::fprintf(stderr, "Assertion '%s' failed.", #condition);         \
            ::gten::detail::die_prefix_done();                                  \
        }                                                                       \
    } while (0)

#define GTEN_ASSERTM(condition, ...)                                            \
    do {                                                                        \
        if (!(condition)) {                                                     \
            ::gten::detail::die_prefix(__FILE__, __LINE__);                     \
            std::fprintf(stderr, __VA_ARGS__);                                  \
            ::gten::detail::die_prefix_done();                                  \
        }                                                                       \
    } while (0)

// A C-ABI call that must succeed (include/gten_hip.h returns 0 on success).
#define GTEN_HIP_OK(call)                                                       \
    do {                                                                        \
        const int gten_rc_ = (call);                                            \
        GTEN_ASSERTM(gten_rc_ == 0, "%s failed (%d): %s", #call, gten_rc_, gten_hip_last_error()); \
    } while (0)
