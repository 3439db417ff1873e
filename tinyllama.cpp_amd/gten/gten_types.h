// gten_types.h -- scalar types of the gten API (same names and enum order as
// gten/gten_types.h:15-33 of the reference, which is what the C-ABI dtype
// codes in include/gten_hip.h follow).
#pragma once

#include <cstdint>
#include <cstring>

#include "log.h"

namespace gten {

typedef int32_t Int32;
typedef uint16_t Float16;
typedef int8_t Qint8;
typedef uint8_t Qint4;

enum class Dtype { Int32, Float16, Float32, Qint8, Qint4 };

static const Dtype kInt32 = Dtype::Int32;
static const Dtype kFloat16 = Dtype::Float16;
static const Dtype kFloat32 = Dtype::Float32;
static const Dtype kQint8 = Dtype::Qint8;
static const Dtype kQint4 = Dtype::Qint4;

inline const char* dtype_str(Dtype dtype)
{
    switch (dtype) {
    case Dtype::Int32: return "Int32";
    case Dtype::Float16: return "Float16";
    case Dtype::Float32: return "Float32";
    case Dtype::Qint8: return "Qint8";
    case Dtype::Qint4: return "Qint4";
    }
    GTEN_ASSERT(false);
    return "";
}

// code passed across the C-ABI
inline int dtype_code(Dtype d) { return static_cast<int>(d); }

// Host-side half conversions (IEEE, round to nearest even; NaN -> 0x7E00 with
// the sign kept, as gten/gten_types.h:99-119).  Only used off the hot path:
// printing tensors and the synthetic-weight writer.
inline float fp16_to_fp32(Float16 h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t bits;
    if (e == 0) {
        if (m == 0) {
            bits = sign;
        } else {
            const float v = (float)m * 5.9604644775390625e-8f;
            std::memcpy(&bits, &v, 4);
            bits |= sign;
        }
    } else if (e == 31) {
        bits = sign | 0x7f800000u | (m << 13);
    } else {
        bits = sign | ((e + 112u) << 23) | (m << 13);
    }
    float out;
    std::memcpy(&out, &bits, 4);
    return out;
}

inline Float16 fp32_to_fp16(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    uint32_t a = x & 0x7fffffffu;
    if (a > 0x7f800000u) return (Float16)(sign | 0x7e00u);
    if (a >= 0x477ff000u) return (Float16)(sign | 0x7c00u);
    if (a < 0x38800000u) {
        float t;
        std::memcpy(&t, &a, 4);
        t += 0.5f;
        uint32_t tb;
        std::memcpy(&tb, &t, 4);
        return (Float16)(sign | (tb - 0x3f000000u));
    }
    a += 0xc8000fffu + ((a >> 13) & 1u);
    return (Float16)(sign | (a >> 13));
}

} // namespace gten
