// synth.h -- deterministic synthetic TinyLlama weights and the offline weight
// quantizers, so that benchmarks and parity fixtures need a seed, not a file.
//
// Real TinyLlama weights cannot be fetched here (no network; the reference's
// model_dl.py downloads them).  Values: fp32 N(0, 0.02^2) for every linear /
// embedding matrix and 1 + N(0, 0.05^2) for norm vectors (the recipe BASELINE.md
// section 2 was measured with), from a counter-based integer hash, so any
// element is a pure function of (seed, tensor index, element index) and the
// GPU box regenerates identical bytes without torch or numpy RNG streams.
//
// Quantizers follow the reference converter tinyllama_to_gten.py:24-148:
// per 32-element block delta = absmax/127 (Q8) or absmax/7 (Q4) in f32,
// q = round_half_to_even(x / delta), stored delta = fp16(delta); Q4 adds 7 and
// packs elements [0,16) into high nibbles, [16,32) into low nibbles.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../gten/gten_types.h"
#include "../gten/quants.h"

namespace gten {
namespace synth {

// Version of everything below that decides a synthetic weight's bytes (generator, quantizers, tensor order): part of the name of
// the per-node weight cache (TinyLlama::load_synthetic_cached), so that a cache written by another generator is never read.
// BUMP IT with any such change.
constexpr int kFormat = 2;

inline uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// Two independent standard normals from one 64-bit hash (Box-Muller in double,
// rounded to f32 once; libm-version independent to well below f32 resolution
// except at astronomically rare ties).
inline void normal_pair(uint64_t seed, uint64_t tensor, uint64_t pair_index, float* a, float* b)
{
    const uint64_t h = mix64(mix64(seed * 0x2545f4914f6cdd1dull + tensor) ^ (pair_index * 0x9e3779b97f4a7c15ull));
    const double u1 = ((double)(h >> 32) + 1.0) / 4294967297.0;      // (0,1)
    const double u2 = ((double)(h & 0xffffffffull) + 0.5) / 4294967296.0;
    const double r = std::sqrt(-2.0 * std::log(u1));
    const double t = 6.283185307179586476925 * u2;
    *a = (float)(r * std::cos(t));
    *b = (float)(r * std::sin(t));
}

// fills out[0..numel) with mean + sigma * N(0,1)
inline void fill_normal(float* out, size_t numel, uint64_t seed, uint64_t tensor, float mean, float sigma)
{
    const size_t pairs = (numel + 1) / 2;
    #pragma omp parallel for schedule(static)
    for (size_t p = 0; p < pairs; p++) {
        float a, b;
        normal_pair(seed, tensor, p, &a, &b);
        out[2 * p] = mean + sigma * a;
        if (2 * p + 1 < numel) out[2 * p + 1] = mean + sigma * b;
    }
}

inline size_t row_bytes(Dtype dt, int cols)
{
    switch (dt) {
    case Dtype::Float16: return (size_t)cols * 2;
    case Dtype::Qint8: return (size_t)(cols / 32) * sizeof(Q8Block);
    case Dtype::Qint4: return (size_t)(cols / 32) * sizeof(Q4Block);
    default: return (size_t)cols * 4;
    }
}

// fp32 [rows][cols] -> storage bytes in .gten payload order
inline void quantize_weight(const float* w, int rows, int cols, Dtype dt, uint8_t* out)
{
    if (dt == Dtype::Float16) {
        Float16* h = reinterpret_cast<Float16*>(out);
        const size_t n = (size_t)rows * cols;
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) h[i] = fp32_to_fp16(w[i]);
        return;
    }
    GTEN_ASSERT(cols % 32 == 0);
    const size_t nblk = (size_t)rows * (cols / 32);
    const float qmax = (dt == Dtype::Qint8) ? 127.0f : 7.0f;
    #pragma omp parallel for schedule(static)
    for (size_t b = 0; b < nblk; b++) {
        const float* x = w + b * 32;
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) amax = std::fmax(amax, std::fabs(x[j]));
        const float delta = amax / qmax;
        const float inv = (delta != 0.0f) ? 1.0f / delta : 0.0f;
        if (dt == Dtype::Qint8) {
            Q8Block* o = reinterpret_cast<Q8Block*>(out) + b;
            o->delta = fp32_to_fp16(delta);
            for (int j = 0; j < 32; j++) o->data[j] = (Qint8)std::nearbyint(x[j] * inv);
        } else {
            Q4Block* o = reinterpret_cast<Q4Block*>(out) + b;
            o->delta = fp32_to_fp16(delta);
            for (int j = 0; j < 16; j++) {
                const int hi = (int)std::nearbyint(x[j] * inv) + 7;
                const int lo = (int)std::nearbyint(x[j + 16] * inv) + 7;
                o->data[j] = (Qint4)((hi << 4) | (lo & 0x0f));
            }
        }
    }
}

// Synthetic prompt / teacher-forcing ids: [1] + LCG ids in [3, 31993), SURVEY 8(d).
inline std::vector<int32_t> synthetic_tokens(int count, uint32_t seed = 12345, int n_vocab = 32003)
{
    std::vector<int32_t> t((size_t)count);
    uint32_t s = seed;
    const uint32_t span = (uint32_t)(n_vocab - 10 - 3);
    for (int i = 0; i < count; i++) {
        s = s * 1664525u + 1013904223u;
        t[(size_t)i] = (i == 0) ? 1 : (int32_t)(3u + (s >> 8) % span);
    }
    return t;
}

} // namespace synth
} // namespace gten
