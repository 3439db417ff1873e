// gten_dev.h -- device-side building blocks shared by the gfx950 kernels.
//
// Wave64 only (MI355X / CDNA4).  Numerics follow the reference's row codec:
// every operator reads rows in their storage dtype to f32, computes in f32 and
// writes the row back in the storage dtype (gten/ops.h:40-96).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gten_hip.h"

#define GTEN_WAVE 64
#define GTEN_QBLK 32      // gten/quants.h:12-15
#define GTEN_Q8_BYTES 34  // gten/quants.h:17-23
#define GTEN_Q4_BYTES 18  // gten/quants.h:25-31

namespace gtd {

// ---- fp16 <-> fp32 (gten/gten_types.h:79-119: IEEE, round to nearest even) ----
__device__ __forceinline__ float h2f(uint16_t h)
{
    _Float16 x;
    __builtin_memcpy(&x, &h, 2);
    return (float)x;
}
// The f32 value is made OPAQUE to the optimiser before the convert: the reference rounds to f32 first and to fp16
// second (gten/ops.h:73-96 writes an f32 row through fp32_to_fp16).  Left transparent, the code generator folds a
// preceding multiply / fma and the convert into ONE v_fma_mix{lo,hi}_f16 -- a single rounding of the exact product,
// one fp16 ulp away on ties -- and whether it does depends on build flags and on the surrounding kernel (round 2: the
// fused decoder's RMSNorm prologue got it, the operator kernel did not).  tests/test_no_packed_f32_cpu.py checks the
// generated code: no v_fma_mix in any kernel.
__device__ __forceinline__ _Float16 f2hv(float f)
{
    asm("" : "+v"(f));
    return (_Float16)f;               // v_cvt_f16_f32, RNE, overflow -> inf
}
__device__ __forceinline__ uint16_t f2h(float f)
{
    const _Float16 x = f2hv(f);
    uint16_t h;
    __builtin_memcpy(&h, &x, 2);
    return h;
}

// ---- once-read weight stream: nontemporal loads (global_load ... nt) ----
// A decode step reads every weight byte exactly once and each byte by ONE workgroup; with the default policy the
// stream displaces what the step DOES re-read from the caches (residual rows, attention partials, K/V).  Measured on
// the fused q4 step (profiles/README.md, round 3).  Activations, K/V and anything re-read keep the default policy.
#ifndef GTEN_NT_WEIGHTS
#define GTEN_NT_WEIGHTS 1
#endif
typedef unsigned gt_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_w16(const void* p)
{
#if GTEN_NT_WEIGHTS
    const gt_u4v v = __builtin_nontemporal_load((const gt_u4v*)p);
#else
    const gt_u4v v = *(const gt_u4v*)p;
#endif
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint16_t ld_w2(const uint16_t* p)
{
#if GTEN_NT_WEIGHTS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

// ---- cross-lane moves on the DPP path (no LDS crossbar round trip) ----
// CTRL: 0xB1 quad_perm[1,0,3,2] (xor 1) | 0x4E quad_perm[2,3,0,1] (xor 2) |
//       0x141 row_half_mirror | 0x140 row_mirror | 0x142 row_bcast15 | 0x143 row_bcast31
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_mov_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
// max / sum over the 4 lanes of a quad (every lane gets the result)
__device__ __forceinline__ float quad_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v));
    return fmaxf(v, dpp_mov<0x4E>(v));
}
__device__ __forceinline__ int quad_sum_i(int v)
{
    v += dpp_mov_i<0xB1>(v);
    return v + dpp_mov_i<0x4E>(v);
}

// ---- maxima of NON-NEGATIVE floats (absolute values, the Q8 absmax): IEEE order is the integer order of the bit
// patterns, so each step is one v_max_i32_dpp instead of mov_dpp + canonicalize + max; rows 16 lanes apart are
// combined by gfx950's v_permlane16_swap instead of a trip through the LDS crossbar.  Exact (a maximum has no order).
template <int CTRL>
__device__ __forceinline__ int nn_max_step(int i)
{
    return max(i, __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float nn_max4(float v)              // the 4 lanes of a quad
{
    int i = __float_as_int(v);
    i = nn_max_step<0xB1>(i);
    i = nn_max_step<0x4E>(i);
    return __int_as_float(i);
}
__device__ __forceinline__ float nn_max8(float v)              // aligned groups of 8 lanes
{
    int i = __float_as_int(nn_max4(v));
    i = nn_max_step<0x141>(i);
    return __int_as_float(i);
}
__device__ __forceinline__ float row16_absmax(float v)         // each row of 16 lanes
{
    int i = __float_as_int(nn_max8(v));
    return __int_as_float(nn_max_step<0x140>(i));
}
__device__ __forceinline__ float nn_max32(float v)             // the two 32-lane halves of the wave
{
    int i = __float_as_int(nn_max8(v));
    i = nn_max_step<0x140>(i);
    const auto r = __builtin_amdgcn_permlane16_swap(i, i, false, false);   // rows {0,0,2,2} | {1,1,3,3}
    return __int_as_float(max((int)r[0], (int)r[1]));
}
__device__ __forceinline__ int sum32_lanes_i(int v)            // integer sum over each 32-lane half
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (int)r[0] + (int)r[1];
}

// ---- wavefront reductions (all 64 lanes end with the same value) ----
// Sum of the 64 lanes in a fixed tree: quads, 8, 16 (DPP mirrors), then the
// four 16-lane rows chained through row_bcast15/31; the total lands in lane 63.
__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    v += dpp_mov<0x142, 0xA>(v);
    v += dpp_mov<0x143, 0xC>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// the same on the DPP path (lanes outside a row_bcast's rows keep their own value: old = v)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_max_dpp(float v)
{
    v = quad_max(v);
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    v = fmaxf(v, dpp_keep<0x142, 0xA>(v));
    v = fmaxf(v, dpp_keep<0x143, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// max over aligned groups of `width` lanes (width = 4, 8, 16 or 32)
template <int WIDTH>
__device__ __forceinline__ float group_max(float v)
{
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum as a perfectly balanced binary tree over the threads in natural
// order (wave tree above, then waves pairwise): with power-of-two thread counts
// the result does not depend on how the same elements are split over 256 or 512
// threads, which is what lets kernels with different geometries agree bit for bit.
__device__ __forceinline__ float block_sum_tree(float v, float* scratch)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = (i < nw) ? scratch[i] : 0.f;
#pragma unroll
    for (int w = 1; w < 16; w <<= 1)
#pragma unroll
        for (int i = 0; i < 16; i += 2 * w) t[i] = t[i] + t[i + w];
    return t[0];
}

// The same tree for a workgroup of NW waves known at compile time (NW = 4 or 8): no branch per wave slot, the
// partials come back as 16-byte LDS reads, and -- with FRESH -- no barrier ahead of the scratch write (the caller
// guarantees nobody is still reading `scratch`, e.g. its first use in the kernel).  Bit-identical to block_sum_tree.
template <int NW, bool FRESH>
__device__ __forceinline__ float block_sum_tree_n(float v, float* scratch)
{
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_sum(v);
    if (!FRESH) __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t[16];
    const float4 a = ((const float4*)scratch)[0];
    t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = a.w;
    if (NW == 8) { const float4 b = ((const float4*)scratch)[1]; t[4] = b.x; t[5] = b.y; t[6] = b.z; t[7] = b.w; }
#pragma unroll
    for (int i = NW; i < 16; i++) t[i] = 0.f;
#pragma unroll
    for (int w = 1; w < 16; w <<= 1)
#pragma unroll
        for (int i = 0; i < 16; i += 2 * w) t[i] = t[i] + t[i + w];
    return t[0];
}

// block-wide sum through LDS scratch (>= 16 floats); every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* scratch)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; i++) t += scratch[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* scratch)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float t = scratch[0];
    for (int i = 1; i < nw; i++) t = fmaxf(t, scratch[i]);
    return t;
}

// block_max / block_sum for a workgroup of NW waves known at compile time, on a scratch area nobody else is using
// (no barrier ahead of the write; give the two reductions of a kernel DIFFERENT scratch words).  Same values, and for
// the sum the same wave order, as block_max / block_sum.
template <int NW>
__device__ __forceinline__ float block_max_n(float v, float* scratch)
{
    static_assert(NW == 4, "4 waves");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_max_dpp(v);
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    const float4 a = *(const float4*)scratch;
    return fmaxf(fmaxf(fmaxf(a.x, a.y), a.z), a.w);
}
template <int NW>
__device__ __forceinline__ float block_sum_n(float v, float* scratch)
{
    static_assert(NW == 4, "4 waves");
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    const float4 a = *(const float4*)scratch;
    float t = 0.f;
    t += a.x; t += a.y; t += a.z; t += a.w;
    return t;
}

// Per-thread part of sum x^2 over an f32 row in LDS.  Every kernel that needs the
// RMSNorm statistic uses THIS element->thread mapping and order (thread t owns
// the 8 consecutive elements of groups t, t+blockDim, ...), so the operator path
// and the fused decode path round identically.  d % 8 == 0.
__device__ __forceinline__ float sumsq_tree8(const float (&v)[8])
{
    return ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
}
__device__ __forceinline__ float sumsq_tree4(const float (&v)[4])
{
    return (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
}
// (combine with block_sum_tree; rows longer than 8 * blockDim add further passes sequentially)
__device__ __forceinline__ float row_sumsq8(const float* row, int d)
{
    float ss = 0.f;
    for (int gi = threadIdx.x; gi * 8 < d; gi += blockDim.x) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = row[gi * 8 + i];
        ss += sumsq_tree8(v);
    }
    return ss;
}

// ---- Q8 activation quantizer pieces (gten/quants.h:52-66) ----
// delta = absmax/127 in f32; the STORED delta is fp16(delta) but the rounding
// scale is 1/delta of the unrounded f32 value; roundf = half away from zero.
struct Q8Scale {
    float scale;      // 1/delta or 0
    uint16_t d16;     // stored delta
    float ddeq;       // fp16_to_fp32(stored delta): what every reader multiplies by
};
// a / 127, correctly rounded, in three instructions instead of the ten of hipcc's IEEE division sequence: with
// y = RN(1/127), q0 = RN(a y), the remainder r = a - 127 q0 is exact in an fma and RN(q0 + r y) is the correctly rounded
// quotient (Markstein).  Checked exhaustively against IEEE division over all 2^23 significands of a binade
// (tests/test_golden_cpu.py::test_div127_markstein); binary scaling carries it to every a whose quotient is a normal
// number -- below that (absmax < 1.5e-36) the block's stored delta is 0 either way.
__device__ __forceinline__ float div127(float a)
{
    const float y = 0x1.020408p-7f;
    const float q0 = a * y;
    const float r = __builtin_fmaf(-127.0f, q0, a);
    return __builtin_fmaf(r, y, q0);
}
// 1 / d, correctly rounded, in three instructions: the hardware estimate (1 ulp) and one Newton step whose residual
// is exact in an fma (Markstein).  Checked against the IEEE division expansion on the device over every significand
// of several binades (gten_hip_selftest_q8scale, tests/test_ops_gpu.py); d is a Q8 block delta: a positive normal number
// far from the ends of the exponent range (d == 0 is handled by the caller).
__device__ __forceinline__ float recip_rn(float d)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ Q8Scale q8_scale_from_absmax(float amax)
{
    Q8Scale s;
    const float delta = div127(amax);
    s.d16 = f2h(delta);
    s.ddeq = h2f(s.d16);
    s.scale = (delta != 0.0f) ? recip_rn(delta) : 0.0f;
    return s;
}
// (int)roundf(x) (half away from zero), exactly, for |x| < 2^22: add the largest float below one half with x's sign (one
// f32 addition, round to nearest even) and truncate (the conversion does).  A fraction below .5 leaves the sum at most
// one ulp short of the next integer, a fraction of .5 or more carries it there.  Checked against floor(|x| + .5) over
// every binary32 value from 2^-30 to 2^22 (tests/test_golden_cpu.py::test_round_half_away_by_one_addition); the build
// never contracts the product in front of it into this addition (-ffp-contract=off, tests/test_no_packed_f32_cpu.py).
__device__ __forceinline__ int round_half_away_i(float x)
{
    return (int)(x + copysignf(0x1.fffffep-2f, x));
}
// a Q8 quant: |x * scale| <= 127 by the scale's construction
__device__ __forceinline__ int q8_round(float x, float scale)
{
    return round_half_away_i(x * scale);
}

// a lane's Q8 block (32 quants as 8 dwords + its stored delta) into a row of 34-byte blocks, lane L = block L: the
// pair's 17 dwords are written by the even lane ([d0 | q0 | d1], dwords 0..8) and the odd lane (its quants, dwords 9..16)
__device__ __forceinline__ void store_q8_block_lane(uint8_t* row, int L, const unsigned (&pq)[8], unsigned d16)
{
    const unsigned d_next = (unsigned)__shfl_down((int)d16, 1, 64);
    unsigned* op = (unsigned*)(row + (size_t)(L >> 1) * 68);
    if (!(L & 1)) {
        op[0] = d16 | (pq[0] << 16);
#pragma unroll
        for (int j = 1; j < 8; j++) op[j] = (pq[j - 1] >> 16) | (pq[j] << 16);
        op[8] = (pq[7] >> 16) | (d_next << 16);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) op[9 + j] = pq[j];
    }
}

// ---- storage rows -> f32 in LDS ----
// Q8 rows are 34-byte blocks (2-byte aligned only): byte loads keep it simple
// and coalesce (32 consecutive lanes read 32 consecutive bytes).
__device__ __forceinline__ float load_elem(const uint8_t* row, int dtype, int i)
{
    if (dtype == GTEN_Q8) {
        const uint8_t* blk = row + (size_t)(i >> 5) * GTEN_Q8_BYTES;
        const uint16_t d = *(const uint16_t*)blk;
        return (float)(int8_t)blk[2 + (i & 31)] * h2f(d);
    } else if (dtype == GTEN_F16) {
        return h2f(((const uint16_t*)row)[i]);
    } else {
        return ((const float*)row)[i];
    }
}

__device__ __forceinline__ void load_row_f32(const uint8_t* row, int dtype, int d, float* dst)
{
    for (int i = threadIdx.x; i < d; i += blockDim.x) dst[i] = load_elem(row, dtype, i);
}

// ---- f32 in LDS -> storage row (gten/ops.h:73-96) ----
// `len` may end in a partial Q8 block (attention probability rows,
// gten/quants.h:103-109): only delta + len%32 quants are written for it.
// blockDim.x must be a multiple of 32 so that a block never straddles a step.
__device__ __forceinline__ void store_row(const float* src, int dtype, int len, uint8_t* out)
{
    if (dtype == GTEN_Q8) {
        const int padded = (len + 31) & ~31;
        for (int i = threadIdx.x; i < padded; i += blockDim.x) {
            const bool ok = i < len;
            const float x = ok ? src[i] : 0.0f;
            const float amax = group_max<32>(fabsf(x));
            const Q8Scale s = q8_scale_from_absmax(amax);
            uint8_t* blk = out + (size_t)(i >> 5) * GTEN_Q8_BYTES;
            if (ok) blk[2 + (i & 31)] = (uint8_t)(int8_t)q8_round(x, s.scale);
            if ((i & 31) == 0) *(uint16_t*)blk = s.d16;
        }
    } else if (dtype == GTEN_F16) {
        for (int i = threadIdx.x; i < len; i += blockDim.x) ((uint16_t*)out)[i] = f2h(src[i]);
    } else {
        for (int i = threadIdx.x; i < len; i += blockDim.x) ((float*)out)[i] = src[i];
    }
}

// Round an f32 row held in LDS through the storage dtype IN PLACE (write +
// read back, without touching global memory).  Same partial-tail rule.
__device__ __forceinline__ void round_row_inplace(float* v, int dtype, int len)
{
    if (dtype == GTEN_Q8) {
        const int padded = (len + 31) & ~31;
        for (int i = threadIdx.x; i < padded; i += blockDim.x) {
            const bool ok = i < len;
            const float x = ok ? v[i] : 0.0f;
            const float amax = group_max<32>(fabsf(x));
            const Q8Scale s = q8_scale_from_absmax(amax);
            if (ok) v[i] = (float)q8_round(x, s.scale) * s.ddeq;
        }
    } else if (dtype == GTEN_F16) {
        for (int i = threadIdx.x; i < len; i += blockDim.x) v[i] = h2f(f2h(v[i]));
    }
}

// ---- packed weight addressing (see include/gten_hip.h) ----
struct PackedW {
    const uint8_t* qs;     // quants
    const uint16_t* ds;    // deltas, [rows][nb]
    int nb;                // blocks per row
};
__device__ __forceinline__ PackedW packed_view(const void* w, int dtype, int rows, int cols)
{
    PackedW p;
    p.nb = cols >> 5;
    p.qs = (const uint8_t*)w;
    const size_t qbytes = (size_t)rows * p.nb * (dtype == GTEN_Q4 ? 16 : 32);
    p.ds = (const uint16_t*)((const uint8_t*)w + qbytes);
    return p;
}

// signed 4x int8 dot with accumulate (v_dot4_i32_i8)
__device__ __forceinline__ int dot4(int a, int b, int c)
{
    return __builtin_amdgcn_sdot4(a, b, c, false);
}

// One Q4 weight block (16 bytes of nibbles: byte i holds element i in the high
// nibble and element i+16 in the low nibble, gten/quants.h:78-90) against one
// Q8 activation block held as 8 dwords.  Returns sum a_i * (w_i - 7) exactly.
__device__ __forceinline__ int dot_q8_q4_block(const int (&a)[8], int asum, const uint4 w)
{
    const unsigned m = 0x0f0f0f0fu;
    int acc = 0;
    acc = dot4(a[0], (int)((w.x >> 4) & m), acc);
    acc = dot4(a[1], (int)((w.y >> 4) & m), acc);
    acc = dot4(a[2], (int)((w.z >> 4) & m), acc);
    acc = dot4(a[3], (int)((w.w >> 4) & m), acc);
    acc = dot4(a[4], (int)(w.x & m), acc);
    acc = dot4(a[5], (int)(w.y & m), acc);
    acc = dot4(a[6], (int)(w.z & m), acc);
    acc = dot4(a[7], (int)(w.w & m), acc);
    return acc - 7 * asum;
}
// The same in two steps, for callers that dot one weight block with several activation
// blocks: split the nibbles once, then 8 dot4 per activation block.
struct Q4Unpacked { int hi[4], lo[4]; };
__device__ __forceinline__ Q4Unpacked q4_unpack(const uint4 w)
{
    const unsigned m = 0x0f0f0f0fu;
    Q4Unpacked u;
    u.hi[0] = (int)((w.x >> 4) & m); u.hi[1] = (int)((w.y >> 4) & m); u.hi[2] = (int)((w.z >> 4) & m); u.hi[3] = (int)((w.w >> 4) & m);
    u.lo[0] = (int)(w.x & m); u.lo[1] = (int)(w.y & m); u.lo[2] = (int)(w.z & m); u.lo[3] = (int)(w.w & m);
    return u;
}
__device__ __forceinline__ int dot_q8_q4_unpacked(const int (&a)[8], int asum, const Q4Unpacked& u)
{
    int acc = 0;
    acc = dot4(a[0], u.hi[0], acc);
    acc = dot4(a[1], u.hi[1], acc);
    acc = dot4(a[2], u.hi[2], acc);
    acc = dot4(a[3], u.hi[3], acc);
    acc = dot4(a[4], u.lo[0], acc);
    acc = dot4(a[5], u.lo[1], acc);
    acc = dot4(a[6], u.lo[2], acc);
    acc = dot4(a[7], u.lo[3], acc);
    return acc - 7 * asum;
}
__device__ __forceinline__ int dot_q8_q8_block(const int (&a)[8], const uint4 w0, const uint4 w1)
{
    int acc = 0;
    acc = dot4(a[0], (int)w0.x, acc);
    acc = dot4(a[1], (int)w0.y, acc);
    acc = dot4(a[2], (int)w0.z, acc);
    acc = dot4(a[3], (int)w0.w, acc);
    acc = dot4(a[4], (int)w1.x, acc);
    acc = dot4(a[5], (int)w1.y, acc);
    acc = dot4(a[6], (int)w1.z, acc);
    acc = dot4(a[7], (int)w1.w, acc);
    return acc;
}

// Q8 activation vector staged in LDS in structure-of-arrays form so that a lane
// can fetch "its" block with two 16-byte reads:
//   q   : [nb][32] int8  (16-byte aligned)
//   d   : [nb] f32       (fp16_to_fp32 of the stored delta)
//   sum : [nb] int32     (sum of the 32 quants, for the Q4 "-7" offset)
struct ActQ8 {
    int8_t* q;
    float* d;
    int* sum;
};
__device__ __forceinline__ size_t actq8_bytes(int nb) { return (size_t)nb * 40; }
__device__ __forceinline__ ActQ8 actq8_carve(uint8_t* lds, int nb)
{
    ActQ8 a;
    a.q = (int8_t*)lds;
    a.d = (float*)(lds + (size_t)nb * 32);
    a.sum = (int*)(lds + (size_t)nb * 36);
    return a;
}

// Stage one Q8 storage row (34-byte blocks in global memory) into ActQ8 form.
__device__ __forceinline__ void stage_q8_row(const uint8_t* row, int nb, ActQ8 a)
{
    for (int i = threadIdx.x; i < nb * 32; i += blockDim.x) {
        const uint8_t* blk = row + (size_t)(i >> 5) * GTEN_Q8_BYTES;
        const int qv = (int)(int8_t)blk[2 + (i & 31)];
        a.q[i] = (int8_t)qv;
        int s = qv;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((i & 31) == 0) {
            a.d[i >> 5] = h2f(*(const uint16_t*)blk);
            a.sum[i >> 5] = s;
        }
    }
}

// Quantize an f32 row in LDS into ActQ8 form (gten/quants.h:52-66 per block).
__device__ __forceinline__ void quantize_to_actq8(const float* v, int nb, ActQ8 a)
{
    for (int i = threadIdx.x; i < nb * 32; i += blockDim.x) {
        const float x = v[i];
        const float amax = group_max<32>(fabsf(x));
        const Q8Scale s = q8_scale_from_absmax(amax);
        const int qv = q8_round(x, s.scale);
        a.q[i] = (int8_t)qv;
        int t = qv;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if ((i & 31) == 0) {
            a.d[i >> 5] = s.ddeq;
            a.sum[i >> 5] = t;
        }
    }
}

// ---- one weight row against the staged activation; whole wave cooperates ----
// Lanes stride over the K blocks (lane L takes blocks L, L+64, ...), so each
// wave instruction reads 1 KiB of contiguous quants.  Result is wave-uniform.
__device__ __forceinline__ float wave_dot_q4(const PackedW w, size_t row, const ActQ8 a)
{
    const int lane = threadIdx.x & 63;
    const uint4* qrow = (const uint4*)(w.qs + row * (size_t)w.nb * 16);
    const uint16_t* drow = w.ds + row * (size_t)w.nb;
    float acc = 0.f;
    for (int b = lane; b < w.nb; b += 64) {
        const uint4 wq = qrow[b];
        const float dw = h2f(drow[b]);
        const int4* ap = (const int4*)(a.q + (size_t)b * 32);
        const int4 a0 = ap[0], a1 = ap[1];
        const int av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const int isum = dot_q8_q4_block(av, a.sum[b], wq);
        acc += (float)isum * (a.d[b] * dw);
    }
    return wave_sum(acc);
}
__device__ __forceinline__ float wave_dot_q8(const PackedW w, size_t row, const ActQ8 a)
{
    const int lane = threadIdx.x & 63;
    const uint4* q0 = (const uint4*)(w.qs + row * (size_t)w.nb * 32);
    const uint4* q1 = q0 + w.nb;
    const uint16_t* drow = w.ds + row * (size_t)w.nb;
    float acc = 0.f;
    for (int b = lane; b < w.nb; b += 64) {
        const uint4 w0 = q0[b], w1 = q1[b];
        const float dw = h2f(drow[b]);
        const int4* ap = (const int4*)(a.q + (size_t)b * 32);
        const int4 a0 = ap[0], a1 = ap[1];
        const int av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const int isum = dot_q8_q8_block(av, w0, w1);
        acc += (float)isum * (a.d[b] * dw);
    }
    return wave_sum(acc);
}
// f16 weights x f16 activations (activations staged as f32 in LDS, exact).
// Lane L takes elements [8L, 8L+8) of every 512-element segment; four segments'
// worth of weights are requested together so a row costs one memory latency per
// 2048 elements instead of one per 512.
__device__ __forceinline__ float wave_dot_f16(const uint16_t* wrow, const float* act, int d)
{
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int e0 = lane * 8; e0 < d; e0 += 2048) {
        uint4 wv[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int e = e0 + 512 * k;
            ok[k] = e < d;
            wv[k] = *(const uint4*)(wrow + (ok[k] ? e : e0));      // clamped: no select on loaded data
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!ok[k]) continue;
            const int e = e0 + 512 * k;
            const unsigned u[4] = {wv[k].x, wv[k].y, wv[k].z, wv[k].w};
            const float4 a0 = *(const float4*)(act + e), a1 = *(const float4*)(act + e + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc += h2f((uint16_t)(u[j] & 0xffffu)) * av[2 * j];
                acc += h2f((uint16_t)(u[j] >> 16)) * av[2 * j + 1];
            }
        }
    }
    return wave_sum(acc);
}

} // namespace gtd
