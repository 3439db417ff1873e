// gten_ops.hip -- the ten gten::ops:: entry points as gfx950 kernels
// (include/gten_hip.h).  These are the general forms: any number of new rows
// [start_pos, n), every dtype pair the reference dispatches on
// (gten/ops.h:482-512).  The single-row decode fast path lives in
// gten_decode.hip and produces the same bytes.
//
// Every kernel follows the reference's row discipline: storage dtype -> f32,
// compute in f32, write the row back in the storage dtype (gten/ops.h:40-96).
#include "gten_dev.h"

#include <vector>
#include <algorithm>
#include "gten_rt.h"

#include <cstdlib>

using namespace gtd;

// gten_hip_set_row_segments (include/gten_hip.h): starts[0 .. n] of the prompts sharing one row matrix; empty = one prompt
static std::vector<int> g_seg;

extern __shared__ __attribute__((aligned(16))) uint8_t g_smem[];

// ------------------------------------------------------------------ pack

// .gten block stream (gten/quants.h:17-31) -> packed planes (include/gten_hip.h)
__global__ void k_pack_q4(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t nblk)
{
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const uint8_t* s = src + b * GTEN_Q4_BYTES;
    uint16_t* ds = (uint16_t*)(dst + nblk * 16);
    ds[b] = *(const uint16_t*)s;
    uint8_t* q = dst + b * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) q[i] = s[2 + i];
}

__global__ void k_pack_q8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int nb)
{
    const size_t nblk = (size_t)rows * nb;
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const size_t row = b / nb, col = b % nb;
    const uint8_t* s = src + b * GTEN_Q8_BYTES;
    uint16_t* ds = (uint16_t*)(dst + nblk * 32);
    ds[b] = *(const uint16_t*)s;
    uint8_t* p0 = dst + row * (size_t)nb * 32 + col * 16;
    uint8_t* p1 = p0 + (size_t)nb * 16;
#pragma unroll
    for (int i = 0; i < 16; i++) { p0[i] = s[2 + i]; p1[i] = s[18 + i]; }
}

// ----------------------------------------------------------------- embed

// ops::token_embed, gten/ops.h:514-564
template <int WT>
__global__ __launch_bounds__(256) void k_embed(const void* __restrict__ w, int n_vocab, const int32_t* __restrict__ tokens,
                                               uint8_t* __restrict__ out, int out_dtype, size_t out_pitch, int d, int start_pos)
{
    const int r = start_pos + blockIdx.x;
    const int tok = tokens[r];
    uint8_t* orow = out + (size_t)r * out_pitch;
    const int nb = d >> 5;
    if (WT == GTEN_F16) {
        // verbatim copy (gten/ops.h:529-532)
        const uint16_t* src = (const uint16_t*)w + (size_t)tok * d;
        for (int i = threadIdx.x; i < d; i += blockDim.x) ((uint16_t*)orow)[i] = src[i];
    } else if (WT == GTEN_Q8) {
        // verbatim block copy (gten/ops.h:519-521), un-doing the load-time repack
        const PackedW p = packed_view(w, GTEN_Q8, n_vocab, d);
        const uint8_t* q0 = p.qs + (size_t)tok * nb * 32;
        for (int i = threadIdx.x; i < d; i += blockDim.x) {
            const int b = i >> 5, e = i & 31;
            uint8_t* blk = orow + (size_t)b * GTEN_Q8_BYTES;
            blk[2 + e] = q0[(size_t)(e >> 4) * nb * 16 + (size_t)b * 16 + (e & 15)];
            if (e == 0) *(uint16_t*)blk = p.ds[(size_t)tok * nb + b];
        }
    } else {
        // Q4 row -> f32 -> re-quantized in the activation dtype (gten/ops.h:522-528)
        float* v = (float*)g_smem;
        const PackedW p = packed_view(w, GTEN_Q4, n_vocab, d);
        const uint8_t* q = p.qs + (size_t)tok * nb * 16;
        for (int i = threadIdx.x; i < d; i += blockDim.x) {
            const int b = i >> 5, e = i & 31;
            const uint8_t byte = q[(size_t)b * 16 + (e & 15)];
            const int nib = (e < 16) ? (byte >> 4) : (byte & 0x0f);
            v[i] = (float)(nib - 7) * h2f(p.ds[(size_t)tok * nb + b]);
        }
        __syncthreads();
        store_row(v, out_dtype, d, orow);
    }
}

// ---------------------------------------------------------------- matmul

// ops::matmul_2d, gten/ops.h:613-670.  One workgroup = one new row x 32
// consecutive output features (= one Q8 output block); each of the 4 waves
// walks 8 weight rows, all 64 lanes striding over the K blocks of a row.
template <int WT>
__global__ __launch_bounds__(256) void k_matmul(const uint8_t* __restrict__ x, size_t x_pitch, const void* __restrict__ w,
                                                uint8_t* __restrict__ out, int out_dtype, size_t out_pitch,
                                                int d_in, int d_out, int start_pos)
{
    float* res = (float*)g_smem;                 // 32 results
    uint8_t* act_mem = g_smem + 128;
    const int r = start_pos + blockIdx.y;
    const int c0 = blockIdx.x * 32;
    const uint8_t* xrow = x + (size_t)r * x_pitch;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nb = d_in >> 5;

    ActQ8 a = actq8_carve(act_mem, nb);
    float* actf = (float*)act_mem;
    if (WT == GTEN_F16) load_row_f32(xrow, GTEN_F16, d_in, actf);
    else stage_q8_row(xrow, nb, a);
    __syncthreads();

    const PackedW pw = packed_view(w, WT, d_out, d_in);
#pragma unroll 2
    for (int j = 0; j < 8; j++) {
        const int c = c0 + wid * 8 + j;
        float v = 0.f;
        if (c < d_out) {
            if (WT == GTEN_F16) v = wave_dot_f16((const uint16_t*)w + (size_t)c * d_in, actf, d_in);
            else if (WT == GTEN_Q8) v = wave_dot_q8(pw, (size_t)c, a);
            else v = wave_dot_q4(pw, (size_t)c, a);
        }
        if (lane == 0) res[wid * 8 + j] = v;
    }
    __syncthreads();

    // write_row_from_float for this 32-wide slice (gten/ops.h:73-96)
    if (threadIdx.x < 32) {
        const int c = c0 + threadIdx.x;
        const bool ok = c < d_out;
        const float v = ok ? res[threadIdx.x] : 0.f;
        uint8_t* orow = out + (size_t)r * out_pitch;
        if (out_dtype == GTEN_Q8) {
            const float amax = group_max<32>(fabsf(v));
            const Q8Scale s = q8_scale_from_absmax(amax);
            uint8_t* blk = orow + (size_t)blockIdx.x * GTEN_Q8_BYTES;
            blk[2 + threadIdx.x] = (uint8_t)(int8_t)q8_round(v, s.scale);
            if (threadIdx.x == 0) *(uint16_t*)blk = s.d16;
        } else if (out_dtype == GTEN_F16) {
            if (ok) ((uint16_t*)orow)[c] = f2h(v);
        } else {
            if (ok) ((float*)orow)[c] = v;
        }
    }
}

// ------------------------------------------------------------ row-wise ops

// ops::rms_norm, gten/ops.h:762-814: out = x / (sqrt(mean x^2) + 1e-6) * w
// a16 (Q8 rows, prompt processing): the f16 copy W.x reads -- f16(quant * stored delta) in the fragment order of
// gten_mfma.hip's k_act_to_f16<FAST> (elements 0,2,1,3 of every four), row 0 = start_pos -- written beside the row itself.
__global__ __launch_bounds__(256) void k_rms_norm(const uint8_t* __restrict__ x, int dtype, size_t x_pitch,
                                                  const uint16_t* __restrict__ w, uint8_t* __restrict__ out,
                                                  size_t out_pitch, int d, int start_pos, _Float16* __restrict__ a16)
{
    float* red = (float*)g_smem;
    float* v = (float*)(g_smem + 64);
    const int r = start_pos + blockIdx.x;
    load_row_f32(x + (size_t)r * x_pitch, dtype, d, v);
    __syncthreads();
    const float ss = block_sum_tree(row_sumsq8(v, d), red);
    // x / (rms + eps) * w evaluated as x * (1 / (rms + eps)) * w: one correctly
    // rounded reciprocal per row instead of a division per element (<= 1 ulp
    // apart before the row is rounded to the activation dtype)
    const float inv = recip_rn(sqrtf(ss / (float)d) + 1e-6f);      // == 1.0f / (...): correctly rounded either way
    for (int i = threadIdx.x; i < d; i += blockDim.x) v[i] = v[i] * inv * h2f(w[i]);
    __syncthreads();
    if (a16) {
        // store_row's Q8 branch, with the rounded value kept for the f16 copy
        uint8_t* orow = out + (size_t)r * out_pitch;
        _Float16* arow = a16 + (size_t)blockIdx.x * d;
        for (int i = threadIdx.x; i < d; i += blockDim.x) {
            const float xv = v[i];
            const Q8Scale s = q8_scale_from_absmax(group_max<32>(fabsf(xv)));
            const int qv = q8_round(xv, s.scale);
            uint8_t* blk = orow + (size_t)(i >> 5) * GTEN_Q8_BYTES;
            blk[2 + (i & 31)] = (uint8_t)(int8_t)qv;
            if ((i & 31) == 0) *(uint16_t*)blk = s.d16;
            arow[(i & ~3) + ((i & 1) << 1) + ((i >> 1) & 1)] = f2hv((float)qv * s.ddeq);
        }
        return;
    }
    store_row(v, dtype, d, out + (size_t)r * out_pitch);
}

// The same operator for Q8 rows of exactly 64 blocks (n_embd 2048), ONE WAVE per row: lane L owns block L -- its 34 bytes
// arrive as nine dwords, the row lives in registers, and the only cross-lane steps are the sum tree and one delta handed
// to the neighbour for the straddling dword of a block pair.  The sum of squares is k_rms_norm's tree exactly: a lane's
// 32 elements are leaves 4L .. 4L+3 of the balanced tree over 256 eight-element partials, reduced in the lane, then
// wave_sum continues the same tree over the lanes -- bit-identical rows (tests/test_ops_gpu.py compares the two kernels).
__global__ __launch_bounds__(64) void k_rms_norm_q8w(const uint8_t* __restrict__ x, size_t x_pitch, const uint16_t* __restrict__ w,
                                                     uint8_t* __restrict__ out, size_t out_pitch, int start_pos, uint4* __restrict__ a16)
{
    constexpr int D = 2048;
    const int L = threadIdx.x, odd = L & 1;
    const int r = start_pos + blockIdx.x;
    const unsigned* pw = (const unsigned*)(x + (size_t)r * x_pitch + (size_t)(L >> 1) * 68) + (odd ? 8 : 0);
    unsigned dw[9];
#pragma unroll
    for (int j = 0; j < 9; j++) dw[j] = pw[j];
    uint4 wq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) wq[j] = ((const uint4*)(w + L * 32))[j];
    const float dx = odd ? h2f((uint16_t)(dw[0] >> 16)) : h2f((uint16_t)(dw[0] & 0xffffu));
    float v[32];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned q = odd ? dw[1 + j] : __builtin_amdgcn_alignbit(dw[j + 1], dw[j], 16);
#pragma unroll
        for (int i = 0; i < 4; i++) v[4 * j + i] = (float)(int)(int8_t)(q >> (8 * i)) * dx;
    }
    float part[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float t8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t8[i] = v[8 * k + i];
        part[k] = sumsq_tree8(t8);
    }
    const float ss = wave_sum((part[0] + part[1]) + (part[2] + part[3]));
    const float inv = recip_rn(sqrtf(ss / (float)D) + 1e-6f);
    const unsigned* wh = (const unsigned*)wq;
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint16_t wb = (uint16_t)((i & 1) ? (wh[i >> 1] >> 16) : (wh[i >> 1] & 0xffffu));
        v[i] = v[i] * inv * h2f(wb);
        amax = fmaxf(amax, fabsf(v[i]));
    }
    const Q8Scale sc = q8_scale_from_absmax(amax);
    unsigned pq[8], hw[16];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        unsigned wd = 0;
        unsigned short hq[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int qv = q8_round(v[4 * j + i], sc.scale);
            wd |= ((unsigned)qv & 0xffu) << (8 * i);
            hq[i] = f2h((float)qv * sc.ddeq);
        }
        pq[j] = wd;
        hw[2 * j] = (unsigned)hq[0] | ((unsigned)hq[2] << 16);
        hw[2 * j + 1] = (unsigned)hq[1] | ((unsigned)hq[3] << 16);
    }
    // the pair's 17 dwords: the even lane writes [d0 | q0 | d1] (dwords 0..8), the odd lane its quants (dwords 9..16)
    const unsigned d_next = (unsigned)__shfl_down((int)sc.d16, 1, 64);
    unsigned* op = (unsigned*)(out + (size_t)r * out_pitch + (size_t)(L >> 1) * 68);
    if (!odd) {
        op[0] = (unsigned)sc.d16 | (pq[0] << 16);
#pragma unroll
        for (int j = 1; j < 8; j++) op[j] = (pq[j - 1] >> 16) | (pq[j] << 16);
        op[8] = (pq[7] >> 16) | (d_next << 16);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) op[9 + j] = pq[j];
    }
    if (a16) {
        uint4* dst = a16 + ((size_t)blockIdx.x * 64 + L) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) dst[j] = make_uint4(hw[4 * j], hw[4 * j + 1], hw[4 * j + 2], hw[4 * j + 3]);
    }
}

// rows this kernel takes: Q8, 2048 wide, 4-byte aligned pairs
static bool rms_norm_q8w_ok(const void* x, size_t x_pitch, const void* w, const void* out, size_t out_pitch, int dtype, int d)
{
    return dtype == GTEN_Q8 && d == 2048 && x_pitch % 4 == 0 && out_pitch % 4 == 0 && ((uintptr_t)x & 3) == 0 && ((uintptr_t)out & 3) == 0 &&
           ((uintptr_t)w & 15) == 0;
}

// ops::rotary_emb, gten/ops.h:714-760 (rotate-half pairing, position = row)
__global__ __launch_bounds__(256) void k_rope(uint8_t* __restrict__ x, int dtype, size_t pitch, int d, int d_head,
                                              int start_pos, const float2* __restrict__ table)
{
    float* v = (float*)g_smem;
    const int r = start_pos + blockIdx.x;
    uint8_t* row = x + (size_t)r * pitch;
    load_row_f32(row, dtype, d, v);
    __syncthreads();
    const int half = d_head >> 1;
    for (int i = threadIdx.x; i < (d >> 1); i += blockDim.x) {
        const int h = i / half, j = i % half;
        const float2 cs = table[(size_t)r * half + j];
        const float x0 = v[h * d_head + j], x1 = v[h * d_head + j + half];
        v[h * d_head + j] = x0 * cs.x - x1 * cs.y;
        v[h * d_head + j + half] = x0 * cs.y + x1 * cs.x;
    }
    __syncthreads();
    store_row(v, dtype, d, row);
}

// the same for TWO matrices in one launch (q and k of a prompt: blockIdx.y picks the matrix)
__global__ __launch_bounds__(256) void k_rope2(uint8_t* __restrict__ x0, size_t pitch0, int d0, uint8_t* __restrict__ x1, size_t pitch1, int d1,
                                               int dtype, int d_head, int start_pos, const float2* __restrict__ table)
{
    float* v = (float*)g_smem;
    const int r = start_pos + blockIdx.x;
    const bool second = blockIdx.y != 0;
    const int d = second ? d1 : d0;
    uint8_t* row = second ? x1 + (size_t)r * pitch1 : x0 + (size_t)r * pitch0;
    load_row_f32(row, dtype, d, v);
    __syncthreads();
    const int half = d_head >> 1;
    for (int i = threadIdx.x; i < (d >> 1); i += blockDim.x) {
        const int h = i / half, j = i % half;
        const float2 cs = table[(size_t)r * half + j];
        const float x0v = v[h * d_head + j], x1v = v[h * d_head + j + half];
        v[h * d_head + j] = x0v * cs.x - x1v * cs.y;
        v[h * d_head + j + half] = x0v * cs.y + x1v * cs.x;
    }
    __syncthreads();
    store_row(v, dtype, d, row);
}

// ... and for Q8 rows with 64-wide heads, ONE WAVE per (row, matrix): lane L owns block L -- the first (even L) or second
// (odd L) half of head L / 2 -- and takes its partner's 32 values from the neighbouring lane; k_rope's arithmetic per
// element, the row in registers, dword loads and stores (8 -> 4 us per launch on a 256-id prompt).
__global__ __launch_bounds__(64) void k_rope2_q8w(uint8_t* __restrict__ x0, size_t pitch0, int nblk0, uint8_t* __restrict__ x1, size_t pitch1, int nblk1,
                                                  int start_pos, const float2* __restrict__ table)
{
    const int L = threadIdx.x, odd = L & 1, r = start_pos + blockIdx.x;
    const bool second = blockIdx.y != 0;
    if (L >= (second ? nblk1 : nblk0)) return;
    uint8_t* row = second ? x1 + (size_t)r * pitch1 : x0 + (size_t)r * pitch0;
    const unsigned* pw = (const unsigned*)(row + (size_t)(L >> 1) * 68) + (odd ? 8 : 0);
    unsigned dw[9];
#pragma unroll
    for (int j = 0; j < 9; j++) dw[j] = pw[j];
    float4 cs[16];                                              // (cos, sin) of this position for j = 0 .. 31, two per load
#pragma unroll
    for (int j = 0; j < 16; j++) cs[j] = ((const float4*)(table + (size_t)r * 32))[j];
    const float dx = odd ? h2f((uint16_t)(dw[0] >> 16)) : h2f((uint16_t)(dw[0] & 0xffffu));
    float v[32];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned q = odd ? dw[1 + j] : __builtin_amdgcn_alignbit(dw[j + 1], dw[j], 16);
#pragma unroll
        for (int i = 0; i < 4; i++) v[4 * j + i] = (float)(int)(int8_t)(q >> (8 * i)) * dx;
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const float other = __shfl_xor(v[j], 1, 64);
        const float c = (j & 1) ? cs[j >> 1].z : cs[j >> 1].x, sn = (j & 1) ? cs[j >> 1].w : cs[j >> 1].y;
        // even lane: x0 * cos - x1 * sin with x1 the partner's; odd lane: x0 * sin + x1 * cos with x0 the partner's
        v[j] = odd ? other * sn + v[j] * c : v[j] * c - other * sn;
        amax = fmaxf(amax, fabsf(v[j]));
    }
    const Q8Scale sc = q8_scale_from_absmax(amax);
    unsigned pq[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        unsigned wd = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) wd |= ((unsigned)q8_round(v[4 * j + i], sc.scale) & 0xffu) << (8 * i);
        pq[j] = wd;
    }
    store_q8_block_lane(row, L, pq, sc.d16);
}
static bool rope_q8w_ok(const void* x, size_t pitch, int dtype, int d, int d_head)
{
    return dtype == GTEN_Q8 && d_head == 64 && d % 64 == 0 && d <= 2048 && pitch % 4 == 0 && ((uintptr_t)x & 3) == 0;
}

enum { EW_SILU = 0, EW_MUL = 1, EW_ADD = 2 };

// ops::silu / mul / add, gten/ops.h:673-711, 816-910
template <int OP>
__global__ __launch_bounds__(256) void k_elementwise(const uint8_t* a, const uint8_t* b,
                                                     uint8_t* out, int dtype, size_t pitch, int d, int start_pos)
{
    float* v = (float*)g_smem;
    const int r = start_pos + blockIdx.x;
    const uint8_t* arow = a + (size_t)r * pitch;
    const uint8_t* brow = (OP == EW_SILU) ? nullptr : b + (size_t)r * pitch;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const float xa = load_elem(arow, dtype, i);
        float o;
        if (OP == EW_SILU) o = xa / (1.0f + expf(-xa));
        else if (OP == EW_MUL) o = xa * load_elem(brow, dtype, i);
        else o = xa + load_elem(brow, dtype, i);
        v[i] = o;
    }
    __syncthreads();   // all reads of this row are done before an in-place write
    store_row(v, dtype, d, out + (size_t)r * pitch);
}

// The same three operators for Q8 rows of many blocks (prompt processing: 2048 rows x 5632 elements per call): one thread
// per PAIR of adjacent Q8 blocks.  A pair is 68 bytes, always 4-byte aligned, so it travels as 17 dwords each way
// (the byte-per-lane loads of k_elementwise reach a fifth of the memory rate); dequantize, operate and re-quantize in
// registers -- the block absmax needs no cross-lane step.  Per element the arithmetic is k_elementwise's: bit-identical.
template <int OP>
__global__ __launch_bounds__(256) void k_elementwise_q8x2(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* out,
                                                          size_t pitch, int pairs_per_row, int start_pos, int total_pairs)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_pairs) return;
    const int r = start_pos + gid / pairs_per_row, pr = gid % pairs_per_row;
    const size_t off = (size_t)r * pitch + (size_t)pr * 68;
    unsigned aw[17], bw[17];
    const unsigned* ap = (const unsigned*)(a + off);
#pragma unroll
    for (int j = 0; j < 17; j++) aw[j] = ap[j];
    if (OP != EW_SILU) {
        const unsigned* bp = (const unsigned*)(b + off);
#pragma unroll
        for (int j = 0; j < 17; j++) bw[j] = bp[j];
    }
    unsigned ow[17];
    unsigned d16o[2];
    unsigned pq[2][8];
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        // pair bytes: [d0 | q0 x32 | d1 | q1 x32]; q0 straddles the dwords by 2 bytes
        const float da = blk ? h2f((uint16_t)(aw[8] >> 16)) : h2f((uint16_t)(aw[0] & 0xffffu));
        const float db = (OP == EW_SILU) ? 0.f : (blk ? h2f((uint16_t)(bw[8] >> 16)) : h2f((uint16_t)(bw[0] & 0xffffu)));
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned qa = blk ? aw[9 + j] : __builtin_amdgcn_alignbit(aw[j + 1], aw[j], 16);
            const unsigned qb = (OP == EW_SILU) ? 0u : (blk ? bw[9 + j] : __builtin_amdgcn_alignbit(bw[j + 1], bw[j], 16));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float xa = (float)(int)(int8_t)(qa >> (8 * i)) * da;
                float o;
                if (OP == EW_SILU) o = xa / (1.0f + expf(-xa));
                else if (OP == EW_MUL) o = xa * ((float)(int)(int8_t)(qb >> (8 * i)) * db);
                else o = xa + ((float)(int)(int8_t)(qb >> (8 * i)) * db);
                v[4 * j + i] = o;
                amax = fmaxf(amax, fabsf(o));
            }
        }
        const Q8Scale sc = q8_scale_from_absmax(amax);
        d16o[blk] = sc.d16;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned w = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) w |= ((unsigned)q8_round(v[4 * j + i], sc.scale) & 0xffu) << (8 * i);
            pq[blk][j] = w;
        }
    }
    ow[0] = d16o[0] | (pq[0][0] << 16);
#pragma unroll
    for (int j = 1; j < 8; j++) ow[j] = (pq[0][j - 1] >> 16) | (pq[0][j] << 16);
    ow[8] = (pq[0][7] >> 16) | (d16o[1] << 16);
#pragma unroll
    for (int j = 0; j < 8; j++) ow[9 + j] = pq[1][j];
    unsigned* op = (unsigned*)(out + off);
#pragma unroll
    for (int j = 0; j < 17; j++) op[j] = ow[j];
}

// silu_inplace(gate) followed by mul_inplace(gate, up) (gten/modules.cpp:244-249) in ONE pass over Q8 block pairs: the
// silu values are rounded to their Q8 block exactly as the first operator stores them, dequantized again, multiplied and
// rounded to the block the second operator stores -- the bytes `gate` ends with are those of the two launches.
__global__ __launch_bounds__(256) void k_silu_mul_q8x2(uint8_t* gate, const uint8_t* __restrict__ up, size_t pitch, int pairs_per_row,
                                                       int start_pos, int total_pairs, uint4* __restrict__ a16)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total_pairs) return;
    const int r = start_pos + gid / pairs_per_row, pr = gid % pairs_per_row;
    const size_t off = (size_t)r * pitch + (size_t)pr * 68;
    unsigned aw[17], bw[17];
    const unsigned* ap = (const unsigned*)(gate + off);
    const unsigned* bp = (const unsigned*)(up + off);
#pragma unroll
    for (int j = 0; j < 17; j++) aw[j] = ap[j];
#pragma unroll
    for (int j = 0; j < 17; j++) bw[j] = bp[j];
    unsigned ow[17];
    unsigned d16o[2];
    unsigned pq[2][8];
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
        const float da = blk ? h2f((uint16_t)(aw[8] >> 16)) : h2f((uint16_t)(aw[0] & 0xffffu));
        const float db = blk ? h2f((uint16_t)(bw[8] >> 16)) : h2f((uint16_t)(bw[0] & 0xffffu));
        float v[32];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned qa = blk ? aw[9 + j] : __builtin_amdgcn_alignbit(aw[j + 1], aw[j], 16);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float xa = (float)(int)(int8_t)(qa >> (8 * i)) * da;
                const float o = xa / (1.0f + expf(-xa));
                v[4 * j + i] = o;
                amax = fmaxf(amax, fabsf(o));
            }
        }
        const Q8Scale s1 = q8_scale_from_absmax(amax);
        const float ds = s1.ddeq;
        amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned qb = blk ? bw[9 + j] : __builtin_amdgcn_alignbit(bw[j + 1], bw[j], 16);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float xs = (float)q8_round(v[4 * j + i], s1.scale) * ds;
                const float o = xs * ((float)(int)(int8_t)(qb >> (8 * i)) * db);
                v[4 * j + i] = o;
                amax = fmaxf(amax, fabsf(o));
            }
        }
        const Q8Scale sc = q8_scale_from_absmax(amax);
        d16o[blk] = sc.d16;
        unsigned hw[16];                                          // the block as f16(quant * delta), elements 0,2,1,3 of every four
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned w = 0;
            unsigned short hq[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int qv = q8_round(v[4 * j + i], sc.scale);
                w |= ((unsigned)qv & 0xffu) << (8 * i);
                hq[i] = f2h((float)qv * sc.ddeq);
            }
            pq[blk][j] = w;
            hw[2 * j] = (unsigned)hq[0] | ((unsigned)hq[2] << 16);
            hw[2 * j + 1] = (unsigned)hq[1] | ((unsigned)hq[3] << 16);
        }
        if (a16) {
            uint4* dst = a16 + ((size_t)(gid / pairs_per_row) * (2 * pairs_per_row) + 2 * pr + blk) * 4;
#pragma unroll
            for (int j = 0; j < 4; j++) dst[j] = make_uint4(hw[4 * j], hw[4 * j + 1], hw[4 * j + 2], hw[4 * j + 3]);
        }
    }
    ow[0] = d16o[0] | (pq[0][0] << 16);
#pragma unroll
    for (int j = 1; j < 8; j++) ow[j] = (pq[0][j - 1] >> 16) | (pq[0][j] << 16);
    ow[8] = (pq[0][7] >> 16) | (d16o[1] << 16);
#pragma unroll
    for (int j = 0; j < 8; j++) ow[9 + j] = pq[1][j];
    unsigned* op = (unsigned*)(gate + off);
#pragma unroll
    for (int j = 0; j < 17; j++) op[j] = ow[j];
}

// ... and for f16 rows: one thread per 8 elements (16 bytes each way); per element k_elementwise's arithmetic
template <int OP>
__global__ __launch_bounds__(256) void k_elementwise_f16x8(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint8_t* out,
                                                           size_t pitch, int groups_per_row, int start_pos, int total_groups)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_groups) return;
    const int r = start_pos + gid / groups_per_row, gr = gid % groups_per_row;
    const size_t off = (size_t)r * pitch + (size_t)gr * 16;
    const uint4 av = *(const uint4*)(a + off);
    uint4 bv = make_uint4(0, 0, 0, 0);
    if (OP != EW_SILU) bv = *(const uint4*)(b + off);
    const unsigned aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
    unsigned ow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        unsigned o2 = 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float xa = h2f((uint16_t)(aw[j] >> (16 * i)));
            float o;
            if (OP == EW_SILU) o = xa / (1.0f + expf(-xa));
            else if (OP == EW_MUL) o = xa * h2f((uint16_t)(bw[j] >> (16 * i)));
            else o = xa + h2f((uint16_t)(bw[j] >> (16 * i)));
            o2 |= (unsigned)f2h(o) << (16 * i);
        }
        ow[j] = o2;
    }
    *(uint4*)(out + off) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}
// silu_inplace(gate) then mul_inplace(gate, up) on f16 rows in one pass: the silu value is rounded to f16 as the first
// operator stores it, widened again, multiplied and rounded as the second stores it
__global__ __launch_bounds__(256) void k_silu_mul_f16x8(uint8_t* gate, const uint8_t* __restrict__ up, size_t pitch, int groups_per_row,
                                                        int start_pos, int total_groups)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= total_groups) return;
    const int r = start_pos + gid / groups_per_row, gr = gid % groups_per_row;
    const size_t off = (size_t)r * pitch + (size_t)gr * 16;
    const uint4 av = *(const uint4*)(gate + off), bv = *(const uint4*)(up + off);
    const unsigned aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
    unsigned ow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        unsigned o2 = 0;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float xa = h2f((uint16_t)(aw[j] >> (16 * i)));
            const float sv = h2f(f2h(xa / (1.0f + expf(-xa))));
            o2 |= (unsigned)f2h(sv * h2f((uint16_t)(bw[j] >> (16 * i)))) << (16 * i);
        }
        ow[j] = o2;
    }
    *(uint4*)(gate + off) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

static bool elementwise_f16x8_ok(const void* a, const void* b, const void* out, int dtype, size_t pitch, int rows, int d)
{
    const auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return dtype == GTEN_F16 && d % 8 == 0 && pitch % 16 == 0 && al(a) && al(out) && (!b || al(b)) && (size_t)rows * d >= 32768;
}
template <int OP>
static int launch_elementwise_f16x8(const void* a, const void* b, void* out, size_t pitch, int n, int d, int start_pos)
{
    const int gpr = d / 8, total = (n - start_pos) * gpr;
    GTR_LAUNCH(KT_ELEMWISE, (k_elementwise_f16x8<OP>), dim3((total + 255) / 256), dim3(256), 0,
               (const uint8_t*)a, (const uint8_t*)b, (uint8_t*)out, pitch, gpr, start_pos, total);
    return 0;
}

// rows x blocks big enough for the block-pair kernel, and everything 4-byte aligned
static bool elementwise_q8x2_ok(const void* a, const void* b, const void* out, int dtype, size_t pitch, int rows, int d)
{
    const auto al = [](const void* p) { return ((uintptr_t)p & 3) == 0; };
    return dtype == GTEN_Q8 && (d / 32) % 2 == 0 && pitch % 4 == 0 && al(a) && al(out) && (!b || al(b)) && (size_t)rows * d >= 32768;
}
template <int OP>
static int launch_elementwise_q8x2(const void* a, const void* b, void* out, size_t pitch, int n, int d, int start_pos)
{
    const int ppr = d / 64, total = (n - start_pos) * ppr;
    GTR_LAUNCH(KT_ELEMWISE, (k_elementwise_q8x2<OP>), dim3((total + 255) / 256), dim3(256), 0,
               (const uint8_t*)a, (const uint8_t*)b, (uint8_t*)out, pitch, ppr, start_pos, total);
    return 0;
}

// -------------------------------------------------------------- attention

// ops::qkv_attn, gten/ops.h:930-1133.  One workgroup per (head, new row).
// LDS: p[n] f32 | q staged | reduction scratch | p.V partials.
__global__ __launch_bounds__(256) void k_attn(const uint8_t* __restrict__ q, const uint8_t* __restrict__ k,
                                              const uint8_t* __restrict__ v, uint8_t* __restrict__ out, int dtype,
                                              size_t q_pitch, size_t kv_pitch, size_t out_pitch,
                                              int n_heads, int n_kv, int d_head, int start_pos, int p_cap)
{
    const int h = blockIdx.x;
    const int r = start_pos + blockIdx.y;
    const int g = h / (n_heads / n_kv);
    const int nk = r + 1;                      // columns c <= r are unmasked (gten/ops.h:957)
    const int nblk = d_head >> 5;
    const size_t head_bytes = (dtype == GTEN_Q8) ? (size_t)nblk * GTEN_Q8_BYTES : (size_t)d_head * 2;

    float* red = (float*)g_smem;                          // 16
    float* qf = red + 16;                                 // d_head floats (f16 mode) / ints (Q8 mode)
    int* qi = (int*)qf;
    float* qd = qf + d_head;                              // nblk
    float* outv = qd + 8;                                 // d_head
    float* part = outv + d_head;                          // 256
    float* p = part + 256;                                // p_cap

    const uint8_t* qrow = q + (size_t)r * q_pitch + (size_t)h * head_bytes;
    if (dtype == GTEN_Q8) {
        if (threadIdx.x < d_head / 4) {
            const int b = threadIdx.x >> 3, j = threadIdx.x & 7;
            const uint16_t* qw = (const uint16_t*)(qrow + (size_t)b * GTEN_Q8_BYTES);
            qi[threadIdx.x] = (int)((unsigned)qw[1 + 2 * j] | ((unsigned)qw[2 + 2 * j] << 16));
            if (j == 0) qd[b] = h2f(qw[0]);
        }
    } else {
        for (int e = threadIdx.x; e < d_head; e += blockDim.x) qf[e] = h2f(((const uint16_t*)qrow)[e]);
    }
    __syncthreads();

    const float scale = 1.0f / sqrtf((float)d_head);
    float lmax = -INFINITY;
    for (int c = threadIdx.x; c < nk; c += blockDim.x) {
        const uint8_t* kp = k + (size_t)c * kv_pitch + (size_t)g * head_bytes;
        float s = 0.f;
        if (dtype == GTEN_Q8) {
            // integer block dots scaled by the two deltas (gten/ops.h:224-316)
            for (int b = 0; b < nblk; b++) {
                const uint16_t* kw = (const uint16_t*)(kp + (size_t)b * GTEN_Q8_BYTES);
                int isum = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int kv4 = (int)((unsigned)kw[1 + 2 * j] | ((unsigned)kw[2 + 2 * j] << 16));
                    isum = dot4(qi[b * 8 + j], kv4, isum);
                }
                s += (float)isum * (qd[b] * h2f(kw[0]));
            }
        } else {
            const uint16_t* k16 = (const uint16_t*)kp;
            for (int e = 0; e < d_head; e++) s += qf[e] * h2f(k16[e]);
        }
        s *= scale;
        p[c] = s;
        lmax = fmaxf(lmax, s);
    }
    const float mx = block_max(lmax, red);
    float lsum = 0.f;
    for (int c = threadIdx.x; c < nk; c += blockDim.x) {
        const float e = expf(p[c] - mx);
        p[c] = e;
        lsum += e;
    }
    const float tot = block_sum(lsum, red);
    for (int c = threadIdx.x; c < nk; c += blockDim.x) p[c] = p[c] / tot;
    __syncthreads();
    // the probability row is stored in the activation dtype (gten/ops.h:996-997)
    round_row_inplace(p, dtype, nk);
    __syncthreads();

    // out[e] = sum_c p[c] * V[c][g][e]  (gten/ops.h:1046-1089)
    const int ngrp = blockDim.x / d_head;
    const int e = threadIdx.x % d_head, grp = threadIdx.x / d_head;
    float acc = 0.f;
    if (grp < ngrp) {
        const uint8_t* vbase = v + (size_t)g * head_bytes;
        for (int c = grp; c < nk; c += ngrp) acc += p[c] * load_elem(vbase + (size_t)c * kv_pitch, dtype, e);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < d_head) {
        float o = 0.f;
        for (int gi = 0; gi < ngrp; gi++) o += part[gi * d_head + threadIdx.x];
        outv[threadIdx.x] = o;
    }
    __syncthreads();
    store_row(outv, dtype, d_head, out + (size_t)r * out_pitch + (size_t)h * head_bytes);
}

// ------------------------------------------------------------------ C-ABI

using namespace gtr;
const int* gtr::row_segments(int* n_segments)
{
    if (n_segments) *n_segments = g_seg.empty() ? 0 : (int)g_seg.size() - 1;
    return g_seg.empty() ? nullptr : g_seg.data();
}

static bool act_dtype_ok(int dt) { return dt == GTEN_F16 || dt == GTEN_Q8; }

// ---- self-test of the quantizer's scale arithmetic (see include/gten_hip.h)
__global__ __launch_bounds__(256) void k_selftest_q8scale(unsigned long long* __restrict__ bad)
{
    const int exps[6] = {-40, -20, -7, 0, 6, 20};                 // binades of the operand
    const unsigned m = blockIdx.x * 256u + threadIdx.x;           // significand bits, 0 .. 2^23 - 1
    unsigned bd = 0, br = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const float a = __uint_as_float(((unsigned)(127 + exps[k]) << 23) | m);
        volatile float av = a;                                    // keep the compiler from folding either side
        const float ref_d = av / 127.0f, ref_r = 1.0f / av;
        bd += (__float_as_uint(gtd::div127(a)) != __float_as_uint(ref_d));
        br += (__float_as_uint(gtd::recip_rn(a)) != __float_as_uint(ref_r));
    }
    if (bd) atomicAdd(bad, (unsigned long long)bd);
    if (br) atomicAdd(bad + 1, (unsigned long long)br);
}

extern "C" {

int gten_hip_selftest_q8scale(unsigned long long* mismatches_div127, unsigned long long* mismatches_recip)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(mismatches_div127 && mismatches_recip, "selftest_q8scale: null argument");
    unsigned long long* d = nullptr;
    GTR_CHECK(hipMalloc((void**)&d, 16));
    GTR_CHECK(hipMemsetAsync(d, 0, 16, gtr::stream()));
    hipLaunchKernelGGL(k_selftest_q8scale, dim3((1u << 23) / 256), dim3(256), 0, gtr::stream(), d);
    GTR_LAUNCHED();
    unsigned long long h[2] = {0, 0};
    GTR_CHECK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, gtr::stream()));
    GTR_CHECK(hipStreamSynchronize(gtr::stream()));
    GTR_CHECK(hipFree(d));
    *mismatches_div127 = h[0];
    *mismatches_recip = h[1];
    return 0;
}


int gten_hip_pack_weight(const void* src_blocks, int dtype, int rows, int cols, void* dst_packed)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(src_blocks && dst_packed && rows > 0 && cols > 0, "pack_weight: bad arguments");
    kv_watch_touch(dst_packed, (size_t)rows * gten_hip_row_bytes(dtype, cols));      // (every writer of device memory: gten_rt.h, watched K / V caches)
    if (dtype == GTEN_F16) {
        GTR_CHECK(hipMemcpyAsync(dst_packed, src_blocks, (size_t)rows * cols * 2, hipMemcpyDeviceToDevice, stream()));
        return 0;
    }
    GTR_REQUIRE(dtype == GTEN_Q8 || dtype == GTEN_Q4, "pack_weight: dtype %d is not a weight dtype", dtype);
    GTR_REQUIRE(cols % 32 == 0, "pack_weight: cols %d not a multiple of the block size 32", cols);
    const size_t nblk = (size_t)rows * (cols / 32);
    const unsigned grid = (unsigned)((nblk + 255) / 256);
    if (dtype == GTEN_Q4)
        GTR_LAUNCH(KT_PACK, k_pack_q4, dim3(grid), dim3(256), 0, (const uint8_t*)src_blocks, (uint8_t*)dst_packed, nblk);
    else
        GTR_LAUNCH(KT_PACK, k_pack_q8, dim3(grid), dim3(256), 0, (const uint8_t*)src_blocks, (uint8_t*)dst_packed, rows, cols / 32);
    return 0;
}

int gten_hip_token_embed(const void* w, int w_dtype, int n_vocab, const int32_t* tokens,
                         void* out, int out_dtype, size_t out_pitch, int n, int d, int start_pos)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(w && tokens && out, "token_embed: null pointer");
    GTR_REQUIRE(n > 0 && start_pos >= 0 && start_pos < n, "token_embed: bad rows n=%d start_pos=%d", n, start_pos);
    GTR_REQUIRE(d > 0 && d % 32 == 0 && d <= 8192, "token_embed: unsupported width %d", d);
    GTR_REQUIRE(act_dtype_ok(out_dtype), "token_embed: bad output dtype %d", out_dtype);
    GTR_REQUIRE(out_pitch >= gten_hip_row_bytes(out_dtype, d), "token_embed: output pitch too small");
    kv_watch_touch(out, (size_t)n * out_pitch);
    const dim3 grid(n - start_pos), block(256);
    if (w_dtype == GTEN_F16) {
        GTR_REQUIRE(out_dtype == GTEN_F16, "token_embed: f16 table needs f16 output (row copy, gten/ops.h:529)");
        GTR_LAUNCH(KT_EMBED, (k_embed<GTEN_F16>), grid, block, 0, w, n_vocab, tokens, (uint8_t*)out, out_dtype, out_pitch, d, start_pos);
    } else if (w_dtype == GTEN_Q8) {
        GTR_REQUIRE(out_dtype == GTEN_Q8, "token_embed: Q8 table needs Q8 output (block copy, gten/ops.h:519)");
        GTR_LAUNCH(KT_EMBED, (k_embed<GTEN_Q8>), grid, block, 0, w, n_vocab, tokens, (uint8_t*)out, out_dtype, out_pitch, d, start_pos);
    } else if (w_dtype == GTEN_Q4) {
        GTR_REQUIRE(out_dtype == GTEN_Q8, "token_embed: Q4 table needs Q8 output (gten/ops.h:523)");
        GTR_LAUNCH(KT_EMBED, (k_embed<GTEN_Q4>), grid, block, (size_t)d * 4, w, n_vocab, tokens, (uint8_t*)out, out_dtype, out_pitch, d, start_pos);
    } else {
        return fail(-4, "token_embed: bad table dtype %d", w_dtype);
    }
    return 0;
}

int gten_hip_matmul_2d(const void* x, int x_dtype, size_t x_pitch, const void* w, int w_dtype,
                       void* out, int out_dtype, size_t out_pitch, int n, int d_in, int d_out, int start_pos)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(x && w && out, "matmul_2d: null pointer");
    GTR_REQUIRE(n > 0 && start_pos >= 0 && start_pos < n, "matmul_2d: bad rows n=%d start_pos=%d", n, start_pos);
    GTR_REQUIRE(d_in > 0 && d_in % 32 == 0 && d_in <= 16384, "matmul_2d: unsupported d_in %d", d_in);
    GTR_REQUIRE(d_out > 0, "matmul_2d: bad d_out %d", d_out);
    const bool pair_ok = (x_dtype == GTEN_F16 && w_dtype == GTEN_F16) || (x_dtype == GTEN_Q8 && (w_dtype == GTEN_Q8 || w_dtype == GTEN_Q4));
    GTR_REQUIRE(pair_ok, "matmul_2d: unsupported dtype pair (%d,%d) (gten/ops.h:482-512)", x_dtype, w_dtype);
    GTR_REQUIRE(out_dtype == GTEN_F16 || out_dtype == GTEN_Q8 || out_dtype == GTEN_F32, "matmul_2d: bad output dtype %d", out_dtype);
    GTR_REQUIRE(out_dtype != GTEN_Q8 || d_out % 32 == 0, "matmul_2d: Q8 output needs d_out %% 32 == 0 (got %d)", d_out);
    GTR_REQUIRE(x_pitch >= gten_hip_row_bytes(x_dtype, d_in), "matmul_2d: input pitch too small");
    GTR_REQUIRE(out_pitch >= gten_hip_row_bytes(out_dtype, d_out), "matmul_2d: output pitch too small");
    GTR_REQUIRE(n - start_pos <= 65535, "matmul_2d: too many new rows");
    kv_watch_touch(out, (size_t)n * out_pitch);       // (the key / value projections write the K / V caches: gten/modules.cpp:188-201)
    // prefill-sized calls go to the matrix cores (gten_mfma.hip); the row-per-workgroup
    // kernel below streams the weights once per row and is meant for a handful of rows
    if (n - start_pos >= GTEN_MFMA_MIN_ROWS && d_in % 128 == 0)   // the MFMA kernel stages 4 quant blocks at a time
        return gten_launch_matmul_mfma(x, x_dtype, x_pitch, w, w_dtype, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    const dim3 grid((d_out + 31) / 32, n - start_pos), block(256);
    const size_t act = (w_dtype == GTEN_F16) ? (size_t)d_in * 4 : (size_t)(d_in / 32) * 40;
    const size_t smem = 128 + act;
    if (w_dtype == GTEN_F16)
        GTR_LAUNCH(KT_MATMUL, (k_matmul<GTEN_F16>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, d_in, d_out, start_pos);
    else if (w_dtype == GTEN_Q8)
        GTR_LAUNCH(KT_MATMUL, (k_matmul<GTEN_Q8>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, d_in, d_out, start_pos);
    else
        GTR_LAUNCH(KT_MATMUL, (k_matmul<GTEN_Q4>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, d_in, d_out, start_pos);
    return 0;
}

static int check_rowwise(const char* op, const void* a, const void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    GTR_REQUIRE(a && out, "%s: null pointer", op);
    GTR_REQUIRE(act_dtype_ok(dtype), "%s: bad activation dtype %d", op, dtype);
    GTR_REQUIRE(n > 0 && start_pos >= 0 && start_pos < n, "%s: bad rows n=%d start_pos=%d", op, n, start_pos);
    GTR_REQUIRE(d > 0 && d % 32 == 0 && d <= 12288, "%s: unsupported width %d", op, d);
    GTR_REQUIRE(pitch >= gten_hip_row_bytes(dtype, d), "%s: pitch too small", op);
    return 0;
}

int gten_hip_rms_norm(const void* x, int dtype, size_t x_pitch, const void* w_f16,
                      void* out, size_t out_pitch, int n, int d, int start_pos)
{
    GTR_NEED_INIT();
    if (int rc = check_rowwise("rms_norm", x, out, dtype, x_pitch, n, d, start_pos)) return rc;
    GTR_REQUIRE(w_f16 && out_pitch >= gten_hip_row_bytes(dtype, d), "rms_norm: bad weight/output");
    kv_watch_touch(out, (size_t)n * out_pitch);
    if (n - start_pos >= 4 && rms_norm_q8w_ok(x, x_pitch, w_f16, out, out_pitch, dtype, d)) {
        GTR_LAUNCH(KT_RMSNORM, k_rms_norm_q8w, dim3(n - start_pos), dim3(64), 0, (const uint8_t*)x, x_pitch, (const uint16_t*)w_f16, (uint8_t*)out, out_pitch,
                   start_pos, (uint4*)nullptr);
        return 0;
    }
    GTR_LAUNCH(KT_RMSNORM, k_rms_norm, dim3(n - start_pos), dim3(256), 64 + (size_t)d * 4,
                       (const uint8_t*)x, dtype, x_pitch, (const uint16_t*)w_f16, (uint8_t*)out, out_pitch, d, start_pos, (_Float16*)nullptr);
    return 0;
}

int gten_hip_rotary_emb(void* x, int dtype, size_t pitch, int n, int d, int d_head, int start_pos)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(g_seg.empty(), "rotary_emb: row segments are set (gten_hip_set_row_segments): only gten_hip_block_rows rotates per prompt");
    if (int rc = check_rowwise("rotary_emb", x, x, dtype, pitch, n, d, start_pos)) return rc;
    GTR_REQUIRE(d_head > 0 && d_head % 2 == 0 && d % d_head == 0, "rotary_emb: bad d_head %d for width %d", d_head, d);
    GTR_REQUIRE(n <= GTEN_ROPE_MAX_POS, "rotary_emb: position %d beyond the table (%d)", n, GTEN_ROPE_MAX_POS);
    kv_watch_touch(x, (size_t)n * pitch);             // (K is rotated in place in its cache: gten/modules.cpp:199)
    const float2* table = nullptr;
    if (int rc = rope_table(d_head, &table)) return rc;
    if (n - start_pos >= 4 && rope_q8w_ok(x, pitch, dtype, d, d_head)) {
        GTR_LAUNCH(KT_ROPE, k_rope2_q8w, dim3(n - start_pos, 1), dim3(64), 0, (uint8_t*)x, pitch, d / 32, (uint8_t*)x, pitch, 0, start_pos, table);
        return 0;
    }
    GTR_LAUNCH(KT_ROPE, k_rope, dim3(n - start_pos), dim3(256), (size_t)d * 4, (uint8_t*)x, dtype, pitch, d, d_head, start_pos, table);
    return 0;
}

int gten_hip_silu(const void* x, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    GTR_NEED_INIT();
    if (int rc = check_rowwise("silu", x, out, dtype, pitch, n, d, start_pos)) return rc;
    kv_watch_touch(out, (size_t)n * pitch);
    if (elementwise_q8x2_ok(x, nullptr, out, dtype, pitch, n - start_pos, d)) return launch_elementwise_q8x2<EW_SILU>(x, nullptr, out, pitch, n, d, start_pos);
    if (elementwise_f16x8_ok(x, nullptr, out, dtype, pitch, n - start_pos, d)) return launch_elementwise_f16x8<EW_SILU>(x, nullptr, out, pitch, n, d, start_pos);
    GTR_LAUNCH(KT_ELEMWISE, (k_elementwise<EW_SILU>), dim3(n - start_pos), dim3(256), (size_t)d * 4,
                       (const uint8_t*)x, (const uint8_t*)nullptr, (uint8_t*)out, dtype, pitch, d, start_pos);
    return 0;
}

int gten_hip_mul(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    GTR_NEED_INIT();
    if (int rc = check_rowwise("mul", a, out, dtype, pitch, n, d, start_pos)) return rc;
    GTR_REQUIRE(b, "mul: null pointer");
    kv_watch_touch(out, (size_t)n * pitch);
    if (elementwise_q8x2_ok(a, b, out, dtype, pitch, n - start_pos, d)) return launch_elementwise_q8x2<EW_MUL>(a, b, out, pitch, n, d, start_pos);
    if (elementwise_f16x8_ok(a, b, out, dtype, pitch, n - start_pos, d)) return launch_elementwise_f16x8<EW_MUL>(a, b, out, pitch, n, d, start_pos);
    GTR_LAUNCH(KT_ELEMWISE, (k_elementwise<EW_MUL>), dim3(n - start_pos), dim3(256), (size_t)d * 4,
                       (const uint8_t*)a, (const uint8_t*)b, (uint8_t*)out, dtype, pitch, d, start_pos);
    return 0;
}

int gten_hip_add(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    GTR_NEED_INIT();
    if (int rc = check_rowwise("add", a, out, dtype, pitch, n, d, start_pos)) return rc;
    GTR_REQUIRE(b, "add: null pointer");
    kv_watch_touch(out, (size_t)n * pitch);
    if (elementwise_q8x2_ok(a, b, out, dtype, pitch, n - start_pos, d)) return launch_elementwise_q8x2<EW_ADD>(a, b, out, pitch, n, d, start_pos);
    if (elementwise_f16x8_ok(a, b, out, dtype, pitch, n - start_pos, d)) return launch_elementwise_f16x8<EW_ADD>(a, b, out, pitch, n, d, start_pos);
    GTR_LAUNCH(KT_ELEMWISE, (k_elementwise<EW_ADD>), dim3(n - start_pos), dim3(256), (size_t)d * 4,
                       (const uint8_t*)a, (const uint8_t*)b, (uint8_t*)out, dtype, pitch, d, start_pos);
    return 0;
}

// the greedy sampler's argmax on the device (tinyllama.cpp:416-424: strict >, the first maximum wins), one row of f32 logits
__global__ __launch_bounds__(1024) void k_argmax_row(const float* __restrict__ v, int n, int32_t* __restrict__ out)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float x = v[i];
        if (x > best) { best = x; idx = i; }                 // (ascending i per thread: its first maximum)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { bv[wid] = best; bi[wid] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[0] = (idx == 0x7fffffff) ? 0 : idx;
    }
}

int gten_hip_argmax_row(const float* logits, int n, int32_t* out)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(logits && out && n > 0, "argmax_row: bad arguments");
    kv_watch_touch(out, 4);
    GTR_LAUNCH(KT_ELEMWISE, k_argmax_row, dim3(1), dim3(1024), 0, logits, n, out);
    return 0;
}

int gten_hip_qkv_attn(const void* q, const void* k, const void* v, void* out, int dtype,
                      size_t q_pitch, size_t kv_pitch, size_t out_pitch,
                      int n, int n_heads, int n_kv_heads, int d_head, int start_pos)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(q && k && v && out, "qkv_attn: null pointer");
    GTR_REQUIRE(g_seg.empty(), "qkv_attn: row segments are set (gten_hip_set_row_segments): only gten_hip_block_rows attends per prompt");
    GTR_REQUIRE(act_dtype_ok(dtype), "qkv_attn: bad activation dtype %d", dtype);
    GTR_REQUIRE(n > 0 && start_pos >= 0 && start_pos < n, "qkv_attn: bad rows n=%d start_pos=%d", n, start_pos);
    GTR_REQUIRE(n_heads > 0 && n_kv_heads > 0 && n_heads % n_kv_heads == 0, "qkv_attn: bad head counts %d/%d", n_heads, n_kv_heads);
    GTR_REQUIRE(d_head % 32 == 0 && d_head >= 32 && d_head <= 256, "qkv_attn: unsupported d_head %d", d_head);
    GTR_REQUIRE(q_pitch >= gten_hip_row_bytes(dtype, n_heads * d_head) && out_pitch >= gten_hip_row_bytes(dtype, n_heads * d_head) &&
                kv_pitch >= gten_hip_row_bytes(dtype, n_kv_heads * d_head), "qkv_attn: pitch too small");
    GTR_REQUIRE(n - start_pos <= 65535 && n <= 12288, "qkv_attn: context %d too long for this kernel", n);
    kv_watch_touch(out, (size_t)n * out_pitch);
    if (dtype == GTEN_Q8 && d_head == 64 && n - start_pos >= GTEN_ATTN_TILED_MIN_ROWS) {
        // prompt processing: 32 rows of a head per workgroup, int8 MFMA scores (same bytes as k_attn below)
        // (fewer than 16 new rows take the row kernel: tests reach it for long inputs by calling in 15-row pieces)
        return gten_launch_attn_tiled(q, k, v, out, q_pitch, kv_pitch, out_pitch, n, n_heads, n_kv_heads, start_pos);
    }
    if (dtype == GTEN_F16 && d_head == 64 && n - start_pos >= GTEN_ATTN_TILED_MIN_ROWS && q_pitch % 16 == 0 && kv_pitch % 16 == 0 &&
        ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)out % 4 == 0) && out_pitch % 4 == 0) {
        return gten_launch_attn_tiled_f16(q, k, v, out, q_pitch, kv_pitch, out_pitch, n, n_heads, n_kv_heads, start_pos);
    }
    const int p_cap = (n + 31) & ~31;
    const size_t smem = (size_t)(16 + d_head + 8 + d_head + 256 + p_cap) * 4;
    GTR_LAUNCH(KT_ATTN, k_attn, dim3(n_heads, n - start_pos), dim3(256), smem,
                       (const uint8_t*)q, (const uint8_t*)k, (const uint8_t*)v, (uint8_t*)out, dtype,
                       q_pitch, kv_pitch, out_pitch, n_heads, n_kv_heads, d_head, start_pos, p_cap);
    return 0;
}

static bool g_block_rows = true;           // gten_hip_set_block_rows

int gten_hip_set_block_rows(int on)
{
    g_block_rows = on != 0;
    return 0;
}

// One AttentionBlock over MANY new rows (gten/modules.cpp:224-254) in 9 launches (12 in the exact form) instead of the 23 of the module-by-module
// sequence: the f16 copy of an input is made once for the matrices that share it, q | k | v and gate | up are one W.x launch
// each, q and k are rotated in one launch, silu and the product are one pass, and the two residual sums ride in the
// epilogues of the o and down projections.  Every buffer ends with the bytes the module sequence leaves in it.
int gten_hip_block_rows(const gten_hip_block_desc* b, int n, int start_pos)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(b, "block_rows: null descriptor");
    const int E = b->n_embd, F = b->n_ffn, rows = n - start_pos;
    const int dh = b->n_heads > 0 ? E / b->n_heads : 0, KV = dh * b->n_kv_heads;
    const bool off = !g_block_rows;
    if (!g_seg.empty())
        GTR_REQUIRE(start_pos == 0 && n == g_seg.back(), "block_rows: row segments cover %d rows, the call has rows [%d, %d)", g_seg.back(), start_pos, n);
    // the prompts of this call: (first row, rows) -- one, unless row segments are set
    std::vector<std::pair<int, int>> segs;
    if (g_seg.empty()) segs.push_back({start_pos, rows});
    else for (size_t k = 0; k + 1 < g_seg.size(); k++) segs.push_back({g_seg[k], g_seg[k + 1] - g_seg[k]});
    const bool seg = !g_seg.empty();
    // what this path computes; everything else stays with the operators (the caller falls back on GTEN_HIP_NOT_HANDLED)
    const bool q8cfg = b->adtype == GTEN_Q8 && (b->wdtype == GTEN_Q8 || b->wdtype == GTEN_Q4);
    const bool f16cfg = b->adtype == GTEN_F16 && b->wdtype == GTEN_F16;
    if (off || !(q8cfg || f16cfg) || rows < GTEN_MFMA_MIN_ROWS ||
        rows > 65535 || n > (seg ? GTEN_SEG_MAX_ROWS : GTEN_ROPE_MAX_POS) || start_pos < 0 || dh != 64 || E % 128 != 0 || F % 128 != 0 || KV % 32 != 0 || (F / 32) % 2 != 0 ||
        b->n_heads % b->n_kv_heads != 0)
        return GTEN_HIP_NOT_HANDLED;
    const void* ptrs[] = {b->attn_norm_w, b->wq, b->wk, b->wv, b->wo, b->ffn_norm_w, b->wgate, b->wup, b->wdown, b->inp, b->attn_norm_out,
                          b->q, b->k, b->v, b->attn_out, b->o, b->h, b->ffn_norm_out, b->gate, b->up, b->down, b->out};
    for (const void* p : ptrs) GTR_REQUIRE(p && ((uintptr_t)p & 15) == 0, "block_rows: null or unaligned pointer");
    if (kv_watch_any()) {
        // every buffer this call writes (k and v are the K / V caches); the operators it is composed of touch only their own outputs
        const size_t rE = gten_hip_row_bytes(b->adtype, E), rKV = gten_hip_row_bytes(b->adtype, KV), rF = gten_hip_row_bytes(b->adtype, F);
        const std::pair<const void*, size_t> outs[] = {{b->attn_norm_out, rE}, {b->q, rE}, {b->k, rKV}, {b->v, rKV}, {b->attn_out, rE}, {b->o, rE}, {b->h, rE},
                                                       {b->ffn_norm_out, rE}, {b->gate, rF}, {b->up, rF}, {b->down, rE}, {b->out, rE}};
        for (const auto& o : outs) kv_watch_touch(o.first, (size_t)n * o.second);
    }
    if (f16cfg) {
        // the f16 configuration: no f16 copies to make (the rows are the operands), otherwise the same composition
        const size_t hE = (size_t)E * 2, hKV = (size_t)KV * 2, hF = (size_t)F * 2;
        int rc;
        if ((rc = gten_hip_rms_norm(b->inp, GTEN_F16, hE, b->attn_norm_w, b->attn_norm_out, hE, n, E, start_pos))) return rc;
        {
            MfmaMats m;
            m.n = 3;
            m.w[0] = b->wq; m.out[0] = b->q; m.out_pitch[0] = hE; m.d_out[0] = E;
            m.w[1] = b->wk; m.out[1] = b->k; m.out_pitch[1] = hKV; m.d_out[1] = KV;
            m.w[2] = b->wv; m.out[2] = b->v; m.out_pitch[2] = hKV; m.d_out[2] = KV;
            if ((rc = gten_launch_matmul_mfma_multi(b->attn_norm_out, hE, GTEN_F16, m, GTEN_F16, n, E, start_pos, false))) return rc;
        }
        {
            const float2* table = nullptr;
            if ((rc = rope_table(dh, &table))) return rc;
            // (row segments: every prompt from position 0 -- base pointers moved to its first row, start_pos 0)
            for (const auto& sg : segs) {
                const size_t r0 = seg ? (size_t)sg.first : 0;
                GTR_LAUNCH(KT_ROPE, k_rope2, dim3(sg.second, 2), dim3(256), (size_t)E * 4, (uint8_t*)b->q + r0 * hE, hE, E, (uint8_t*)b->k + r0 * hKV, hKV, KV,
                           GTEN_F16, dh, seg ? 0 : start_pos, table);
            }
        }
        for (const auto& sg : segs) {
            const size_t r0 = seg ? (size_t)sg.first : 0;
            if ((rc = gten_launch_attn_tiled_f16((const uint8_t*)b->q + r0 * hE, (const uint8_t*)b->k + r0 * hKV, (const uint8_t*)b->v + r0 * hKV,
                                                 (uint8_t*)b->attn_out + r0 * hE, hE, hKV, hE, seg ? sg.second : n, b->n_heads, b->n_kv_heads, seg ? 0 : start_pos))) return rc;
        }
        {
            MfmaMats m;
            m.n = 1; m.w[0] = b->wo; m.out[0] = b->o; m.out_pitch[0] = hE; m.d_out[0] = E;
            m.resid = b->inp; m.sum_out = b->h; m.resid_pitch = hE;
            if ((rc = gten_launch_matmul_mfma_multi(b->attn_out, hE, GTEN_F16, m, GTEN_F16, n, E, start_pos, false))) return rc;
        }
        if ((rc = gten_hip_rms_norm(b->h, GTEN_F16, hE, b->ffn_norm_w, b->ffn_norm_out, hE, n, E, start_pos))) return rc;
        {
            MfmaMats m;
            m.n = 2;
            m.w[0] = b->wgate; m.out[0] = b->gate; m.out_pitch[0] = hF; m.d_out[0] = F;
            m.w[1] = b->wup; m.out[1] = b->up; m.out_pitch[1] = hF; m.d_out[1] = F;
            if ((rc = gten_launch_matmul_mfma_multi(b->ffn_norm_out, hE, GTEN_F16, m, GTEN_F16, n, E, start_pos, false))) return rc;
        }
        {
            const int gpr = F / 8, total = rows * gpr;
            GTR_LAUNCH(KT_ELEMWISE, k_silu_mul_f16x8, dim3((total + 255) / 256), dim3(256), 0, (uint8_t*)b->gate, (const uint8_t*)b->up, hF, gpr, start_pos, total);
        }
        {
            MfmaMats m;
            m.n = 1; m.w[0] = b->wdown; m.out[0] = b->down; m.out_pitch[0] = hE; m.d_out[0] = E;
            m.resid = b->h; m.sum_out = b->out; m.resid_pitch = hE;
            if ((rc = gten_launch_matmul_mfma_multi(b->gate, hF, GTEN_F16, m, GTEN_F16, n, F, start_pos, false))) return rc;
        }
        return 0;
    }
    const size_t pE = gten_hip_row_bytes(GTEN_Q8, E), pKV = gten_hip_row_bytes(GTEN_Q8, KV), pF = gten_hip_row_bytes(GTEN_Q8, F);
    int rc;
    // fast form: the producers of the W.x inputs write the f16 copy themselves (one launch less per projection input);
    // exact form: the copy is integers + a delta table, made by its own launch
    const bool fold = !prefill_exact();
    _Float16* a16 = nullptr;
    if (fold && (rc = gten_mfma_scratch(rows, E > F ? E : F, (uint8_t**)&a16))) return rc;
    auto norm = [&](const void* x, const void* w, void* out) -> int {
        if (!fold) {
            if (int e = gten_hip_rms_norm(x, GTEN_Q8, pE, w, out, pE, n, E, start_pos)) return e;
            return gten_mfma_convert(out, pE, n, E, start_pos);
        }
        if (rms_norm_q8w_ok(x, pE, w, out, pE, GTEN_Q8, E))
            GTR_LAUNCH(KT_RMSNORM, k_rms_norm_q8w, dim3(rows), dim3(64), 0, (const uint8_t*)x, pE, (const uint16_t*)w, (uint8_t*)out, pE, start_pos, (uint4*)a16);
        else
            GTR_LAUNCH(KT_RMSNORM, k_rms_norm, dim3(rows), dim3(256), 64 + (size_t)E * 4, (const uint8_t*)x, GTEN_Q8, pE, (const uint16_t*)w, (uint8_t*)out,
                       pE, E, start_pos, a16);
        return 0;
    };
    bool normed = false;
    // attention half
    if ((rc = norm(b->inp, b->attn_norm_w, b->attn_norm_out))) return rc;
    {
        MfmaMats m;
        m.n = 3;
        m.w[0] = b->wq; m.out[0] = b->q; m.out_pitch[0] = pE; m.d_out[0] = E;
        m.w[1] = b->wk; m.out[1] = b->k; m.out_pitch[1] = pKV; m.d_out[1] = KV;
        m.w[2] = b->wv; m.out[2] = b->v; m.out_pitch[2] = pKV; m.d_out[2] = KV;
        if ((rc = gten_launch_matmul_mfma_multi(b->attn_norm_out, pE, b->wdtype, m, GTEN_Q8, n, E, start_pos, true))) return rc;
    }
    {
        const float2* table = nullptr;
        if ((rc = rope_table(dh, &table))) return rc;
        const bool wave_rows = rope_q8w_ok(b->q, pE, GTEN_Q8, E, dh) && rope_q8w_ok(b->k, pKV, GTEN_Q8, KV, dh);
        // (row segments: every prompt from position 0 -- base pointers moved to its first row, start_pos 0)
        for (const auto& sg : segs) {
            const size_t r0 = seg ? (size_t)sg.first : 0;
            uint8_t* qs = (uint8_t*)b->q + r0 * pE;
            uint8_t* ks = (uint8_t*)b->k + r0 * pKV;
            if (wave_rows)
                GTR_LAUNCH(KT_ROPE, k_rope2_q8w, dim3(sg.second, 2), dim3(64), 0, qs, pE, E / 32, ks, pKV, KV / 32, seg ? 0 : start_pos, table);
            else
                GTR_LAUNCH(KT_ROPE, k_rope2, dim3(sg.second, 2), dim3(256), (size_t)E * 4, qs, pE, E, ks, pKV, KV, GTEN_Q8, dh, seg ? 0 : start_pos, table);
        }
    }
    for (const auto& sg : segs) {
        const size_t r0 = seg ? (size_t)sg.first : 0;
        if ((rc = gten_launch_attn_tiled((const uint8_t*)b->q + r0 * pE, (const uint8_t*)b->k + r0 * pKV, (const uint8_t*)b->v + r0 * pKV,
                                         (uint8_t*)b->attn_out + r0 * pE, pE, pKV, pE, seg ? sg.second : n, b->n_heads, b->n_kv_heads, seg ? 0 : start_pos,
                                         fold ? (void*)(a16 + r0 * (size_t)E) : nullptr))) return rc;
    }
    {
        MfmaMats m;
        m.n = 1; m.w[0] = b->wo; m.out[0] = b->o; m.out_pitch[0] = pE; m.d_out[0] = E;
        m.resid = b->inp; m.sum_out = b->h; m.resid_pitch = pE;
        // (fast form: the plane sums of a shared K loop continue into the FFN's RMSNorm -- k_splitk_finish_norm)
        if (fold) { m.norm_w = b->ffn_norm_w; m.norm_out = b->ffn_norm_out; m.norm_out_pitch = pE; m.norm_a16 = a16; m.norm_done = &normed; }
        if ((rc = gten_launch_matmul_mfma_multi(b->attn_out, pE, b->wdtype, m, GTEN_Q8, n, E, start_pos, fold))) return rc;
    }
    // feed-forward half
    if (!normed && (rc = norm(b->h, b->ffn_norm_w, b->ffn_norm_out))) return rc;
    {
        MfmaMats m;
        m.n = 2;
        m.w[0] = b->wgate; m.out[0] = b->gate; m.out_pitch[0] = pF; m.d_out[0] = F;
        m.w[1] = b->wup; m.out[1] = b->up; m.out_pitch[1] = pF; m.d_out[1] = F;
        if ((rc = gten_launch_matmul_mfma_multi(b->ffn_norm_out, pE, b->wdtype, m, GTEN_Q8, n, E, start_pos, true))) return rc;
    }
    {
        const int ppr = F / 64, total = rows * ppr;
        // (a thread walks 64 elements: short prompts get one-wave workgroups so that every CU has one)
        const int nt = total < 256 * 256 ? 64 : 256;
        GTR_LAUNCH(KT_ELEMWISE, k_silu_mul_q8x2, dim3((total + nt - 1) / nt), dim3(nt), 0, (uint8_t*)b->gate, (const uint8_t*)b->up, pF, ppr, start_pos, total, fold ? (uint4*)a16 : (uint4*)nullptr);
    }
    {
        MfmaMats m;
        m.n = 1; m.w[0] = b->wdown; m.out[0] = b->down; m.out_pitch[0] = pE; m.d_out[0] = E;
        m.resid = b->h; m.sum_out = b->out; m.resid_pitch = pE;
        if ((rc = gten_launch_matmul_mfma_multi(b->gate, pF, b->wdtype, m, GTEN_Q8, n, F, start_pos, fold))) return rc;
    }
    return 0;
}

int gten_hip_row_segments_ok(int n_embd, int n_ffn, int n_heads, int n_kv_heads, int wdtype, int adtype)
{
    if (!g_block_rows || n_heads <= 0 || n_kv_heads <= 0) return 0;
    const bool off = false;
    const int dh = n_embd / n_heads, KV = dh * n_kv_heads;
    const bool q8cfg = adtype == GTEN_Q8 && (wdtype == GTEN_Q8 || wdtype == GTEN_Q4), f16cfg = adtype == GTEN_F16 && wdtype == GTEN_F16;
    return !off && (q8cfg || f16cfg) && dh == 64 && n_embd % 128 == 0 && n_ffn % 128 == 0 && KV % 32 == 0 && (n_ffn / 32) % 2 == 0 && n_heads % n_kv_heads == 0;
}

int gten_hip_set_row_segments(const int32_t* starts, int n_segments)
{
    GTR_NEED_INIT();
    if (n_segments <= 0) { g_seg.clear(); return 0; }
    GTR_REQUIRE(starts && starts[0] == 0, "set_row_segments: starts[0] must be 0");
    for (int k = 0; k < n_segments; k++)
        GTR_REQUIRE(starts[k + 1] - starts[k] >= GTEN_MFMA_MIN_ROWS, "set_row_segments: segment %d has %d rows (at least %d)", k, starts[k + 1] - starts[k], GTEN_MFMA_MIN_ROWS);
    for (int k = 0; k < n_segments; k++)
        GTR_REQUIRE(starts[k + 1] - starts[k] <= GTEN_ROPE_MAX_POS, "set_row_segments: segment %d has %d rows (at most %d: the RoPE table)", k, starts[k + 1] - starts[k],
                    GTEN_ROPE_MAX_POS);
    GTR_REQUIRE(starts[n_segments] <= GTEN_SEG_MAX_ROWS, "set_row_segments: %d rows in all (at most %d)", starts[n_segments], GTEN_SEG_MAX_ROWS);
    g_seg.assign(starts, starts + n_segments + 1);
    return 0;
}

// n ranges by value in the kernel arguments: no table to upload, nothing for the host to keep alive
struct CopyRanges { gten_hip_copy_range r[GTEN_HIP_MAX_COPY_RANGES]; };
__global__ __launch_bounds__(256) void k_copy_ranges(const CopyRanges t)
{
    const gten_hip_copy_range r = t.r[blockIdx.y];
    const uintptr_t al = (uintptr_t)r.dst | (uintptr_t)r.src | r.bytes;      // (uniform per range)
    const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    if ((al & 15) == 0) for (size_t i = t0; i < (r.bytes >> 4); i += step) ((uint4*)r.dst)[i] = ((const uint4*)r.src)[i];
    else if ((al & 3) == 0) for (size_t i = t0; i < (r.bytes >> 2); i += step) ((unsigned*)r.dst)[i] = ((const unsigned*)r.src)[i];
    else for (size_t i = t0; i < r.bytes; i += step) ((uint8_t*)r.dst)[i] = ((const uint8_t*)r.src)[i];
}
int gten_hip_copy_ranges(const gten_hip_copy_range* ranges, int n)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(ranges && n >= 0 && n <= GTEN_HIP_MAX_COPY_RANGES, "copy_ranges: %d ranges (at most %d)", n, GTEN_HIP_MAX_COPY_RANGES);
    if (n == 0) return 0;
    CopyRanges t{};
    size_t longest = 0;
    for (int i = 0; i < n; i++) {
        GTR_REQUIRE(ranges[i].dst && ranges[i].src, "copy_ranges: range %d is null", i);
        kv_watch_touch(ranges[i].dst, ranges[i].bytes);
        t.r[i] = ranges[i];
        longest = std::max(longest, ranges[i].bytes);
    }
    const int gx = (int)std::min<size_t>(32, std::max<size_t>(1, (longest / 16 + 255) / 256));
    GTR_LAUNCH(KT_ELEMWISE, k_copy_ranges, dim3(gx, n), dim3(256), 0, t);
    return 0;
}

} // extern "C"
