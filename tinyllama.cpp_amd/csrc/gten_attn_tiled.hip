// gten_attn_tiled.hip -- ops::qkv_attn for MANY new rows (prompt processing), Q8 activations, d_head 64.
//
// Replaces, for that case, the row-at-a-time k_attn of gten_ops.hip (which re-reads the whole K/V
// prefix once per (head, row): 9 GB of L2 traffic per block at n = 2048).  Reference: gten/ops.h:930-1116
// (qk scores, causal mask, softmax, probabilities rounded to the activation dtype, p.V).
//
// One workgroup = 32 query rows of one head.  The K/V prefix is walked in 256-position tiles, three
// times: row maxima, sums of exponentials, probabilities.  Scores are recomputed each time instead
// of kept (32 x 2048 f32 would not fit LDS), which costs almost nothing because a Q8 score IS an
// int8 matrix product: one v_mfma_i32_16x16x32_i8 is exactly one 32-wide quant block of 16 rows x
// 16 positions, and the two block sums are scaled by delta_q * delta_k in f32 as the scalar code
// does.  p.V stays on the VALU (f32 mul + add per term, the bound of this kernel): V's per-position
// deltas sit inside the sum, so it is not an integer matrix product, and the row kernel's association
// (which the fused decoder shares) fixes the order of the adds.
//
// BIT-IDENTICAL to k_attn (and therefore to the fused decoder for contexts <= 256): every
// floating-point reduction is laid out so that it reproduces k_attn's association --
//   * sum of exponentials: k_attn thread t adds positions t, t+256, ...; here lane (wave w,
//     column group cg, lane%16) owns position 64w + 16cg + lane%16 of every tile: the same 256
//     running sums, then the same tree (16-lane DPP rows, (R0+R1)+(R2+R3), waves in order);
//   * p.V: four accumulators per output for positions = 0,1,2,3 (mod 4), summed in that order.
#include "gten_rt.h"
#include "gten_dev.h"

using namespace gtd;
using namespace gtr;

namespace {

constexpr int AT_ROWS = 32;        // query rows per workgroup
constexpr int AT_TILE = 256;       // context positions per tile
constexpr int AT_VSUB = 64;        // positions of V staged at a time
constexpr int AT_PPITCH = 260;     // floats per probability row in LDS (16-byte aligned rows)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
// quant bytes sit at 2-byte aligned offsets inside a 34-byte block: unaligned vector loads (gfx9 global
// memory takes them)
struct __attribute__((packed, aligned(2))) U2 { uint32_t v[2]; };
struct __attribute__((packed, aligned(2))) U4 { uint32_t v[4]; };

// maximum over each row of 16 lanes, any sign (all lanes active): four v_max_f32 with a DPP operand.  Written out because
// the compiler spells a step as v_mov 0 + v_mov_dpp + a canonicalising v_max + v_max -- 16 instructions where 4 do, in a
// kernel bound by VALU issue; the two wait states a DPP read needs behind the VALU write of its source are in the text
// (the hazard recogniser does not look inside it).
__device__ __forceinline__ float row16_max(float v)
{
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"                       // ... and behind the LAST write too: whatever the compiler places next (a DPP or
        : "+v"(v));                     // readlane consumer of v) gets its wait states -- it pads nothing behind an asm
    return v;
}
// the first four steps of wave_sum(): every lane of a 16-lane row ends with the row's sum
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ long as_long(const U2& u)
{
    long r;
    __builtin_memcpy(&r, &u, 8);
    return r;
}

struct QFrag {
    long a[2][2];          // [row group][quant block]: MFMA A operand, row = lane % 16, k = 8 * (lane / 16) ..
    float d[2][2][4];      // deltas of the 4 output rows of this lane, times 1 / sqrt(64): [row group][block][i]
};

// scores of 2 x 16 query rows against the 16 positions of one column group
__device__ __forceinline__ void tile_scores(const QFrag& q, const uint8_t* __restrict__ kslice, int lq, float (&s)[2][4])
{
    const U2 b0 = *(const U2*)(kslice + 2 + 8 * lq);
    const U2 b1 = *(const U2*)(kslice + 36 + 8 * lq);
    const float kd0 = h2f(*(const uint16_t*)kslice), kd1 = h2f(*(const uint16_t*)(kslice + 34));
    const long kb0 = as_long(b0), kb1 = as_long(b1);
    const v4i z = {0, 0, 0, 0};
#pragma unroll
    for (int rg = 0; rg < 2; rg++) {
        const v4i i0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(q.a[rg][0], kb0, z, 0, 0, 0);
        const v4i i1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(q.a[rg][1], kb1, z, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
            // gten/ops.h:224-316: block sums scaled by both deltas, their sum by 1 / sqrt(64) -- q.d carries that factor: a power
            // of two commutes with every rounding on the way (no term is near the subnormal range), the same bits in 7
            // instructions per score instead of 9
            s[rg][i] = (float)i0[i] * (q.d[rg][0][i] * kd0) + (float)i1[i] * (q.d[rg][1][i] * kd1);
    }
}

// FAST (the default; gten_hip_set_prefill_exact(1) selects the exact form): p.V on the matrix cores.  The probabilities
// -- still rounded to Q8 blocks along the context exactly as above -- are kept as f16 rows, a V sub-tile is staged
// TRANSPOSED as f16 values [element][position], and one v_mfma_f32_16x16x32_f16 adds 32 positions of 16 rows x 16
// elements inside the matrix core: 16 MFMAs per wave and 256-position tile instead of ~2700 VALU instructions per
// thread.  Each operand element carries one fp16 rounding (relative 2^-11) and the sum follows the core's order, not
// k_attn's four stride-4 accumulators: outputs agree with the exact form to f32 / fp16 rounding noise (inside the
// model band: tests/test_prefill_gpu.py against the reference's full-size prompt goldens), not byte for byte.
constexpr int AT_PHPITCH = AT_TILE + 8;      // halfs per probability row (528 bytes: 16 rows x 16-byte reads cover all banks)
constexpr int AT_VTPITCH = AT_VSUB + 8;      // halfs per element row of the transposed V sub-tile (144 bytes)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_attn_tiled_q8(const uint8_t* __restrict__ q, const uint8_t* __restrict__ k,
                                                       const uint8_t* __restrict__ v, uint8_t* __restrict__ out,
                                                       size_t q_pitch, size_t kv_pitch, size_t out_pitch,
                                                       int n_heads, int n_kv, int n, int start_pos, _Float16* __restrict__ a16)
{
    constexpr int P_BYTES = FAST ? AT_ROWS * AT_PHPITCH * 2 : AT_ROWS * AT_PPITCH * 4;
    constexpr int V_BYTES = FAST ? 64 * AT_VTPITCH * 2 + AT_ROWS * 64 * 4 : AT_VSUB * 64 * 4;
    __shared__ __attribute__((aligned(16))) uint8_t s_mem[P_BYTES + V_BYTES];
    float* s_p = (float*)s_mem;                                 // exact: f32 probability rows
    float* s_v = (float*)(s_mem + P_BYTES);                     // exact: f32 V sub-tile [position][element]
    _Float16* s_ph = (_Float16*)s_mem;                          // fast: f16 probability rows
    _Float16* s_vt = (_Float16*)(s_mem + P_BYTES);              // fast: f16 V sub-tile, transposed [element][position]
    float* s_o = (float*)(s_mem + P_BYTES + 64 * AT_VTPITCH * 2);   // fast: the output tile [row][element] on its way to the epilogue
    __shared__ float s_red[4 * AT_ROWS];
    __shared__ float s_row[AT_ROWS];

    const int h = blockIdx.x;
    const int rt = gridDim.y - 1 - blockIdx.y;            // long (late) row tiles are scheduled first
    const int r0 = start_pos + rt * AT_ROWS;
    const int g = h / (n_heads / n_kv);
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lc = l & 15, lq = l >> 4;
    const int r_last = min(r0 + AT_ROWS, n) - 1;          // last row of this tile: positions 0..r_last matter
    const int ntile = r_last / AT_TILE + 1;
    const uint8_t* kbase = k + (size_t)g * 68;
    const uint8_t* vbase = v + (size_t)g * 68;

    QFrag qf;
#pragma unroll
    for (int rg = 0; rg < 2; rg++) {
        const uint8_t* qs = q + (size_t)min(r0 + 16 * rg + lc, n - 1) * q_pitch + (size_t)h * 68;
        qf.a[rg][0] = as_long(*(const U2*)(qs + 2 + 8 * lq));
        qf.a[rg][1] = as_long(*(const U2*)(qs + 36 + 8 * lq));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint8_t* qr = q + (size_t)min(r0 + 16 * rg + 4 * lq + i, n - 1) * q_pitch + (size_t)h * 68;
            qf.d[rg][0][i] = h2f(*(const uint16_t*)qr) * 0.125f;          // (1 / sqrt(64): tile_scores)
            qf.d[rg][1][i] = h2f(*(const uint16_t*)(qr + 34)) * 0.125f;
        }
    }
    const int row_base = r0 + 4 * lq;                      // this lane's rows: row_base + 16 rg + i

    float mx[2][4], tot[2][4];
    if (FAST) {
        // ---- fast form: ONE walk over the K / V prefix.  The row maximum runs along (mx); a tile's probabilities are formed
        //      against the maximum so far, e = exp(s - mx), rounded to Q8 blocks (the quants are scale-free; the stored block
        //      delta is rounded at the tile's scale instead of the row's: relative 2^-11, the decode path's deviation 4) and
        //      multiplied into V at once; when the maximum grows, the sums so far and the output accumulators are rescaled by
        //      exp(old - new).  The row sum (lsum, per lane over the positions it owns) divides the output at the end.
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) { mx[rg][i] = -INFINITY; tot[rg][i] = 1.0f; }
    } else {
    // ---- pass 0: row maxima
#pragma unroll
    for (int rg = 0; rg < 2; rg++)
#pragma unroll
        for (int i = 0; i < 4; i++) mx[rg][i] = -INFINITY;
    for (int t = 0; t < ntile; t++) {
#pragma unroll
        for (int cg = 0; cg < 4; cg++) {
            const int c = t * AT_TILE + 64 * w + 16 * cg + lc;
            if (c - lc > r_last) continue;                 // wave-uniform: nothing of this column group is visible
            float s[2][4];
            tile_scores(qf, kbase + (size_t)min(c, n - 1) * kv_pitch, lq, s);
#pragma unroll
            for (int rg = 0; rg < 2; rg++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (c <= row_base + 16 * rg + i) mx[rg][i] = fmaxf(mx[rg][i], s[rg][i]);   // causal mask, gten/ops.h:957
        }
    }
#pragma unroll
    for (int rg = 0; rg < 2; rg++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            mx[rg][i] = row16_max(mx[rg][i]);
            if (lc == 0) s_red[w * AT_ROWS + 16 * rg + 4 * lq + i] = mx[rg][i];
        }
    __syncthreads();
    if (threadIdx.x < AT_ROWS)
        s_row[threadIdx.x] = fmaxf(fmaxf(s_red[threadIdx.x], s_red[AT_ROWS + threadIdx.x]),
                                   fmaxf(s_red[2 * AT_ROWS + threadIdx.x], s_red[3 * AT_ROWS + threadIdx.x]));
    __syncthreads();
#pragma unroll
    for (int rg = 0; rg < 2; rg++)
#pragma unroll
        for (int i = 0; i < 4; i++) mx[rg][i] = s_row[16 * rg + 4 * lq + i];
    __syncthreads();

    // ---- pass 1: sums of exponentials, one running sum per (row, position mod 256)
    {
        float ls[2][4][4];
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int cg = 0; cg < 4; cg++)
#pragma unroll
                for (int i = 0; i < 4; i++) ls[rg][cg][i] = 0.f;
        for (int t = 0; t < ntile; t++) {
#pragma unroll
            for (int cg = 0; cg < 4; cg++) {
                const int c = t * AT_TILE + 64 * w + 16 * cg + lc;
                if (c - lc > r_last) continue;
                float s[2][4];
                tile_scores(qf, kbase + (size_t)min(c, n - 1) * kv_pitch, lq, s);
#pragma unroll
                for (int rg = 0; rg < 2; rg++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        ls[rg][cg][i] += (c <= row_base + 16 * rg + i) ? (FAST ? __expf(s[rg][i] - mx[rg][i]) : expf(s[rg][i] - mx[rg][i])) : 0.f;
            }
        }
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float r0s = row16_sum(ls[rg][0][i]), r1s = row16_sum(ls[rg][1][i]);
                const float r2s = row16_sum(ls[rg][2][i]), r3s = row16_sum(ls[rg][3][i]);
                const float ws = (r3s + r2s) + (r1s + r0s);          // wave_sum()'s last two steps
                if (lc == 0) s_red[w * AT_ROWS + 16 * rg + 4 * lq + i] = ws;
            }
        __syncthreads();
        if (threadIdx.x < AT_ROWS) {
            float t = 0.f;
            for (int i = 0; i < 4; i++) t += s_red[i * AT_ROWS + threadIdx.x];   // block_sum(): waves in order
            s_row[threadIdx.x] = t;
        }
        __syncthreads();
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) tot[rg][i] = s_row[16 * rg + 4 * lq + i];
    }
    }
    // (fast form above: the hardware exponential and one reciprocal per row instead of a division per probability -- a few ulp on
    //  values that are about to be rounded to 8 bits.  Measured and not kept: the scores as f16 matrix products with the deltas
    //  folded into f16 copies of q and K -- no per-score scaling arithmetic, 13.9 -> 13.2 ms for a 2048-id prompt -- moves the
    //  outputs beyond the operator's band: a score error of 2^-11 relative is an error of the same size in EVERY probability of
    //  the row, tests/test_ops_gpu.py.)
    // ---- pass 2: probabilities in the activation dtype, times V
    const int ep = threadIdx.x & 31, rq = threadIdx.x >> 5;      // outputs: rows 4 rq .. 4 rq + 3, elements 2 ep, 2 ep + 1
    v2f acc[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[rr][j] = (v2f){0.f, 0.f};
    v4f macc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // fast: [row tile], rows 16 rt + 4 lq + i, element 16 w + lc
    float lsum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // fast: this lane's share of the rows' sums of exponentials

    // fast form: a V sub-tile's bytes are requested one sub-tile ahead (the first of a tile before its scores), so that the
    // staging below never waits for memory
    U4 vraw = {{0, 0, 0, 0}};
    float vdel = 0.f;
    auto vload = [&](int c0) {
        const int pos = threadIdx.x >> 2, qtr = threadIdx.x & 3, b = qtr >> 1;
        const uint8_t* vs_ = vbase + (size_t)min(c0 + pos, n - 1) * kv_pitch;
        vraw = *(const U4*)(vs_ + 2 + 34 * b + 16 * (qtr & 1));
        vdel = h2f(*(const uint16_t*)(vs_ + 34 * b));
    };
    for (int t = 0; t < ntile; t++) {
        float p[2][4][4];
        if (FAST) vload(t * AT_TILE);
#pragma unroll
        for (int cg = 0; cg < 4; cg++) {
            const int c = t * AT_TILE + 64 * w + 16 * cg + lc;
            if (c - lc > r_last) {
#pragma unroll
                for (int rg = 0; rg < 2; rg++)
#pragma unroll
                    for (int i = 0; i < 4; i++) p[rg][cg][i] = FAST ? -INFINITY : 0.f;
                continue;
            }
            float s[2][4];
            tile_scores(qf, kbase + (size_t)min(c, n - 1) * kv_pitch, lq, s);
#pragma unroll
            for (int rg = 0; rg < 2; rg++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bool vis = c <= row_base + 16 * rg + i;
                    if (FAST) p[rg][cg][i] = vis ? s[rg][i] : -INFINITY;            // (the score; the probability below)
                    else p[rg][cg][i] = vis ? expf(s[rg][i] - mx[rg][i]) / tot[rg][i] : 0.f;
                }
        }
        if (FAST) {
            // the tile's row maxima (16 lanes, then the four waves through LDS), the running maximum, the rescale
#pragma unroll
            for (int rg = 0; rg < 2; rg++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float mt = row16_max(fmaxf(fmaxf(p[rg][0][i], p[rg][1][i]), fmaxf(p[rg][2][i], p[rg][3][i])));
                    if (lc == 0) s_red[w * AT_ROWS + 16 * rg + 4 * lq + i] = mt;
                }
            __syncthreads();
            if (threadIdx.x < AT_ROWS)
                s_row[threadIdx.x] = fmaxf(fmaxf(s_red[threadIdx.x], s_red[AT_ROWS + threadIdx.x]),
                                           fmaxf(s_red[2 * AT_ROWS + threadIdx.x], s_red[3 * AT_ROWS + threadIdx.x]));
            __syncthreads();
#pragma unroll
            for (int rg = 0; rg < 2; rg++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float mn = fmaxf(mx[rg][i], s_row[16 * rg + 4 * lq + i]);
                    const float sc = __expf(mx[rg][i] - mn);                        // (0 on the first tile: exp(-inf))
                    macc[rg][i] *= sc;
                    lsum[rg][i] *= sc;
                    mx[rg][i] = mn;
#pragma unroll
                    for (int cg = 0; cg < 4; cg++) {
                        const float e = __expf(p[rg][cg][i] - mn);                  // (masked: exp(-inf) = 0)
                        p[rg][cg][i] = e;
                        lsum[rg][i] += e;
                    }
                }
        }
        // the probability row is stored as Q8 blocks along the context (gten/ops.h:996-997): a block = two
        // column groups x 16 lanes; masked entries are zeros, exactly what the partial tail block sees
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int bp = 0; bp < 2; bp++) {
                    const float amax = row16_absmax(fmaxf(fabsf(p[rg][2 * bp][i]), fabsf(p[rg][2 * bp + 1][i])));
                    const Q8Scale sc = q8_scale_from_absmax(amax);
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const float x = p[rg][2 * bp + u][i];
                        const float pq = (float)q8_round(x, sc.scale) * sc.ddeq;
                        if (FAST) s_ph[(16 * rg + 4 * lq + i) * AT_PHPITCH + 64 * w + 16 * (2 * bp + u) + lc] = f2hv(pq);
                        else s_p[(16 * rg + 4 * lq + i) * AT_PPITCH + 64 * w + 16 * (2 * bp + u) + lc] = pq;
                    }
                }
        __syncthreads();

        for (int vs = 0; vs < AT_TILE / AT_VSUB; vs++) {
            const int c0 = t * AT_TILE + vs * AT_VSUB;
            if (c0 > r_last) break;
            {
                // stage 64 positions of this kv head's V slice as f32: thread = (position, quarter of the 64 elements)
                const int pos = threadIdx.x >> 2, qtr = threadIdx.x & 3, b = qtr >> 1;
                const uint8_t* vs_ = vbase + (size_t)min(c0 + pos, n - 1) * kv_pitch;
                U4 raw = vraw;
                float d = vdel;
                if (!FAST) {
                    raw = *(const U4*)(vs_ + 2 + 34 * b + 16 * (qtr & 1));
                    d = h2f(*(const uint16_t*)(vs_ + 34 * b));
                }
                if (FAST) {
                    // the sub-tile transposed, as f16 values: element 16 qtr + 4 kk + j of this position
#pragma unroll
                    for (int kk = 0; kk < 4; kk++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            // (column ^ 16 x (row / 16): the four quarters of a position -- rows 16 apart, 576 dwords, the SAME bank
                            //  with any 16-byte aligned pitch -- land 8 dwords apart instead of on top of each other)
                            s_vt[(16 * qtr + 4 * kk + j) * AT_VTPITCH + (pos ^ (16 * qtr))] = f2hv((float)(int8_t)(raw.v[kk] >> (8 * j)) * d);
                    if (vs + 1 < AT_TILE / AT_VSUB && c0 + AT_VSUB <= r_last) vload(c0 + AT_VSUB);
                } else {
                    float* dst = s_v + pos * 64 + 16 * qtr;
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        v4f o4;
#pragma unroll
                        for (int j = 0; j < 4; j++) o4[j] = (float)(int8_t)(raw.v[kk] >> (8 * j)) * d;
                        *(v4f*)(dst + 4 * kk) = o4;
                    }
                }
            }
            __syncthreads();
            if (FAST) {
                // wave w: elements 16 w .. 16 w + 15 of both 16-row tiles; 32 positions per matrix instruction
                // (positions past r_last meet p = 0; their V rows are clamped reads of real rows)
#pragma unroll
                for (int ks = 0; ks < AT_VSUB / 32; ks++) {
                    const h8 bv = *(const h8*)(s_vt + (16 * w + lc) * AT_VTPITCH + ((32 * ks + 8 * lq) ^ (16 * w)));
#pragma unroll
                    for (int rt2 = 0; rt2 < 2; rt2++) {
                        const h8 av = *(const h8*)(s_ph + (16 * rt2 + lc) * AT_PHPITCH + vs * AT_VSUB + 32 * ks + 8 * lq);
                        macc[rt2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, macc[rt2], 0, 0, 0);
                    }
                }
                __syncthreads();
                continue;
            }
            const int lim = min(AT_VSUB, r_last - c0 + 1);       // positions past r_last carry p = 0 for every row
#pragma unroll 4
            for (int c4 = 0; c4 < lim; c4 += 4) {
                v2f vv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) vv[j] = *(const v2f*)(s_v + (c4 + j) * 64 + 2 * ep);
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const v4f pp = *(const v4f*)(s_p + (4 * rq + rr) * AT_PPITCH + vs * AT_VSUB + c4);
                    // plain f32 mul + add: v_pk_mul_f32 / v_pk_add_f32 issue at half rate on gfx950 (measured: no gain)
#pragma unroll
                    for (int j = 0; j < 4; j++) { acc[rr][j].x = acc[rr][j].x + pp[j] * vv[j].x; acc[rr][j].y = acc[rr][j].y + pp[j] * vv[j].y; }
                }
            }
            __syncthreads();
        }
    }

    // ---- output rows in the activation dtype (store_row, gten/ops.h:73-96): a Q8 block = 16 lanes x 2 elements
    if (FAST) {
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float ws = row16_sum(lsum[rg][i]);
                if (lc == 0) s_red[w * AT_ROWS + 16 * rg + 4 * lq + i] = ws;
            }
#pragma unroll
        for (int rt2 = 0; rt2 < 2; rt2++)
#pragma unroll
            for (int i = 0; i < 4; i++) s_o[(16 * rt2 + 4 * lq + i) * 64 + 16 * w + lc] = macc[rt2][i];
        __syncthreads();
        if (threadIdx.x < AT_ROWS) {
            float t = 0.f;
            for (int i = 0; i < 4; i++) t += s_red[i * AT_ROWS + threadIdx.x];
            s_row[threadIdx.x] = recip_rn(t);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        v2f o = (v2f){0.f, 0.f};
        if (FAST) {
            o = *(const v2f*)(s_o + (4 * rq + rr) * 64 + 2 * ep);
            const float rl = s_row[4 * rq + rr];
            o.x *= rl; o.y *= rl;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) o += acc[rr][j];
        }
        const float amax = row16_absmax(fmaxf(fabsf(o.x), fabsf(o.y)));
        const Q8Scale sc = q8_scale_from_absmax(amax);
        const int row = r0 + 4 * rq + rr;
        if (row < n) {
            uint8_t* blk = out + (size_t)row * out_pitch + (size_t)h * 68 + (size_t)(ep >> 4) * GTEN_Q8_BYTES;
            const unsigned b0 = (unsigned)(uint8_t)(int8_t)q8_round(o.x, sc.scale), b1 = (unsigned)(uint8_t)(int8_t)q8_round(o.y, sc.scale);
            *(uint16_t*)(blk + 2 + 2 * (ep & 15)) = (uint16_t)(b0 | (b1 << 8));
            if ((ep & 15) == 0) *(uint16_t*)blk = sc.d16;
            if (a16) {
                // the f16 copy the o projection reads (gten_mfma.hip's fragment order: elements 0,2,1,3 of every four)
                _Float16* ar = a16 + (size_t)(row - start_pos) * (n_heads * 64) + h * 64 + ((2 * ep) & ~3) + (ep & 1);
                ar[0] = f2hv((float)(int)(int8_t)b0 * sc.ddeq);
                ar[2] = f2hv((float)(int)(int8_t)b1 * sc.ddeq);
            }
        }
    }
}

// ---- the same walk for f16 activations (the f16 configuration: tinyllama.cpp:258-265).  A score is the f16 x f16
// products of a row pair, exact in f32, added inside the matrix core: two v_mfma_f32_16x16x32_f16 per 16 x 16 scores.
// The core's order of additions is not the scalar loop's, so unlike the Q8 kernel this one agrees with k_attn to f32
// summation-order noise, not byte for byte (tests: the oracle tolerance of test_qkv_attn, the f16 goldens).  Probabilities
// and outputs are rounded to f16 (gten/ops.h:996-997, 73-96); p.V keeps k_attn's four accumulators.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct QFragH {
    h8 a[2][2];            // [row group][k step of 32]: row = lane % 16, elements 32 s + 8 (lane / 16) ..
};

__device__ __forceinline__ void tile_scores_f16(const QFragH& q, const uint8_t* __restrict__ kslice, int lq, float (&s)[2][4])
{
    const h8 b0 = *(const h8*)(kslice + 16 * lq), b1 = *(const h8*)(kslice + 64 + 16 * lq);
#pragma unroll
    for (int rg = 0; rg < 2; rg++) {
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(q.a[rg][0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(q.a[rg][1], b1, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) s[rg][i] = acc[i] * 0.125f;     // 1 / sqrt(64)
    }
}

// FAST (the default): p.V on the matrix cores as in k_attn_tiled_q8<true> -- the f16 probabilities and the f16 V elements
// ARE the operands (no extra rounding at all: only the order of the f32 additions differs from the four stride-4
// accumulators), the V sub-tile staged transposed, one v_mfma_f32_16x16x32_f16 per 32 positions of 16 rows x 16 elements.
template <bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_attn_tiled_f16(const uint8_t* __restrict__ q, const uint8_t* __restrict__ k,
                                                        const uint8_t* __restrict__ v, uint8_t* __restrict__ out,
                                                        size_t q_pitch, size_t kv_pitch, size_t out_pitch,
                                                        int n_heads, int n_kv, int n, int start_pos)
{
    __shared__ __attribute__((aligned(16))) float s_p[AT_ROWS * AT_PPITCH];
    __shared__ __attribute__((aligned(16))) float s_v[AT_VSUB * 64];
    _Float16* s_ph = (_Float16*)s_p;                                     // fast: f16 probability rows [32][AT_PHPITCH] ...
    float* s_o = s_p + AT_ROWS * AT_PHPITCH / 2;                         // ... and behind them the output tile [32][64]
    _Float16* s_vt = (_Float16*)s_v;                                     // fast: the V sub-tile transposed [element][position]
    static_assert(AT_ROWS * AT_PHPITCH / 2 + AT_ROWS * 64 <= AT_ROWS * AT_PPITCH && 64 * AT_VTPITCH * 2 <= AT_VSUB * 64 * 4, "fast-form tiles fit");
    __shared__ float s_red[4 * AT_ROWS];
    __shared__ float s_row[AT_ROWS];

    const int h = blockIdx.x;
    const int rt = gridDim.y - 1 - blockIdx.y;
    const int r0 = start_pos + rt * AT_ROWS;
    const int g = h / (n_heads / n_kv);
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6, lc = l & 15, lq = l >> 4;
    const int r_last = min(r0 + AT_ROWS, n) - 1;
    const int ntile = r_last / AT_TILE + 1;
    const uint8_t* kbase = k + (size_t)g * 128;
    const uint8_t* vbase = v + (size_t)g * 128;

    QFragH qf;
#pragma unroll
    for (int rg = 0; rg < 2; rg++) {
        const uint8_t* qs = q + (size_t)min(r0 + 16 * rg + lc, n - 1) * q_pitch + (size_t)h * 128;
        qf.a[rg][0] = *(const h8*)(qs + 16 * lq);
        qf.a[rg][1] = *(const h8*)(qs + 64 + 16 * lq);
    }
    const int row_base = r0 + 4 * lq;

    float mx[2][4], tot[2][4];
    if (FAST) {
        // ---- fast form: maxima and sums of exponentials in one walk (running pairs per lane, joined at the end), the
        //      hardware exponential, one reciprocal per row -- as in k_attn_tiled_q8<true>
        float rm[2][4], rl[2][4];
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) { rm[rg][i] = -INFINITY; rl[rg][i] = 0.f; }
        for (int t = 0; t < ntile; t++) {
#pragma unroll
            for (int cg = 0; cg < 4; cg++) {
                const int c = t * AT_TILE + 64 * w + 16 * cg + lc;
                if (c - lc > r_last) continue;
                float s[2][4];
                tile_scores_f16(qf, kbase + (size_t)min(c, n - 1) * kv_pitch, lq, s);
#pragma unroll
                for (int rg = 0; rg < 2; rg++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (c <= row_base + 16 * rg + i) {
                            const float mn = fmaxf(rm[rg][i], s[rg][i]);
                            rl[rg][i] = rl[rg][i] * __expf(rm[rg][i] - mn) + __expf(s[rg][i] - mn);
                            rm[rg][i] = mn;
                        }
            }
        }
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float m16 = row16_max(rm[rg][i]);
                if (lc == 0) s_red[w * AT_ROWS + 16 * rg + 4 * lq + i] = m16;
            }
        __syncthreads();
        if (threadIdx.x < AT_ROWS)
            s_row[threadIdx.x] = fmaxf(fmaxf(s_red[threadIdx.x], s_red[AT_ROWS + threadIdx.x]),
                                       fmaxf(s_red[2 * AT_ROWS + threadIdx.x], s_red[3 * AT_ROWS + threadIdx.x]));
        __syncthreads();
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) mx[rg][i] = s_row[16 * rg + 4 * lq + i];
        __syncthreads();
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float ws = row16_sum(rl[rg][i] * __expf(rm[rg][i] - mx[rg][i]));
                if (lc == 0) s_red[w * AT_ROWS + 16 * rg + 4 * lq + i] = ws;
            }
        __syncthreads();
        if (threadIdx.x < AT_ROWS) {
            float t = 0.f;
            for (int i = 0; i < 4; i++) t += s_red[i * AT_ROWS + threadIdx.x];
            s_row[threadIdx.x] = t;
        }
        __syncthreads();
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) tot[rg][i] = recip_rn(s_row[16 * rg + 4 * lq + i]);
    } else {
    // ---- pass 0: row maxima
#pragma unroll
    for (int rg = 0; rg < 2; rg++)
#pragma unroll
        for (int i = 0; i < 4; i++) mx[rg][i] = -INFINITY;
    for (int t = 0; t < ntile; t++) {
#pragma unroll
        for (int cg = 0; cg < 4; cg++) {
            const int c = t * AT_TILE + 64 * w + 16 * cg + lc;
            if (c - lc > r_last) continue;
            float s[2][4];
            tile_scores_f16(qf, kbase + (size_t)min(c, n - 1) * kv_pitch, lq, s);
#pragma unroll
            for (int rg = 0; rg < 2; rg++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (c <= row_base + 16 * rg + i) mx[rg][i] = fmaxf(mx[rg][i], s[rg][i]);
        }
    }
#pragma unroll
    for (int rg = 0; rg < 2; rg++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            mx[rg][i] = row16_max(mx[rg][i]);
            if (lc == 0) s_red[w * AT_ROWS + 16 * rg + 4 * lq + i] = mx[rg][i];
        }
    __syncthreads();
    if (threadIdx.x < AT_ROWS)
        s_row[threadIdx.x] = fmaxf(fmaxf(s_red[threadIdx.x], s_red[AT_ROWS + threadIdx.x]),
                                   fmaxf(s_red[2 * AT_ROWS + threadIdx.x], s_red[3 * AT_ROWS + threadIdx.x]));
    __syncthreads();
#pragma unroll
    for (int rg = 0; rg < 2; rg++)
#pragma unroll
        for (int i = 0; i < 4; i++) mx[rg][i] = s_row[16 * rg + 4 * lq + i];
    __syncthreads();

    // ---- pass 1: sums of exponentials (k_attn's 256 running sums per row and its tree, as in the Q8 kernel)
    {
        float ls[2][4][4];
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int cg = 0; cg < 4; cg++)
#pragma unroll
                for (int i = 0; i < 4; i++) ls[rg][cg][i] = 0.f;
        for (int t = 0; t < ntile; t++) {
#pragma unroll
            for (int cg = 0; cg < 4; cg++) {
                const int c = t * AT_TILE + 64 * w + 16 * cg + lc;
                if (c - lc > r_last) continue;
                float s[2][4];
                tile_scores_f16(qf, kbase + (size_t)min(c, n - 1) * kv_pitch, lq, s);
#pragma unroll
                for (int rg = 0; rg < 2; rg++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        ls[rg][cg][i] += (c <= row_base + 16 * rg + i) ? expf(s[rg][i] - mx[rg][i]) : 0.f;
            }
        }
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float r0s = row16_sum(ls[rg][0][i]), r1s = row16_sum(ls[rg][1][i]);
                const float r2s = row16_sum(ls[rg][2][i]), r3s = row16_sum(ls[rg][3][i]);
                const float ws = (r3s + r2s) + (r1s + r0s);
                if (lc == 0) s_red[w * AT_ROWS + 16 * rg + 4 * lq + i] = ws;
            }
        __syncthreads();
        if (threadIdx.x < AT_ROWS) {
            float t = 0.f;
            for (int i = 0; i < 4; i++) t += s_red[i * AT_ROWS + threadIdx.x];
            s_row[threadIdx.x] = t;
        }
        __syncthreads();
#pragma unroll
        for (int rg = 0; rg < 2; rg++)
#pragma unroll
            for (int i = 0; i < 4; i++) tot[rg][i] = s_row[16 * rg + 4 * lq + i];
    }

    }
    // ---- pass 2: probabilities rounded to f16, times V
    const int ep = threadIdx.x & 31, rq = threadIdx.x >> 5;
    v2f acc[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[rr][j] = (v2f){0.f, 0.f};
    v4f macc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // fast: [row tile], rows 16 rt + 4 lq + i, element 16 w + lc

    for (int t = 0; t < ntile; t++) {
#pragma unroll
        for (int cg = 0; cg < 4; cg++) {
            const int c = t * AT_TILE + 64 * w + 16 * cg + lc;
            float s[2][4];
            const bool any = c - lc <= r_last;
            if (any) tile_scores_f16(qf, kbase + (size_t)min(c, n - 1) * kv_pitch, lq, s);
#pragma unroll
            for (int rg = 0; rg < 2; rg++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float pr = (any && c <= row_base + 16 * rg + i) ? (FAST ? __expf(s[rg][i] - mx[rg][i]) * tot[rg][i] : expf(s[rg][i] - mx[rg][i]) / tot[rg][i]) : 0.f;
                    if (FAST) ((uint16_t*)s_ph)[(16 * rg + 4 * lq + i) * AT_PHPITCH + 64 * w + 16 * cg + lc] = f2h(pr);
                    else s_p[(16 * rg + 4 * lq + i) * AT_PPITCH + 64 * w + 16 * cg + lc] = h2f(f2h(pr));
                }
        }
        __syncthreads();

        for (int vs = 0; vs < AT_TILE / AT_VSUB; vs++) {
            const int c0 = t * AT_TILE + vs * AT_VSUB;
            if (c0 > r_last) break;
            {
                // stage 64 positions of this kv head's V slice as f32: thread = (position, quarter of the 64 elements)
                const int pos = threadIdx.x >> 2, qtr = threadIdx.x & 3;
                const uint8_t* vs_ = vbase + (size_t)min(c0 + pos, n - 1) * kv_pitch + 32 * qtr;
                const uint4 r0w = *(const uint4*)vs_, r1w = *(const uint4*)(vs_ + 16);
                const unsigned raw[8] = {r0w.x, r0w.y, r0w.z, r0w.w, r1w.x, r1w.y, r1w.z, r1w.w};
                float* dst = s_v + pos * 64 + 16 * qtr;
                if (FAST) {
#pragma unroll
                    for (int e = 0; e < 16; e++) ((uint16_t*)s_vt)[(16 * qtr + e) * AT_VTPITCH + (pos ^ (16 * qtr))] = (uint16_t)(raw[e >> 1] >> (16 * (e & 1)));   // (swizzle: see k_attn_tiled_q8)
                } else
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    v4f o4;
                    o4[0] = h2f((uint16_t)(raw[2 * kk] & 0xffffu)); o4[1] = h2f((uint16_t)(raw[2 * kk] >> 16));
                    o4[2] = h2f((uint16_t)(raw[2 * kk + 1] & 0xffffu)); o4[3] = h2f((uint16_t)(raw[2 * kk + 1] >> 16));
                    *(v4f*)(dst + 4 * kk) = o4;
                }
            }
            __syncthreads();
            if (FAST) {
                // wave w: elements 16 w .. 16 w + 15 of both 16-row tiles (positions past r_last meet p = 0)
#pragma unroll
                for (int ks = 0; ks < AT_VSUB / 32; ks++) {
                    const h8 bv = *(const h8*)(s_vt + (16 * w + lc) * AT_VTPITCH + ((32 * ks + 8 * lq) ^ (16 * w)));
#pragma unroll
                    for (int rt2 = 0; rt2 < 2; rt2++) {
                        const h8 av = *(const h8*)(s_ph + (16 * rt2 + lc) * AT_PHPITCH + vs * AT_VSUB + 32 * ks + 8 * lq);
                        macc[rt2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, macc[rt2], 0, 0, 0);
                    }
                }
                __syncthreads();
                continue;
            }
            const int lim = min(AT_VSUB, r_last - c0 + 1);
#pragma unroll 4
            for (int c4 = 0; c4 < lim; c4 += 4) {
                v2f vv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) vv[j] = *(const v2f*)(s_v + (c4 + j) * 64 + 2 * ep);
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const v4f pp = *(const v4f*)(s_p + (4 * rq + rr) * AT_PPITCH + vs * AT_VSUB + c4);
#pragma unroll
                    for (int j = 0; j < 4; j++) { acc[rr][j].x = acc[rr][j].x + pp[j] * vv[j].x; acc[rr][j].y = acc[rr][j].y + pp[j] * vv[j].y; }
                }
            }
            __syncthreads();
        }
    }

    // ---- output rows as f16
    if (FAST) {
#pragma unroll
        for (int rt2 = 0; rt2 < 2; rt2++)
#pragma unroll
            for (int i = 0; i < 4; i++) s_o[(16 * rt2 + 4 * lq + i) * 64 + 16 * w + lc] = macc[rt2][i];
        __syncthreads();
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        v2f o = (v2f){0.f, 0.f};
        if (FAST) o = *(const v2f*)(s_o + (4 * rq + rr) * 64 + 2 * ep);
        else
#pragma unroll
        for (int j = 0; j < 4; j++) o += acc[rr][j];
        const int row = r0 + 4 * rq + rr;
        if (row < n)
            *(unsigned*)(out + (size_t)row * out_pitch + (size_t)h * 128 + 4 * ep) = (unsigned)f2h(o.x) | ((unsigned)f2h(o.y) << 16);
    }
}

} // namespace

int gten_launch_attn_tiled_f16(const void* q, const void* k, const void* v, void* out, size_t q_pitch, size_t kv_pitch,
                               size_t out_pitch, int n, int n_heads, int n_kv_heads, int start_pos)
{
    const int rows = n - start_pos;
    const dim3 grid(n_heads, (rows + AT_ROWS - 1) / AT_ROWS);
    if (gtr::prefill_exact())
        GTR_LAUNCH(KT_ATTN_TILED, k_attn_tiled_f16<false>, grid, dim3(256), 0, (const uint8_t*)q, (const uint8_t*)k, (const uint8_t*)v,
                   (uint8_t*)out, q_pitch, kv_pitch, out_pitch, n_heads, n_kv_heads, n, start_pos);
    else
        GTR_LAUNCH(KT_ATTN_TILED, k_attn_tiled_f16<true>, grid, dim3(256), 0, (const uint8_t*)q, (const uint8_t*)k, (const uint8_t*)v,
                   (uint8_t*)out, q_pitch, kv_pitch, out_pitch, n_heads, n_kv_heads, n, start_pos);
    return 0;
}

int gten_launch_attn_tiled(const void* q, const void* k, const void* v, void* out, size_t q_pitch, size_t kv_pitch,
                           size_t out_pitch, int n, int n_heads, int n_kv_heads, int start_pos, void* a16)
{
    const int rows = n - start_pos;
    const dim3 grid(n_heads, (rows + AT_ROWS - 1) / AT_ROWS);
    if (gtr::prefill_exact())
        GTR_LAUNCH(KT_ATTN_TILED, k_attn_tiled_q8<false>, grid, dim3(256), 0, (const uint8_t*)q, (const uint8_t*)k, (const uint8_t*)v,
                   (uint8_t*)out, q_pitch, kv_pitch, out_pitch, n_heads, n_kv_heads, n, start_pos, (_Float16*)nullptr);
    else
        GTR_LAUNCH(KT_ATTN_TILED, k_attn_tiled_q8<true>, grid, dim3(256), 0, (const uint8_t*)q, (const uint8_t*)k, (const uint8_t*)v,
                   (uint8_t*)out, q_pitch, kv_pitch, out_pitch, n_heads, n_kv_heads, n, start_pos, (_Float16*)a16);
    return 0;
}
