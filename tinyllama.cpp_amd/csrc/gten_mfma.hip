// gten_mfma.hip -- multi-row W.x (prefill) on the matrix cores.
//
// ops::matmul_2d (gten/ops.h:613-670) for >= 16 new rows.  The reference's
// quantized contraction is, per 32-wide block, an EXACT integer dot product
// scaled by the two block deltas (gten/ops.h:224-479).  One
// v_mfma_f32_16x16x32_f16 spans exactly one such block along K: the int8 / int4
// quants are exact in f16, 32 products of magnitude <= 127*127 sum exactly in
// f32, so the MFMA result IS the reference's integer block sum for a 16x16 tile
// of (rows x output features); it is then scaled by da[row]*dw[col] and added to
// an f32 accumulator in block order -- the scalar build's order
// (gten/ops.h:296-312), to the bit.  f16 x f16 simply accumulates inside the MFMA.
//
// Tile: one workgroup = 2 x 2 waves, each wave a register tile of WM x WN MFMA
// tiles (64 x 64 outputs for <4,4>): an activation fragment read from LDS feeds WN
// MFMAs and a dequantized weight fragment feeds WM, so neither LDS bandwidth nor
// the dequantization bounds the loop -- the per-block rescale does (3 plain f32
// instructions per output, the price of the reference's exact block order).
// Memory side: every global access is a 16-byte piece of a contiguous run.  Q8
// activations are expanded ONCE per call to f16 rows + f32 deltas (k_act_to_f16,
// a few microseconds) instead of once per column of workgroups; both operands of
// a stage (4 quant blocks) are requested a stage ahead into registers and land in
// a double-buffered LDS stage (one barrier per stage).  The first version loaded
// weight fragments straight from global memory, 8 bytes per lane from 16 rows per
// instruction: the kernel was bound by L1 tag lookups at 1/5 of its VALU limit.
// The K order inside a fragment is permuted (0,2,1,3 per 4 quants) on BOTH
// operands, which turns the nibble/byte -> f16 expansion into mask-and-or instead
// of byte shuffles.  Results leave straight from the accumulators: a Q8 output
// block is two adjacent 16-wide tiles of the same wave (DPP row maximum).
//
// FAST form (the default for quantized weights; gten_hip_set_prefill_exact(1) selects the exact form above): the block
// deltas are folded into the f16 operands -- activations are expanded once per call to f16(q * da), the weight tile is
// expanded ONCE PER WORKGROUP to f16(q * dw) as its pieces come back from memory and lies in LDS in the activation rows'
// layout, so the K loop is fragment reads and matrix instructions only -- and the products accumulate ACROSS the K
// blocks inside the matrix core.  No per-block rescale (12 of the exact form's 21 VALU instructions per MFMA), no
// separate block-sum registers (so the 4 x 4 register tile fits).  What it costs: each operand element carries one
// fp16 rounding (relative 2^-11, a fifth of the Q8 quantization step's own noise) and the sum is no longer in the
// scalar build's order -- results agree with the exact form to ~1e-3 relative, inside the q8 / q4 logit band
// (tests/test_prefill_gpu.py holds both forms to the reference's full-size prompt goldens).
#include "gten_dev.h"
#include "gten_rt.h"

#include <algorithm>

using namespace gtd;

extern __shared__ __attribute__((aligned(16))) uint8_t g_smem[];

#ifndef GTEN_MFMA_NS
#define GTEN_MFMA_NS 2           // register sets of the fast form's K loop = stages requested ahead (measured: profiles/README.md, round 3)
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// two small integers held in the 16-bit halves of `bits | 0x6400` (= 1024 + v) -> exact f16 (v - bias)
__device__ __forceinline__ unsigned pk_int_to_f16(unsigned biased_pair, float bias)
{
    half2_t h;
    __builtin_memcpy(&h, &biased_pair, 4);
    const half2_t b = {(_Float16)bias, (_Float16)bias};
    h = h - b;
    unsigned out;
    __builtin_memcpy(&out, &h, 4);
    return out;
}

// four int8 in a dword -> four exact f16 in the fragment order (q0, q2 | q1, q3)
__device__ __forceinline__ void int8x4_to_f16(unsigned w, unsigned& lo, unsigned& hi)
{
    lo = pk_int_to_f16((w & 0x00ff00ffu) ^ 0x64806480u, 1024.0f + 128.0f);          // bytes 0, 2 via (q + 128)
    hi = pk_int_to_f16(((w >> 8) & 0x00ff00ffu) ^ 0x64806480u, 1024.0f + 128.0f);   // bytes 1, 3
}

// (a & mask) | bits in one instruction: the mask rides in a VGPR, the constant in an SGPR (gfx9 VOP3
// takes no literal and a single scalar operand)
__device__ __forceinline__ unsigned and_or(unsigned a, unsigned mask_vgpr, unsigned bits)
{
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(mask_vgpr), "s"(bits));
    return r;
}

// two f16 products in one instruction (v_pk_mul_f16)
__device__ __forceinline__ unsigned pk_mul_f16(unsigned a, unsigned b)
{
    half2_t x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    x = x * y;
    unsigned out;
    __builtin_memcpy(&out, &x, 4);
    return out;
}

// 8 consecutive quants of one weight row (two dwords) -> 8 f16 in the fragment order
template <int WT>
__device__ __forceinline__ half8 weight_frag(const uint2 bytes, int nibble_shift, unsigned nib_mask)
{
    unsigned r[4];
    if (WT == GTEN_Q4) {
        // (nibble - 7), gten/quants.h:78-90; elements 0-15 of a block are the high nibbles
        r[0] = pk_int_to_f16(and_or(bytes.x >> nibble_shift, nib_mask, 0x64006400u), 1024.0f + 7.0f);
        r[1] = pk_int_to_f16(and_or(bytes.x >> (nibble_shift + 8), nib_mask, 0x64006400u), 1024.0f + 7.0f);
        r[2] = pk_int_to_f16(and_or(bytes.y >> nibble_shift, nib_mask, 0x64006400u), 1024.0f + 7.0f);
        r[3] = pk_int_to_f16(and_or(bytes.y >> (nibble_shift + 8), nib_mask, 0x64006400u), 1024.0f + 7.0f);
    } else {
        int8x4_to_f16(bytes.x, r[0], r[1]);
        int8x4_to_f16(bytes.y, r[2], r[3]);
    }
    half8 out;
    __builtin_memcpy(&out, r, 16);
    return out;
}


template <int WT, int WM, int WN, int KB_ = 4>
struct MfmaCfg {
    static constexpr bool QUANT = (WT != GTEN_F16);
    static constexpr int BM = 32 * WM, BN = 32 * WN;       // workgroup tile: 2 x 2 waves
    static constexpr int KB = KB_;                         // quant blocks per stage (4 = 128 K; 2 halves the LDS stage: one more workgroup per CU)
    // bytes per staged activation row: KB blocks of 64 B, NO padding; the 16-byte piece c of row r lies at slot c ^ (r & SWZ).
    // A wave's ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
    // + 32 (MI355X_MICROARCH.md, LDS) -- not in 16 consecutive lanes: under that grouping round 2's padded pitch (KB * 64 + 16)
    // put two fragment rows on the same banks in every group (8 LDS cycles per read instead of 4; the fragment reads of two
    // workgroups per CU then took as long as their matrix instructions).  With the XOR the 16 lanes of every group hit 16
    // different 16-byte bank groups for every kb.
    static constexpr int APITCH = KB * 64;
    static constexpr int SWZ = KB * 4 - 1;                 // pieces per row - 1 (7 or 15)
    static constexpr int WROW = (WT == GTEN_Q4) ? KB * 16 : (WT == GTEN_Q8 ? KB * 32 : 0);   // weight bytes per feature per stage
    static constexpr int WPITCH = WROW + 8;
    static constexpr int DA_PIECES = (BM * KB + 255) / 256;
    // (the delta regions are padded to one slot per staging thread: every thread loads and stores, no branch)
    static constexpr int A_BYTES = BM * APITCH, DA_BYTES = QUANT ? DA_PIECES * 256 * 4 : 0;
    static constexpr int W_BYTES = QUANT ? BN * WPITCH : 0, DW_BYTES = QUANT ? KB * 256 * 4 : 0;
    static constexpr int STAGE = A_BYTES + DA_BYTES + W_BYTES + DW_BYTES;
    static constexpr int A_PIECES = BM * KB * 4 / 256;     // 16-byte pieces of the activation tile per thread
    static constexpr int W_PIECES = QUANT ? (BN * WROW / 16 + 255) / 256 : 0;
    static constexpr size_t smem() { return (size_t)2 * STAGE; }
    // fast form: the weight tile lies in LDS EXPANDED, f16(quant * delta), rows of KB blocks like the activation rows --
    // each element is expanded once per workgroup (as its piece comes back from memory), not once per wave and K block
    static constexpr int W16PITCH = KB * 64;               // (same layout and swizzle as the activation rows)
    static constexpr int STAGE_FAST = A_BYTES + BN * W16PITCH;
    static constexpr size_t smem_fast() { return (size_t)2 * STAGE_FAST; }
};

// Q8 activation rows [start_pos, n) -> f16 rows in the fragment order (q0,q2,q1,q3 per 4) + f32 block deltas.
// One thread per (row, quant block).
// FAST: the values q * delta rounded to f16 (the deltas folded in; `da` is not written).
template <bool FAST>
__global__ __launch_bounds__(256) void k_act_to_f16(const uint8_t* __restrict__ x, size_t x_pitch, int rows, int nb, int start_pos,
                                                    uint4* __restrict__ a16, float* __restrict__ da)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nb) return;
    const int r = idx / nb, b = idx - r * nb;
    const uint16_t* blk = (const uint16_t*)(x + (size_t)(start_pos + r) * x_pitch + (size_t)b * GTEN_Q8_BYTES);   // 2-byte aligned
    unsigned q[8];
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = (unsigned)blk[1 + 2 * i] | ((unsigned)blk[2 + 2 * i] << 16);
    uint4* dst = a16 + (size_t)idx * 4;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        unsigned f[4];
        int8x4_to_f16(q[i], f[0], f[1]);
        int8x4_to_f16(q[i + 1], f[2], f[3]);
        if (FAST) {
            const unsigned d2 = (unsigned)blk[0] | ((unsigned)blk[0] << 16);
#pragma unroll
            for (int k = 0; k < 4; k++) f[k] = pk_mul_f16(f[k], d2);
        }
        dst[i >> 1] = make_uint4(f[0], f[1], f[2], f[3]);
    }
    if (!FAST) da[idx] = h2f(blk[0]);
}

// The same, one thread per PAIR of adjacent blocks: a pair is 68 bytes, always 4-byte aligned, i.e. 17 dwords instead of
// 2 x 17 two-byte loads (rows of an even number of blocks at a 4-byte aligned pitch: every activation width of the model).
template <bool FAST>
__global__ __launch_bounds__(256) void k_act_to_f16_x2(const uint8_t* __restrict__ x, size_t x_pitch, int rows, int nb, int start_pos,
                                                       uint4* __restrict__ a16, float* __restrict__ da)
{
    const int np = nb >> 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * np) return;
    const int r = idx / np, pr = idx - r * np;
    const unsigned* pw = (const unsigned*)(x + (size_t)(start_pos + r) * x_pitch + (size_t)pr * 68);
    unsigned w[17];
#pragma unroll
    for (int i = 0; i < 17; i++) w[i] = pw[i];
    unsigned q[2][8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        q[0][i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], 16);      // block 0's quants straddle the dwords by 2 bytes
        q[1][i] = w[9 + i];
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const size_t bi = (size_t)r * nb + 2 * pr + b;
        uint4* dst = a16 + bi * 4;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            unsigned f[4];
            int8x4_to_f16(q[b][i], f[0], f[1]);
            int8x4_to_f16(q[b][i + 1], f[2], f[3]);
            if (FAST) {
                const unsigned dh = b ? (w[8] >> 16) : (w[0] & 0xffffu);
                const unsigned d2 = dh | (dh << 16);
#pragma unroll
                for (int k = 0; k < 4; k++) f[k] = pk_mul_f16(f[k], d2);
            }
            dst[i >> 1] = make_uint4(f[0], f[1], f[2], f[3]);
        }
        if (!FAST) da[bi] = h2f((uint16_t)(b ? (w[8] >> 16) : (w[0] & 0xffffu)));
    }
}

// Up to three weight matrices that share ONE input (q | k | v, gate | up) in one launch: column tiles [tile0[m], tile0[m+1])
// belong to matrix m.  resid / sum_out (one matrix, Q8 rows): the residual sum behind the projection, ops::add's arithmetic
// on the row just rounded -- sum_out = Q8(resid + Q8(W x)).
struct MatArgs {
    const void* w[3];
    uint8_t* out[3];
    size_t out_pitch[3];
    int d_out[3];
    int tile0[3];
    int n_mats;
    const uint8_t* resid;
    uint8_t* sum_out;
    size_t resid_pitch;
    // fast form, few workgroups: gridDim.z workgroups share a tile's K loop, each writes its f32 partial sums to plane
    // blockIdx.z of `partial` ([z][new row][sum of d_out]); k_splitk_finish adds the planes in order and rounds the rows
    float* partial;
    int part_pitch;
    int part_col0[3];
};

// A wave tile that lies wholly inside a Q8 output (every tile of a prompt GEMM but those on the last rows / features): the same
// arithmetic as mfma_epilogue below, per output two multiplies, one addition and the conversions, no bounds predicate and no
// address arithmetic in the vector unit: the row and block of a store are a scalar offset (SALU) on a buffer descriptor, the
// lane's share (its row group and byte) one VGPR for the whole tile.  The generic form spent 60 VALU instructions per output
// on predicates and 64-bit addresses, three fifths of all VALU work of a kernel that is bound by VALU issue.
template <int WM, int WN, bool RESID>
__device__ __forceinline__ void mfma_epilogue_q8_full(floatx4 (&acc)[WM][WN], const MatArgs& ms, uint8_t* __restrict__ out, const int out_pitch,
                                                      const int row_base, const int col0, const int g, const int l16, const int start_pos)
{
    const size_t boff = (size_t)(col0 >> 5) * GTEN_Q8_BYTES;
    const int span = (16 * WM - 1) * out_pitch + (WN / 2) * GTEN_Q8_BYTES;
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)(start_pos + row_base) * out_pitch + boff), 0, span, 0x00020000);
    // a block's delta leaves from the first lane of its row: the other lanes' offset lies beyond the descriptor's range, where a
    // buffer store is dropped -- no exec-mask branch between the blocks, so the scheduler may interleave them
    constexpr int DROP = 0x40000000;
    const int vd = 4 * g * out_pitch, vq = vd + 2 + l16, vdp = l16 == 0 ? vd : DROP;
    const int rpitch = (int)ms.resid_pitch;
    const int rspan = (16 * WM - 1) * rpitch + (WN / 2) * GTEN_Q8_BYTES;
    const size_t roff = (size_t)(start_pos + row_base) * ms.resid_pitch + boff;
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(ms.resid + (RESID ? roff : 0)), 0, RESID ? rspan : 0, 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(ms.sum_out + (RESID ? roff : 0)), 0, RESID ? rspan : 0, 0x00020000);
    const int rd = 4 * g * rpitch, rq = rd + 2 + l16, rdp = l16 == 0 ? rd : DROP;
#pragma unroll
    for (int t = 0; t < WM; t++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int j = 0; j < WN; j += 2) {
                const int so = (16 * t + i) * out_pitch + (j >> 1) * GTEN_Q8_BYTES;
                const float v0 = acc[t][j][i], v1 = acc[t][j + 1][i];
                const Q8Scale sc = q8_scale_from_absmax(row16_absmax(fmaxf(fabsf(v0), fabsf(v1))));
                const int q0 = q8_round(v0, sc.scale), q1 = q8_round(v1, sc.scale);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q0, rs_o, vq, so, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q1, rs_o, vq, so + 16, 0);
                __builtin_amdgcn_raw_buffer_store_b16(sc.d16, rs_o, vdp, so, 0);
                if (RESID) {
                    // ops::add on the block just stored (gten/ops.h:816-860): both operands from their stored bytes
                    const int ro = (16 * t + i) * rpitch + (j >> 1) * GTEN_Q8_BYTES;
                    const float dr = h2f(__builtin_amdgcn_raw_buffer_load_b16(rs_r, rd, ro, 0)), dq = sc.ddeq;
                    const float r0 = (float)(int)(int8_t)__builtin_amdgcn_raw_buffer_load_b8(rs_r, rq, ro, 0);
                    const float r1 = (float)(int)(int8_t)__builtin_amdgcn_raw_buffer_load_b8(rs_r, rq, ro + 16, 0);
                    const float s0 = r0 * dr + (float)q0 * dq, s1 = r1 * dr + (float)q1 * dq;
                    const Q8Scale ss = q8_scale_from_absmax(row16_absmax(fmaxf(fabsf(s0), fabsf(s1))));
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q8_round(s0, ss.scale), rs_s, rq, ro, 0);
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)q8_round(s1, ss.scale), rs_s, rq, ro + 16, 0);
                    __builtin_amdgcn_raw_buffer_store_b16(ss.d16, rs_s, rdp, ro, 0);
                }
            }
        }
    }
}

// Results leave straight from the accumulators (shared by the prompt GEMM kernels): either this workgroup's f32 sums into its
// plane of a shared K loop, or the rows written in the output dtype (gten/ops.h:73-96), a Q8 output block = two adjacent
// 16-wide tiles of the wave, with the residual sum behind a single Q8 / f16 projection.  row_base: first row of this wave
// (relative to start_pos), col0: its first feature.
template <int WM, int WN>
__device__ __forceinline__ void mfma_epilogue(floatx4 (&acc)[WM][WN], const MatArgs& ms, const int mi, uint8_t* __restrict__ out, const size_t out_pitch,
                                              const int d_out, const int out_dtype, const int rows, const int row_base, const int col0, const int g,
                                              const int l16, const int start_pos)
{
    if (ms.partial) {
        // K shared with other workgroups: this one's f32 sums into its plane
        float* plane = ms.partial + (size_t)blockIdx.z * rows * ms.part_pitch + ms.part_col0[mi];
#pragma unroll
        for (int t = 0; t < WM; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = row_base + 16 * t + 4 * g + i;
                if (r >= rows) continue;
#pragma unroll
                for (int j = 0; j < WN; j++) {
                    const int c = col0 + 16 * j + l16;
                    if (c < d_out) plane[(size_t)r * ms.part_pitch + c] = acc[t][j][i];
                }
            }
        return;
    }
    // ---- rows written in the output dtype straight from the accumulators (gten/ops.h:73-96)
    {
        const int rb = __builtin_amdgcn_readfirstlane(row_base), cb = __builtin_amdgcn_readfirstlane(col0);
        if (out_dtype == GTEN_Q8 && rb + 16 * WM <= rows && cb + 16 * WN <= d_out && out_pitch < (size_t)(1 << 24)
            && (!ms.resid || ms.resid_pitch < (size_t)(1 << 24))) {
            if (ms.resid) mfma_epilogue_q8_full<WM, WN, true>(acc, ms, out, (int)out_pitch, rb, cb, g, l16, start_pos);
            else mfma_epilogue_q8_full<WM, WN, false>(acc, ms, out, (int)out_pitch, rb, cb, g, l16, start_pos);
            return;
        }
    }
#pragma unroll
    for (int t = 0; t < WM; t++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = row_base + 16 * t + 4 * g + i;
            const bool rok = r < rows;
            uint8_t* orow = out + (size_t)(start_pos + (rok ? r : rows - 1)) * out_pitch;
#pragma unroll
            for (int j = 0; j < WN; j += 2) {
                const int c0 = col0 + 16 * j + l16, c1 = c0 + 16;
                const bool ok0 = rok && c0 < d_out, ok1 = rok && c1 < d_out;
                const float v0 = ok0 ? acc[t][j][i] : 0.f, v1 = ok1 ? acc[t][j + 1][i] : 0.f;
                if (out_dtype == GTEN_Q8) {
                    // one 32-wide output block = tiles j, j + 1 of this wave
                    const Q8Scale sc = q8_scale_from_absmax(row16_absmax(fmaxf(fabsf(v0), fabsf(v1))));
                    uint8_t* blk = orow + (size_t)((col0 + 16 * j) >> 5) * GTEN_Q8_BYTES;
                    const int q0 = q8_round(v0, sc.scale), q1 = q8_round(v1, sc.scale);
                    if (ok0) blk[2 + l16] = (uint8_t)(int8_t)q0;
                    if (ok1) blk[18 + l16] = (uint8_t)(int8_t)q1;
                    if (ok0 && l16 == 0) *(uint16_t*)blk = sc.d16;
                    if (ms.resid) {
                        // ops::add on the block just stored (gten/ops.h:816-860 via k_elementwise's arithmetic): both operands
                        // dequantized from their stored bytes, f32 sum, rounded to a Q8 block again
                        const size_t boff = (size_t)((col0 + 16 * j) >> 5) * GTEN_Q8_BYTES;
                        const uint8_t* rb = ms.resid + (size_t)(start_pos + (rok ? r : rows - 1)) * ms.resid_pitch + boff;
                        const float dr = h2f(*(const uint16_t*)rb), dq = sc.ddeq;
                        const float s0 = ok0 ? (float)(int)(int8_t)rb[2 + l16] * dr + (float)q0 * dq : 0.f;
                        const float s1 = ok1 ? (float)(int)(int8_t)rb[18 + l16] * dr + (float)q1 * dq : 0.f;
                        const Q8Scale ss = q8_scale_from_absmax(row16_absmax(fmaxf(fabsf(s0), fabsf(s1))));
                        uint8_t* sb = ms.sum_out + (size_t)(start_pos + (rok ? r : rows - 1)) * ms.resid_pitch + boff;
                        if (ok0) sb[2 + l16] = (uint8_t)(int8_t)q8_round(s0, ss.scale);
                        if (ok1) sb[18 + l16] = (uint8_t)(int8_t)q8_round(s1, ss.scale);
                        if (ok0 && l16 == 0) *(uint16_t*)sb = ss.d16;
                    }
                } else if (out_dtype == GTEN_F16) {
                    const uint16_t h0 = f2h(v0), h1 = f2h(v1);
                    if (ok0) ((uint16_t*)orow)[c0] = h0;
                    if (ok1) ((uint16_t*)orow)[c1] = h1;
                    if (ms.resid) {
                        // ops::add on the f16 values just stored: both operands from their stored halves, f32 sum, one rounding
                        const size_t ro = (size_t)(start_pos + (rok ? r : rows - 1)) * ms.resid_pitch;
                        const uint16_t* rr16 = (const uint16_t*)(ms.resid + ro);
                        uint16_t* so16 = (uint16_t*)(ms.sum_out + ro);
                        if (ok0) so16[c0] = f2h(h2f(rr16[c0]) + h2f(h0));
                        if (ok1) so16[c1] = f2h(h2f(rr16[c1]) + h2f(h1));
                    }
                } else {
                    if (ok0) ((float*)orow)[c0] = v0;
                    if (ok1) ((float*)orow)[c1] = v1;
                }
            }
        }
    }
}

// (2 waves per SIMD = a 256-VGPR budget: the block sums then come back in VGPRs instead of AGPRs, which
//  would cost four v_accvgpr_read per MFMA in a loop that is bound by VALU issue)
// a16: f16 activation rows of the NEW rows (row 0 = start_pos), pitch d_in * 2 (quantized) or x itself (f16 weights)
template <int WT, int WM, int WN, int KB_, bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_matmul_mfma(
    const uint8_t* __restrict__ a16, size_t a_pitch, const float* __restrict__ da_rows, const MatArgs ms,
    int out_dtype, int n, int d_in, int start_pos)
{
    using C = MfmaCfg<WT, WM, WN, KB_>;
    constexpr int BM = C::BM, BN = C::BN, KB = C::KB, APITCH = C::APITCH, WPITCH = C::WPITCH, WROW = C::WROW;
    constexpr bool QUANT = C::QUANT;
    constexpr bool EXACT = QUANT && !FAST;                   // per-block rescale in the scalar build's order
    static_assert(QUANT || !FAST, "the fast form is about quantized weights");
    // the weight tile lies in LDS as f16 rows in the activation rows' layout: fast form (expanded) and f16 weights (copied)
    constexpr bool LDSW = FAST || !QUANT;
    constexpr int STAGE = LDSW ? C::STAGE_FAST : C::STAGE, W16PITCH = C::W16PITCH;

    // which of the (up to three) matrices sharing this input the workgroup's column tile belongs to
    int mi = 0;
    if (ms.n_mats > 1 && (int)blockIdx.x >= ms.tile0[1]) mi = 1;
    if (ms.n_mats > 2 && (int)blockIdx.x >= ms.tile0[2]) mi = 2;
    const void* __restrict__ w = ms.w[mi];
    uint8_t* __restrict__ out = ms.out[mi];
    const size_t out_pitch = ms.out_pitch[mi];
    const int d_out = ms.d_out[mi];

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    const int g = lane >> 4, l16 = lane & 15;
    const int rows = n - start_pos;
    const int row0 = blockIdx.y * BM;                        // relative to start_pos
    const int colw = ((int)blockIdx.x - ms.tile0[mi]) * BN;  // first feature of the workgroup
    const int col0 = colw + wc * 16 * WN;                    // first feature of this wave
    const int nb = d_in >> 5;
    const int nstage_all = nb / KB;
    // this workgroup's share of the K loop: stages [s0, s0 + nstage)
    const int s0 = (int)blockIdx.z * nstage_all / (int)gridDim.z;
    const int nstage = ((int)blockIdx.z + 1) * nstage_all / (int)gridDim.z - s0;
    const PackedW pw = packed_view(w, WT, d_out, d_in);
    const int nibble_shift = (g < 2) ? 4 : 0;
    unsigned nib_mask = 0x000f000fu;
    asm volatile("" : "+v"(nib_mask));                       // keep it in a register (see and_or)

    // ---- staging roles: 16-byte pieces, consecutive threads = consecutive pieces of a row.  All loads are
    //      buffer loads: descriptor (SGPRs) + lane offset (one VGPR, fixed for the kernel) + stage offset (one SGPR);
    //      every thread issues every load (out-of-range pieces read 0 or a neighbour and are never used), so the
    //      loop has no branch around a memory instruction and the in-order vmcnt bookkeeping stays exact.
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a16, 0, (int)((size_t)rows * a_pitch), 0x00020000);
    const auto rs_da = __builtin_amdgcn_make_buffer_rsrc((void*)da_rows, 0, QUANT ? rows * nb * 4 : 0, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(QUANT ? (void*)pw.qs : (void*)w, 0,
                                                        QUANT ? (int)((size_t)d_out * nb * (WT == GTEN_Q4 ? 16 : 32)) : (int)((size_t)d_out * d_in * 2), 0x00020000);
    const auto rs_dw = __builtin_amdgcn_make_buffer_rsrc((void*)pw.ds, 0, QUANT ? d_out * nb * 2 : 0, 0x00020000);
    // activations: piece p -> (row p / (KB*4), piece-in-row p % (KB*4)); a stage of a row is KB * 64 contiguous bytes
    unsigned a_src[C::A_PIECES], a_dst[C::A_PIECES];
#pragma unroll
    for (int k = 0; k < C::A_PIECES; k++) {
        const int p = threadIdx.x + 256 * k, r = p / (KB * 4), c = p % (KB * 4);
        a_src[k] = (unsigned)min(row0 + r, rows - 1) * (unsigned)a_pitch + c * 16;
        a_dst[k] = r * APITCH + ((c ^ (r & C::SWZ)) * 16);
    }
    // activation deltas: one float per (row, block): BM * KB of them, DA_PIECES per thread (slots past them are padding)
    constexpr int DA_PIECES = C::DA_PIECES;
    unsigned da_src[DA_PIECES], da_dst[DA_PIECES];
#pragma unroll
    for (int k = 0; k < DA_PIECES; k++) {
        const int p = threadIdx.x + 256 * k, r = p / KB, b = p % KB;
        da_src[k] = ((unsigned)min(row0 + r, rows - 1) * nb + b) * 4;
        da_dst[k] = (p < BM * KB) ? b * BM + r : p;
    }
    // weights: piece p -> (feature p / (WROW/16), piece-in-row); Q8 rows are two planes of nb * 16 bytes
    // (f16 weights: a feature's stage is KB * 64 contiguous bytes = KB * 4 pieces, copied as they are)
    constexpr int NWP = QUANT ? C::W_PIECES : BN * KB * 4 / 256;
    constexpr int WP = NWP > 0 ? NWP : 1;
    static_assert(!QUANT || BN * (WROW / 16) == C::W_PIECES * 256, "weight tile must be a whole number of pieces per thread");
    static_assert(QUANT || BN * KB * 4 == NWP * 256, "weight tile must be a whole number of pieces per thread");
    unsigned w_src[WP], w_dst[WP];
#pragma unroll
    for (int k = 0; k < NWP; k++) {
        const int p = threadIdx.x + 256 * k, pr = QUANT ? WROW / 16 : KB * 4, f = p / pr, c = p % pr;
        const unsigned frow = (unsigned)min(colw + f, d_out - 1);
        if (WT == GTEN_Q4) w_src[k] = frow * nb * 16 + c * 16;
        else if (WT == GTEN_Q8) w_src[k] = frow * nb * 32 + (c / KB) * nb * 16 + (c % KB) * 16;     // plane c / KB, block c % KB
        else w_src[k] = frow * (unsigned)d_in * 2 + c * 16;
        w_dst[k] = QUANT ? f * WPITCH + c * 16 : f * W16PITCH + ((c ^ (f & C::SWZ)) * 16);
    }
    // weight deltas: thread -> feature t: KB halves per stage (threads past BN fill padding slots)
    const unsigned dw_src = (unsigned)min(colw + (int)threadIdx.x, d_out - 1) * nb * 2;
    // fast form: the delta of every PIECE's block (one dword holding it; which half: pd_hi), and where the piece's
    // expanded elements go
    unsigned pd_src[WP], pd_hi[WP], w16_dst[WP], w16_pi[WP], w16_fm[WP];
#pragma unroll
    for (int k = 0; k < C::W_PIECES; k++) {
        constexpr int pr = (WROW / 16) > 0 ? WROW / 16 : 1;      // (f16 weights have no packed rows: W_PIECES is 0 and this loop is empty)
        const int p = threadIdx.x + 256 * k, f = p / pr, c = p % pr;
        const unsigned frow = (unsigned)min(colw + f, d_out - 1);
        const int blk = (WT == GTEN_Q4) ? c : c % KB, pl = (WT == GTEN_Q4) ? 0 : c / KB;
        const unsigned di = frow * nb + blk;                 // + s * KB per stage: KB is even, the parity is the piece's own
        pd_src[k] = (di & ~1u) * 2;
        pd_hi[k] = di & 1u;
        w16_dst[k] = f * W16PITCH;                          // row; the expanded 16-byte pieces blk * 4 + pl * 2 + q go to slot piece ^ (f & SWZ)
        w16_pi[k] = blk * 4 + pl * 2;
        w16_fm[k] = f & C::SWZ;
    }

    struct Raw { v4i_t a[C::A_PIECES]; v4i_t w[WP]; float da[DA_PIECES]; v2i_t dw; int pd[FAST ? WP : 1]; };
    auto load_stage = [&](int s, Raw& r) {
#pragma unroll
        for (int k = 0; k < C::A_PIECES; k++) r.a[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_src[k], s * (KB * 64), 0);
        if (!QUANT) {
#pragma unroll
            for (int k = 0; k < NWP; k++) r.w[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_src[k], s * (KB * 64), 0);
        }
        if (QUANT) {
#pragma unroll
            for (int k = 0; k < C::W_PIECES; k++) r.w[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_src[k], s * (KB * 16), 0);
            if (EXACT) {
#pragma unroll
                for (int k = 0; k < DA_PIECES; k++) r.da[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_da, da_src[k], s * (KB * 4), 0));
            }
            if (FAST) {
#pragma unroll
                for (int k = 0; k < C::W_PIECES; k++) r.pd[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_dw, pd_src[k], s * (KB * 2), 0);
            } else if (KB == 4) r.dw = __builtin_amdgcn_raw_buffer_load_b64(rs_dw, dw_src, s * (KB * 2), 0);
            else r.dw[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_dw, dw_src, s * (KB * 2), 0);
        }
    };
    auto store_stage = [&](const Raw& r, int buf) {
        uint8_t* sa = g_smem + buf * STAGE;
#pragma unroll
        for (int k = 0; k < C::A_PIECES; k++) *(v4i_t*)(sa + a_dst[k]) = r.a[k];
        if (!QUANT) {
            uint8_t* sw16 = sa + C::A_BYTES;
#pragma unroll
            for (int k = 0; k < NWP; k++) *(v4i_t*)(sw16 + w_dst[k]) = r.w[k];
        } else if (FAST) {
            // a piece = 16 bytes of quants of ONE block: expanded to f16, times the block's delta (one fp16 rounding per
            // element), in the fragment order of the activation rows
            uint8_t* sw16 = sa + C::A_BYTES;
#pragma unroll
            for (int k = 0; k < C::W_PIECES; k++) {
                const unsigned dbits = pd_hi[k] ? ((unsigned)r.pd[k] >> 16) : ((unsigned)r.pd[k] & 0xffffu);
                const unsigned d2 = dbits | (dbits << 16);
                const uint2 lo = make_uint2((unsigned)r.w[k][0], (unsigned)r.w[k][1]), hi = make_uint2((unsigned)r.w[k][2], (unsigned)r.w[k][3]);
                half8 e[WT == GTEN_Q4 ? 4 : 2];
                if (WT == GTEN_Q4) {
                    e[0] = weight_frag<WT>(lo, 4, nib_mask); e[1] = weight_frag<WT>(hi, 4, nib_mask);      // elements 0..15: high nibbles
                    e[2] = weight_frag<WT>(lo, 0, nib_mask); e[3] = weight_frag<WT>(hi, 0, nib_mask);      // elements 16..31: low nibbles
                } else {
                    e[0] = weight_frag<WT>(lo, 0, nib_mask); e[1] = weight_frag<WT>(hi, 0, nib_mask);
                }
#pragma unroll
                for (int q = 0; q < (WT == GTEN_Q4 ? 4 : 2); q++) {
                    unsigned u[4];
                    __builtin_memcpy(u, &e[q], 16);
#pragma unroll
                    for (int i = 0; i < 4; i++) u[i] = pk_mul_f16(u[i], d2);
                    *(v4i_t*)(sw16 + w16_dst[k] + (((w16_pi[k] + q) ^ w16_fm[k]) * 16)) = (v4i_t){(int)u[0], (int)u[1], (int)u[2], (int)u[3]};
                }
            }
        } else if (QUANT) {
            float* sda = (float*)(sa + C::A_BYTES);
            uint8_t* sw = sa + C::A_BYTES + C::DA_BYTES;
            float* sdw = (float*)(sw + C::W_BYTES);
            if (EXACT) {
#pragma unroll
                for (int k = 0; k < DA_PIECES; k++) sda[da_dst[k]] = r.da[k];
            }
#pragma unroll
            for (int k = 0; k < C::W_PIECES; k++) {
                *(v2i_t*)(sw + w_dst[k]) = (v2i_t){r.w[k][0], r.w[k][1]};
                *(v2i_t*)(sw + w_dst[k] + 8) = (v2i_t){r.w[k][2], r.w[k][3]};
            }
            sdw[0 * 256 + threadIdx.x] = h2f((uint16_t)((unsigned)r.dw[0] & 0xffffu));
            sdw[1 * 256 + threadIdx.x] = h2f((uint16_t)((unsigned)r.dw[0] >> 16));
            if (KB == 4) {
                sdw[2 * 256 + threadIdx.x] = h2f((uint16_t)((unsigned)r.dw[1] & 0xffffu));
                sdw[3 * 256 + threadIdx.x] = h2f((uint16_t)((unsigned)r.dw[1] >> 16));
            }
        }
    };

    floatx4 acc[WM][WN];
#pragma unroll
    for (int t = 0; t < WM; t++)
#pragma unroll
        for (int j = 0; j < WN; j++) acc[t][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // Stage s is computed from LDS buffer s & 1 while stage s + 1 waits in registers (stored behind the compute) and
    // stage s + NS is requested: NS register sets, NS stages of memory latency covered.  Round 2 ran NS = 2 -- at 16 MFMAs
    // per wave and stage (64 x 64 tiles) a stage is ~0.3 us of work against 1-2 us from request to data, and every stage of
    // the K loop waited for memory (19 us per launch at 256 rows); the sets cost 21 VGPRs each (fast form).
    constexpr int NS = FAST ? GTEN_MFMA_NS : 2;
    Raw raw[NS];
#pragma unroll
    for (int u = 0; u < NS; u++) load_stage(s0 + u, raw[u]);
    store_stage(raw[0], 0);
    __syncthreads();

    auto stage_body = [&](int s, Raw& fetch, const Raw& land) {
        const int buf = s & 1;
        const uint8_t* sa = g_smem + buf * STAGE;
        const float* sda = (const float*)(sa + C::A_BYTES);
        const uint8_t* sw = sa + C::A_BYTES + (LDSW ? 0 : C::DA_BYTES);      // (fast form / f16 weights: the f16 tile, rows of W16PITCH bytes)
        const float* sdw = (const float*)(sw + C::W_BYTES);
        load_stage(s0 + s + NS, fetch);
        if constexpr (LDSW) {
            // Fast form (and f16 weights): nothing but fragment reads and matrix instructions in the K loop.  The fragments of block kb + 1 are
            // requested before the matrix instructions of block kb are issued, and the NEXT stage's expansion (store_stage:
            // vector instructions + LDS writes into the other buffer, last read two barriers ago) sits in front of them, so
            // the scheduler can slide it into the matrix pipe's shadow instead of running it behind the last MFMA.
            half8 af[2][WM], bf[2][WN];
            auto frags = [&](int kb, half8 (&a)[WM], half8 (&b)[WN]) {
#pragma unroll
                for (int j = 0; j < WN; j++) b[j] = *(const half8*)(sw + (wc * 16 * WN + 16 * j + l16) * W16PITCH + (((kb * 4 + g) ^ (l16 & C::SWZ)) * 16));
#pragma unroll
                for (int t = 0; t < WM; t++) a[t] = *(const half8*)(sa + (wr * 16 * WM + 16 * t + l16) * APITCH + (((kb * 4 + g) ^ (l16 & C::SWZ)) * 16));
            };
            frags(0, af[0], bf[0]);
            // The next stage's tile is stored BEHIND the first block's matrix instructions: LDS operations complete in order and
            // s_waitcnt lgkmcnt counts them all, so with the eight ds_write_b128 (13 cycles each) issued ahead of them the
            // first matrix instructions waited for the store queue, not for their fragments.  2048-id prompt: W.x 6.51 -> 6.27-6.35 ms,
            // 512 ids 2.76 -> 2.65 (behind the LAST block's matrix instructions instead: 6.45 / 2.63).  In this order the compiler
            // sinks the prefetch requests among the matrix instructions and waits vmcnt(0) at the loop header; with the requests
            // pinned at the top (no such wait: vmcnt(7) / (6) at the store) W.x measured 6.46 ms, with the store made unconditional
            // as well 6.52 -- the loop does not wait for memory.
            // (Also built into THIS kernel and measured: only the activation tile by LDS-DMA (inline assembly, the swizzle made by
            // the source addresses, one stage ahead into the other buffer, a counted s_waitcnt vmcnt ahead of the stage's barrier;
            // 148 instead of 188 VGPRs, half the ds_write_b128): the same bits, W.x 6.97 against 6.40 ms at 2048 ids, 2.88 against
            // 2.67 at 512 -- one stage of lookahead does not cover the tile's latency, the register path's two stages do.)
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                if (kb + 1 < KB) frags(kb + 1, af[(kb + 1) & 1], bf[(kb + 1) & 1]);
#pragma unroll
                for (int t = 0; t < WM; t++)
#pragma unroll
                    for (int j = 0; j < WN; j++) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kb & 1][t], bf[kb & 1][j], acc[t][j], 0, 0, 0);
                if (kb == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < nstage) store_stage(land, buf ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // (measured and not kept: the expansion made unconditional -- one basic block -- and woven between the matrix
            //  instructions with sched_group_barrier: 2048-id prompt 13.2 -> 15.1 ms; woven by hand, one 8-element unit behind
            //  every row of four matrix instructions, pinned with scheduling barriers: 12.4 -> 12.9 ms.  The two waves of a
            //  SIMD already overlap each other's expansion and matrix phases.  Nor does expanding the whole matrix once per call
            //  into an f16 scratch and running the f16-weights kernel pay, even at 2048 rows: 12.4 -> 13.4 ms -- the expansion
            //  launch and the 3.5 x larger weight stream cost more than the per-workgroup expansion they replace.)
            __syncthreads();
            return;
        }
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            // ---- operands of this quant block
            half8 bf[WN];
            float dw[WN];
#pragma unroll
            for (int j = 0; j < WN; j++) {
                if (FAST) {
                    bf[j] = *(const half8*)(sw + (wc * 16 * WN + 16 * j + l16) * W16PITCH + (((kb * 4 + g) ^ (l16 & C::SWZ)) * 16));
                } else if (QUANT) {
                    const int f = wc * 16 * WN + 16 * j + l16;
                    const uint2 by = (WT == GTEN_Q4) ? *(const uint2*)(sw + f * WPITCH + kb * 16 + (g & 1) * 8)
                                                     : *(const uint2*)(sw + f * WPITCH + (g >> 1) * (KB * 16) + kb * 16 + (g & 1) * 8);
                    bf[j] = weight_frag<WT>(by, nibble_shift, nib_mask);
                    dw[j] = sdw[kb * 256 + f];
                }
            }
            half8 af[WM];
            float4 da4[WM];
#pragma unroll
            for (int t = 0; t < WM; t++) {
                const int trow = wr * 16 * WM + 16 * t;
                af[t] = *(const half8*)(sa + (trow + l16) * APITCH + (((kb * 4 + g) ^ (l16 & C::SWZ)) * 16));
                if (EXACT) da4[t] = *(const float4*)(sda + kb * BM + trow + g * 4);
            }
            // ---- matrix phase: every MFMA of the block is issued before any result is touched, so the rescale
            //      never waits on the matrix pipe (and the SIMD's other wave fills it meanwhile)
            floatx4 isum[EXACT ? WM : 1][EXACT ? WN : 1];
#pragma unroll
            for (int t = 0; t < WM; t++)
#pragma unroll
                for (int j = 0; j < WN; j++) {
                    if (EXACT) {
                        const floatx4 z = {0.f, 0.f, 0.f, 0.f};
                        isum[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t], bf[j], z, 0, 0, 0);   // exact integer block sums
                    } else {
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t], bf[j], acc[t][j], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
            // ---- vector phase: dot += isum * da * dw, left to right like the scalar build (gten/ops.h:311).
            //      Plain f32 on purpose: packed f32 beside MFMAs is the slower form on gfx950 (build.py)
            if (EXACT) {
#pragma unroll
                for (int t = 0; t < WM; t++) {
                    const float da[4] = {da4[t].x, da4[t].y, da4[t].z, da4[t].w};
#pragma unroll
                    for (int j = 0; j < WN; j++)
#pragma unroll
                        for (int i = 0; i < 4; i++) acc[t][j][i] = acc[t][j][i] + (isum[t][j][i] * da[i]) * dw[j];
                }
                // the accumulators are pinned here: without it the whole stage's rescale sinks behind its last MFMA
#pragma unroll
                for (int t = 0; t < WM; t++)
#pragma unroll
                    for (int j = 0; j < WN; j++) asm volatile("" : "+v"(acc[t][j]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 < nstage) store_stage(land, buf ^ 1);    // stage s + 1 into the buffer last read two barriers ago
        __syncthreads();
    };
    for (int s = 0; s < nstage; s += NS) {
#pragma unroll
        for (int u = 0; u < NS; u++)
            if (s + u < nstage) stage_body(s + u, raw[u], raw[(u + 1) % NS]);
    }

    mfma_epilogue<WM, WN>(acc, ms, mi, out, out_pitch, d_out, out_dtype, rows, row0 + wr * 16 * WM, col0, g, l16, start_pos);
}

// The planes of a shared K loop -> rows: per element the planes are added in order (((p0 + p1) + p2) + p3), then the row
// is rounded exactly as k_matmul_mfma's own epilogue rounds it (Q8 block of 32 = 32 lanes; the residual sum behind it).
__global__ __launch_bounds__(256) void k_splitk_finish(const MatArgs ms, int n_planes, int rows, int start_pos)
{
    const int nbo = ms.part_pitch >> 5;                                           // Q8 blocks per row over all matrices of the launch
    const int gb = blockIdx.x * 8 + (threadIdx.x >> 5), e = threadIdx.x & 31;     // one Q8 block per 32 lanes
    if (gb >= rows * nbo) return;
    const int r = gb / nbo, bg = gb - r * nbo;
    int mi = 0;
    if (ms.n_mats > 1 && bg * 32 >= ms.part_col0[1]) mi = 1;
    if (ms.n_mats > 2 && bg * 32 >= ms.part_col0[2]) mi = 2;
    const int b = bg - (ms.part_col0[mi] >> 5);
    const float* p = ms.partial + (size_t)r * ms.part_pitch + bg * 32 + e;
    float v = p[0];
    for (int z = 1; z < n_planes; z++) v = v + p[(size_t)z * rows * ms.part_pitch];
    const Q8Scale sc = q8_scale_from_absmax(group_max<32>(fabsf(v)));
    const int qv = q8_round(v, sc.scale);
    uint8_t* blk = ms.out[mi] + (size_t)(start_pos + r) * ms.out_pitch[mi] + (size_t)b * GTEN_Q8_BYTES;
    blk[2 + e] = (uint8_t)(int8_t)qv;
    if (e == 0) *(uint16_t*)blk = sc.d16;
    if (ms.resid) {
        const uint8_t* rb = ms.resid + (size_t)(start_pos + r) * ms.resid_pitch + (size_t)b * GTEN_Q8_BYTES;
        const float sv = (float)(int)(int8_t)rb[2 + e] * h2f(*(const uint16_t*)rb) + (float)qv * sc.ddeq;
        const Q8Scale ss = q8_scale_from_absmax(group_max<32>(fabsf(sv)));
        uint8_t* sb = ms.sum_out + (size_t)(start_pos + r) * ms.resid_pitch + (size_t)b * GTEN_Q8_BYTES;
        sb[2 + e] = (uint8_t)(int8_t)q8_round(sv, ss.scale);
        if (e == 0) *(uint16_t*)sb = ss.d16;
    }
}

// k_splitk_finish for ONE 2048-wide projection with its residual sum, one WAVE per row (lane L = block L), continued into
// the RMSNorm that reads the sum (gten/modules.cpp:236-240: inp_res, then ffn_norm): the planes are added in order and the
// row is rounded (the projection's module tensor), the residual sum is rounded (the Residual's tensor), and RMSNorm runs on
// the values it would read back -- k_rms_norm_q8w's arithmetic, tree and all -- leaving its Q8 row and, for the next W.x,
// the f16 copy.  The bytes of three launches (plane sums, RMSNorm, conversion) from one.
__global__ __launch_bounds__(64) void k_splitk_finish_norm(const MatArgs ms, int n_planes, int rows, int start_pos, const uint16_t* __restrict__ nw,
                                                           uint8_t* __restrict__ nout, size_t nout_pitch, uint4* __restrict__ a16)
{
    constexpr int D = 2048;
    const int L = threadIdx.x, odd = L & 1, r = blockIdx.x;
    const float* p = ms.partial + (size_t)r * ms.part_pitch + ms.part_col0[0] + L * 32;
    float v[32];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float4 t = ((const float4*)p)[j];
        v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    }
    for (int z = 1; z < n_planes; z++) {
        const float* pz = p + (size_t)z * rows * ms.part_pitch;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float4 t = ((const float4*)pz)[j];
            v[4 * j] = v[4 * j] + t.x; v[4 * j + 1] = v[4 * j + 1] + t.y; v[4 * j + 2] = v[4 * j + 2] + t.z; v[4 * j + 3] = v[4 * j + 3] + t.w;
        }
    }
    // the residual row's block (nine dwords) and the norm weights, requested beside the planes
    const unsigned* rp = (const unsigned*)(ms.resid + (size_t)(start_pos + r) * ms.resid_pitch + (size_t)(L >> 1) * 68) + (odd ? 8 : 0);
    unsigned rw[9];
#pragma unroll
    for (int j = 0; j < 9; j++) rw[j] = rp[j];
    uint4 wq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) wq[j] = ((const uint4*)(nw + L * 32))[j];
    // 1. the projection's row
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) amax = fmaxf(amax, fabsf(v[i]));
    const Q8Scale sc = q8_scale_from_absmax(amax);
    const float dr = odd ? h2f((uint16_t)(rw[0] >> 16)) : h2f((uint16_t)(rw[0] & 0xffffu));
    unsigned pq[8];
    float amax2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const unsigned qr = odd ? rw[1 + j] : __builtin_amdgcn_alignbit(rw[j + 1], rw[j], 16);
        unsigned wd = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int qv = q8_round(v[4 * j + i], sc.scale);
            wd |= ((unsigned)qv & 0xffu) << (8 * i);
            // 2. ops::add on the stored values (k_splitk_finish's expression)
            v[4 * j + i] = (float)(int)(int8_t)(qr >> (8 * i)) * dr + (float)qv * sc.ddeq;
            amax2 = fmaxf(amax2, fabsf(v[4 * j + i]));
        }
        pq[j] = wd;
    }
    store_q8_block_lane(ms.out[0] + (size_t)(start_pos + r) * ms.out_pitch[0], L, pq, sc.d16);
    const Q8Scale ss = q8_scale_from_absmax(amax2);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        unsigned wd = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int qv = q8_round(v[4 * j + i], ss.scale);
            wd |= ((unsigned)qv & 0xffu) << (8 * i);
            v[4 * j + i] = (float)qv * ss.ddeq;                 // what RMSNorm reads back from the stored sum
        }
        pq[j] = wd;
    }
    store_q8_block_lane(ms.sum_out + (size_t)(start_pos + r) * ms.resid_pitch, L, pq, ss.d16);
    // 3. RMSNorm of the sum (k_rms_norm_q8w)
    float part[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float t8[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t8[i] = v[8 * k + i];
        part[k] = sumsq_tree8(t8);
    }
    const float sq = wave_sum((part[0] + part[1]) + (part[2] + part[3]));
    const float inv = recip_rn(sqrtf(sq / (float)D) + 1e-6f);
    const unsigned* wh = (const unsigned*)wq;
    float amax3 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint16_t wb = (uint16_t)((i & 1) ? (wh[i >> 1] >> 16) : (wh[i >> 1] & 0xffffu));
        v[i] = v[i] * inv * h2f(wb);
        amax3 = fmaxf(amax3, fabsf(v[i]));
    }
    const Q8Scale sn = q8_scale_from_absmax(amax3);
    unsigned hw[16];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        unsigned wd = 0;
        unsigned short hq[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int qv = q8_round(v[4 * j + i], sn.scale);
            wd |= ((unsigned)qv & 0xffu) << (8 * i);
            hq[i] = f2h((float)qv * sn.ddeq);
        }
        pq[j] = wd;
        hw[2 * j] = (unsigned)hq[0] | ((unsigned)hq[2] << 16);
        hw[2 * j + 1] = (unsigned)hq[1] | ((unsigned)hq[3] << 16);
    }
    store_q8_block_lane(nout + (size_t)(start_pos + r) * nout_pitch, L, pq, sn.d16);
    if (a16) {
        uint4* dst = a16 + ((size_t)r * 64 + L) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) dst[j] = make_uint4(hw[4 * j], hw[4 * j + 1], hw[4 * j + 2], hw[4 * j + 3]);
    }
}

// f16 copy of the new Q8 activation rows, owned by the library and reused call after call: ONE SET PER STREAM of the
// library (prompts are processed on stream 1 beside the decode slices of stream 0, and nothing stops a caller from running
// prompt-sized calls on both: two streams must never share a scratch whose contents live from one launch to the next)
static int act_scratch(size_t a_bytes, size_t d_bytes, uint8_t** a16, float** da)
{
    using namespace gtr;
    static uint8_t* bufs_a[2] = {nullptr, nullptr};
    static float* bufs_d[2] = {nullptr, nullptr};
    static size_t caps_a[2] = {0, 0}, caps_d[2] = {0, 0};
    const int si = stream_index();
    uint8_t*& buf_a = bufs_a[si];
    float*& buf_d = bufs_d[si];
    size_t& cap_a = caps_a[si];
    size_t& cap_d = caps_d[si];
    if (a_bytes > cap_a) {
        if (buf_a) { GTR_CHECK(hipStreamSynchronize(stream())); GTR_CHECK(hipFree(buf_a)); }
        buf_a = nullptr; cap_a = 0;
        GTR_CHECK(hipMalloc((void**)&buf_a, a_bytes + a_bytes / 2));
        cap_a = a_bytes + a_bytes / 2;
    }
    if (d_bytes > cap_d) {
        if (buf_d) { GTR_CHECK(hipStreamSynchronize(stream())); GTR_CHECK(hipFree(buf_d)); }
        buf_d = nullptr; cap_d = 0;
        GTR_CHECK(hipMalloc((void**)&buf_d, d_bytes + d_bytes / 2));
        cap_d = d_bytes + d_bytes / 2;
    }
    *a16 = buf_a; *da = buf_d;
    return 0;
}

static int partial_scratch(size_t bytes, float** out)
{
    using namespace gtr;
    static float* bufs[2] = {nullptr, nullptr};      // (one per stream, as act_scratch)
    static size_t caps[2] = {0, 0};
    float*& buf = bufs[stream_index()];
    size_t& cap = caps[stream_index()];
    if (bytes > cap) {
        if (buf) { GTR_CHECK(hipStreamSynchronize(stream())); GTR_CHECK(hipFree(buf)); }
        buf = nullptr; cap = 0;
        GTR_CHECK(hipMalloc((void**)&buf, bytes + bytes / 2));
        cap = bytes + bytes / 2;
    }
    *out = buf;
    return 0;
}

// measured, whole prompts of 16 / 64 / 256 / 512 ids (q4): 3.09 / 3.67 / 4.26 / 5.21 ms without sharing, 2.22 / 2.55 / 3.59 /
// 5.09 ms with these factors (twice the factors: 3.79 ms at 256 ids; sharing the wide gate | up loops too: no gain)
static int splitk_factor(int rows, int d_in, int d_out)
{
    // (2048 rows, round 3: factors 2 / 2, 2 / 4, 1 / 2, 1 / 4 for K = 2048 / 5632 -- 6.92 / 7.08 / 6.55 / 6.68 ms of W.x against 6.67)
    // (with the lean epilogue the unshared loops won back what the plane sums cost: sharing up to 512 / 256 / 128 / 0 rows -- W.x of a
    //  384-id prompt 2.75 / 2.55 / 2.57 / 2.59 ms, 512 ids 3.08 / 2.76 / 2.75 / 2.75, 256 ids 2.20 / 2.21 / 2.27 / 2.30, 128 ids 1.76 / 1.75 /
    //  1.74 / 2.03: shared up to 256 rows)
    if (rows > 256 || d_out > 2560 || d_out % 32 != 0) return 1;
    return (d_in >= 4096 ? 4 : 2) * (rows <= 128 ? 2 : 1);
}

// the f16 copy of rows [start_pos, n) of a Q8 activation matrix (fast form: deltas folded in); gten_mfma_convert
static int convert_rows(const void* x, size_t x_pitch, int n, int d_in, int start_pos, bool fast, uint8_t** buf_out, float** da_out)
{
    using namespace gtr;
    const int rows = n - start_pos, nb = d_in / 32;
    uint8_t* buf = nullptr;
    float* da = nullptr;
    if (int rc = act_scratch((size_t)rows * d_in * 2, (size_t)rows * nb * 4, &buf, &da)) return rc;
    const bool x2 = nb % 2 == 0 && x_pitch % 4 == 0 && ((uintptr_t)x % 4) == 0;
    const dim3 grid(x2 ? (rows * (nb / 2) + 255) / 256 : (rows * nb + 255) / 256);
    if (fast) {
        if (x2) GTR_LAUNCH(KT_MATMUL_MFMA, k_act_to_f16_x2<true>, grid, dim3(256), 0, (const uint8_t*)x, x_pitch, rows, nb, start_pos, (uint4*)buf, da);
        else GTR_LAUNCH(KT_MATMUL_MFMA, k_act_to_f16<true>, grid, dim3(256), 0, (const uint8_t*)x, x_pitch, rows, nb, start_pos, (uint4*)buf, da);
    } else {
        if (x2) GTR_LAUNCH(KT_MATMUL_MFMA, k_act_to_f16_x2<false>, grid, dim3(256), 0, (const uint8_t*)x, x_pitch, rows, nb, start_pos, (uint4*)buf, da);
        else GTR_LAUNCH(KT_MATMUL_MFMA, k_act_to_f16<false>, grid, dim3(256), 0, (const uint8_t*)x, x_pitch, rows, nb, start_pos, (uint4*)buf, da);
    }
    *buf_out = buf; *da_out = da;
    return 0;
}

// the scratch as the last convert_rows left it (the caller vouches that it holds THIS input: gten_block.hip)
static int converted_rows(int rows, int d_in, uint8_t** buf_out, float** da_out)
{
    return act_scratch((size_t)rows * d_in * 2, (size_t)rows * (d_in / 32) * 4, buf_out, da_out);
}

// the launch of one tile configuration
template <int WT, int WM, int WN, bool FAST, int KB_ = 4>
static int launch_cfg(const void* x, size_t x_pitch, const gtr::MfmaMats& m, int out_dtype, int n, int d_in, int start_pos, bool converted)
{
    using namespace gtr;
    using C = MfmaCfg<WT, WM, WN, KB_>;
    constexpr int TBM = C::BM, TBN = C::BN, TNT = 256;
    const size_t smem = (FAST || !C::QUANT) ? C::smem_fast() : C::smem();
    static bool attr_set = false;
    if (!attr_set) {
        GTR_CHECK(hipFuncSetAttribute((const void*)k_matmul_mfma<WT, WM, WN, KB_, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const int rows = n - start_pos;
    const uint8_t* a16 = (const uint8_t*)x + (size_t)start_pos * x_pitch;
    size_t a_pitch = x_pitch;
    float* da = nullptr;
    if (C::QUANT) {
        uint8_t* buf = nullptr;
        if (int rc = converted ? converted_rows(rows, d_in, &buf, &da) : convert_rows(x, x_pitch, n, d_in, start_pos, FAST, &buf, &da)) return rc;
        a16 = buf; a_pitch = (size_t)d_in * 2;
    }
    MatArgs ms{};
    int tiles = 0;
    for (int k = 0; k < m.n; k++) {
        ms.w[k] = m.w[k]; ms.out[k] = (uint8_t*)m.out[k]; ms.out_pitch[k] = m.out_pitch[k]; ms.d_out[k] = m.d_out[k]; ms.tile0[k] = tiles;
        tiles += (m.d_out[k] + TBN - 1) / TBN;
    }
    ms.n_mats = m.n;
    ms.resid = (const uint8_t*)m.resid; ms.sum_out = (uint8_t*)m.sum_out; ms.resid_pitch = m.resid_pitch;
    // Few new rows and narrow outputs leave most CUs without a workgroup and every workgroup alone with a long K loop whose
    // loads it cannot cover: the K loop is then shared by 2 (4 for d_in >= 4096) workgroups.  The factor depends on (rows,
    // d_in, d_out of the matrix) only, so a matrix gets the same sums whether it is launched alone or beside its siblings.
    int ks = 1;
    // (row segments set: never -- the factor depends on the row count, and a prompt must get the same bits whatever
    //  shares the row matrix with it, gten_hip_set_row_segments)
    if (FAST && out_dtype == GTEN_Q8 && !row_segments(nullptr)) {
        ks = splitk_factor(rows, d_in, m.d_out[0]);
        for (int k = 1; k < m.n; k++)
            if (splitk_factor(rows, d_in, m.d_out[k]) != ks) ks = 0;              // (mixed: the caller launches them one by one)
        GTR_REQUIRE(ks != 0, "matmul: matrices of one launch disagree about sharing the K loop");
        if ((d_in / 32 / C::KB) < 2 * ks) ks = 1;
    }
    const dim3 grid(tiles, (rows + TBM - 1) / TBM, ks), block(TNT);
    if (ks > 1) {
        int cols = 0;
        for (int k = 0; k < m.n; k++) { ms.part_col0[k] = cols; cols += (m.d_out[k] + 31) & ~31; }
        float* planes = nullptr;
        if (int rc = partial_scratch((size_t)ks * rows * cols * 4, &planes)) return rc;
        ms.partial = planes; ms.part_pitch = cols;
        GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<WT, WM, WN, KB_, FAST>), grid, block, smem, a16, a_pitch, (const float*)da, ms, out_dtype, n, d_in, start_pos);
        const auto al4 = [](const void* q, size_t pitch) { return ((uintptr_t)q & 3) == 0 && pitch % 4 == 0; };
        if (m.norm_w && m.n == 1 && m.resid && m.d_out[0] == 2048 && al4(m.out[0], m.out_pitch[0]) && al4(m.resid, m.resid_pitch) &&
            al4(m.sum_out, m.resid_pitch) && al4(m.norm_out, m.norm_out_pitch) && ((uintptr_t)m.norm_w & 15) == 0) {
            GTR_LAUNCH(KT_MATMUL_MFMA, k_splitk_finish_norm, dim3(rows), dim3(64), 0, ms, ks, rows, start_pos, (const uint16_t*)m.norm_w,
                       (uint8_t*)m.norm_out, m.norm_out_pitch, (uint4*)m.norm_a16);
            if (m.norm_done) *m.norm_done = true;
            return 0;
        }
        const int blocks = rows * (cols / 32);
        GTR_LAUNCH(KT_MATMUL_MFMA, k_splitk_finish, dim3((blocks + 7) / 8), dim3(256), 0, ms, ks, rows, start_pos);
        return 0;
    }
    GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<WT, WM, WN, KB_, FAST>), grid, block, smem, a16, a_pitch, (const float*)da, ms, out_dtype, n, d_in, start_pos);
    return 0;
}

// exact (per-block rescale, the scalar build's order) or fast (deltas folded into the operands): gten_hip_set_prefill_exact
static bool g_prefill_exact = false;

// the largest tile that still gives the chip enough workgroups
template <int WT, bool FAST>
static int launch_wt(const void* x, size_t x_pitch, const gtr::MfmaMats& m, int out_dtype, int n, int d_in, int start_pos, bool converted)
{
    const int rows = n - start_pos;
    auto wgs = [&](int bm, int bn) {
        int t = 0;
        for (int k = 0; k < m.n; k++) t += (m.d_out[k] + bn - 1) / bn;
        return t * ((rows + bm - 1) / bm);
    };
    // (fast form: two-block stages for the 128-feature tiles -- the expanded weight tile is four times the packed one)
#define MF_GO(WM_, WN_) return launch_cfg<WT, WM_, WN_, FAST, (((FAST || WT == GTEN_F16) && WN_ == 4) ? 2 : 4)>(x, x_pitch, m, out_dtype, n, d_in, start_pos, converted)
    // (exact form, measured at 2048 rows, q4, W.x total: <2,4> 14.2 ms | <2,2> 15.3 | <4,2> 19.3 | <4,4> 26.2 | <2,4> with
    //  2-block stages 15.3: the 4-row-tile variants fit one workgroup per CU only, and occupancy matters more than the
    //  amortised nibble expansion once the kernel is VALU-bound at 2 waves per SIMD)
    if constexpr (FAST || WT == GTEN_F16) {
        // 128 x 128 outputs per workgroup on two-block stages (two workgroups per CU): 2048 rows, q4, W.x total 10.7 ms against
        // 11.6 for <2,4> on four-block stages and 12.5 for <4,4> on four-block stages (one workgroup per CU)
        // (the 2048-wide projections give exactly 256 of these tiles, one workgroup per CU; 64 x 128 tiles -- 512 workgroups --
        //  measured the same: 13.0 against 13.2 ms for a 2048-id prompt)
        // (round 3, measured and not kept: 128 x 128 tiles for EVERY projection from 128 rows up, their K loops shared by 2-8
        //  workgroups until 200 / 400 / 800 workgroups are in flight: 128 rows 1.79 -> 2.25 / 2.37 / 2.36 ms of W.x, 256: 2.27 -> 2.71 /
        //  2.84 / 3.17, 512: 3.26 -> 3.22 / 3.63 / 3.85, 1024: 4.42 -> 4.63 / 4.73 / 6.03 -- the 64 x 64 tiles with their factors stay)
        // (round 3, measured and not kept: a kernel with EVERY global byte moved by LDS-DMA (`buffer_load ... lds` as inline
        //  assembly with hand-counted s_waitcnt vmcnt, three-slot rings for the activation tile and the packed weights, the
        //  weight tile expanded LDS -> registers -> LDS, one barrier per stage), 256 x 128 outputs on eight waves: the same
        //  bits, gate | up 138.6 us against 140.4 at 2048 rows, but the 2048-wide projections 98 us against 75 (128 tiles on
        //  256 CUs), W.x 8.6 ms against 7.6 -- and still 7.4 against 6.6 once both had the lean epilogue.  With parts of its
        //  loop knocked out (W.x ms, base 8.60): no MFMA 8.42, no fragment reads 7.71, no weight expansion 7.04, no DMA 7.03,
        //  no barrier 8.08, no epilogue 5.9 -- the matrix pipe is the smallest term; what the kernels wait for adds up from
        //  VALU issue (the epilogue), the LDS store path and exposed latencies at 2 waves per SIMD: DESIGN.md 3.3.)
        // (64 x 128 tiles only for the launches with fewer than 300 / 512 / 1024 of the 128 x 128 tiles: W.x 6.40 / 6.42 / 6.43 ms at 2048 rows
        //  against 6.44, 3.91 / 3.91 / 4.22 at 1024 rows against 3.91 -- the tile shape is not what this kernel waits for)
        if (rows > 64 && wgs(128, 128) >= 256) return launch_cfg<WT, 4, 4, FAST, 2>(x, x_pitch, m, out_dtype, n, d_in, start_pos, converted);
    }
    if (rows > 32 && wgs(64, 128) >= 384) MF_GO(2, 4);
    // (measured and not kept: 32-row tiles for launches of fewer than 256 / 512 of the 64 x 64 tiles -- a 256-id prompt 5.10 ->
    //  5.32 ms: the few workgroups wait for their loads, not for the matrix pipe; sharing the K loop is what helps, splitk_factor)
    if (rows > 32) MF_GO(2, 2);
    MF_GO(1, 2);
#undef MF_GO
}

bool gtr::prefill_exact() { return g_prefill_exact; }

extern "C" int gten_hip_set_prefill_exact(int on)
{
    g_prefill_exact = on != 0;
    return 0;
}

int gten_launch_matmul_mfma_multi(const void* x, size_t x_pitch, int w_dtype, const gtr::MfmaMats& m, int out_dtype,
                                  int n, int d_in, int start_pos, bool converted)
{
    if (w_dtype == GTEN_F16) return launch_wt<GTEN_F16, false>(x, x_pitch, m, out_dtype, n, d_in, start_pos, converted);
    if (g_prefill_exact) {
        if (w_dtype == GTEN_Q8) return launch_wt<GTEN_Q8, false>(x, x_pitch, m, out_dtype, n, d_in, start_pos, converted);
        return launch_wt<GTEN_Q4, false>(x, x_pitch, m, out_dtype, n, d_in, start_pos, converted);
    }
    if (w_dtype == GTEN_Q8) return launch_wt<GTEN_Q8, true>(x, x_pitch, m, out_dtype, n, d_in, start_pos, converted);
    return launch_wt<GTEN_Q4, true>(x, x_pitch, m, out_dtype, n, d_in, start_pos, converted);
}

int gten_mfma_convert(const void* x, size_t x_pitch, int n, int d_in, int start_pos)
{
    uint8_t* buf;
    float* da;
    return convert_rows(x, x_pitch, n, d_in, start_pos, !g_prefill_exact, &buf, &da);
}

int gten_mfma_scratch(int rows, int d_max, uint8_t** a16)
{
    float* da;
    return act_scratch((size_t)rows * d_max * 2, (size_t)rows * (d_max / 32) * 4, a16, &da);
}

int gten_launch_matmul_mfma(const void* x, int x_dtype, size_t x_pitch, const void* w, int w_dtype,
                            void* out, int out_dtype, size_t out_pitch, int n, int d_in, int d_out, int start_pos)
{
    (void)x_dtype;
    gtr::MfmaMats m{};
    m.n = 1; m.w[0] = w; m.out[0] = out; m.out_pitch[0] = out_pitch; m.d_out[0] = d_out;
    return gten_launch_matmul_mfma_multi(x, x_pitch, w_dtype, m, out_dtype, n, d_in, start_pos, false);
}
