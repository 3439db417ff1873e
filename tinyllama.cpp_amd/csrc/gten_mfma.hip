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
// deltas are folded into the f16 operands -- activations are expanded once per call to f16(q * da), a weight fragment
// is multiplied by its block's dw as it leaves LDS (four v_pk_mul_f16) -- and the products accumulate ACROSS the K
// blocks inside the matrix core.  No per-block rescale (12 of the exact form's 21 VALU instructions per MFMA), no
// separate block-sum registers (so the 4 x 4 register tile fits).  What it costs: each operand element carries one
// fp16 rounding (relative 2^-11, a fifth of the Q8 quantization step's own noise) and the sum is no longer in the
// scalar build's order -- results agree with the exact form to ~1e-3 relative, inside the q8 / q4 logit band
// (tests/test_prefill_gpu.py holds both forms to the reference's full-size prompt goldens).
#include "gten_dev.h"
#include "gten_rt.h"

using namespace gtd;

extern __shared__ __attribute__((aligned(16))) uint8_t g_smem[];

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

__device__ __forceinline__ float row16_absmax(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    return fmaxf(v, dpp_mov<0x140>(v));
}

template <int WT, int WM, int WN, int KB_ = 4>
struct MfmaCfg {
    static constexpr bool QUANT = (WT != GTEN_F16);
    static constexpr int BM = 32 * WM, BN = 32 * WN;       // workgroup tile: 2 x 2 waves
    static constexpr int KB = KB_;                         // quant blocks per stage (4 = 128 K; 2 halves the LDS stage: one more workgroup per CU)
    static constexpr int APITCH = KB * 64 + 16;            // bytes per staged activation row (odd multiple of 16: conflict-free b128 reads)
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

// (2 waves per SIMD = a 256-VGPR budget: the block sums then come back in VGPRs instead of AGPRs, which
//  would cost four v_accvgpr_read per MFMA in a loop that is bound by VALU issue)
// a16: f16 activation rows of the NEW rows (row 0 = start_pos), pitch d_in * 2 (quantized) or x itself (f16 weights)
template <int WT, int WM, int WN, int KB_, bool FAST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_matmul_mfma(
    const uint8_t* __restrict__ a16, size_t a_pitch, const float* __restrict__ da_rows, const void* __restrict__ w,
    uint8_t* __restrict__ out, int out_dtype, size_t out_pitch, int n, int d_in, int d_out, int start_pos)
{
    using C = MfmaCfg<WT, WM, WN, KB_>;
    constexpr int BM = C::BM, BN = C::BN, KB = C::KB, APITCH = C::APITCH, WPITCH = C::WPITCH, WROW = C::WROW;
    constexpr bool QUANT = C::QUANT;
    constexpr bool EXACT = QUANT && !FAST;                   // per-block rescale in the scalar build's order
    static_assert(QUANT || !FAST, "the fast form is about quantized weights");

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    const int g = lane >> 4, l16 = lane & 15;
    const int rows = n - start_pos;
    const int row0 = blockIdx.y * BM;                        // relative to start_pos
    const int colw = blockIdx.x * BN;                        // first feature of the workgroup
    const int col0 = colw + wc * 16 * WN;                    // first feature of this wave
    const int nb = d_in >> 5;
    const int nstage = nb / KB;
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
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)pw.qs, 0, QUANT ? (int)((size_t)d_out * nb * (WT == GTEN_Q4 ? 16 : 32)) : 0, 0x00020000);
    const auto rs_dw = __builtin_amdgcn_make_buffer_rsrc((void*)pw.ds, 0, QUANT ? d_out * nb * 2 : 0, 0x00020000);
    // activations: piece p -> (row p / (KB*4), piece-in-row p % (KB*4)); a stage of a row is KB * 64 contiguous bytes
    unsigned a_src[C::A_PIECES], a_dst[C::A_PIECES];
#pragma unroll
    for (int k = 0; k < C::A_PIECES; k++) {
        const int p = threadIdx.x + 256 * k, r = p / (KB * 4), c = p % (KB * 4);
        a_src[k] = (unsigned)min(row0 + r, rows - 1) * (unsigned)a_pitch + c * 16;
        a_dst[k] = r * APITCH + c * 16;
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
    constexpr int WP = C::W_PIECES > 0 ? C::W_PIECES : 1;
    static_assert(!QUANT || BN * (WROW / 16) == C::W_PIECES * 256, "weight tile must be a whole number of pieces per thread");
    unsigned w_src[WP], w_dst[WP];
#pragma unroll
    for (int k = 0; k < C::W_PIECES; k++) {
        const int p = threadIdx.x + 256 * k, pr = WROW / 16, f = p / pr, c = p % pr;
        const unsigned frow = (unsigned)min(colw + f, d_out - 1);
        if (WT == GTEN_Q4) w_src[k] = frow * nb * 16 + c * 16;
        else w_src[k] = frow * nb * 32 + (c / KB) * nb * 16 + (c % KB) * 16;     // plane c / KB, block c % KB
        w_dst[k] = f * WPITCH + c * 16;
    }
    // weight deltas: thread -> feature t: KB halves per stage (threads past BN fill padding slots)
    const unsigned dw_src = (unsigned)min(colw + (int)threadIdx.x, d_out - 1) * nb * 2;

    struct Raw { v4i_t a[C::A_PIECES]; v4i_t w[WP]; float da[DA_PIECES]; v2i_t dw; };
    auto load_stage = [&](int s, Raw& r) {
#pragma unroll
        for (int k = 0; k < C::A_PIECES; k++) r.a[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_src[k], s * (KB * 64), 0);
        if (QUANT) {
#pragma unroll
            for (int k = 0; k < C::W_PIECES; k++) r.w[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_src[k], s * (KB * 16), 0);
            if (EXACT) {
#pragma unroll
                for (int k = 0; k < DA_PIECES; k++) r.da[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_da, da_src[k], s * (KB * 4), 0));
            }
            if (KB == 4) r.dw = __builtin_amdgcn_raw_buffer_load_b64(rs_dw, dw_src, s * (KB * 2), 0);
            else r.dw[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_dw, dw_src, s * (KB * 2), 0);
        }
    };
    auto store_stage = [&](const Raw& r, int buf) {
        uint8_t* sa = g_smem + buf * C::STAGE;
#pragma unroll
        for (int k = 0; k < C::A_PIECES; k++) *(v4i_t*)(sa + a_dst[k]) = r.a[k];
        if (QUANT) {
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

    // f16 weights: fragments straight from global memory (16 bytes per lane, 64-byte runs per row)
    size_t wrow16[WN];
#pragma unroll
    for (int j = 0; j < WN; j++) wrow16[j] = (size_t)min(col0 + 16 * j + l16, d_out - 1) * d_in + g * 8;

    // Stage s is computed from LDS buffer s & 1 while stage s + 1 waits in registers (stored behind the compute)
    // and stage s + 2 is requested: two stages of memory latency are covered by one stage of work each.
    Raw raw0, raw1;
    load_stage(0, raw0);
    store_stage(raw0, 0);
    load_stage(1, raw1);
    __syncthreads();

    auto stage_body = [&](int s, Raw& fetch, const Raw& land) {
        const int buf = s & 1;
        const uint8_t* sa = g_smem + buf * C::STAGE;
        const float* sda = (const float*)(sa + C::A_BYTES);
        const uint8_t* sw = sa + C::A_BYTES + C::DA_BYTES;
        const float* sdw = (const float*)(sw + C::W_BYTES);
        load_stage(s + 2, fetch);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            // ---- operands of this quant block
            half8 bf[WN];
            float dw[WN];
#pragma unroll
            for (int j = 0; j < WN; j++) {
                if (QUANT) {
                    const int f = wc * 16 * WN + 16 * j + l16;
                    const uint2 by = (WT == GTEN_Q4) ? *(const uint2*)(sw + f * WPITCH + kb * 16 + (g & 1) * 8)
                                                     : *(const uint2*)(sw + f * WPITCH + (g >> 1) * (KB * 16) + kb * 16 + (g & 1) * 8);
                    bf[j] = weight_frag<WT>(by, nibble_shift, nib_mask);
                    dw[j] = sdw[kb * 256 + f];
                    if (FAST) {
                        // the block's weight delta folded into the fragment (exact f16 value, one rounding per element)
                        const _Float16 dh = (_Float16)dw[j];
                        bf[j] = bf[j] * (half8){dh, dh, dh, dh, dh, dh, dh, dh};
                    }
                } else {
                    bf[j] = *(const half8*)((const uint16_t*)w + wrow16[j] + (size_t)(s * KB + kb) * 32);
                }
            }
            half8 af[WM];
            float4 da4[WM];
#pragma unroll
            for (int t = 0; t < WM; t++) {
                const int trow = wr * 16 * WM + 16 * t;
                af[t] = *(const half8*)(sa + (trow + l16) * APITCH + kb * 64 + g * 16);
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
    for (int s = 0; s < nstage; s += 2) {
        stage_body(s, raw0, raw1);
        if (s + 1 < nstage) stage_body(s + 1, raw1, raw0);
    }

    // ---- rows written in the output dtype straight from the accumulators (gten/ops.h:73-96)
#pragma unroll
    for (int t = 0; t < WM; t++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = row0 + wr * 16 * WM + 16 * t + 4 * g + i;
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
                    if (ok0) blk[2 + l16] = (uint8_t)(int8_t)q8_round(v0, sc.scale);
                    if (ok1) blk[18 + l16] = (uint8_t)(int8_t)q8_round(v1, sc.scale);
                    if (ok0 && l16 == 0) *(uint16_t*)blk = sc.d16;
                } else if (out_dtype == GTEN_F16) {
                    if (ok0) ((uint16_t*)orow)[c0] = f2h(v0);
                    if (ok1) ((uint16_t*)orow)[c1] = f2h(v1);
                } else {
                    if (ok0) ((float*)orow)[c0] = v0;
                    if (ok1) ((float*)orow)[c1] = v1;
                }
            }
        }
    }
}

// f16 copy of the new Q8 activation rows, owned by the library and reused call after call (one stream)
static int act_scratch(size_t a_bytes, size_t d_bytes, uint8_t** a16, float** da)
{
    using namespace gtr;
    static uint8_t* buf_a = nullptr;
    static float* buf_d = nullptr;
    static size_t cap_a = 0, cap_d = 0;
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

template <int WT, int WM, int WN, bool FAST, int KB_ = 4>
static int launch_cfg(const void* x, size_t x_pitch, const void* w, void* out, int out_dtype, size_t out_pitch,
                      int n, int d_in, int d_out, int start_pos)
{
    using namespace gtr;
    using C = MfmaCfg<WT, WM, WN, KB_>;
    static bool attr_set = false;
    if (!attr_set) {
        GTR_CHECK(hipFuncSetAttribute((const void*)k_matmul_mfma<WT, WM, WN, KB_, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::smem()));
        attr_set = true;
    }
    const int rows = n - start_pos, nb = d_in / 32;
    const uint8_t* a16 = (const uint8_t*)x + (size_t)start_pos * x_pitch;
    size_t a_pitch = x_pitch;
    float* da = nullptr;
    if (C::QUANT) {
        uint8_t* buf = nullptr;
        if (int rc = act_scratch((size_t)rows * d_in * 2, (size_t)rows * nb * 4, &buf, &da)) return rc;
        if (nb % 2 == 0 && x_pitch % 4 == 0 && ((uintptr_t)x % 4) == 0)
            GTR_LAUNCH(KT_MATMUL_MFMA, k_act_to_f16_x2<FAST>, dim3((rows * (nb / 2) + 255) / 256), dim3(256), 0, (const uint8_t*)x, x_pitch, rows, nb,
                       start_pos, (uint4*)buf, da);
        else
            GTR_LAUNCH(KT_MATMUL_MFMA, k_act_to_f16<FAST>, dim3((rows * nb + 255) / 256), dim3(256), 0, (const uint8_t*)x, x_pitch, rows, nb, start_pos,
                       (uint4*)buf, da);
        a16 = buf; a_pitch = (size_t)d_in * 2;
    }
    const dim3 grid((d_out + C::BN - 1) / C::BN, (rows + C::BM - 1) / C::BM), block(256);
    GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<WT, WM, WN, KB_, FAST>), grid, block, C::smem(), a16, a_pitch, (const float*)da, w, (uint8_t*)out,
               out_dtype, out_pitch, n, d_in, d_out, start_pos);
    return 0;
}

// exact (per-block rescale, the scalar build's order) or fast (deltas folded into the operands): gten_hip_set_prefill_exact
static bool g_prefill_exact = false;

// the largest tile that still gives the chip enough workgroups
template <int WT, bool FAST>
static int launch_wt(const void* x, size_t x_pitch, const void* w, void* out, int out_dtype, size_t out_pitch,
                     int n, int d_in, int d_out, int start_pos)
{
    const int rows = n - start_pos;
    auto wgs = [&](int bm, int bn) { return ((d_out + bn - 1) / bn) * ((rows + bm - 1) / bm); };
#define MF_GO(WM_, WN_) return launch_cfg<WT, WM_, WN_, FAST>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos)
    // (exact form, measured at 2048 rows, q4, W.x total: <2,4> 14.2 ms | <2,2> 15.3 | <4,2> 19.3 | <4,4> 26.2 | <2,4> with
    //  2-block stages 15.3: the 4-row-tile variants fit one workgroup per CU only, and occupancy matters more than the
    //  amortised nibble expansion once the kernel is VALU-bound at 2 waves per SIMD)
    if constexpr (FAST) {
        // 128 x 128 outputs per workgroup on two-block stages (two workgroups per CU): 2048 rows, q4, W.x total 10.7 ms against
        // 11.6 for <2,4> on four-block stages and 12.5 for <4,4> on four-block stages (one workgroup per CU)
        if (rows > 64 && wgs(128, 128) >= 256) return launch_cfg<WT, 4, 4, FAST, 2>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    }
    if (rows > 32 && wgs(64, 128) >= 384) MF_GO(2, 4);
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

int gten_launch_matmul_mfma(const void* x, int x_dtype, size_t x_pitch, const void* w, int w_dtype,
                            void* out, int out_dtype, size_t out_pitch, int n, int d_in, int d_out, int start_pos)
{
    (void)x_dtype;
    if (w_dtype == GTEN_F16) return launch_wt<GTEN_F16, false>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    if (g_prefill_exact) {
        if (w_dtype == GTEN_Q8) return launch_wt<GTEN_Q8, false>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
        return launch_wt<GTEN_Q4, false>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    }
    if (w_dtype == GTEN_Q8) return launch_wt<GTEN_Q8, true>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    return launch_wt<GTEN_Q4, true>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
}
