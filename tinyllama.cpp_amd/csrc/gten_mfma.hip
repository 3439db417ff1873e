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
// Tile: one workgroup = 128 rows x 64 output features, 4 waves; wave w owns
// feature tile w (16 features) for all eight 16-row tiles, so each dequantized
// weight fragment feeds 8 MFMAs.  Four quant blocks (128 K) of the 128 rows are
// converted to f16 once per workgroup and staged in LDS per barrier pair
// (272-byte row pitch: conflict-free 16-byte fragment reads); the per-block
// rescale runs as packed f32 math (two accumulator lanes per instruction).
#include "gten_dev.h"
#include "gten_rt.h"

using namespace gtd;

extern __shared__ __attribute__((aligned(16))) uint8_t g_smem[];

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

#define MF_BM 128               // rows per workgroup (8 MFMA row tiles per wave)
#define MF_BN 64                // output features per workgroup (one 16-wide tile per wave)
#define MF_KB 4                 // quant blocks (of 32) staged per barrier pair
#define MF_APITCH 272           // bytes per staged activation row: 128 halves + 16 pad (conflict-free b128 reads)

typedef float float2_t __attribute__((ext_vector_type(2)));

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

// four int8 in a dword -> four exact f16 in K order (two dwords)
__device__ __forceinline__ void int8x4_to_f16(unsigned w, unsigned& lo, unsigned& hi)
{
    lo = pk_int_to_f16(__builtin_amdgcn_perm(0, w, 0x0c010c00) ^ 0x64806480u, 1024.0f + 128.0f);   // bytes 0,1 via (q + 128)
    hi = pk_int_to_f16(__builtin_amdgcn_perm(0, w, 0x0c030c02) ^ 0x64806480u, 1024.0f + 128.0f);   // bytes 2,3
}

// 8 consecutive quants of one weight row -> 8 f16 in K order
template <int WT>
__device__ __forceinline__ half8 weight_frag(const uint2 bytes, bool low_nibbles)
{
    unsigned r[4];
    const unsigned w[2] = {bytes.x, bytes.y};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned src = w[i >> 1];
        const unsigned pair = (i & 1) ? __builtin_amdgcn_perm(0, src, 0x0c030c02) : __builtin_amdgcn_perm(0, src, 0x0c010c00);
        if (WT == GTEN_Q4) {
            const unsigned nib = low_nibbles ? (pair & 0x000f000fu) : ((pair >> 4) & 0x000f000fu);
            r[i] = pk_int_to_f16(nib | 0x64006400u, 1024.0f + 7.0f);          // (nibble - 7), gten/quants.h:78-90
        } else {
            r[i] = pk_int_to_f16(pair ^ 0x64806480u, 1024.0f + 128.0f);        // int8 via (q + 128)
        }
    }
    half8 out;
    __builtin_memcpy(&out, r, 16);
    return out;
}

template <int WT>
__global__ __launch_bounds__(256) void k_matmul_mfma(const uint8_t* __restrict__ x, size_t x_pitch, const void* __restrict__ w,
                                                     uint8_t* __restrict__ out, int out_dtype, size_t out_pitch,
                                                     int n, int d_in, int d_out, int start_pos)
{
    uint8_t* lds_a = g_smem;                                        // 128 rows x 272 B: MF_KB blocks of f16 activations
    float* lds_da = (float*)(g_smem + MF_BM * MF_APITCH);           // [MF_KB][128] activation deltas
    float* lds_out = (float*)g_smem;                                // epilogue: 128 x 65 f32 (reuses the stage)

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row0 = start_pos + blockIdx.y * MF_BM;
    const int col0 = blockIdx.x * MF_BN;
    const int nb = d_in >> 5;
    const int g = lane >> 4, l16 = lane & 15;

    const int col = col0 + wid * 16 + l16;
    const int colc = col < d_out ? col : d_out - 1;
    const PackedW pw = packed_view(w, WT, d_out, d_in);

    // staging role: thread -> (row, pair of quant blocks) of the stage: 128 rows x 2 pairs
    const int srow = threadIdx.x >> 1, spair = threadIdx.x & 1;
    const int grow = row0 + srow < n ? row0 + srow : n - 1;
    const uint8_t* xrow = x + (size_t)grow * x_pitch;

    floatx4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = (floatx4){0.f, 0.f, 0.f, 0.f};

    for (int b0 = 0; b0 < nb; b0 += MF_KB) {
        // ---- weight fragments of the stage's blocks: 8 quants of feature `colc`, K slice g of each block
        half8 bf[MF_KB];
        float dw[MF_KB];
#pragma unroll
        for (int kb = 0; kb < MF_KB; kb++) {
            const int b = b0 + kb;
            dw[kb] = 1.0f;
            if (WT == GTEN_F16) {
                bf[kb] = *(const half8*)((const uint16_t*)w + (size_t)colc * d_in + b * 32 + g * 8);
            } else if (WT == GTEN_Q4) {
                const uint2 by = *(const uint2*)(pw.qs + ((size_t)colc * nb + b) * 16 + (g & 1) * 8);
                bf[kb] = weight_frag<GTEN_Q4>(by, g >= 2);
                dw[kb] = h2f(pw.ds[(size_t)colc * nb + b]);
            } else {
                const uint2 by = *(const uint2*)(pw.qs + (size_t)colc * nb * 32 + (size_t)(g >> 1) * nb * 16 + (size_t)b * 16 + (g & 1) * 8);
                bf[kb] = weight_frag<GTEN_Q8>(by, false);
                dw[kb] = h2f(pw.ds[(size_t)colc * nb + b]);
            }
        }
        // ---- stage MF_KB activation blocks of the 128 rows as f16 (+ their deltas)
        __syncthreads();
        if (WT == GTEN_F16) {
            // 128 halves of this row = 256 B; this thread copies its half (8 x 16 B)
            const uint4* src = (const uint4*)(xrow + (size_t)b0 * 64 + spair * 128);
            uint4* dst = (uint4*)(lds_a + srow * MF_APITCH + spair * 128);
#pragma unroll
            for (int i = 0; i < 8; i++) dst[i] = src[i];
        } else {
            // two consecutive 34-byte blocks = 68 bytes, 4-byte aligned: [d0 | q0 x32 | d1 | q1 x32]
            const unsigned* src = (const unsigned*)(xrow + (size_t)(b0 + spair * 2) * GTEN_Q8_BYTES);
            unsigned wv[17];
#pragma unroll
            for (int i = 0; i < 17; i++) wv[i] = src[i];
            uint4* dst = (uint4*)(lds_a + srow * MF_APITCH + spair * 128);
#pragma unroll
            for (int i = 0; i < 8; i += 2) {       // block 0: quants straddle dwords by two bytes
                unsigned f[4];
                int8x4_to_f16(__builtin_amdgcn_alignbit(wv[i + 1], wv[i], 16), f[0], f[1]);
                int8x4_to_f16(__builtin_amdgcn_alignbit(wv[i + 2], wv[i + 1], 16), f[2], f[3]);
                dst[i >> 1] = make_uint4(f[0], f[1], f[2], f[3]);
            }
#pragma unroll
            for (int i = 0; i < 8; i += 2) {       // block 1: quants are dword aligned
                unsigned f[4];
                int8x4_to_f16(wv[9 + i], f[0], f[1]);
                int8x4_to_f16(wv[10 + i], f[2], f[3]);
                dst[4 + (i >> 1)] = make_uint4(f[0], f[1], f[2], f[3]);
            }
            lds_da[(spair * 2) * MF_BM + srow] = h2f((uint16_t)(wv[0] & 0xffffu));
            lds_da[(spair * 2 + 1) * MF_BM + srow] = h2f((uint16_t)(wv[8] >> 16));
        }
        __syncthreads();
        // ---- MF_KB blocks x 8 row tiles against this wave's feature tile
#pragma unroll
        for (int kb = 0; kb < MF_KB; kb++) {
            const float2_t dw2 = {dw[kb], dw[kb]};
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const half8 af = *(const half8*)(lds_a + (t * 16 + l16) * MF_APITCH + kb * 64 + g * 16);
                if (WT == GTEN_F16) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[kb], acc[t], 0, 0, 0);
                } else {
                    const floatx4 z = {0.f, 0.f, 0.f, 0.f};
                    const floatx4 isum = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[kb], z, 0, 0, 0);   // exact integer block sums
                    const float4 da = *(const float4*)(lds_da + kb * MF_BM + t * 16 + g * 4);
                    // dot += isum * da * dw, left to right like the scalar build (gten/ops.h:311), two lanes per instruction
                    const float2_t i01 = {isum[0], isum[1]}, i23 = {isum[2], isum[3]};
                    const float2_t d01 = {da.x, da.y}, d23 = {da.z, da.w};
                    float2_t a01 = {acc[t][0], acc[t][1]}, a23 = {acc[t][2], acc[t][3]};
                    a01 = a01 + (i01 * d01) * dw2;
                    a23 = a23 + (i23 * d23) * dw2;
                    acc[t][0] = a01[0]; acc[t][1] = a01[1]; acc[t][2] = a23[0]; acc[t][3] = a23[1];
                }
            }
        }
    }

    // ---- epilogue: tile -> LDS -> rows written in the output dtype (gten/ops.h:73-96)
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) lds_out[(t * 16 + g * 4 + i) * 65 + wid * 16 + l16] = acc[t][i];
    __syncthreads();
    const int half = lane >> 5, e = lane & 31;               // half wave = one 32-wide output block
    for (int rr = wid; rr < MF_BM; rr += 4) {
        const int r = row0 + rr;
        const int c = col0 + half * 32 + e;
        const bool ok = (r < n) && (c < d_out);
        const float v = ok ? lds_out[rr * 65 + half * 32 + e] : 0.f;
        uint8_t* orow = out + (size_t)(r < n ? r : n - 1) * out_pitch;
        if (out_dtype == GTEN_Q8) {
            const Q8Scale s = q8_scale_from_absmax(group_max<32>(fabsf(v)));
            if (ok) {
                uint8_t* blk = orow + (size_t)((col0 >> 5) + half) * GTEN_Q8_BYTES;
                blk[2 + e] = (uint8_t)(int8_t)q8_round(v, s.scale);
                if (e == 0) *(uint16_t*)blk = s.d16;
            }
        } else if (out_dtype == GTEN_F16) {
            if (ok) ((uint16_t*)orow)[c] = f2h(v);
        } else {
            if (ok) ((float*)orow)[c] = v;
        }
    }
}

int gten_launch_matmul_mfma(const void* x, int x_dtype, size_t x_pitch, const void* w, int w_dtype,
                            void* out, int out_dtype, size_t out_pitch, int n, int d_in, int d_out, int start_pos)
{
    using namespace gtr;
    (void)x_dtype;
    const dim3 grid((d_out + MF_BN - 1) / MF_BN, (n - start_pos + MF_BM - 1) / MF_BM), block(256);
    const size_t smem = (size_t)MF_BM * MF_APITCH + (size_t)MF_KB * MF_BM * 4;     // >= the 128 x 65 f32 epilogue tile
    if (w_dtype == GTEN_F16)
        GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<GTEN_F16>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    else if (w_dtype == GTEN_Q8)
        GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<GTEN_Q8>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    else
        GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<GTEN_Q4>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    return 0;
}
