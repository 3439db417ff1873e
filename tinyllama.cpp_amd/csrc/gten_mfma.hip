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
// Tile: one workgroup = 64 rows x 64 output features, 4 waves; wave w owns
// feature tile w (16 features) for all four 16-row tiles, so each converted
// weight fragment (block dequant staged in registers) feeds 4 MFMAs.  The
// activation block of the 64 rows is converted to f16 once per workgroup and
// staged in LDS (80-byte row pitch: conflict-free 16-byte fragment reads).
#include "gten_dev.h"
#include "gten_rt.h"

using namespace gtd;

extern __shared__ __attribute__((aligned(16))) uint8_t g_smem[];

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

#define MF_BM 64
#define MF_BN 64
#define MF_APITCH 80            // bytes per staged activation row (32 halves + pad)

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

// 8 consecutive quants of one weight row -> 8 f16 in K order
template <int WT>
__device__ __forceinline__ half8 weight_frag(const uint2 bytes, bool low_nibbles)
{
    unsigned r[4];
    const unsigned w[2] = {bytes.x, bytes.y};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // bytes (2i, 2i+1) of the 8 -> [b_lo, 0, b_hi, 0]
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
    uint8_t* lds_a = g_smem;                                   // 64 rows x 80 B
    float* lds_da = (float*)(g_smem + MF_BM * MF_APITCH);      // 64 activation deltas of the current block
    float* lds_out = lds_da + MF_BM;                           // 64 x 65 f32 (epilogue)

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int row0 = start_pos + blockIdx.y * MF_BM;
    const int col0 = blockIdx.x * MF_BN;
    const int nb = d_in >> 5;
    const int g = lane >> 4, l16 = lane & 15;

    // this lane's weight row (feature) and its slice of each block
    const int col = col0 + wid * 16 + l16;
    const int colc = col < d_out ? col : d_out - 1;
    const PackedW pw = packed_view(w, WT, d_out, d_in);

    // staging role: thread -> (row, 8-element part) of the activation block
    const int srow = threadIdx.x >> 2, spart = threadIdx.x & 3;
    const int grow = row0 + srow < n ? row0 + srow : n - 1;
    const uint8_t* xrow = x + (size_t)grow * x_pitch;

    floatx4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (floatx4){0.f, 0.f, 0.f, 0.f};

    for (int b = 0; b < nb; b++) {
        // ---- weight fragment of this block: 8 quants of feature `colc`, K slice g
        half8 bf;
        float dw = 1.0f;
        if (WT == GTEN_F16) {
            bf = *(const half8*)((const uint16_t*)w + (size_t)colc * d_in + b * 32 + g * 8);
        } else if (WT == GTEN_Q4) {
            const uint2 by = *(const uint2*)(pw.qs + ((size_t)colc * nb + b) * 16 + (g & 1) * 8);
            bf = weight_frag<GTEN_Q4>(by, g >= 2);
            dw = h2f(pw.ds[(size_t)colc * nb + b]);
        } else {
            const uint2 by = *(const uint2*)(pw.qs + (size_t)colc * nb * 32 + (size_t)(g >> 1) * nb * 16 + (size_t)b * 16 + (g & 1) * 8);
            bf = weight_frag<GTEN_Q8>(by, false);
            dw = h2f(pw.ds[(size_t)colc * nb + b]);
        }
        // ---- stage the activation block of the 64 rows as f16
        __syncthreads();
        if (WT == GTEN_F16) {
            *(uint4*)(lds_a + srow * MF_APITCH + spart * 16) = *(const uint4*)(xrow + (size_t)b * 64 + spart * 16);
        } else {
            const uint16_t* q16 = (const uint16_t*)(xrow + (size_t)b * GTEN_Q8_BYTES + 2 + spart * 8);
            unsigned r[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned two = q16[i];                                   // bytes (2i, 2i+1)
                const unsigned pair = (two & 0xffu) | ((two & 0xff00u) << 8);
                r[i] = pk_int_to_f16(pair ^ 0x64806480u, 1024.0f + 128.0f);
            }
            *(uint4*)(lds_a + srow * MF_APITCH + spart * 16) = make_uint4(r[0], r[1], r[2], r[3]);
            if (spart == 0) lds_da[srow] = h2f(*(const uint16_t*)(xrow + (size_t)b * GTEN_Q8_BYTES));
        }
        __syncthreads();
        // ---- 4 row tiles x this wave's feature tile
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const half8 af = *(const half8*)(lds_a + (t * 16 + l16) * MF_APITCH + g * 16);
            if (WT == GTEN_F16) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[t], 0, 0, 0);
            } else {
                const floatx4 z = {0.f, 0.f, 0.f, 0.f};
                const floatx4 isum = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, z, 0, 0, 0);   // exact integer block sums
                const float4 da = *(const float4*)(lds_da + t * 16 + g * 4);
                // dot += isum * da * dw, left to right like the scalar build (gten/ops.h:311)
                acc[t][0] += isum[0] * da.x * dw;
                acc[t][1] += isum[1] * da.y * dw;
                acc[t][2] += isum[2] * da.z * dw;
                acc[t][3] += isum[3] * da.w * dw;
            }
        }
    }

    // ---- epilogue: tile -> LDS -> rows written in the output dtype (gten/ops.h:73-96)
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; t++)
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
    const size_t smem = MF_BM * MF_APITCH + MF_BM * 4 + (size_t)MF_BM * 65 * 4;
    if (w_dtype == GTEN_F16)
        GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<GTEN_F16>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    else if (w_dtype == GTEN_Q8)
        GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<GTEN_Q8>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    else
        GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<GTEN_Q4>), grid, block, smem, (const uint8_t*)x, x_pitch, w, (uint8_t*)out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    return 0;
}
