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
// Activations of KB quant blocks are converted to f16 once per workgroup into a
// double-buffered LDS stage (one barrier per stage); the raw weight bytes and the
// raw activation bytes of the NEXT stage are requested before the current stage is
// computed.  The K order inside a fragment is permuted (0,2,1,3 per 4 quants) on
// BOTH operands, which turns the nibble/byte -> f16 expansion into mask-and-or
// instead of byte shuffles.  Results leave straight from the accumulators: a Q8
// output block is two adjacent 16-wide tiles of the same wave (DPP row maximum).
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

template <int WT, int WM, int WN>
struct MfmaCfg {
    static constexpr int BM = 32 * WM, BN = 32 * WN;       // workgroup tile: 2 x 2 waves
    static constexpr int KB = 4;                           // quant blocks per stage; staging threads = BM rows x 2 block pairs
    static constexpr int APITCH = KB * 64 + 16;            // bytes per staged row (odd multiple of 16: conflict-free b128 reads)
    static constexpr int STAGE = BM * APITCH + KB * BM * 4;  // f16 activations + f32 deltas
    static constexpr size_t smem() { return (size_t)2 * STAGE; }
};

// (2 waves per SIMD = a 256-VGPR budget: the block sums then come back in VGPRs instead of AGPRs, which
//  would cost four v_accvgpr_read per MFMA in a loop that is bound by VALU issue)
template <int WT, int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_matmul_mfma(const uint8_t* __restrict__ x, size_t x_pitch, const void* __restrict__ w,
                                                     uint8_t* __restrict__ out, int out_dtype, size_t out_pitch,
                                                     int n, int d_in, int d_out, int start_pos)
{
    using C = MfmaCfg<WT, WM, WN>;
    constexpr int BM = C::BM, KB = C::KB, APITCH = C::APITCH, PAIRS = KB / 2;
    constexpr bool QUANT = (WT != GTEN_F16);

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 1, wc = wid & 1;
    const int g = lane >> 4, l16 = lane & 15;
    const int row0 = start_pos + blockIdx.y * BM;
    const int col0 = blockIdx.x * C::BN + wc * 16 * WN;
    const int nb = d_in >> 5;
    const int nstage = (nb + KB - 1) / KB;
    const PackedW pw = packed_view(w, WT, d_out, d_in);
    const int nibble_shift = (g < 2) ? 4 : 0;
    unsigned nib_mask = 0x000f000fu;
    asm volatile("" : "+v"(nib_mask));                       // keep it in a register (see and_or)

    // this lane's weight rows (one per feature tile, clamped) as 32-bit byte offsets: every load below is
    // "uniform base of the stage (SGPRs) + lane offset (one VGPR) + immediate", no per-load address arithmetic
    unsigned wq_off[WN], wd_off[WN];
    size_t wrow[WN];
#pragma unroll
    for (int j = 0; j < WN; j++) {
        const int col = col0 + 16 * j + l16;
        wrow[j] = (size_t)(col < d_out ? col : d_out - 1);
        if (WT == GTEN_Q4) wq_off[j] = (unsigned)wrow[j] * nb * 16 + (g & 1) * 8;
        else wq_off[j] = (unsigned)wrow[j] * nb * 32 + (g >> 1) * nb * 16 + (g & 1) * 8;
        wd_off[j] = (unsigned)wrow[j] * nb * 2;
    }

    // staging role: thread -> (row, pair of quant blocks) of the stage (the first 2 BM threads)
    const bool stager = (int)threadIdx.x < BM * PAIRS;
    const int srow = stager ? threadIdx.x / PAIRS : 0, spair = threadIdx.x % PAIRS;
    const unsigned x_off = (unsigned)((row0 + srow < n ? row0 + srow : n - 1) - start_pos) * (unsigned)x_pitch + spair * (QUANT ? 2 * GTEN_Q8_BYTES : 128);
    const uint8_t* x0 = x + (size_t)start_pos * x_pitch;

    floatx4 acc[WM][WN];
#pragma unroll
    for (int t = 0; t < WM; t++)
#pragma unroll
        for (int j = 0; j < WN; j++) acc[t][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // ---- raw bytes of one stage, in registers
    struct ARaw { unsigned q[17]; uint4 h[8]; };
    struct BRaw { uint2 q[WN][KB]; uint2 d[WN][KB / 4]; };
    auto load_a = [&](int s, ARaw& a) {
        if (!stager) return;
        const int sn = min(s, nstage - 1);                              // past the end: re-reads the last stage (never stored)
        if (QUANT) {
            // two consecutive 34-byte blocks = 68 bytes, 4-byte aligned: [d0 | q0 x32 | d1 | q1 x32]
            const uint8_t* sbase = x0 + (size_t)sn * (KB * GTEN_Q8_BYTES);
#pragma unroll
            for (int i = 0; i < 17; i++) a.q[i] = *(const unsigned*)(sbase + x_off + 4 * i);
        } else {
            const uint8_t* sbase = x0 + (size_t)sn * (KB * 64);
#pragma unroll
            for (int i = 0; i < 8; i++) a.h[i] = *(const uint4*)(sbase + x_off + 16 * i);
        }
    };
    auto store_a = [&](const ARaw& a, int buf) {
        if (!stager) return;
        uint8_t* lds_a = g_smem + buf * C::STAGE;
        float* lds_da = (float*)(lds_a + BM * APITCH);
        uint4* dst = (uint4*)(lds_a + srow * APITCH + spair * 128);
        if (QUANT) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {       // block 0: quants straddle dwords by two bytes
                unsigned f[4];
                int8x4_to_f16(__builtin_amdgcn_alignbit(a.q[i + 1], a.q[i], 16), f[0], f[1]);
                int8x4_to_f16(__builtin_amdgcn_alignbit(a.q[i + 2], a.q[i + 1], 16), f[2], f[3]);
                dst[i >> 1] = make_uint4(f[0], f[1], f[2], f[3]);
            }
#pragma unroll
            for (int i = 0; i < 8; i += 2) {       // block 1: quants are dword aligned
                unsigned f[4];
                int8x4_to_f16(a.q[9 + i], f[0], f[1]);
                int8x4_to_f16(a.q[10 + i], f[2], f[3]);
                dst[4 + (i >> 1)] = make_uint4(f[0], f[1], f[2], f[3]);
            }
            lds_da[(spair * 2) * BM + srow] = h2f((uint16_t)(a.q[0] & 0xffffu));
            lds_da[(spair * 2 + 1) * BM + srow] = h2f((uint16_t)(a.q[8] >> 16));
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) dst[i] = a.h[i];
        }
    };
    // weight bytes: ONE stage-deep register buffer refilled slot by slot -- the bytes of (stage s + 1, block kb)
    // are requested right after those of (s, kb) were expanded, a whole stage of compute ahead of their use
    auto load_bq = [&](int s, int kb, int j) -> uint2 {
        const uint8_t* sbase = pw.qs + (size_t)min(s, nstage - 1) * (KB * 16);
        return *(const uint2*)(sbase + wq_off[j] + kb * 16);
    };
    auto load_bd = [&](int s, int k4, int j) -> uint2 {
        const uint8_t* sbase = (const uint8_t*)pw.ds + (size_t)min(s, nstage - 1) * (KB * 2);
        return *(const uint2*)(sbase + wd_off[j] + k4 * 8);
    };

    ARaw araw;
    BRaw braw;
    load_a(0, araw);
    if (QUANT) {
#pragma unroll
        for (int j = 0; j < WN; j++) {
#pragma unroll
            for (int kb = 0; kb < KB; kb++) braw.q[j][kb] = load_bq(0, kb, j);
#pragma unroll
            for (int k4 = 0; k4 < KB / 4; k4++) braw.d[j][k4] = load_bd(0, k4, j);
        }
    }
    store_a(araw, 0);
    __syncthreads();

    // expand block `kb` of the buffered stage `st` into MFMA fragments and refill its slot with stage st + 1
    // (past the end: an unused re-read -- no branch in the instruction stream)
    half8 bf[WN];
    float dw[WN];
    auto prep_b = [&](int st, int kb) {
#pragma unroll
        for (int j = 0; j < WN; j++) {
            if (QUANT) {
                bf[j] = weight_frag<WT>(braw.q[j][kb], nibble_shift, nib_mask);
                const unsigned dpair = (kb & 2) ? braw.d[j][kb >> 2].y : braw.d[j][kb >> 2].x;
                dw[j] = h2f((uint16_t)((kb & 1) ? (dpair >> 16) : (dpair & 0xffffu)));
                braw.q[j][kb] = load_bq(st + 1, kb, j);
                if ((kb & 3) == 3) braw.d[j][kb >> 2] = load_bd(st + 1, kb >> 2, j);
            } else {
                bf[j] = *(const half8*)((const uint16_t*)w + wrow[j] * d_in + (size_t)min(st * KB + kb, nb - 1) * 32 + g * 8);
            }
        }
    };
    prep_b(0, 0);

    for (int s = 0; s < nstage; s++) {
        const int buf = s & 1;
        const uint8_t* lds_a = g_smem + buf * C::STAGE;
        const float* lds_da = (const float*)(lds_a + BM * APITCH);
        const bool more = s + 1 < nstage;
        load_a(s + 1, araw);                     // (past the end: re-reads the last pair, never stored)
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            // ---- matrix phase: every MFMA of this quant block is issued before any result is touched, so the
            //      rescale below never waits on the matrix pipe (and the SIMD's other wave fills it meanwhile)
            half8 af[WM];
            float4 da4[WM];
#pragma unroll
            for (int t = 0; t < WM; t++) {
                const int trow = wr * 16 * WM + 16 * t;
                af[t] = *(const half8*)(lds_a + (trow + l16) * APITCH + kb * 64 + g * 16);
                if (QUANT) da4[t] = *(const float4*)(lds_da + kb * BM + trow + g * 4);
            }
            floatx4 isum[WM][WN];
            float dwc[WN];
#pragma unroll
            for (int t = 0; t < WM; t++)
#pragma unroll
                for (int j = 0; j < WN; j++) {
                    if (QUANT) {
                        const floatx4 z = {0.f, 0.f, 0.f, 0.f};
                        isum[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t], bf[j], z, 0, 0, 0);   // exact integer block sums
                    } else {
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t], bf[j], acc[t][j], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int j = 0; j < WN; j++) dwc[j] = dw[j];
            __builtin_amdgcn_sched_barrier(0);
            // ---- vector phase: fragments of the next block, then this block's rescale
            if (kb + 1 < KB) prep_b(s, kb + 1); else prep_b(s + 1, 0);
            if (QUANT) {
#pragma unroll
                for (int t = 0; t < WM; t++) {
                    const float da[4] = {da4[t].x, da4[t].y, da4[t].z, da4[t].w};
#pragma unroll
                    for (int j = 0; j < WN; j++)
#pragma unroll
                        for (int i = 0; i < 4; i++)     // dot += isum * da * dw, left to right like the scalar build (gten/ops.h:311);
                            acc[t][j][i] = acc[t][j][i] + (isum[t][j][i] * da[i]) * dwc[j];   // plain f32 on purpose (build.py)
                }
                // the accumulators are pinned here: without it the whole stage's rescale sinks behind its last MFMA
#pragma unroll
                for (int t = 0; t < WM; t++)
#pragma unroll
                    for (int j = 0; j < WN; j++) asm volatile("" : "+v"(acc[t][j]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_a(araw, buf ^ 1);    // last read two barriers ago
        __syncthreads();
    }

    // ---- rows written in the output dtype straight from the accumulators (gten/ops.h:73-96)
#pragma unroll
    for (int t = 0; t < WM; t++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = row0 + wr * 16 * WM + 16 * t + 4 * g + i;
            const bool rok = r < n;
            uint8_t* orow = out + (size_t)(rok ? r : n - 1) * out_pitch;
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

template <int WT, int WM, int WN>
static int launch_cfg(const void* x, size_t x_pitch, const void* w, void* out, int out_dtype, size_t out_pitch,
                      int n, int d_in, int d_out, int start_pos)
{
    using namespace gtr;
    using C = MfmaCfg<WT, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        GTR_CHECK(hipFuncSetAttribute((const void*)k_matmul_mfma<WT, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::smem()));
        attr_set = true;
    }
    const dim3 grid((d_out + C::BN - 1) / C::BN, (n - start_pos + C::BM - 1) / C::BM), block(256);
    GTR_LAUNCH(KT_MATMUL_MFMA, (k_matmul_mfma<WT, WM, WN>), grid, block, C::smem(), (const uint8_t*)x, x_pitch, w, (uint8_t*)out,
               out_dtype, out_pitch, n, d_in, d_out, start_pos);
    return 0;
}

// the largest tile that still gives the chip enough workgroups
template <int WT>
static int launch_wt(const void* x, size_t x_pitch, const void* w, void* out, int out_dtype, size_t out_pitch,
                     int n, int d_in, int d_out, int start_pos)
{
    const int rows = n - start_pos;
    auto wgs = [&](int bm, int bn) { return ((d_out + bn - 1) / bn) * ((rows + bm - 1) / bm); };
    static const int forced = [] { const char* e = std::getenv("GTEN_HIP_MFMA_CFG"); return e ? atoi(e) : 0; }();   // tuning aid: 24 | 22 | 12
#define MF_GO(WM_, WN_) return launch_cfg<WT, WM_, WN_>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos)
    // (<4,4> and <4,2> register tiles do not fit the 256-VGPR budget of the VGPR-destination MFMA form: they spill)
    if (forced == 24) MF_GO(2, 4);
    if (forced == 22) MF_GO(2, 2);
    if (forced == 12) MF_GO(1, 2);
    if (rows > 32 && wgs(64, 128) >= 384) MF_GO(2, 4);
    if (rows > 32) MF_GO(2, 2);
    MF_GO(1, 2);
#undef MF_GO
}

int gten_launch_matmul_mfma(const void* x, int x_dtype, size_t x_pitch, const void* w, int w_dtype,
                            void* out, int out_dtype, size_t out_pitch, int n, int d_in, int d_out, int start_pos)
{
    (void)x_dtype;
    if (w_dtype == GTEN_F16) return launch_wt<GTEN_F16>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    if (w_dtype == GTEN_Q8) return launch_wt<GTEN_Q8>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
    return launch_wt<GTEN_Q4>(x, x_pitch, w, out, out_dtype, out_pitch, n, d_in, d_out, start_pos);
}
