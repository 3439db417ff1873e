// gten_decode_wxp.h: the narrow projections of the f16 configuration's wide decode step (o: K = n_embd, down: K = n_ffn -> n_embd) in EIGHT
// K planes of 64-feature workgroups (round 5) -- part of the single-token decode translation unit: included by gten_decode.hip.
//
// k_dec_mmv_f16 gives a workgroup 16 output features over its K range for all rows of the lane: 90 KB of f16 weights and rows x K range x 2
// bytes of activations for down at 128 rows -- 720 KB, eight times the weights -- on 219 registers, one workgroup per CU, every wave
// requesting nine fragments per step of eight matrix instructions: 26.9 us for 23 MB of weights.  k_dec_wxp_f16<NBK>: a workgroup owns 64
// output features over ONE EIGHTH of K (NBK = K / 256 steps of 32 elements) for all rows:
//   * its weights (64 rows x NBK steps x 64 bytes: 90 KB for down) arrive in one round trip, 16-byte pieces copied into LDS in
//     matrix-operand order (the XOR placement of gten_decode_ffn.h: conflict-free fragment reads);
//   * wave w owns row tile w: its NBK activation fragments (16 bytes per lane and step of its rows, requested ahead of the weights)
//     each feed four matrix instructions, one per feature tile, accumulating across the steps inside the matrix core -- no
//     cross-wave sum;
//   * the f32 sums of the plane leave as they are; the staging launch behind (PRO_RESID) adds the eight planes in order.
// n_embd / 64 x 8 workgroups (256 for TinyLlama), each reading rows x K / 8 x 2 bytes of activations: a quarter of the bytes per CU.
// The sums differ from k_dec_mmv_f16's in the association of the f32 additions only (eight plane sums of NBK steps each, added in plane
// order, instead of two planes of eight wave ranges); EVERY f16 decoder of 16+ sequences takes this kernel for the shape, so lanes of
// 128 rows, 64-sequence and 16-sequence decoders agree bit for bit (tests/test_multiseq_gpu.py, tests/test_ffn_streamed_gpu.py).
// (The q4 / q8 analogue was built and measured in this round and is not in the library: +1.8 % for 1.1 GB more planes per step,
//  tools/experiments/gten_decode_wxp.h; for f16 the same change is 26.9 -> 9 us on the largest launch of the step.)
// Up to three matrices concatenated along the output (q | k | v: every one a multiple of 64 wide, so a workgroup's 64 columns lie in one):
// n_mats, rest.w1 / w2 / d_out1 / d_out2 as in k_dec_mmv_f16; the planes of q | k | v are added by the attention launch (k_dec_attn_hm_f16).
template <int NBK>
__global__ __launch_bounds__(512, 2) void k_dec_wxp_f16(const uint16_t* __restrict__ a_h, const uint16_t* __restrict__ w0, float* __restrict__ out, const int d_in,
                                                        const int d_out0, const int S, const int out_cols, const int plane_floats, const int frt,
                                                        const int n_mats, const MmvRest rest)
{
    int colw = (int)blockIdx.x * 64, d_out = d_out0;
    const uint16_t* w = w0;
    if (n_mats > 1 && colw >= d_out0) {
        colw -= d_out0; w = (const uint16_t*)rest.w1; d_out = rest.d_out1;
        if (n_mats > 2 && colw >= rest.d_out1) { colw -= rest.d_out1; w = (const uint16_t*)rest.w2; d_out = rest.d_out2; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int kb0 = (int)blockIdx.y * NBK;                        // this plane's first step of 32 elements
    uint8_t* xb = g_smem;                                         // [NBK][4 tiles][4 k-groups][16 columns][16 B]

    // ---- this wave's activation fragments: rows 16 wid + l16 (frt tiles were staged), elements 32 step + 8 g .. + 7
    const bool has_rows = wid < frt;                              // (uniform per wave)
    uint4 aw[NBK];
    if (has_rows) {
        const uint16_t* arow = a_h + (size_t)(16 * wid + l16) * d_in + (size_t)kb0 * 32 + 8 * g;
#pragma unroll
        for (int k = 0; k < NBK; k++) aw[k] = *(const uint4*)(arow + k * 32);
    }
    // ---- the weights: piece p = (row p / (4 NBK) of the 64, step, k-group): a row's 4 NBK pieces are one contiguous run of 64 NBK bytes
    constexpr int NPC = 64 * NBK * 4, PPT = (NPC + 511) / 512;
    uint4 raw[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int p = min((int)threadIdx.x + 512 * j, NPC - 1), rr = p / (4 * NBK), wi = p - rr * (4 * NBK);
        const size_t wrow = (size_t)min(colw + rr, d_out - 1);
        // (w may come out of the argument struct: say that it is global memory, or the loads are FLAT ones -- which may alias LDS, so
        //  every piece would have to land before the first ds_write: the pieces went through scratch)
        typedef int wxp_v4i __attribute__((ext_vector_type(4)));
        typedef const wxp_v4i __attribute__((address_space(1)))* gmem_w4;
        const wxp_v4i v = *(gmem_w4)(uintptr_t)(w + wrow * d_in + (size_t)kb0 * 32 + wi * 8);
        raw[j] = make_uint4((unsigned)v[0], (unsigned)v[1], (unsigned)v[2], (unsigned)v[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int p = (int)threadIdx.x + 512 * j;
        if (p < NPC) {
            const int rr = p / (4 * NBK), wi = p - rr * (4 * NBK), bc = wi >> 2, gg = wi & 3, f_own = rr >> 4, sr = rr & 15;
            *(uint4*)(xb + (size_t)(bc * 4 + f_own) * 1024 + (size_t)gg * 256 + (size_t)((sr ^ bc) & 15) * 16) = raw[j];
        }
    }
    __syncthreads();
    if (!has_rows) return;

    mmvh_f4 acc[4];
#pragma unroll
    for (int f = 0; f < 4; f++) acc[f] = (mmvh_f4){0.f, 0.f, 0.f, 0.f};
    const uint8_t* rb = xb + (size_t)g * 256;
#pragma unroll
    for (int k = 0; k < NBK; k++) {
        mmvh_h8 ah;
        __builtin_memcpy(&ah, &aw[k], 16);
#pragma unroll
        for (int f = 0; f < 4; f++) {
            mmvh_h8 bh;
            const uint4 b = *(const uint4*)(rb + (size_t)(k * 4 + f) * 1024 + (size_t)((l16 ^ k) & 15) * 16);
            __builtin_memcpy(&bh, &b, 16);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[f], 0, 0, 0);
        }
    }
    // lane (l16, g): rows 16 wid + 4 g + i, columns 64 blockIdx.x + 16 f + l16 of plane blockIdx.y
    float* po = out + (size_t)blockIdx.y * plane_floats;
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const int cm = colw + 16 * f + l16, col = (int)blockIdx.x * 64 + 16 * f + l16;     // column inside its matrix / of the concatenated row
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 16 * wid + 4 * g + i;
            if (r < S && cm < d_out) po[(size_t)r * out_cols + col] = acc[f][i];
        }
    }
}
