// EXPERIMENT, not part of the library (round 5; HISTORY.md, round 5: built, every equality and band test green, +1.8 % at 256 sequences for
// +1.1 GB of f32 planes per step: not kept).  Kept here as the starting point of the wide W.x rework DESIGN.md section 7 describes.
// gten_decode_wxp.h: the narrow projections of a wide decode step (down: K = n_ffn -> n_embd) in EIGHT K planes, q4 weights (round 5) --
// part of the single-token decode translation unit: included by gten_decode.hip.
//
// What bounds k_dec_mmvh on these launches is not their weights: a workgroup owns 16 output features over its K range for all rows
// of the lane -- 25 KB of weights and rows x K range x 2 bytes of activation fragments (0.72 MB for down in two planes at 128 rows).
// 256 such workgroups pull 184 MB per launch through the L2s, 0.72 MB per CU at ~85 GB/s: that is the 8.3 us of the launch.
// k_dec_wxp_q4<NBK>: a workgroup owns 64 output features over ONE EIGHTH of K (NBK = n_ffn / 256 quant blocks) for all rows:
//   * its weights (64 rows x NBK blocks: 25 KB of nibbles + deltas for down) arrive in one round trip, a 16-byte piece per thread
//     and step, and are expanded ONCE to f16((n - 7) dw) in matrix-operand order into LDS (NBK x 4 KB, the XOR placement of
//     gten_decode_ffn.h: conflict-free fragment reads);
//   * wave w owns row tile w (16 sequences): its NBK activation fragments (16 bytes per lane and block, requested before the
//     weights so that they fly behind the expansion) each feed four matrix instructions, one per feature tile, accumulating
//     across the blocks inside the matrix core -- no cross-wave sum, no LDS for the activations;
//   * the f32 sums of the plane leave as they are; the staging launch behind (PRO_RESID) adds the eight planes in order;
//   * blockIdx.x = plane: workgroups are dealt to the eight XCDs round-robin, so an XCD's workgroups all read the SAME eighth of
//     the activation fragments -- one L2 fetches it once, instead of every L2 fetching all of them.
// n_embd / 64 x 8 workgroups (256 for TinyLlama), each reading rows x K / 8 x 2 bytes of fragments: 46 MB per launch instead of
// 184.  The sums differ from k_dec_mmvh's in the association of the f32 additions only (eight plane sums of NBK blocks each, added
// in plane order, instead of two planes of eight wave slices) -- EVERY decoder of 16+ sequences takes this kernel for the shape, so
// lanes of 128 rows, 64-sequence decoders and 16-sequence decoders still agree bit for bit (tests/test_multiseq_gpu.py).
template <int NBK>
__global__ __launch_bounds__(512, 2) void k_dec_wxp_q4(const uint16_t* __restrict__ a_ah, const void* __restrict__ w, float* __restrict__ out, const int d_in,
                                                       const int d_out, const int S, const int out_cols, const int plane_floats, const int frt, const int plane_x)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int bplane = plane_x ? (int)blockIdx.x : (int)blockIdx.y, bgroup = plane_x ? (int)blockIdx.y : (int)blockIdx.x;
    const int nb = d_in >> 5, kb0 = bplane * NBK;        // quant blocks per weight row; this plane's first block
    uint8_t* xb = g_smem;                                         // [NBK][4 tiles][4 k-groups][16 columns][16 B]
    const PackedW pw = packed_view(w, GTEN_Q4, d_out, d_in);

    // ---- this wave's activation fragments: row tile `wid` (frt tiles were staged), 1 KB per quant block
    const bool has_rows = wid < frt;                              // (uniform per wave)
    uint4 aw[NBK];
    if (has_rows) {
        const uint8_t* ab = (const uint8_t*)a_ah + ((size_t)kb0 * frt + wid) * 1024 + (size_t)lane * 16;
#pragma unroll
        for (int k = 0; k < NBK; k++) aw[k] = *(const uint4*)(ab + (size_t)k * frt * 1024);
    }
    // ---- the weights: piece p = (row p / NBK of the 64, block p % NBK), a row's NBK pieces are one contiguous run
    constexpr int NPC = 64 * NBK, PPT = (NPC + 511) / 512;
    uint4 raw[PPT];
    unsigned rawd[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int p = min((int)threadIdx.x + 512 * j, NPC - 1), rr = p / NBK, bc = p - rr * NBK;
        const unsigned wrow = (unsigned)min(bgroup * 64 + rr, d_out - 1);
        const unsigned bi = wrow * (unsigned)nb + (unsigned)(kb0 + bc);
        raw[j] = *(const uint4*)(pw.qs + (size_t)bi * 16);
        rawd[j] = *(const uint16_t*)((const uint8_t*)pw.ds + (size_t)bi * 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    // f16((n - 7) dw) of a piece's 32 elements as the four 16-byte fragments k-group 0 .. 3 reads (k_dec_mmvh's element order)
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int p = (int)threadIdx.x + 512 * j;
        if (p < NPC) {
            const int rr = p / NBK, bc = p - rr * NBK, f_own = rr >> 4, sr = rr & 15;
            const unsigned d2 = rawd[j] | (rawd[j] << 16);
            const unsigned src[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
            uint8_t* base = xb + (size_t)(bc * 4 + f_own) * 1024 + (size_t)((sr ^ bc) & 15) * 16;
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                const int nshift = (gg < 2) ? 4 : 0;
                const unsigned x = src[(gg & 1) * 2] >> nshift, y = src[(gg & 1) * 2 + 1] >> nshift;
                uint4 u;
                u.x = mmvh_scale((x & 0x000f000fu) | 0x64006400u, 1031.0f, d2);
                u.y = mmvh_scale(((x >> 8) & 0x000f000fu) | 0x64006400u, 1031.0f, d2);
                u.z = mmvh_scale((y & 0x000f000fu) | 0x64006400u, 1031.0f, d2);
                u.w = mmvh_scale(((y >> 8) & 0x000f000fu) | 0x64006400u, 1031.0f, d2);
                *(uint4*)(base + gg * 256) = u;
            }
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
    // lane (l16, g): rows 16 wid + 4 g + i, columns 64 blockIdx.y + 16 f + l16 of plane blockIdx.x
    float* po = out + (size_t)bplane * plane_floats;
#pragma unroll
    for (int f = 0; f < 4; f++) {
        const int col = bgroup * 64 + 16 * f + l16;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = 16 * wid + 4 * g + i;
            if (r < S && col < d_out) po[(size_t)r * out_cols + col] = acc[f][i];
        }
    }
}
