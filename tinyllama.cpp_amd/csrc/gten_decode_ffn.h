// gten_decode_ffn.h: gate | up of a 128-row lane with the slice's silu . mul chain, q4 weights, K = 2048 (round 5) -- part of the
// single-token decode translation unit: included by gten_decode.hip.
//
// k_dec_mmvh<Q4, 8, 4, true> (gten_decode_wide_wx.h) runs this launch in three phases that every one of its 176 workgroups goes
// through in lockstep: the whole 72 KB weight slab arrives (4.7 us of the launch's 15), eight waves each multiply ONE eighth of K
// for all 128 rows (4.6 us: at the matrix pipe's rate for the 176 CUs that have a workgroup), and the eight partial sums of every
// output cross LDS to be added before the silu . mul chain can start (5 us).  k_dec_ffn_q4 computes THE SAME SUMS -- bit for bit:
// the same f16 operands (mmvh_scale), the same eight K slices each accumulated inside the matrix core from zero, added in the same
// order -- organised the other way round:
//   * a wave owns ONE ROW TILE (16 sequences) and all 64 weight rows of the slice over the WHOLE K: the eight slice sums of an
//     output are formed one after the other in the same lane and added as they come -- no cross-wave sum, and the silu . mul
//     chain runs on the accumulators (a row's 32 features of a Q8 block sit in one 16-lane row of the wave x two feature
//     tiles: the block maxima are DPP row steps);
//   * the weights are STREAMED: one K slice (8 quant blocks x 64 rows = 8 KB of nibbles + deltas) is one 16-byte piece per thread,
//     expanded ONCE per workgroup to f16((n - 7) dw) in matrix-operand order into a double-buffered LDS chunk (32 KB), two slices
//     ahead in flight -- the first matrix instruction waits for 8 KB, not for 72;
//   * an activation fragment of the wave's row tile (straight from L2, requested a whole slice ahead into the register its
//     predecessor just left) feeds four matrix instructions, one per feature tile, each against a conflict-free ds_read_b128.
// One barrier per slice.  Selected for full 128-row lanes at d_in = 2048; every other shape keeps k_dec_mmvh (the same bits).
// Measured (256 sequences = two lanes, per launch / per step): 15.1 us / 3.08-3.12 ms -> 12.2 us / 3.05 ms.  With the kernel cut to
// 128 registers so that two workgroups share a CU (hipcc spills 48 of them): 17.5 us -- not kept.  What is left of the launch:
// ~2.5 us until the first slice is there, ~4.5 us of matrix instructions on the 176 CUs that have a workgroup, ~2.2 us of the
// silu . mul chain (eight outputs per lane x ~75 dependent instructions: the reference's expf and division and four Q8 roundings).
#ifndef FFN_OCC
#define FFN_OCC 2             // waves per SIMD the register budget is set for (4 = two workgroups per CU within 128 registers: 48 spilled, 17.5 against 12.2 us)
#endif

__device__ __forceinline__ float ffn_row16_max(float v)         // maximum over the 16 lanes of a row (non-negative values)
{
    return row16_absmax(v);
}

// SILU: the FFN slice (32 gate + 32 up rows, the chain in the epilogue, f16 fragments out).  !SILU: 64 consecutive rows of ONE
// matrix (w_gate; the lm_head), the raw f32 sums out (out_frag = float rows of out_cols floats) -- k_dec_mmvh<Q4, 8, 4, false>'s bits.
template <bool SILU>
__global__ __launch_bounds__(512, FFN_OCC) void k_dec_ffn_q4(const uint16_t* __restrict__ a_ah, const void* __restrict__ w_gate, const void* __restrict__ w_up,
                                                       uint16_t* __restrict__ out_frag, const int d_in, const int n_ffn, const int S, const int out_cols)
{
    constexpr int RT = 8, NBW = 8;                                // row tiles of the lane; quant blocks per K slice (d_in = 2048: 64 blocks = 8 slices)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int nb = d_in >> 5;
    uint8_t* xb = g_smem;                                         // [2][NBW][4 tiles][64 lanes][16 B]: expanded weight fragments of a slice

    // ---- this thread's piece of every slice: weight row rr (0 .. 63: tiles 0, 1 = gate rows, 2, 3 = up rows of the slice), block bc
    const int rr = threadIdx.x >> 3, bc = threadIdx.x & 7, f_own = rr >> 4, sr = rr & 15;
    const PackedW pw = packed_view((!SILU || f_own < 2) ? w_gate : w_up, GTEN_Q4, n_ffn, d_in);
    const size_t wrow = SILU ? (size_t)blockIdx.x * 32 + 16 * (f_own & 1) + sr : (size_t)min((int)blockIdx.x * 64 + rr, n_ffn - 1);
    // (32-bit offsets from the matrices' uniform bases: one address register per request instead of two)
    const unsigned qoff = (unsigned)wrow * (unsigned)nb * 16u + (unsigned)bc * 16u, doff = ((unsigned)wrow * (unsigned)nb + (unsigned)bc) * 2u;
    uint4 raw[2];
    unsigned rawd[2];
    auto request_w = [&](int c, int slot) {
        raw[slot] = *(const uint4*)(pw.qs + (qoff + (unsigned)(c * NBW * 16)));
        rawd[slot] = *(const uint16_t*)((const uint8_t*)pw.ds + (doff + (unsigned)(c * NBW * 2)));
    };
    // f16((n - 7) dw) of the piece's 32 elements, as the four 16-byte fragments k-group g = 0 .. 3 reads (elements 8 g .. 8 g + 7 in
    // the order 0 2 1 3 4 6 5 7: k_dec_mmvh's), into buffer `buf`: fragment (block bc, tile f_own, k-group g, column sr) at slot
    // sr ^ bc of its 16-slot row (the XOR spreads the eight blocks a wave writes at once over the banks; a reader undoes it)
    auto expand = [&](int slot, int buf) {
        const unsigned d2 = rawd[slot] | (rawd[slot] << 16);
        const unsigned src[4] = {raw[slot].x, raw[slot].y, raw[slot].z, raw[slot].w};
        uint8_t* base = xb + (size_t)buf * (NBW * 4 * 1024) + (size_t)(bc * 4 + f_own) * 1024 + (size_t)((sr ^ bc) & 15) * 16;
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
    };
    // ---- this wave's activation fragments: row tile `wid`, 1 KB per quant block
    const unsigned aoff = (unsigned)lane * 16u + (unsigned)wid * 1024u;      // bytes
    // (ONE set of eight fragment registers: a block's fragment is requested again -- for the next slice -- as soon as its four
    //  matrix instructions are issued, so the kernel stays within 128 registers and two workgroups share a CU: the two lanes' launches,
    //  176 workgroups each, then run side by side instead of in two rounds)
    uint4 aw[NBW];
    request_w(0, 0);
#pragma unroll
    for (int k = 0; k < NBW; k++) aw[k] = *(const uint4*)((const uint8_t*)a_ah + (aoff + (unsigned)(k * RT * 1024)));
    request_w(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    expand(0, 0);
    __syncthreads();

    mmvh_f4 total[4];
#pragma unroll
    for (int f = 0; f < 4; f++) total[f] = (mmvh_f4){0.f, 0.f, 0.f, 0.f};
    constexpr int n_slices = 8;                                   // (d_in = 2048: the launcher selects this kernel for that width only)
#pragma unroll
    for (int c = 0; c < n_slices; c++) {                          // (fully unrolled: the two register sets are picked at compile time)
        const int cur = c & 1;
        if (c + 1 < n_slices) {
            expand(cur ^ 1, cur ^ 1);                             // slice c + 1 -> the buffer slice c - 1 was read from (every wave is past the barrier behind it)
            if (c + 2 < n_slices) request_w(c + 2, cur);
        }
        mmvh_f4 acc[4];
#pragma unroll
        for (int f = 0; f < 4; f++) acc[f] = (mmvh_f4){0.f, 0.f, 0.f, 0.f};
        const uint8_t* rb = xb + (size_t)cur * (NBW * 4 * 1024) + (size_t)g * 256;
#pragma unroll
        for (int k = 0; k < NBW; k++) {
            mmvh_h8 ah;
            __builtin_memcpy(&ah, &aw[k], 16);
#pragma unroll
            for (int f = 0; f < 4; f++) {
                mmvh_h8 bh;
                const uint4 b = *(const uint4*)(rb + (size_t)(k * 4 + f) * 1024 + (size_t)((l16 ^ k) & 15) * 16);
                __builtin_memcpy(&bh, &b, 16);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[f], 0, 0, 0);
            }
            if (c + 1 < n_slices) aw[k] = *(const uint4*)((const uint8_t*)a_ah + (aoff + (unsigned)(((c + 1) * NBW + k) * RT * 1024)));
        }
        // the slice sums join the running totals in slice order: ((((0 + s0) + s1) + ...) -- k_dec_mmvh's wave order
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int i = 0; i < 4; i++) total[f][i] = total[f][i] + acc[f][i];
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);                        // (nothing of the next slice is scheduled into this one: registers)
    }

    if (!SILU) {
        // lane (l16, g): rows 16 wid + 4 g + i, columns 64 blockIdx.x + 16 f + l16
        float* out = (float*)out_frag;
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int col = (int)blockIdx.x * 64 + 16 * f + l16;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 16 * wid + 4 * g + i;
                if (r < S && col < n_ffn) out[(size_t)r * out_cols + col] = total[f][i];
            }
        }
        return;
    }
    // ---- the slice's silu(gate) * up chain with every rounding the modules make (k_dec_silumul_rows' arithmetic), on the
    //      accumulators: lane (l16, g) holds rows 16 wid + 4 g + i, features l16 (tiles 0 | 2) and 16 + l16 (tiles 1 | 3); a row's
    //      Q8 block = its 16-lane row x the two tiles
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float gv[2], uv[2], v[2];
        {
            const Q8Scale s = q8_scale_from_absmax(ffn_row16_max(fmaxf(fabsf(total[0][i]), fabsf(total[1][i]))));     // gate projection written
            gv[0] = (float)q8_round(total[0][i], s.scale) * s.ddeq; gv[1] = (float)q8_round(total[1][i], s.scale) * s.ddeq;
        }
        {
            const float s0 = gv[0] / (1.0f + expf(-gv[0])), s1 = gv[1] / (1.0f + expf(-gv[1]));                         // silu in place
            const Q8Scale s = q8_scale_from_absmax(ffn_row16_max(fmaxf(fabsf(s0), fabsf(s1))));
            gv[0] = (float)q8_round(s0, s.scale) * s.ddeq; gv[1] = (float)q8_round(s1, s.scale) * s.ddeq;
        }
        {
            const Q8Scale s = q8_scale_from_absmax(ffn_row16_max(fmaxf(fabsf(total[2][i]), fabsf(total[3][i]))));     // up projection written
            uv[0] = (float)q8_round(total[2][i], s.scale) * s.ddeq; uv[1] = (float)q8_round(total[3][i], s.scale) * s.ddeq;
        }
        v[0] = gv[0] * uv[0]; v[1] = gv[1] * uv[1];                                                                      // mul in place, then written:
        const Q8Scale sc = q8_scale_from_absmax(ffn_row16_max(fmaxf(fabsf(v[0]), fabsf(v[1]))));
        const int r = 16 * wid + 4 * g + i;
        if (r < S) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = 16 * h + l16, kp = (k & ~3) | ((k & 1) << 1) | ((k >> 1) & 1);
                const int qv = q8_round(v[h], sc.scale);
                out_frag[(((size_t)blockIdx.x * RT + wid) * 64 + (kp >> 3) * 16 + (r & 15)) * 8 + (kp & 7)] = f2h((float)qv * sc.ddeq);
            }
        }
    }
}
