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
#include <type_traits>
#ifndef FFN_OCC
#define FFN_OCC 4             // waves per SIMD the register budget is set for: 128 registers, two workgroups per CU (the K loop rolled: no spill)
#endif

__device__ __forceinline__ float ffn_row16_max(float v)         // maximum over the 16 lanes of a row (non-negative values)
{
    return row16_absmax(v);
}

// SILU: the FFN slice (32 gate + 32 up rows, the chain in the epilogue, f16 fragments out).  !SILU: 64 consecutive rows of ONE
// matrix (w_gate; the lm_head), the raw f32 sums out (out_frag = float rows of out_cols floats) -- k_dec_mmvh<Q4, 8, 4, false>'s bits.
// PART: a lane of 1 .. 4 row tiles (16 .. 64 sequences; `frt` tiles were staged): waves frt .. 7 expand weights and keep the barriers, the
// matrix work is the first frt waves' -- the same sums as k_dec_mmvh<Q4, frt, 4, ..> (whose eight waves split K into the same eight slices).
template <bool SILU, bool PART = false>
__global__ __launch_bounds__(512, FFN_OCC) void k_dec_ffn_q4(const uint16_t* __restrict__ a_ah, const void* __restrict__ w_gate, const void* __restrict__ w_up,
                                                       uint16_t* __restrict__ out_frag, const int d_in, const int n_ffn, const int S, const int out_cols, const int frt)
{
    constexpr int NBW = 8;                                        // quant blocks per K slice (d_in = 2048: 64 blocks = 8 slices)
    const int RT = PART ? frt : 8;                                // row tiles of the lane = the fragment stride of the staging
    const bool rows = !PART || (int)(threadIdx.x >> 6) < frt;     // (uniform per wave) this wave owns a row tile
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
    if (rows) {
#pragma unroll
        for (int k = 0; k < NBW; k++) aw[k] = *(const uint4*)((const uint8_t*)a_ah + (aoff + (unsigned)(k * RT * 1024)));
    }
    request_w(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    expand(0, 0);
    __syncthreads();

    mmvh_f4 total[4];
#pragma unroll
    for (int f = 0; f < 4; f++) total[f] = (mmvh_f4){0.f, 0.f, 0.f, 0.f};
    constexpr int n_slices = 8;                                   // (d_in = 2048: the launcher selects this kernel for that width only)
    // one slice; MORE1 / MORE2: a slice / two slices follow (compile-time: no branch around a request); c may be a run-time value
    auto slice = [&](const int c, const int cur, auto more1, auto more2) {
        constexpr bool MORE1 = decltype(more1)::value, MORE2 = decltype(more2)::value;
        if (MORE1) {
            expand(cur ^ 1, cur ^ 1);                             // slice c + 1 -> the buffer slice c - 1 was read from (every wave is past the barrier behind it)
            if (MORE2) request_w(c + 2, cur);
        }
        if (rows) {
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
            if (MORE1) aw[k] = *(const uint4*)((const uint8_t*)a_ah + (aoff + (unsigned)(((c + 1) * NBW + k) * RT * 1024)));
        }
        // the slice sums join the running totals in slice order: ((((0 + s0) + s1) + ...) -- k_dec_mmvh's wave order
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int i = 0; i < 4; i++) total[f][i] = total[f][i] + acc[f][i];
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);                        // (nothing of the next slice is scheduled into this one: registers)
    };
    // slices in pairs (the two LDS buffers and weight-register slots alternate) as a ROLLED loop, the last pair peeled: unrolled eight
    // times hipcc kept 180 registers live across the slices, rolled it needs two thirds of that -- two workgroups per CU
    {
        using Y = std::true_type;
        using N = std::false_type;
#pragma unroll 1
        for (int c2 = 0; c2 < n_slices - 2; c2 += 2) {
            slice(c2, 0, Y{}, Y{});
            slice(c2 + 1, 1, Y{}, Y{});
        }
        slice(n_slices - 2, 0, Y{}, N{});
        slice(n_slices - 1, 1, N{}, N{});
    }

    if (!rows) return;
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

// ---------------------------------------------------------------------------------------------------------------------------
// The same for Q8 WEIGHTS (round 5; SILU: gate | up + chain, !SILU: the lm_head as k_dec_mmvh<Q8, 8, 2, false> at ks = 1 sums it).  A Q8 slice's slab (64 rows x 64 blocks x 32 B + deltas = 139 KB, + the cross-wave sums) does not fit
// LDS, so the q8 configuration's wide decoders ran gate | up as k_dec_mmvh<Q8, 8, 2, false> in two K planes (352 x 2 workgroups) and
// the silu . mul chain as a second launch (k_dec_silumul_rows) on the raw sums: 24 us per block and lane at 128 rows against q4's 12.
// Streamed, nothing has to fit: a K slice is FOUR quant blocks x 64 rows = 512 pieces of 16 bytes (one per thread: row, block, half of
// the block's 32 quants), expanded to f16(q dw) into a double-buffered 16 KB chunk.  THE SAME SUMS as the pair it replaces, bit for
// bit: sixteen slices of four blocks -- the eight wave slices of k_dec_mmvh's first plane, then those of the second -- each
// accumulated in the matrix core from zero, added in order inside their plane, the two plane sums added as k_dec_silumul_rows adds
// them, then that kernel's chain on the accumulators (tests/test_ffn_streamed_gpu.py).
template <bool SILU, bool PART = false>
__global__ __launch_bounds__(512, FFN_OCC) void k_dec_ffn_q8(const uint16_t* __restrict__ a_ah, const void* __restrict__ w_gate, const void* __restrict__ w_up,
                                                       uint16_t* __restrict__ out_frag, const int d_in, const int n_ffn, const int S, const int out_cols, const int frt)
{
    const int RT = PART ? frt : 8;                                // row tiles of the lane = the fragment stride of the staging (PART: as k_dec_ffn_q4)
    const bool rows = !PART || (int)(threadIdx.x >> 6) < frt;     // (uniform per wave) this wave owns a row tile
    // SILU: slices of 4 blocks (two K planes of eight wave slices: k_dec_mmvh<Q8, 8, 2, false> at ks = 2), one piece per thread.
    // !SILU (the lm_head: 64 consecutive rows of ONE matrix, raw f32 sums out): slices of 8 blocks (ONE plane of eight wave slices,
    // ks = 1), two pieces per thread.
    constexpr int NBW = SILU ? 4 : 8, PPT = SILU ? 1 : 2;         // quant blocks per K slice; 16-byte pieces per thread and slice
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int nb = d_in >> 5;
    uint8_t* xb = g_smem;                                         // [2][NBW][4 tiles][64 lanes][16 B]

    // ---- this thread's pieces of every slice: piece = (weight row rr of the 64, block bc of the slice, half hf of the block's 32 quants)
    unsigned qoff[PPT], doff[PPT];
    const PackedW pwg = packed_view(w_gate, GTEN_Q8, n_ffn, d_in), pwu = packed_view(SILU ? w_up : w_gate, GTEN_Q8, n_ffn, d_in);
    const bool up_rows = SILU && (threadIdx.x >> 8) != 0;         // (SILU: pieces of tiles 2, 3 -- rows 32 .. 63 of the 64 -- come from w_up)
    const uint8_t* qbase = up_rows ? pwu.qs : pwg.qs;
    const uint8_t* dbase = (const uint8_t*)(up_rows ? pwu.ds : pwg.ds);
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int p = (int)threadIdx.x + 512 * j;
        const int rr = p / (2 * NBW), bc = (p >> 1) % NBW, hf = p & 1, f_own = rr >> 4, sr = rr & 15;
        const size_t wrow = SILU ? (size_t)blockIdx.x * 32 + 16 * (f_own & 1) + sr : (size_t)min((int)blockIdx.x * 64 + rr, n_ffn - 1);
        // a packed Q8 row: the first 16 quants of every block (nb x 16 B), then the last 16 (include/gten_hip.h); deltas apart
        qoff[j] = (unsigned)wrow * (unsigned)nb * 32u + (unsigned)hf * (unsigned)nb * 16u + (unsigned)bc * 16u;
        doff[j] = ((unsigned)wrow * (unsigned)nb + (unsigned)bc) * 2u;
    }
    uint4 raw[2][PPT];
    unsigned rawd[2][PPT];
    auto request_w = [&](int c, int slot) {
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            raw[slot][j] = *(const uint4*)(qbase + (qoff[j] + (unsigned)(c * NBW * 16)));
            rawd[slot][j] = *(const uint16_t*)(dbase + (doff[j] + (unsigned)(c * NBW * 2)));
        }
    };
    // f16(q dw) of a piece's 16 quants = k-groups 2 hf and 2 hf + 1 of the block's fragments (elements in the order 0 2 1 3 4 6 5 7:
    // k_dec_mmvh's), placed as in k_dec_ffn_q4
    auto expand = [&](int slot, int buf) {
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            const unsigned d2 = rawd[slot][j] | (rawd[slot][j] << 16);
            const unsigned src[4] = {raw[slot][j].x, raw[slot][j].y, raw[slot][j].z, raw[slot][j].w};
            const int p = (int)threadIdx.x + 512 * j, rr = p / (2 * NBW), bc = (p >> 1) % NBW, hf = p & 1, f_own = rr >> 4, sr = rr & 15;
            uint8_t* base = xb + (size_t)buf * (NBW * 4 * 1024) + (size_t)(bc * 4 + f_own) * 1024 + (size_t)((sr ^ bc) & 15) * 16 + (size_t)hf * 512;
#pragma unroll
            for (int gg = 0; gg < 2; gg++) {
                const unsigned x = src[gg * 2], y = src[gg * 2 + 1];
                uint4 u;
                u.x = mmvh_scale((x & 0x00ff00ffu) ^ 0x64806480u, 1152.0f, d2);
                u.y = mmvh_scale(((x >> 8) & 0x00ff00ffu) ^ 0x64806480u, 1152.0f, d2);
                u.z = mmvh_scale((y & 0x00ff00ffu) ^ 0x64806480u, 1152.0f, d2);
                u.w = mmvh_scale(((y >> 8) & 0x00ff00ffu) ^ 0x64806480u, 1152.0f, d2);
                *(uint4*)(base + gg * 256) = u;
            }
        }
    };
    const unsigned aoff = (unsigned)lane * 16u + (unsigned)wid * 1024u;      // this wave's activation fragments: row tile `wid`, 1 KB per block
    uint4 aw[NBW];
    request_w(0, 0);
    if (rows) {
#pragma unroll
        for (int k = 0; k < NBW; k++) aw[k] = *(const uint4*)((const uint8_t*)a_ah + (aoff + (unsigned)(k * RT * 1024)));
    }
    request_w(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    expand(0, 0);
    __syncthreads();

    mmvh_f4 total[2][4], tot[4];                                  // [K plane of k_dec_mmvh][feature tile]; the running plane
#pragma unroll
    for (int f = 0; f < 4; f++) tot[f] = (mmvh_f4){0.f, 0.f, 0.f, 0.f};
    constexpr int n_slices = 64 / NBW;                            // (d_in = 2048: the launcher selects this kernel for that width only)
    // one slice: LAST = nothing behind it to expand / request (compile-time: no branch around a request); c may be a run-time value
    auto slice = [&](const int c, const int cur, auto more1, auto more2) {
        constexpr bool MORE1 = decltype(more1)::value, MORE2 = decltype(more2)::value;
        if (MORE1) {
            expand(cur ^ 1, cur ^ 1);                             // slice c + 1 -> the buffer slice c - 1 was read from
            if (MORE2) request_w(c + 2, cur);
        }
        if (rows) {
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
            if (MORE1) aw[k] = *(const uint4*)((const uint8_t*)a_ah + (aoff + (unsigned)(((c + 1) * NBW + k) * RT * 1024)));
        }
        // the slice sums join their PLANE's running total in slice order (k_dec_mmvh's wave order inside a K plane)
#pragma unroll
        for (int f = 0; f < 4; f++)
#pragma unroll
            for (int i = 0; i < 4; i++) tot[f][i] = tot[f][i] + acc[f][i];
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    using T = std::true_type;
    using F = std::false_type;
    // slices 0 .. 13 in pairs (the two LDS buffers and register slots alternate), 14 and 15 peeled; the plane changes behind slice 7
#pragma unroll 1
    for (int c2 = 0; c2 < n_slices - 2; c2 += 2) {
        slice(c2, 0, T{}, T{});
        slice(c2 + 1, 1, T{}, T{});
        if (SILU && c2 == 6) {                                    // (the second K plane starts from zero)
#pragma unroll
            for (int f = 0; f < 4; f++) { total[0][f] = tot[f]; tot[f] = (mmvh_f4){0.f, 0.f, 0.f, 0.f}; }
        }
    }
    slice(n_slices - 2, 0, T{}, F{});
    slice(n_slices - 1, 1, F{}, F{});
#pragma unroll
    for (int f = 0; f < 4; f++) total[1][f] = tot[f];
    if (!rows) return;
    if (!SILU) {
        // lane (l16, g): rows 16 wid + 4 g + i, columns 64 blockIdx.x + 16 f + l16 (one K plane: the eight slice sums as they stand)
        float* out = (float*)out_frag;
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int col = (int)blockIdx.x * 64 + 16 * f + l16;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 16 * wid + 4 * g + i;
                if (r < S && col < n_ffn) out[(size_t)r * out_cols + col] = total[1][f][i];
            }
        }
        return;
    }
    // ---- k_dec_silumul_rows on the accumulators: plane 0 + plane 1, then the chain of k_dec_ffn_q4's epilogue
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float tg[2], tu[2], gv[2], uv[2], v[2];
        tg[0] = total[0][0][i] + total[1][0][i]; tg[1] = total[0][1][i] + total[1][1][i];
        tu[0] = total[0][2][i] + total[1][2][i]; tu[1] = total[0][3][i] + total[1][3][i];
        {
            const Q8Scale s = q8_scale_from_absmax(ffn_row16_max(fmaxf(fabsf(tg[0]), fabsf(tg[1]))));                   // gate projection written
            gv[0] = (float)q8_round(tg[0], s.scale) * s.ddeq; gv[1] = (float)q8_round(tg[1], s.scale) * s.ddeq;
        }
        {
            const float s0 = gv[0] / (1.0f + expf(-gv[0])), s1 = gv[1] / (1.0f + expf(-gv[1]));                         // silu in place
            const Q8Scale s = q8_scale_from_absmax(ffn_row16_max(fmaxf(fabsf(s0), fabsf(s1))));
            gv[0] = (float)q8_round(s0, s.scale) * s.ddeq; gv[1] = (float)q8_round(s1, s.scale) * s.ddeq;
        }
        {
            const Q8Scale s = q8_scale_from_absmax(ffn_row16_max(fmaxf(fabsf(tu[0]), fabsf(tu[1]))));                   // up projection written
            uv[0] = (float)q8_round(tu[0], s.scale) * s.ddeq; uv[1] = (float)q8_round(tu[1], s.scale) * s.ddeq;
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

// ---------------------------------------------------------------------------------------------------------------------------
// f16 WEIGHTS x f16 ACTIVATIONS (the f16 configuration's wide decoders run lanes of 64 rows; round 5): gate | up and the silu . mul chain
// as one streamed launch in place of k_dec_mmv_f16 (704 workgroups of 16 features in two K planes, every one of them reading all the
// lane's activation rows) + k_dec_silumul_rows_f16.  Nothing to expand: a K slice is four 32-element steps x 64 weight rows = 1024
// pieces of 16 bytes (two per thread: row, step, k-group -- 256 contiguous bytes per row), COPIED into the double-buffered chunk in
// matrix-operand order; a wave's activation fragment is 16 bytes of its row tile's f16 rows (k_dec_mmv_f16's operands, natural element
// order).  THE SAME SUMS as the pair, bit for bit: sixteen slices of four steps -- the eight wave ranges of k_dec_mmv_f16's first K
// plane, then the second's -- each accumulated in the matrix core from zero, added in order inside their plane; then
// k_dec_silumul_rows_f16's arithmetic per element (plane 0 + plane 1, rounded to f16 where the modules store: gate, silu, up, product).
// `frt` row tiles (1 .. 4: the f16 lanes are 64 rows): waves frt .. 7 copy weights and keep the barriers.
template <bool SILU>
__global__ __launch_bounds__(512, FFN_OCC) void k_dec_ffn_f16(const uint16_t* __restrict__ a_h, const uint16_t* __restrict__ w_gate, const uint16_t* __restrict__ w_up,
                                                        uint16_t* __restrict__ out_h, const int d_in, const int n_ffn, const int S, const int frt, const int out_cols)
{
    // SILU: slices of 4 steps (two K planes of eight wave ranges: k_dec_mmv_f16 at ks = 2), rows 0 .. 31 of the 64 from w_gate, 32 .. 63 from
    // w_up.  !SILU (the lm_head: 64 consecutive rows of ONE matrix, raw f32 sums out): slices of 8 steps (ONE plane, ks = 1).
    constexpr int NBW = SILU ? 4 : 8, PPT = NBW / 2;              // 32-element steps per K slice; 16-byte pieces per thread and slice
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const bool rows = wid < frt;                                  // (uniform per wave) this wave owns a row tile
    uint8_t* xb = g_smem;                                         // [2][NBW][4 tiles][4 k-groups][16 columns][16 B]

    // ---- this thread's pieces of every slice: piece p = threadIdx.x + 512 j = (weight row p / (4 NBW) of the 64, step, k-group): a row's
    //      4 NBW pieces are one contiguous run.  SILU: rows 0 .. 31 are gate rows (tiles 0, 1), 32 .. 63 up rows (tiles 2, 3).
    unsigned woff[PPT], dst[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int p = (int)threadIdx.x + 512 * j, rr = p / (4 * NBW), wi = p % (4 * NBW), bc = wi >> 2, gg = wi & 3, f_own = rr >> 4, sr = rr & 15;
        const size_t wrow = SILU ? (size_t)blockIdx.x * 32 + (rr & 31) : (size_t)min((int)blockIdx.x * 64 + rr, n_ffn - 1);
        woff[j] = (unsigned)((wrow * d_in + bc * 32 + gg * 8) * 2);                      // (bytes from the matrix)
        dst[j] = (unsigned)((bc * 4 + f_own) * 1024 + gg * 256 + ((sr ^ bc) & 15) * 16);
    }
    uint4 rw0[PPT], rw1[PPT];                                     // two register sets (indexed by compile-time constants only)
    auto request_w = [&](int c, auto slot) {
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            // (SILU: pieces j < PPT / 2 lie in rows 0 .. 31 -- 512 j / (4 NBW) -- the others in rows 32 .. 63)
            const uint8_t* wm = (const uint8_t*)((SILU && j >= PPT / 2) ? w_up : w_gate);
            const uint4 v = *(const uint4*)(wm + (woff[j] + (unsigned)(c * (NBW * 64))));
            if constexpr (decltype(slot)::value == 0) rw0[j] = v; else rw1[j] = v;
        }
    };
    auto place = [&](auto slot) {                                 // slot s -> LDS buffer s
        constexpr int SL = decltype(slot)::value;
#pragma unroll
        for (int j = 0; j < PPT; j++) *(uint4*)(xb + (size_t)SL * (NBW * 4 * 1024) + dst[j]) = SL == 0 ? rw0[j] : rw1[j];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // ---- this wave's activation fragments: rows 16 wid + l16, elements 32 step + 8 g .. + 7 (k_dec_mmv_f16's A operand)
    const uint8_t* arow = (const uint8_t*)(a_h + (size_t)(16 * wid + l16) * d_in + 8 * g);
    uint4 aw[NBW];
    request_w(0, I0{});
    if (rows) {
#pragma unroll
        for (int k = 0; k < NBW; k++) aw[k] = *(const uint4*)(arow + (size_t)k * 64);
    }
    request_w(1, I1{});
    __builtin_amdgcn_sched_barrier(0);
    place(I0{});
    __syncthreads();

    mmvh_f4 total[2][4], tot[4];                                  // [K plane of k_dec_mmv_f16][feature tile]; the running plane
#pragma unroll
    for (int f = 0; f < 4; f++) tot[f] = (mmvh_f4){0.f, 0.f, 0.f, 0.f};
    const int n_slices = d_in / (32 * NBW);                       // (2048: sixteen; the launcher requires two planes of eight)
    auto slice = [&](const int c, auto curc, auto more1, auto more2) {
        constexpr bool MORE1 = decltype(more1)::value, MORE2 = decltype(more2)::value;
        constexpr int cur = decltype(curc)::value;
        if (MORE1) {
            place(std::integral_constant<int, cur ^ 1>{});
            if (MORE2) request_w(c + 2, curc);
        }
        if (rows) {
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
                if (MORE1) aw[k] = *(const uint4*)(arow + (size_t)((c + 1) * NBW + k) * 64);
            }
#pragma unroll
            for (int f = 0; f < 4; f++)
#pragma unroll
                for (int i = 0; i < 4; i++) tot[f][i] = tot[f][i] + acc[f][i];
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        using Y = std::true_type;
        using N = std::false_type;
#pragma unroll 1
        for (int c2 = 0; c2 < n_slices - 2; c2 += 2) {
            slice(c2, I0{}, Y{}, Y{});
            slice(c2 + 1, I1{}, Y{}, Y{});
            if (SILU && c2 + 2 == n_slices / 2) {                 // (the second K plane starts from zero)
#pragma unroll
                for (int f = 0; f < 4; f++) { total[0][f] = tot[f]; tot[f] = (mmvh_f4){0.f, 0.f, 0.f, 0.f}; }
            }
        }
        slice(n_slices - 2, I0{}, Y{}, N{});
        slice(n_slices - 1, I1{}, N{}, N{});
    }
    if (!rows) return;
    if (!SILU) {
        // lane (l16, g): rows 16 wid + 4 g + i, columns 64 blockIdx.x + 16 f + l16 (one K plane: the eight range sums as they stand)
        float* out = (float*)out_h;
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const int col = (int)blockIdx.x * 64 + 16 * f + l16;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 16 * wid + 4 * g + i;
                if (r < S && col < n_ffn) out[(size_t)r * out_cols + col] = tot[f][i];
            }
        }
        return;
    }
    // ---- k_dec_silumul_rows_f16 on the accumulators: lane (l16, g) holds rows 16 wid + 4 g + i, feature 32 blockIdx.x + 16 h + l16 as gate
    //      (tile h) and up (tile 2 + h)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = 16 * wid + 4 * g + i;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float gt = h2f(f2h(total[0][h][i] + tot[h][i]));
            gt = h2f(f2h(gt / (1.0f + expf(-gt))));
            const float ut = h2f(f2h(total[0][2 + h][i] + tot[2 + h][i]));
            if (r < S) out_h[(size_t)r * n_ffn + (size_t)blockIdx.x * 32 + 16 * h + l16] = f2h(gt * ut);
        }
    }
}
