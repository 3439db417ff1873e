// gten_decode_wide_wx.h: W.x of 16 .. 128 sequences per lane on the matrix cores (k_dec_mmvh, the folded form that ships; k_dec_mmv,
// the exact form behind gten_hip_set_decode_exact; k_dec_mmv_f16; the silu.mul row kernels) -- part of the single-token decode translation unit: included by gten_decode.hip (which owns the includes, the LDS
// symbol, the launch macros and the host side).  Split out in round 4; the code is unchanged.
// ------------------------------------------- W.x kernel, many sequences (matrix cores)
//
// k_dec_mmv<WT, RT>: the W.x of a decode step for up to 16 RT sequences, GEMV-shaped.  A workgroup owns 16
// output features for ALL rows; its eight waves split K eight ways.
//   * the workgroup's whole weight slab (16 features x K: 16-90 KB) is requested at kernel entry as coalesced
//     16-byte pieces -- ONE memory round trip for all of it, like the single-sequence GEMV kernels -- and parked in
//     LDS (pieces XOR-swizzled by row so that the 16 rows of a fragment read do not share banks);
//   * the activations arrive from the staging launches in MFMA-fragment order (q8_stage_frag): one 512-byte
//     coalesced load per (row tile, quant block) and wave, deltas / block sums as [block][row];
//   * one v_mfma_i32_16x16x32_i8 per (row tile, quant block) = the exact integer dot of 16 rows x 16 features
//     over one 32-wide block (Q4 nibbles are turned into int8 (n - 7) byte-parallel, 5 instructions per dword,
//     shared by the row tiles).  Block sums are scaled (isum * da) * dw and accumulated
//     in block order inside a wave; the eight K slices are added in wave order (deterministic).
// The k_matmul_mfma tiles of gten_mfma.hip run this problem at ~1 us per 128 K of serial chain on 40-350
// workgroups (17 / 41 us for K = 2048 / 5632 at 32 sequences); this shape has d_out / 16 workgroups and a chain of K / 8.
struct MmvArgs {
    const int8_t* aq; const float* ad;                        // fragment-major staging (ActFrag): quants, [block][row] deltas
    const void* w[3]; int d_out[3]; int n_mats;               // concatenated outputs (q|k|v, gate|up): multiples of 16 except the last
    float* out; int out_cols;                                 // raw f32 rows, pitch in floats
    int S, d_in;
    int ks, plane;                                            // K split: slices (0 / 1 = none) and floats between their output planes
};

typedef int mmv_v4i __attribute__((ext_vector_type(4)));
#define MMV_MAXP 11            // 16-byte weight pieces per thread: 16 features x 5632 B (Q8, K = 5632) / 512 threads / 16
#define MMV_MAXD 6             // 16-byte pieces of the activation-delta table per thread: 176 blocks x 64 rows x 4 B / 512 / 16

// FT = 16-feature tiles per workgroup: 1 for the launches that have about one workgroup per CU anyway (q|k|v, o,
// down); 4 (Q4) / 2 (Q8) for gate|up and the lm_head at K = 2048, whose 704 / 2001 sixteen-feature workgroups would
// run in several rounds -- each activation fragment and delta then feeds FT matrix instructions.
// (leading scalar arguments: preloaded into SGPRs by the command processor, see GemvHot; the second and third
// matrix of a concatenated launch travel in the struct behind them)
// K SPLIT: gridDim.y workgroups share a feature tile, each takes nb / gridDim.y consecutive quant blocks (its eight
// waves split THAT range) and writes its own plane of partial sums (plane p at out + p * plane floats); the consumer
// adds the planes in order.  Why: q|k|v, o and down have 128-160 feature tiles -- half the CUs idle, one wave per SIMD
// with nothing to hide its LDS -> MFMA -> rescale latencies behind, and every workgroup reading ALL of the
// activations; two slices put two workgroups on a CU and halve each one's chain and activation traffic.
struct MmvRest { const void* w1; const void* w2; int d_out1, d_out2; int plane; };

template <int WT, int RT, int CB, int FT>
__global__ __launch_bounds__(512) void k_dec_mmv(const int8_t* __restrict__ a_aq, const float* __restrict__ a_ad, const void* __restrict__ a_w0,
                                                 float* __restrict__ a_out, const int a_d_in, const int a_d_out0, const int a_out_cols,
                                                 const int a_S, const int a_n_mats, const MmvRest rest)
{
    constexpr int SP = 16 * RT;                               // padded row count
    constexpr int FR = 16 * FT;                               // features per workgroup
    constexpr int NPF = MMV_MAXP / FT;                        // 16-byte weight pieces per thread and feature tile
    const int nb = a_d_in >> 5;                               // quant blocks of a row
    const int nbs = nb / (int)gridDim.y, b_lo = (int)blockIdx.y * nbs;   // this workgroup's slice of them
    const int nbw = nbs >> 3;                                 // blocks per wave (nbs % 8 == 0)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int rowb = nb * (WT == GTEN_Q4 ? 16 : 32);          // weight bytes per feature (in HBM)
    const int rowl = nbs * (WT == GTEN_Q4 ? 16 : 32);         // ... of this slice (in LDS)
    uint8_t* wl = g_smem;                                     // [FR][rowl], 16-byte pieces swizzled: slot = piece ^ (row & 7)
    float* red = (float*)g_smem;                              // [8][SP][16] -- over the slab, once the K loop is done
    uint16_t* dwl = (uint16_t*)(wl + max((size_t)FR * rowl, (size_t)8 * SP * 64));   // [FR][nbs] weight deltas
    float* daT = (float*)(dwl + (size_t)FR * nbs);            // [nbs][SP] activation deltas

    // which matrix (uniform)
    int colw = blockIdx.x * FR, colbase = 0, m = 0;
    if (a_n_mats > 1 && colw >= a_d_out0) {
        colw -= a_d_out0; colbase = a_d_out0; m = 1;
        if (a_n_mats > 2 && colw >= rest.d_out1) { colw -= rest.d_out1; colbase += rest.d_out1; m = 2; }
    }
    const void* w = (m == 0) ? a_w0 : (m == 1) ? rest.w1 : rest.w2;
    const int d_out = (m == 0) ? a_d_out0 : (m == 1) ? rest.d_out1 : rest.d_out2;
    const PackedW pw = packed_view(w, WT, d_out, a_d_in);

    // ---- 1. everything this workgroup will read, requested at once (one memory round trip):
    //         the weight slab (32 threads per feature row, pieces c0 + 32 k), its deltas, the delta table of the
    //         activations, and this wave's first chunk of activation fragments
    const int ppr = rowl >> 4;                                // pieces per feature row of the slice (<= 32 * NPF)
    const int sr = threadIdx.x >> 5, c0 = threadIdx.x & 31;
    uint4 wp[FT][NPF];                                        // (every slot defined: a conditionally written array is left in scratch memory by hipcc)
#pragma unroll
    for (int f = 0; f < FT; f++)
#pragma unroll
        for (int k = 0; k < NPF; k++) wp[f][k] = make_uint4(0, 0, 0, 0);
    unsigned dwv[FT][3];                                      // nb / 2 <= 88 dwords per row
#pragma unroll
    for (int f = 0; f < FT; f++) {
        const size_t frow = (size_t)min(colw + 16 * f + sr, d_out - 1);
        const uint8_t* srow = pw.qs + frow * rowb;
#pragma unroll
        for (int k = 0; k < NPF; k++)
            if (32 * k < ppr) {
                // local piece -> piece of the row in HBM (Q4: one 16-byte piece per block; Q8: two planes of nb pieces)
                const int lp = min(c0 + 32 * k, ppr - 1);
                const int gp = (WT == GTEN_Q4) ? b_lo + lp : (lp < nbs ? b_lo + lp : nb + b_lo + (lp - nbs));
                wp[f][k] = *(const uint4*)(srow + (size_t)gp * 16);
            }
        const unsigned* drow = (const unsigned*)(pw.ds + frow * nb + b_lo);
#pragma unroll
        for (int k = 0; k < 3; k++) dwv[f][k] = drow[min(c0 + 32 * k, (nbs >> 1) - 1)];
    }
    const int ndp = nbs * SP / 4;                             // 16-byte pieces of the slice of the [nb][SP] delta table (same layout in LDS)
    uint4 dap[MMV_MAXD];
#pragma unroll
    for (int k = 0; k < MMV_MAXD; k++) dap[k] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < MMV_MAXD; k++)
        if (512 * k < ndp) dap[k] = ((const uint4*)(a_ad + (size_t)b_lo * SP))[min((int)threadIdx.x + 512 * k, ndp - 1)];
    const int b0 = b_lo + wid * nbw;
    const int8_t* afr = a_aq + (size_t)lane * 8;              // fragment order: 512 contiguous bytes per (block, row tile)
    uint2 araw[RT][CB];
    auto request = [&](int bb) {
#pragma unroll
        for (int c = 0; c < CB; c++)
#pragma unroll
            for (int t = 0; t < RT; t++) araw[t][c] = *(const uint2*)(afr + ((size_t)min(bb + c, nb - 1) * RT + t) * 512);
    };
    request(b0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. park slab and tables in LDS
#pragma unroll
    for (int f = 0; f < FT; f++) {
        const int r = 16 * f + sr;
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int c = c0 + 32 * k;
            if (c < ppr) *(uint4*)(wl + (size_t)r * rowl + (size_t)(c ^ (r & 7)) * 16) = wp[f][k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int c = c0 + 32 * k;
            if (c < (nbs >> 1)) ((unsigned*)dwl)[r * (nbs >> 1) + c] = dwv[f][k];
        }
    }
#pragma unroll
    for (int k = 0; k < MMV_MAXD; k++) {
        const int p = (int)threadIdx.x + 512 * k;
        if (p < ndp) ((uint4*)daT)[p] = dap[k];
    }
    __syncthreads();

    // ---- 3. this wave's K slice
    const mmv_v4i zero4 = {0, 0, 0, 0};
    const int nshift = (g < 2) ? 4 : 0;
    float acc[FT][RT][4];
#pragma unroll
    for (int f = 0; f < FT; f++)
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[f][t][i] = 0.f;
    const uint8_t* wrow = wl + (size_t)l16 * rowl + (g & 1) * 8;        // feature tile f: + 16 f rows (same swizzle: (16 f + l16) & 7 == l16 & 7)
    for (int bb = b0; bb < b0 + nbw; bb += CB) {
        uint2 aqv[RT][CB];
#pragma unroll
        for (int c = 0; c < CB; c++)
#pragma unroll
            for (int t = 0; t < RT; t++) aqv[t][c] = araw[t][c];
        if (bb + CB < b0 + nbw) request(bb + CB);               // next chunk in flight during this one's math
#pragma unroll
        for (int c = 0; c < CB; c++) {
            const int b = min(bb + c, b_lo + nbs - 1) - b_lo;    // block inside the workgroup's slice (LDS index)
            const bool live = bb + c < b0 + nbw;                 // blocks past this wave's slice (ragged last chunk) are scaled by zero
            long bl[FT];
            float dwf[FT];
#pragma unroll
            for (int f = 0; f < FT; f++) {
                dwf[f] = live ? h2f(dwl[(16 * f + l16) * nbs + b]) : 0.f;
                const uint8_t* wr = wrow + (size_t)16 * f * rowl;
                if (WT == GTEN_Q4) {
                    // nibble - 7 as int8, byte-parallel: (n | 0x80) - 7 never borrows across bytes, ^ 0x80 restores the sign
                    const uint2 by = *(const uint2*)(wr + (size_t)(b ^ (l16 & 7)) * 16);
                    const unsigned x = ((((by.x >> nshift) & 0x0f0f0f0fu) | 0x80808080u) - 0x07070707u) ^ 0x80808080u;
                    const unsigned y = ((((by.y >> nshift) & 0x0f0f0f0fu) | 0x80808080u) - 0x07070707u) ^ 0x80808080u;
                    bl[f] = (long)(((unsigned long)y << 32) | x);
                } else {
                    // Q8 rows are two planes of nb 16-byte pieces: elements 0-15, then 16-31
                    const int piece = (g >> 1) * nbs + b;
                    const uint2 by = *(const uint2*)(wr + (size_t)(piece ^ (l16 & 7)) * 16);
                    bl[f] = (long)(((unsigned long)by.y << 32) | by.x);
                }
            }
#pragma unroll
            for (int t = 0; t < RT; t++) {
                const long al = (long)(((unsigned long)aqv[t][c].y << 32) | aqv[t][c].x);
                const float4 da4 = *(const float4*)(daT + (size_t)b * SP + 16 * t + 4 * g);
                const float da[4] = {da4.x, da4.y, da4.z, da4.w};
#pragma unroll
                for (int f = 0; f < FT; f++) {
                    const mmv_v4i isum = __builtin_amdgcn_mfma_i32_16x16x32_i8(al, bl[f], zero4, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[f][t][i] = acc[f][t][i] + ((float)isum[i] * da[i]) * dwf[f];
                }
            }
        }
    }

    // ---- 4. the eight K slices, added in wave order (the slab is dead: `red` lies over it), one feature tile at a time
#pragma unroll
    for (int f = 0; f < FT; f++) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) red[(wid * SP + 16 * t + 4 * g + i) * 16 + l16] = acc[f][t][i];
        __syncthreads();
        for (int idx = threadIdx.x; idx < SP * 16; idx += 512) {
            const int r = idx >> 4, c = idx & 15;
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) v += red[(q * SP + r) * 16 + c];
            if (r < a_S && colw + 16 * f + c < d_out)
                a_out[(size_t)blockIdx.y * rest.plane + (size_t)r * a_out_cols + colbase + colw + 16 * f + c] = v;
        }
    }
}

// ---- k_dec_mmvh: k_dec_mmv with the block deltas folded into f16 operands (the fast form of gten_mfma.hip).
// k_dec_mmv spends 25 vector instructions per matrix instruction (PMC at 64 sequences) -- the i32 block sums of every
// (row tile, feature tile, quant block) converted and scaled (isum * da) * dw on the VALU.  Here the staging launches leave
// the activations as f16(q * da) fragments, a weight fragment becomes f16((n - 7) * dw) as it leaves the slab (once per
// block and feature tile, shared by the row tiles), and v_mfma_f32_16x16x32_f16 accumulates ACROSS the blocks of a wave's K
// slice: no delta table, no per-block arithmetic on the outputs.  One fp16 rounding per operand element (relative 2^-11);
// the sequences' logits stay inside the wide path's band (tests/test_multiseq_oracle_gpu.py).  gten_hip_set_decode_exact(1): k_dec_mmv.
typedef _Float16 mmvh_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 mmvh_h8 __attribute__((ext_vector_type(8)));
typedef float mmvh_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mmvh_scale(unsigned biased_pair, float bias, unsigned d2)
{
    mmvh_h2 h, d;
    __builtin_memcpy(&h, &biased_pair, 4);
    __builtin_memcpy(&d, &d2, 4);
    const mmvh_h2 b = {(_Float16)bias, (_Float16)bias};
    h = (h - b) * d;
    unsigned out;
    __builtin_memcpy(&out, &h, 4);
    return out;
}

// SILU (gate | up, FT = 4): the workgroup owns ONE 32-wide slice of the FFN -- tiles 0, 1 its gate rows, tiles 2, 3 its up
// rows, the whole K range -- and runs the slice's silu(gate) * up chain (k_dec_silumul_rows' arithmetic: every rounding the
// modules make) in its epilogue, writing the f16 fragments the down projection reads: one launch instead of two.
template <int WT, int RT, int FT, bool SILU>
// (round 4: the one-tile launches at 64 / 128 rows -- q|k|v, o, down -- keep to 128 registers, two workgroups per CU: half the
//  activation fragments in flight per wave, but a second workgroup's requests fly while the first computes -- 180 / 162
//  registers and one workgroup per CU before: 64 sequences 37.3 k -> 38.4 k tok/s, 256 sequences 71.0 k -> 73.8 k; the two-tile
//  launch (lm_head) at 64 rows likewise: 38.4 -> 38.7 k; at 128 rows it would spill 34 registers)
__global__ __launch_bounds__(512, ((FT == 1 && RT >= 4) || (FT == 2 && RT == 4 && !SILU)) ? 4 : 2) void k_dec_mmvh(const uint16_t* __restrict__ a_ah, const void* __restrict__ a_w0, float* __restrict__ a_out,
                                                  const int a_d_in, const int a_d_out0, const int a_out_cols, const int a_S, const int a_n_mats,
                                                  const MmvRest rest)
{
    // (more than four row tiles: activation fragments two blocks ahead instead of four -- registers; the one-tile launches from
    //  four row tiles up: one or two blocks ahead, see the launch bounds)
    constexpr int SP = 16 * RT, FR = 16 * FT, CB = (FT == 1 && RT >= 4) ? ((RT > 4) ? 1 : 2) : (FT == 2 && RT == 4 && !SILU) ? 2 : ((RT > 4) ? ((FT > 2) ? 1 : 2) : 4);
    constexpr int NPF = MMV_MAXP / FT;
    const int nb = a_d_in >> 5;
    const int nbs = nb / (int)gridDim.y, b_lo = (int)blockIdx.y * nbs;
    const int nbw = nbs >> 3;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    const int rowb = nb * (WT == GTEN_Q4 ? 16 : 32);
    const int rowl = nbs * (WT == GTEN_Q4 ? 16 : 32);
    uint8_t* wl = g_smem;                                     // [FR][rowl], 16-byte pieces swizzled: slot = piece ^ (row & 7)
    float* red = (float*)g_smem;                              // [8][SP][16] -- over the slab, once the K loop is done
    uint16_t* dwl = (uint16_t*)(wl + max((size_t)FR * rowl, (size_t)8 * SP * 64));   // [FR][nbs] weight deltas

    static_assert(!SILU || FT == 4, "the fused FFN epilogue owns two gate and two up tiles");
    int colw = blockIdx.x * FR, colbase = 0, m = 0;
    if (!SILU && a_n_mats > 1 && colw >= a_d_out0) {
        colw -= a_d_out0; colbase = a_d_out0; m = 1;
        if (a_n_mats > 2 && colw >= rest.d_out1) { colw -= rest.d_out1; colbase += rest.d_out1; m = 2; }
    }
    if (SILU) colw = blockIdx.x * 32;                         // the FFN slice (rows of both matrices)
    const void* w = (m == 0) ? a_w0 : (m == 1) ? rest.w1 : rest.w2;
    const int d_out = (m == 0) ? a_d_out0 : (m == 1) ? rest.d_out1 : rest.d_out2;
    const PackedW pw = packed_view(w, WT, d_out, a_d_in);
    const PackedW pwu = packed_view(SILU ? rest.w1 : w, WT, d_out, a_d_in);      // (SILU: the up matrix, as wide as the gate matrix)

    // ---- 1. the slab, its deltas and this wave's first activation fragments: one memory round trip
    const int ppr = rowl >> 4;
    const int sr = threadIdx.x >> 5, c0 = threadIdx.x & 31;
    uint4 wp[FT][NPF];
#pragma unroll
    for (int f = 0; f < FT; f++)
#pragma unroll
        for (int k = 0; k < NPF; k++) wp[f][k] = make_uint4(0, 0, 0, 0);
    unsigned dwv[FT][3];
#pragma unroll
    for (int f = 0; f < FT; f++) {
        const size_t frow = SILU ? (size_t)min(colw + 16 * (f & 1) + sr, d_out - 1) : (size_t)min(colw + 16 * f + sr, d_out - 1);
        const PackedW& pm = (SILU && f >= 2) ? pwu : pw;
        const uint8_t* srow = pm.qs + frow * rowb;
#pragma unroll
        for (int k = 0; k < NPF; k++)
            if (32 * k < ppr) {
                const int lp = min(c0 + 32 * k, ppr - 1);
                const int gp = (WT == GTEN_Q4) ? b_lo + lp : (lp < nbs ? b_lo + lp : nb + b_lo + (lp - nbs));
                // (default cache policy on purpose: nontemporal requests here measured 36.5 k = 36.5 k tok/s at 64 sequences and
                //  57.0 -> 56.0 k at 256 -- a lane's neighbours find the slab in the memory-side cache)
                wp[f][k] = *(const uint4*)(srow + (size_t)gp * 16);
            }
        const unsigned* drow = (const unsigned*)(pm.ds + frow * nb + b_lo);
#pragma unroll
        for (int k = 0; k < 3; k++) dwv[f][k] = drow[min(c0 + 32 * k, (nbs >> 1) - 1)];
    }
    const int b0 = b_lo + wid * nbw;
    const uint16_t* afr = a_ah + (size_t)lane * 8;            // fragment order: 1024 contiguous bytes per (block, row tile)
    uint4 araw[RT][CB];
    auto request = [&](int bb) {
#pragma unroll
        for (int c = 0; c < CB; c++)
#pragma unroll
            for (int t = 0; t < RT; t++) araw[t][c] = *(const uint4*)(afr + ((size_t)min(bb + c, nb - 1) * RT + t) * 512);
    };
    request(b0);
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. park the slab and its deltas in LDS
#pragma unroll
    for (int f = 0; f < FT; f++) {
        const int r = 16 * f + sr;
#pragma unroll
        for (int k = 0; k < NPF; k++) {
            const int c = c0 + 32 * k;
            if (c < ppr) *(uint4*)(wl + (size_t)r * rowl + (size_t)(c ^ (r & 7)) * 16) = wp[f][k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int c = c0 + 32 * k;
            if (c < (nbs >> 1)) ((unsigned*)dwl)[r * (nbs >> 1) + c] = dwv[f][k];
        }
    }
    __syncthreads();

    // ---- 3. this wave's K slice, accumulated inside the matrix core
    const int nshift = (g < 2) ? 4 : 0;
    mmvh_f4 acc[FT][RT];
#pragma unroll
    for (int f = 0; f < FT; f++)
#pragma unroll
        for (int t = 0; t < RT; t++) acc[f][t] = (mmvh_f4){0.f, 0.f, 0.f, 0.f};
    const uint8_t* wrow = wl + (size_t)l16 * rowl + (g & 1) * 8;
    for (int bb = b0; bb < b0 + nbw; bb += CB) {
        uint4 aqv[RT][CB];
#pragma unroll
        for (int c = 0; c < CB; c++)
#pragma unroll
            for (int t = 0; t < RT; t++) aqv[t][c] = araw[t][c];
        if (bb + CB < b0 + nbw) request(bb + CB);
#pragma unroll
        for (int c = 0; c < CB; c++) {
            const int b = min(bb + c, b_lo + nbs - 1) - b_lo;
            const bool live = bb + c < b0 + nbw;                 // blocks past this wave's slice meet a zero delta
            mmvh_h8 bh[FT];
#pragma unroll
            for (int f = 0; f < FT; f++) {
                const unsigned dbits = live ? (unsigned)dwl[(16 * f + l16) * nbs + b] : 0u;
                const unsigned d2 = dbits | (dbits << 16);
                const uint8_t* wr = wrow + (size_t)16 * f * rowl;
                unsigned u[4];
                if (WT == GTEN_Q4) {
                    const uint2 by = *(const uint2*)(wr + (size_t)(b ^ (l16 & 7)) * 16);
                    const unsigned x = by.x >> nshift, y = by.y >> nshift;
                    u[0] = mmvh_scale((x & 0x000f000fu) | 0x64006400u, 1031.0f, d2);            // nibbles of bytes 0, 2: elements (0, 2)
                    u[1] = mmvh_scale(((x >> 8) & 0x000f000fu) | 0x64006400u, 1031.0f, d2);     // bytes 1, 3: (1, 3)
                    u[2] = mmvh_scale((y & 0x000f000fu) | 0x64006400u, 1031.0f, d2);
                    u[3] = mmvh_scale(((y >> 8) & 0x000f000fu) | 0x64006400u, 1031.0f, d2);
                } else {
                    const int piece = (g >> 1) * nbs + b;
                    const uint2 by = *(const uint2*)(wr + (size_t)(piece ^ (l16 & 7)) * 16);
                    u[0] = mmvh_scale((by.x & 0x00ff00ffu) ^ 0x64806480u, 1152.0f, d2);         // int8 + 128 in the low bits of 1024 + ...
                    u[1] = mmvh_scale(((by.x >> 8) & 0x00ff00ffu) ^ 0x64806480u, 1152.0f, d2);
                    u[2] = mmvh_scale((by.y & 0x00ff00ffu) ^ 0x64806480u, 1152.0f, d2);
                    u[3] = mmvh_scale(((by.y >> 8) & 0x00ff00ffu) ^ 0x64806480u, 1152.0f, d2);
                }
                __builtin_memcpy(&bh[f], u, 16);
            }
#pragma unroll
            for (int t = 0; t < RT; t++) {
                mmvh_h8 ah;
                __builtin_memcpy(&ah, &aqv[t][c], 16);
#pragma unroll
                for (int f = 0; f < FT; f++) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[f], acc[f][t], 0, 0, 0);
            }
        }
    }

    // ---- 4. the eight K slices, added in wave order
    if (SILU) {
        float* fin = (float*)(dwl + (size_t)FR * nbs);        // [4][SP][16]: the slice's gate and up sums
#pragma unroll
        for (int f = 0; f < FT; f++) {
            __syncthreads();
#pragma unroll
            for (int t = 0; t < RT; t++)
#pragma unroll
                for (int i = 0; i < 4; i++) red[(wid * SP + 16 * t + 4 * g + i) * 16 + l16] = acc[f][t][i];
            __syncthreads();
            for (int idx = threadIdx.x; idx < SP * 16; idx += 512) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < 8; q++) v += red[q * SP * 16 + idx];
                fin[f * SP * 16 + idx] = v;
            }
        }
        __syncthreads();
        // 32 lanes = the 32 elements of one sequence's slice = one Q8 block
        uint16_t* oh = (uint16_t*)a_out;
        for (int idx = threadIdx.x; idx < SP * 32; idx += 512) {
            const int r = idx >> 5, k = idx & 31;
            float gv = act_round32(fin[((k >> 4) * SP + r) * 16 + (k & 15)], false);           // gate projection written in the activation dtype
            // (round 5, measured and not kept: the hardware exponential and a rounded reciprocal here -- gate|up 15.1 -> 14.8 us per
            //  launch at 128 rows, the 256-sequence step 3.07-3.12 -> 3.11-3.12 ms: the chain is not what the epilogue waits for)
            gv = act_round32(gv / (1.0f + expf(-gv)), false);                                    // silu in place
            const float uv = act_round32(fin[((2 + (k >> 4)) * SP + r) * 16 + (k & 15)], false);    // up projection written
            const float v = gv * uv;                                                             // mul in place, then written:
            const Q8Scale sc = q8_scale_from_absmax(max32(fabsf(v)));
            const int qv = q8_round(v, sc.scale);
            const int kp = (k & ~3) | ((k & 1) << 1) | ((k >> 1) & 1);
            if (r < a_S) oh[(((size_t)blockIdx.x * RT + (r >> 4)) * 64 + (kp >> 3) * 16 + (r & 15)) * 8 + (kp & 7)] = f2h((float)qv * sc.ddeq);
        }
        return;
    }
#pragma unroll
    for (int f = 0; f < FT; f++) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) red[(wid * SP + 16 * t + 4 * g + i) * 16 + l16] = acc[f][t][i];
        __syncthreads();
        for (int idx = threadIdx.x; idx < SP * 16; idx += 512) {
            const int r = idx >> 4, c = idx & 15;
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q++) v += red[(q * SP + r) * 16 + c];
            if (r < a_S && colw + 16 * f + c < d_out)
                a_out[(size_t)blockIdx.y * rest.plane + (size_t)r * a_out_cols + colbase + colw + 16 * f + c] = v;
        }
    }
}

// silu(write(gate)) * write(up) for S staged rows (wide multi-sequence decode: the gate and up projections come
// from k_dec_mmv as raw f32 rows [gate | up]); written in the fragment-major staging for the down projection.
// Same chain as the EPI_SILUMUL epilogues above (gten/modules.cpp:238-247).  One thread per element, 32 lanes =
// one Q8 block.
__global__ __launch_bounds__(256) void k_dec_silumul_rows(const float* __restrict__ gu_raw, int n_ffn, int rt, int plane,
                                                          int8_t* __restrict__ out_q, float* __restrict__ out_d, int* __restrict__ out_sum, int h16)
{
    const int q = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;         // n_ffn % 256 == 0
    const float* row = gu_raw + (size_t)q * 2 * n_ffn;
    // (`plane` floats further: the second K-split plane of k_dec_mmv's partial sums; 0 = a single plane)
    const float g0 = row[e], u0 = row[n_ffn + e], g1 = row[plane + e], u1 = row[plane + n_ffn + e];
    float g = act_round32(plane ? g0 + g1 : g0, false);                    // gate projection written in the activation dtype
    g = act_round32(g / (1.0f + expf(-g)), false);                         // silu in place
    const float u = act_round32(plane ? u0 + u1 : u0, false);              // up projection written
    const float v = g * u;                                                 // mul in place, then written:
    const Q8Scale sc = q8_scale_from_absmax(max32(fabsf(v)));
    const int qv = q8_round(v, sc.scale);
    const int qs = sum32_i(qv);
    const int b = e >> 5, k = e & 31;
    if (h16) {                                                             // k_dec_mmvh's f16 fragments (elements 0,2,1,3 of every four)
        const int kp = (k & ~3) | ((k & 1) << 1) | ((k >> 1) & 1);
        ((uint16_t*)out_q)[(((size_t)b * rt + (q >> 4)) * 64 + (kp >> 3) * 16 + (q & 15)) * 8 + (kp & 7)] = f2h((float)qv * sc.ddeq);
        return;
    }
    out_q[(((size_t)b * rt + (q >> 4)) * 64 + (k >> 3) * 16 + (q & 15)) * 8 + (k & 7)] = (int8_t)qv;
    if (k == 0) { out_d[(size_t)b * 16 * rt + q] = sc.ddeq; out_sum[(size_t)b * 16 * rt + q] = qs; }
}

// ---- the same launch shape for f16 weights x f16 activations (wide decode of the f16 configuration): a workgroup owns 16
// output features for all rows (sequences), its eight waves split the workgroup's K range (gridDim.y K slices, as in
// k_dec_mmv), one v_mfma_f32_16x16x32_f16 per (row tile, 32 elements of K) accumulating in the matrix core.  No LDS in the
// K loop: a weight fragment is used once (16 bytes straight from HBM: lane (l16, g) reads feature l16, elements 8 g .. 8 g + 7
// of the step), an activation fragment is 16 bytes of the staged f16 row (L2).  Requests run CBK steps ahead of the
// arithmetic.  The eight K ranges are added in wave order, the slices by the consumer: deterministic.
typedef _Float16 mmv_h8 __attribute__((ext_vector_type(8)));
typedef float mmv_f4 __attribute__((ext_vector_type(4)));

template <int RT>
__global__ __launch_bounds__(512) void k_dec_mmv_f16(const uint16_t* __restrict__ a_ah, const void* __restrict__ a_w0, float* __restrict__ a_out,
                                                     const int a_d_in, const int a_d_out0, const int a_out_cols, const int a_S,
                                                     const int a_n_mats, const MmvRest rest)
{
    constexpr int SP = 16 * RT, CBK = RT > 4 ? 2 : 4;         // (eight row tiles: two steps ahead -- the fragments of four would not fit the registers)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l16 = lane & 15, g = lane >> 4;
    float* red = (float*)g_smem;                              // [8][SP][16]
    int colw = blockIdx.x * 16, colbase = 0, m = 0;
    if (a_n_mats > 1 && colw >= a_d_out0) {
        colw -= a_d_out0; colbase = a_d_out0; m = 1;
        if (a_n_mats > 2 && colw >= rest.d_out1) { colw -= rest.d_out1; colbase += rest.d_out1; m = 2; }
    }
    const uint16_t* w = (const uint16_t*)((m == 0) ? a_w0 : (m == 1) ? rest.w1 : rest.w2);
    const int d_out = (m == 0) ? a_d_out0 : (m == 1) ? rest.d_out1 : rest.d_out2;
    const int ksl = a_d_in / (int)gridDim.y, kw = ksl >> 3;   // K elements of this workgroup / of each wave (kw % 32 == 0)
    const int k_lo = (int)blockIdx.y * ksl + wid * kw, steps = kw >> 5;
    const uint16_t* wrow = w + (size_t)min(colw + l16, d_out - 1) * a_d_in + k_lo + 8 * g;
    const uint16_t* arow = a_ah + (size_t)l16 * a_d_in + k_lo + 8 * g;
    mmv_f4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) acc[t] = mmv_f4{0.f, 0.f, 0.f, 0.f};
    uint4 bw[CBK], aw[CBK][RT];
    auto request = [&](int s0) {
#pragma unroll
        for (int c = 0; c < CBK; c++) {
            const int st = min(s0 + c, steps - 1);
            bw[c] = *(const uint4*)(wrow + 32 * st);
#pragma unroll
            for (int t = 0; t < RT; t++) aw[c][t] = *(const uint4*)(arow + (size_t)16 * t * a_d_in + 32 * st);
        }
    };
    request(0);
    for (int s0 = 0; s0 < steps; s0 += CBK) {
        uint4 bq[CBK], aq[CBK][RT];
#pragma unroll
        for (int c = 0; c < CBK; c++) {
            bq[c] = bw[c];
#pragma unroll
            for (int t = 0; t < RT; t++) aq[c][t] = aw[c][t];
        }
        if (s0 + CBK < steps) request(s0 + CBK);
#pragma unroll
        for (int c = 0; c < CBK; c++) {
            if (s0 + c < steps) {                             // (uniform) the ragged last chunk
                const mmv_h8 bh = __builtin_bit_cast(mmv_h8, bq[c]);
#pragma unroll
                for (int t = 0; t < RT; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mmv_h8, aq[c][t]), bh, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) red[(wid * SP + 16 * t + 4 * g + i) * 16 + l16] = acc[t][i];
    __syncthreads();
    for (int idx = threadIdx.x; idx < SP * 16; idx += 512) {
        const int r = idx >> 4, c = idx & 15;
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) v += red[(q * SP + r) * 16 + c];
        if (r < a_S && colw + c < d_out) a_out[(size_t)blockIdx.y * rest.plane + (size_t)r * a_out_cols + colbase + colw + c] = v;
    }
}

// silu(write(gate)) * write(up) for S rows, f16 activations: element-wise (gten/modules.cpp:238-247), written as the f16
// rows the down projection's matrix-core launch reads
__global__ __launch_bounds__(256) void k_dec_silumul_rows_f16(const float* __restrict__ gu_raw, int n_ffn, int plane, uint16_t* __restrict__ out_h)
{
    const int q = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
    const float* row = gu_raw + (size_t)q * 2 * n_ffn;
    const float g0 = row[e], u0 = row[n_ffn + e], g1 = row[plane + e], u1 = row[plane + n_ffn + e];
    float g = h2f(f2h(plane ? g0 + g1 : g0));
    g = h2f(f2h(g / (1.0f + expf(-g))));
    const float u = h2f(f2h(plane ? u0 + u1 : u0));
    out_h[(size_t)q * n_ffn + e] = f2h(g * u);
}
