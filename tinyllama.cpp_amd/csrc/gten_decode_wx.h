// gten_decode_wx.h: the W.x kernels of the decode step for 1 .. 8 sequences (k_dec_gemv8 with its register prologues and
// epilogues, k_dec_gemvm) -- part of the single-token decode translation unit: included by gten_decode.hip (which owns the includes, the LDS
// symbol, the launch macros and the host side).  Split out in round 4; the code is unchanged.
// ------------------------------------------------------------ W.x kernels
//
// One kernel template for every W.x launch of the step, all three configurations:
//   * the wave's weight rows are requested from HBM FIRST, then the prologue
//     runs while they are in flight (hipcc's __syncthreads() here is
//     lgkmcnt(0)+s_barrier, so the loads stay outstanding across it);
//   * the prologue keeps 8 consecutive elements per thread in registers, so a
//     Q8 block is one quad of lanes and its absmax / sum are two DPP steps;
//   * the residual rows between kernels are kept as f32 (exact storage values).

// a device pointer that travelled as a 64-bit kernel argument (hot arguments, below): rebuilt through the global
// address space -- a pointer made from an integer would otherwise be a generic one and every access a FLAT access
template <typename T>
__device__ __forceinline__ const T* from_word(unsigned long long w)
{
    return (const T*)(const __attribute__((address_space(1))) T*)w;
}
// stores through a pointer hipcc only knows as generic (cache rows): global, not FLAT (a FLAT store also counts on
// the LDS counter, so the next barrier would wait for its round trip)
template <typename T>
__device__ __forceinline__ void store_global(void* p, T v)
{
    *(__attribute__((address_space(1))) T*)(uintptr_t)p = v;
}

struct Gemv8Args {
    const DecStep* step;
    const uint8_t* qs[3];         // per matrix: packed quants
    const uint16_t* ds[3];        // per matrix: deltas
    int rows[3];
    int n_mats;
    int d_in;                     // multiple of 32
    float* out;
    // prologue inputs
    const void* table; int n_vocab; const int32_t* tokens;     // PRO_EMBED
    const float* res_a; const float* res_raw;                   // PRO_RESID (f32 rows)
    float* x_out;                                               // PRO_EMBED/PRO_RESID: new residual row, f32
    const uint16_t* norm_w;
    const float* att_part; int d_head, n_chunks;                // PRO_ATT / PRO_ATTW (d_head a power of two)
    const float* att_stats; int stats_stride;                   // PRO_ATTW: [n_heads][n_chunks] (max, sum) of k_dec_attn_one64
    int d_head_shift;
    float* best_val; int* best_idx;                             // lm_head: per-wave running argmax (may be null)
    // EPI_SILUMUL writes / PRO_ACTQ8 reads the staged FFN activation in HBM (ActQ8 layout)
    int8_t* act_q; float* act_d; int* act_sum;
    float* act_f;                                               // same for f16 activations: f32 row of exact f16 values
    // multi-sequence decode (EPI_STAGE / k_dec_gemvm): element strides between consecutive sequences
    int raw_stride;               // res_raw / out rows
    int raw_plane;                // staging launches: floats to the second K-split plane of res_raw (0: a single plane)
    int raw_nplanes;              // ... WXP_PLANES when the producer was k_dec_wxp_f16 (raw_plane floats apart), else 0: one or two planes
    int tok_stride;               // token id rows
    int part_stride;              // att_part
    int best_stride;              // best_val / best_idx
    // k_dec_gemvm + EPI_SILUMUL: where the FFN activation is staged (its INPUT stage is act_*)
    int8_t* out_q; float* out_d; int* out_sum; float* out_f;
    int frag_rt;                  // EPI_STAGE_FRAG: row tiles (of 16 sequences) of the fragment-major staging (k_dec_mmv)
    int frag_h16;                 // ... as f16(quant * delta) fragments (k_dec_mmvh) instead of int8 + delta table
    // PRO_EMBED (the step's first launch) copies the RoPE rotation of the current position where the score
    // kernels find it without knowing the position: rope_now[seq][rope_half]
    const float2* rope; float2* rope_now; int rope_half;
};

// ---- prologue building blocks: a thread owns EPT (8 or 4) consecutive elements, so a
// Q8 block is a group of 32/EPT (4 or 8) adjacent lanes and its absmax / sum are DPP steps.
template <int EPT> __device__ __forceinline__ float grp_max(float v)   // v >= 0 (an absolute maximum)
{
    return (EPT == 4) ? nn_max8(v) : nn_max4(v);
}
template <int EPT> __device__ __forceinline__ int grp_sum_i(int v)
{
    v = quad_sum_i(v);
    if (EPT == 4) v += dpp_mov_i<0x141>(v);
    return v;
}
template <int EPT> __device__ __forceinline__ float grp_sum(float v)   // balanced tree, natural order
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    if (EPT == 4) v += dpp_mov<0x141>(v);
    return v;
}

// write_row_from_float + read_row_to_float of one Q8 block spread over its lane group:
// v <- q * fp16(delta)   (gten/quants.h:52-76)
template <int EPT> __device__ __forceinline__ void q8_roundN(float (&v)[EPT])
{
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; i++) amax = fmaxf(amax, fabsf(v[i]));
    const Q8Scale s = q8_scale_from_absmax(grp_max<EPT>(amax));
#pragma unroll
    for (int i = 0; i < EPT; i++) v[i] = (float)q8_round(v[i], s.scale) * s.ddeq;
}

// "written in the activation dtype and read back": Q8 block rounding or f16 rounding
template <int WT, int EPT> __device__ __forceinline__ void act_roundN(float (&v)[EPT])
{
    if (WT == GTEN_F16) {
#pragma unroll
        for (int i = 0; i < EPT; i++) v[i] = h2f(f2h(v[i]));
    } else {
        q8_roundN<EPT>(v);
    }
}

// quantize the group's block and stage it for the dot products
template <int EPT> __device__ __forceinline__ void q8_stageN(const float (&v)[EPT], int b, int sub, ActQ8 a)
{
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; i++) amax = fmaxf(amax, fabsf(v[i]));
    const Q8Scale s = q8_scale_from_absmax(grp_max<EPT>(amax));
    int q[EPT], sum = 0;
#pragma unroll
    for (int i = 0; i < EPT; i++) { q[i] = q8_round(v[i], s.scale); sum += q[i]; }
    const int lo = (q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) | ((q[3] & 0xff) << 24);
    if (EPT == 8) {
        int2 pk;
        pk.x = lo;
        pk.y = (q[EPT - 4] & 0xff) | ((q[EPT - 3] & 0xff) << 8) | ((q[EPT - 2] & 0xff) << 16) | ((q[EPT - 1] & 0xff) << 24);
        *(int2*)(a.q + (size_t)b * 32 + sub * 8) = pk;
    } else {
        *(int*)(a.q + (size_t)b * 32 + sub * 4) = lo;
    }
    sum = grp_sum_i<EPT>(sum);
    if (sub == 0) { a.d[b] = s.ddeq; a.sum[b] = sum; }
}

// The same block, staged for the matrix-core W.x of many sequences (k_dec_mmv): quants in MFMA-fragment order
// [block][row tile][lane = 16 * (k % 32 / 8) + row % 16][8 bytes] -- one 512-byte coalesced load per (row tile,
// block) and wave -- and the block deltas / sums as [block][row] so that a lane's four rows are one 16-byte load.
struct ActFrag {
    int8_t* q; float* d; int* sum;
    int rt, row;
    int h16;                      // k_dec_mmvh: the fragments as f16(quant * delta) instead (16 bytes per lane, elements 0,2,1,3 of every four)
};
template <int EPT> __device__ __forceinline__ void q8_stage_frag(const float (&v)[EPT], int b, int sub, ActFrag a)
{
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < EPT; i++) amax = fmaxf(amax, fabsf(v[i]));
    const Q8Scale s = q8_scale_from_absmax(grp_max<EPT>(amax));
    int q[EPT], sum = 0;
#pragma unroll
    for (int i = 0; i < EPT; i++) { q[i] = q8_round(v[i], s.scale); sum += q[i]; }
    const int k = sub * EPT;                                   // first element of this thread inside the block
    if (a.h16) {
        uint16_t* hd = (uint16_t*)a.q + ((((size_t)b * a.rt + (a.row >> 4)) * 64 + (k >> 3) * 16 + (a.row & 15)) * 8 + (k & 7));
        unsigned hw[EPT / 2];
#pragma unroll
        for (int i = 0; i < EPT; i += 4) {
            hw[i / 2] = (unsigned)f2h((float)q[i] * s.ddeq) | ((unsigned)f2h((float)q[i + 2] * s.ddeq) << 16);
            hw[i / 2 + 1] = (unsigned)f2h((float)q[i + 1] * s.ddeq) | ((unsigned)f2h((float)q[i + 3] * s.ddeq) << 16);
        }
        if (EPT == 8) *(uint4*)hd = make_uint4(hw[0], hw[1], hw[EPT / 2 - 2], hw[EPT / 2 - 1]);
        else *(uint2*)hd = make_uint2(hw[0], hw[1]);
        return;
    }
    int8_t* dst = a.q + ((((size_t)b * a.rt + (a.row >> 4)) * 64 + (k >> 3) * 16 + (a.row & 15)) * 8 + (k & 7));
    const int lo = (q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) | ((q[3] & 0xff) << 24);
    if (EPT == 8) {
        int2 pk;
        pk.x = lo;
        pk.y = (q[EPT - 4] & 0xff) | ((q[EPT - 3] & 0xff) << 8) | ((q[EPT - 2] & 0xff) << 16) | ((q[EPT - 1] & 0xff) << 24);
        *(int2*)dst = pk;
    } else {
        *(int*)dst = lo;
    }
    sum = grp_sum_i<EPT>(sum);
    if (sub == 0) { a.d[(size_t)b * 16 * a.rt + a.row] = s.ddeq; a.sum[(size_t)b * 16 * a.rt + a.row] = sum; }
}

template <int EPT> __device__ __forceinline__ void ldN(const float* p, float (&v)[EPT])
{
    const float4 a = ((const float4*)p)[0];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    if (EPT == 8) {
        const float4 b = ((const float4*)p)[1];
        v[EPT - 4] = b.x; v[EPT - 3] = b.y; v[EPT - 2] = b.z; v[EPT - 1] = b.w;
    }
}
template <int EPT> __device__ __forceinline__ void stN(float* p, const float (&v)[EPT])
{
    ((float4*)p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    if (EPT == 8) ((float4*)p)[1] = make_float4(v[EPT - 4], v[EPT - 3], v[EPT - 2], v[EPT - 1]);
}
template <int EPT> __device__ __forceinline__ float sumsq_treeN(const float (&v)[EPT])
{
    float s = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    if (EPT == 8) s = s + ((v[EPT - 4] * v[EPT - 4] + v[EPT - 3] * v[EPT - 3]) + (v[EPT - 2] * v[EPT - 2] + v[EPT - 1] * v[EPT - 1]));
    return s;
}

// max / integer sum over the 32 lanes of a half wave (lane = element of one Q8 block)
__device__ __forceinline__ float max32(float v) { return nn_max32(v); }      // v >= 0 (absolute values)
__device__ __forceinline__ int sum32_i(int v) { return sum32_lanes_i(v); }
// one value per lane, the half wave is one Q8 block: v <- q * fp16(delta)
__device__ __forceinline__ float q8_round32(float v)
{
    const Q8Scale s = q8_scale_from_absmax(max32(fabsf(v)));
    return (float)q8_round(v, s.scale) * s.ddeq;
}
__device__ __forceinline__ float act_round32(float v, bool f16)
{
    return f16 ? h2f(f2h(v)) : q8_round32(v);
}

// NT threads (256 or 512): the prologue row (<= 2048 elements) is spread over all of them,
// EPT = 2048/NT elements each, so a 512-thread workgroup puts two waves on every SIMD and
// its prologue issues at twice the rate of a 256-thread one.
// EPI_RAW:     wave w owns rows [(NW*blockIdx.x + w) * R, +R) of the concatenated matrices
// EPI_SILUMUL: 8 waves, block = one 32-wide slice of the FFN: waves 0-3 its gate rows, waves 4-7
//              its up rows (R = 8); the slice's silu(gate)*up chain runs ONCE here, in the
//              epilogue, and is stored quantized for the down projection (PRO_ACTQ8)
// HOT ARGUMENTS.  A kernel reads its arguments with scalar loads before it can form a single address: one more
// memory round trip at the head of every launch of the chain (~0.3 us each, tools/microbench_launch_floor.hip built
// with and without preloading).  gfx950's command processor can PRELOAD the first 14 dwords of the argument segment
// into SGPRs while the waves are being created (-mllvm -amdgpu-kernarg-preload-count, build.py) -- but only leading
// scalar arguments, never a by-value struct.  So the launch passes the few words the first requests are formed from
// as seven leading 64-bit scalars (GemvHot), ahead of the full struct; what a prologue kind does not need carries
// its small integers instead.
struct GemvHot {
    const void* p0;               // PRO_RESID res_raw | PRO_EMBED table | PRO_ATT(W) att_part | PRO_ACTQ8 act_q (f16: act_f)
    const void* p1;               // PRO_RESID res_a   | PRO_EMBED tokens | PRO_ATT {d_head_shift, n_chunks} | PRO_ACTQ8 act_d
    const void* p2;               // PRO_RESID / PRO_EMBED norm_w | PRO_ACTQ8 act_sum | PRO_ATTW att_stats
    const uint8_t* qs0; const uint16_t* ds0;
    int d_in, rows0;
    const DecStep* step;
};
static_assert(sizeof(GemvHot) == 56, "seven 64-bit scalars: the preloadable part of the argument segment");
union GemvHotWords {
    unsigned long long w[7];
    GemvHot h;
    __host__ __device__ GemvHotWords() : w{0, 0, 0, 0, 0, 0, 0} {}
};

// NM: matrices concatenated along the output rows -- 1 (o, down, lm_head: the weight requests are formed from the
// preloaded words alone), or 0 = a.n_mats at run time (q|k|v, gate|up)
template <int WT, int PRO, int NCH, int R, int EPI, int NT, int NM = 0>
__global__ __launch_bounds__(NT) void k_dec_gemv8(const unsigned long long h0, const unsigned long long h1, const unsigned long long h2,
                                                  const unsigned long long h3, const unsigned long long h4, const unsigned long long h5,
                                                  const unsigned long long h6, const Gemv8Args a)
{
    // (rebuilt word by word, through the global address space: a pointer made from an integer would otherwise be
    // a generic one and every load through it a FLAT load)
    // k_dec_gemv8's words: p0 | p1 | p2 | quants of matrix 0 | of matrix 1 | d_in, rows0, rows1, rows2 (16 bits each) |
    // step word -- or, for PRO_RESID (which never needs the position), the quants of matrix 2.  The deltas of a matrix
    // follow its quants in the packed layout (include/gten_hip.h), so the weight requests of q|k|v and gate|up need
    // nothing from the argument struct either.
    GemvHot hot;
    hot.p0 = from_word<void>(h0); hot.p1 = from_word<void>(h1); hot.p2 = from_word<void>(h2);
    hot.qs0 = from_word<uint8_t>(h3);
    const uint8_t* hot_qs1 = from_word<uint8_t>(h4);
    hot.d_in = (int)(h5 & 0xffffu); hot.rows0 = (int)((h5 >> 16) & 0xffffu);
    const int hot_rows1 = (int)((h5 >> 32) & 0xffffu), hot_rows2 = (int)(h5 >> 48);
    const uint8_t* hot_qs2 = (PRO == PRO_RESID) ? from_word<uint8_t>(h6) : a.qs[2];
    hot.step = (PRO == PRO_RESID) ? nullptr : from_word<DecStep>(h6);
    constexpr int WBYTES = (WT == GTEN_Q4) ? 16 : 32;                       // quant bytes per block (unused for f16 weights)
    hot.ds0 = (const uint16_t*)(hot.qs0 + (size_t)hot.rows0 * (hot.d_in >> 5) * WBYTES);
    constexpr int EPT = 2048 / NT;                // prologue elements per thread (d <= 2048 unless PRO_ACTQ8)
    constexpr int LPB = 32 / EPT;                 // lanes per Q8 block
    constexpr int NW = NT / 64;
    static_assert(NT == 256 || NT == 512, "256 or 512 threads");
    // (round 3, built and measured: the FFN slice launch on 1024 threads -- sixteen waves of four rows, the prologue on the
    //  first 512 threads, the same bits -- 6.23 against 5.86 us per launch: not kept)
    static_assert(EPI != EPI_SILUMUL || NT == 512, "the FFN slice epilogue wants 8 waves");
    constexpr bool F16W = (WT == GTEN_F16);        // f16 weights <=> f16 activations (tinyllama.cpp:258-265)
    const int d = hot.d_in, nb = d >> 5;
    ActStage s = carve_stage((PRO == PRO_ACTQ8 && !F16W) ? 32 : d);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // EPI_STAGE: one workgroup per SEQUENCE runs only the prologue and leaves the staged vector in
    // HBM for the multi-sequence W.x kernel (k_dec_gemvm), which then needs no prologue of its own
    constexpr bool STG = (EPI == EPI_STAGE || EPI == EPI_STAGE_FRAG);
    const int seq = STG ? blockIdx.x : 0;
    const int n = (PRO == PRO_RESID) ? 0 : hot.step[seq].n;      // (PRO_RESID never uses the position)
    const float* res_raw = (const float*)hot.p0 + (size_t)seq * a.raw_stride;     // PRO_RESID only
    const float* res_a = (const float*)hot.p1 + (size_t)seq * d;
    float* x_out = a.x_out ? a.x_out + (size_t)seq * d : nullptr;
    const int32_t* tokens = (const int32_t*)hot.p1 + (size_t)seq * a.tok_stride;        // PRO_EMBED only
    constexpr bool ATT = (PRO == PRO_ATT || PRO == PRO_ATTW);
    const float* att_part = (const float*)hot.p0 + (size_t)seq * a.part_stride;      // PRO_ATT / PRO_ATTW only
    const float* att_stats = (const float*)hot.p2 + (size_t)seq * a.stats_stride;   // PRO_ATTW only
    const uint16_t* norm_w = (const uint16_t*)hot.p2;
    const int att_shift = (int)((uintptr_t)hot.p1 & 0xff), att_chunks = (int)(((uintptr_t)hot.p1 >> 8) & 0xffff), att_dh = 1 << att_shift;
    if (STG) {
        s.q8.q = a.act_q + (size_t)seq * d;
        s.q8.d = a.act_d + (size_t)seq * nb;
        s.q8.sum = a.act_sum + (size_t)seq * nb;
        if (F16W) s.row = a.act_f + (size_t)seq * d;
    }
    const bool stores_x = STG || blockIdx.x == 0;
    const int gi = threadIdx.x, base = gi * EPT, blk = gi / LPB, sub = gi % LPB;
    const bool on = base < d;                     // lanes past the row re-read group 0 (never used):
    const int sbase = on ? base : 0;              // an unconditional load has no select on its result

    // ---- 1. request the prologue's inputs (they come back first: vmcnt is in order)
    float pin0[EPT], pin1[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) { pin0[i] = 0.f; pin1[i] = 0.f; }
    if (PRO == PRO_RESID) { ldN<EPT>(res_raw + sbase, pin0); ldN<EPT>(res_a + sbase, pin1); }
    // staging launches of the wide path: the producer (k_dec_mmv) may have split K over two workgroups -- the second
    // plane of partial sums is requested unconditionally (plane 0: the same row again) and added below
    float pin0b[(STG && PRO == PRO_RESID) ? EPT : 1];
    if constexpr (STG && PRO == PRO_RESID) ldN<EPT>(res_raw + a.raw_plane + sbase, pin0b);
    // ... or eight (k_dec_wxp_f16, gten_decode_wxp.h): planes 2 .. 7 (uniform branch; a staging launch requests nothing behind it)
    float pinx[(STG && PRO == PRO_RESID) ? WXP_PLANES - 2 : 1][EPT];
    if constexpr (STG && PRO == PRO_RESID) {
        if (a.raw_nplanes == WXP_PLANES) {
#pragma unroll
            for (int q = 2; q < WXP_PLANES; q++) ldN<EPT>(res_raw + (size_t)q * a.raw_plane + sbase, pinx[q - 2]);
        }
    }
    // (norm_w is required for PRO_EMBED / PRO_RESID: a null check here would be a branch whose join
    // makes hipcc wait for every outstanding load BEFORE the weight rows below are even requested)
    unsigned nw[4] = {0, 0, 0, 0};
    if (PRO == PRO_EMBED || PRO == PRO_RESID) {
        if (EPT == 8) { const uint4 t = *(const uint4*)(norm_w + sbase); nw[0] = t.x; nw[1] = t.y; nw[2] = t.z; nw[3] = t.w; }
        else { const uint2 t = *(const uint2*)(norm_w + sbase); nw[0] = t.x; nw[1] = t.y; }
    }
    // PRO_ATT: all chunk partials of this thread's elements, requested at once (chunks past the context hold
    // stale but readable data and are dropped by a select below; decoder_create: n_chunks <= DEC_ATT_MAXCH)
    float apart[ATT ? DEC_ATT_MAXCH : 1][EPT];
    float2 cst[PRO == PRO_ATTW ? DEC_ATT_MAXCH : 1];             // PRO_ATTW: (max, sum) of this thread's head, every chunk
    if (PRO == PRO_ATTW) {
        const unsigned h = (unsigned)sbase >> att_shift;
#pragma unroll
        for (int j = 0; j < DEC_ATT_MAXCH; j++) cst[j] = ((const float2*)att_stats)[h * (unsigned)att_chunks + (unsigned)min(j, att_chunks - 1)];
    }
    if (ATT) {
        // (32-bit index arithmetic, the head width as a shift: one integer multiply ahead of the eight requests, not seventeen)
        const unsigned h = (unsigned)sbase >> att_shift, e = (unsigned)sbase & (unsigned)(att_dh - 1);
        const unsigned row0 = h * (unsigned)att_chunks;
#pragma unroll
        for (int j = 0; j < DEC_ATT_MAXCH; j++)
            ldN<EPT>(att_part + (((row0 + (unsigned)min(j, att_chunks - 1)) << att_shift) + e), apart[j]);
    }
    unsigned emb[4] = {0, 0, 0, 0};
    float emb_delta = 0.f;
    if (PRO == PRO_EMBED && F16W) {
        const uint16_t* src = (const uint16_t*)hot.p0 + (size_t)tokens[n - 1] * d + sbase;
        if (EPT == 8) { const uint4 t = *(const uint4*)src; emb[0] = t.x; emb[1] = t.y; emb[2] = t.z; emb[3] = t.w; }
        else { const uint2 t = *(const uint2*)src; emb[0] = t.x; emb[1] = t.y; }
    }
    float2 rot_now = make_float2(1.f, 0.f);
    if (PRO == PRO_EMBED) rot_now = a.rope[(size_t)(n - 1) * a.rope_half + (threadIdx.x & (a.rope_half - 1))];
    if (PRO == PRO_EMBED && !F16W) {
        const int tok = tokens[n - 1];
        const int sb = on ? blk : 0, ssub = on ? sub : 0;
        // high nibbles are elements 0..15, low nibbles 16..31 (gten/quants.h:78-90); Q8 planes hold 16 bytes each
        const int byte0 = (ssub * EPT) & 15;      // first source byte inside the 16-byte half
        const uint8_t* src;
        const uint16_t* dsp;
        if (WT == GTEN_Q4) {
            src = (const uint8_t*)hot.p0 + ((size_t)tok * nb + sb) * 16 + byte0;
            dsp = (const uint16_t*)((const uint8_t*)hot.p0 + (size_t)a.n_vocab * nb * 16);
        } else {
            src = (const uint8_t*)hot.p0 + (size_t)tok * nb * 32 + (size_t)((ssub * EPT) >> 4) * nb * 16 + (size_t)sb * 16 + byte0;
            dsp = (const uint16_t*)((const uint8_t*)hot.p0 + (size_t)a.n_vocab * nb * 32);
        }
        emb[0] = *(const unsigned*)src;
        if (EPT == 8) emb[1] = *(const unsigned*)(src + 4);
        emb_delta = h2f(dsp[(size_t)tok * nb + sb]);
    }

    // ---- 2. request this wave's weight rows; they stay in flight during the prologue
    const int rows0 = hot.rows0, rows1 = (NM != 1) ? hot_rows1 : 0, rows2 = (NM != 1) ? hot_rows2 : 0;
    const int total = rows0 + rows1 + rows2;
    const int r0 = (EPI == EPI_SILUMUL) ? (wid >> 2) * rows0 + blockIdx.x * 32 + (wid & 3) * R
                                        : (blockIdx.x * NW + wid) * R;
    uint4 wq[R][NCH], wq1[R][NCH];
    uint16_t wd[R][NCH];
    // the wave's R rows are consecutive rows of ONE matrix (the launchers require the row counts of concatenated
    // matrices to be multiples of R): the matrix is chosen once per wave, not once per row -- the requests of the
    // gate|up launch (8 rows per wave) used to trickle out over ~300 instructions of per-row pointer selection
    int lr0 = r0, rows_m = rows0;
    const uint8_t* qbase = hot.qs0;
    if (lr0 >= rows0 && rows1 > 0) {
        lr0 -= rows0; qbase = hot_qs1; rows_m = rows1;
        if (lr0 >= rows1 && rows2 > 0) { lr0 -= rows1; qbase = hot_qs2; rows_m = rows2; }
    }
    const uint16_t* dbase = (const uint16_t*)(qbase + (size_t)rows_m * nb * WBYTES);
#pragma unroll
    for (int j = 0; j < R; j++) {
        if (STG) break;                           // no W.x in a staging launch
        const int lr = min(lr0 + j, rows_m - 1);  // clamp: always a valid row, the result of a row past the end is discarded
        const uint16_t* drow = dbase + (size_t)lr * nb;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            // out-of-range K blocks read block 0 (finite data) and are zeroed through
            // the activation scale below: no select on loaded data, so nothing waits here
            const int b = (c * 64 + lane < nb) ? c * 64 + lane : 0;
            if (F16W) {
                // f16 rows: NCH counts 512-element segments, lane takes 8 halves of each
                const int e = (c * 512 + lane * 8 < d) ? c * 512 + lane * 8 : 0;
                wq[j][c] = ld_w16((const uint16_t*)qbase + (size_t)lr * d + e);
                wq1[j][c] = make_uint4(0, 0, 0, 0);
                wd[j][c] = 0;
                continue;
            }
            if (WT == GTEN_Q4) {
                wq[j][c] = ld_w16((const uint4*)(qbase + (size_t)lr * nb * 16) + b);
            } else {
                const uint4* q0 = (const uint4*)(qbase + (size_t)lr * nb * 32);
                wq[j][c] = ld_w16(q0 + b);
                wq1[j][c] = ld_w16(q0 + nb + b);
            }
            wd[j][c] = ld_w2(drow + b);
        }
    }
    __builtin_amdgcn_sched_barrier(0);            // keep every request above ahead of the prologue's arithmetic

    // ---- 3. prologue: the element-wise chain of the reference, on chip
    //         (PRO_ACTQ8: nothing to do, the input was staged in HBM by the producer's epilogue)
    if (PRO != PRO_ACTQ8) {
        float v[EPT];
        float ss = 0.f;
        if (PRO == PRO_EMBED) {
#pragma unroll
            for (int i = 0; i < EPT; i++) {
                if (F16W) {
                    v[i] = h2f((uint16_t)((i & 1) ? (emb[i >> 1] >> 16) : (emb[i >> 1] & 0xffffu)));   // row copied verbatim
                } else {
                    const unsigned byte = (emb[i >> 2] >> ((i & 3) * 8)) & 0xffu;
                    if (WT == GTEN_Q4) v[i] = (float)((((sub * EPT) < 16) ? (int)(byte >> 4) : (int)(byte & 0x0fu)) - 7) * emb_delta;
                    else v[i] = (float)(int)(int8_t)byte * emb_delta;  // block copied verbatim (gten/ops.h:519-521)
                }
            }
            if (WT == GTEN_Q4) q8_roundN<EPT>(v);         // Q4 row is re-quantized to Q8 (gten/ops.h:522-528)
        } else if (PRO == PRO_RESID) {
#pragma unroll
            for (int i = 0; i < EPT; i++) v[i] = pin0[i];
            if constexpr (STG && PRO == PRO_RESID) {
#pragma unroll
                for (int i = 0; i < EPT; i++) v[i] += a.raw_plane ? pin0b[i] : 0.f;
                if (a.raw_nplanes == WXP_PLANES) {
#pragma unroll
                    for (int q = 2; q < WXP_PLANES; q++)
#pragma unroll
                        for (int i = 0; i < EPT; i++) v[i] += pinx[q - 2][i];
                }
            }
            act_roundN<WT, EPT>(v);                       // Linear output written in the activation dtype
#pragma unroll
            for (int i = 0; i < EPT; i++) v[i] = pin1[i] + v[i];
            act_roundN<WT, EPT>(v);                       // Residual output written in the activation dtype
        } else if (PRO == PRO_ATTW) {                     // chunk-local partials joined with their softmax weights
            const int nch = (n + DEC_CHUNK - 1) / DEC_CHUNK;
            float M = -INFINITY;
#pragma unroll
            for (int j = 0; j < DEC_ATT_MAXCH; j++) M = fmaxf(M, (j < nch) ? cst[j].x : -INFINITY);
            float w[DEC_ATT_MAXCH], L = 0.f;
#pragma unroll
            for (int j = 0; j < DEC_ATT_MAXCH; j++) {
                // (hardware exponential: exp(0) = 1 exactly, so a single chunk keeps weight 1; stale chunks are dropped)
                w[j] = (j < nch) ? cst[j].y * __expf(cst[j].x - M) : 0.f;
                L += w[j];
            }
            const float rL = recip_rn(L);
#pragma unroll
            for (int j = 0; j < DEC_ATT_MAXCH; j++) w[j] = (nch == 1) ? 1.0f : w[j] * rL;
#pragma unroll
            for (int i = 0; i < EPT; i++) v[i] = 0.f;
#pragma unroll
            for (int j = 0; j < DEC_ATT_MAXCH; j++)
#pragma unroll
                for (int i = 0; i < EPT; i++) v[i] += (j < nch) ? w[j] * apart[j][i] : 0.f;
        } else {                                          // PRO_ATT: sum of the per-chunk partials, fixed order
            const int nch = (n + DEC_CHUNK - 1) / DEC_CHUNK;
#pragma unroll
            for (int i = 0; i < EPT; i++) v[i] = 0.f;
#pragma unroll
            for (int j = 0; j < DEC_ATT_MAXCH; j++)
#pragma unroll
                for (int i = 0; i < EPT; i++) v[i] += (j < nch) ? apart[j][i] : 0.f;     // v + 0 == v: same sum as the loop
        }
        if (PRO == PRO_EMBED || PRO == PRO_RESID) {
            if (on) {
                if (x_out && stores_x) stN<EPT>(x_out + base, v);
                ss = sumsq_treeN<EPT>(v);
            }
            // RMSNorm (gten/ops.h:762-778), then the row is written as Q8
            ss = block_sum_tree_n<NW, true>(ss, s.red);             // first use of s.red in this kernel
            // (mean of squares: a power-of-two width divides exactly by an exponent shift -- the same bits as ss / d)
            const float ms = ((d & (d - 1)) == 0) ? __builtin_ldexpf(ss, -__builtin_ctz(d)) : ss / (float)d;
            const float inv = recip_rn(sqrtf(ms) + 1e-6f);            // == 1.0f / (...) (recip_rn), see k_rms_norm
#pragma unroll
            for (int i = 0; i < EPT; i++) {
                const uint16_t hw = (uint16_t)((i & 1) ? (nw[i >> 1] >> 16) : (nw[i >> 1] & 0xffffu));
                v[i] = v[i] * inv * h2f(hw);
            }
        }
        if (F16W) {
            act_roundN<WT, EPT>(v);
            if (EPI == EPI_STAGE_FRAG) {
                // wide f16 decode (k_dec_mmv_f16): the row as f16, [sequence][d] -- an MFMA A operand is then one 16-byte load
                if (on) {
                    uint16_t* dst = (uint16_t*)a.act_q + (size_t)seq * d + base;
                    if (EPT == 8) *(uint4*)dst = make_uint4(f2h(v[0]) | ((unsigned)f2h(v[1]) << 16), f2h(v[2]) | ((unsigned)f2h(v[3]) << 16),
                                                            f2h(v[EPT - 4]) | ((unsigned)f2h(v[EPT - 3]) << 16), f2h(v[EPT - 2]) | ((unsigned)f2h(v[EPT - 1]) << 16));
                    else *(uint2*)dst = make_uint2(f2h(v[0]) | ((unsigned)f2h(v[1]) << 16), f2h(v[2]) | ((unsigned)f2h(v[3]) << 16));
                }
            } else if (on) stN<EPT>(s.row + base, v);     // staged as f32 (exact f16 values)
        } else if (on) {
            if (EPI == EPI_STAGE_FRAG) q8_stage_frag<EPT>(v, blk, sub, ActFrag{a.act_q, a.act_d, a.act_sum, a.frag_rt, seq, a.frag_h16});
            else q8_stageN<EPT>(v, blk, sub, s.q8);
        }
        __syncthreads();
    } else if (F16W) {
        // the FFN activation row was stored by the gate/up epilogue: bring it on chip
        for (int i = threadIdx.x * 4; i < d; i += NT * 4) *(float4*)(s.row + i) = *(const float4*)((const float*)hot.p0 + i);
        __syncthreads();
    }

    if (PRO == PRO_EMBED && stores_x && (int)threadIdx.x < a.rope_half) a.rope_now[(size_t)seq * a.rope_half + threadIdx.x] = rot_now;
    if (STG) return;

    // ---- 4. this lane's activation blocks, then the dot products
    int av[NCH][8], asum[NCH];
    float ad[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (F16W) { asum[c] = 0; ad[c] = 0.f; continue; }
        const int b = c * 64 + lane;
        const bool in = b < nb;
        const int bs = in ? b : 0;
        const int8_t* qsrc = (PRO == PRO_ACTQ8) ? (const int8_t*)hot.p0 : s.q8.q;
        const int4* ap = (const int4*)(qsrc + (size_t)bs * 32);
        const int4 a0 = ap[0], a1 = ap[1];
        av[c][0] = a0.x; av[c][1] = a0.y; av[c][2] = a0.z; av[c][3] = a0.w;
        av[c][4] = a1.x; av[c][5] = a1.y; av[c][6] = a1.z; av[c][7] = a1.w;
        const float dd = (PRO == PRO_ACTQ8) ? ((const float*)hot.p1)[bs] : s.q8.d[bs];
        const int sm = (PRO == PRO_ACTQ8) ? ((const int*)hot.p2)[bs] : s.q8.sum[bs];
        ad[c] = in ? dd : 0.f;
        asum[c] = in ? sm : 0;
    }
    float best = -INFINITY;
    int best_i = 0x7fffffff;
    // EPI_SILUMUL: 64 results in LDS that nobody reads any more: red + the start of the row for
    // Q8 activations (the staged vector lives in the Q8 area), the Q8 area for f16 (it lives in the row)
    float* res = F16W ? (float*)((uint8_t*)s.row + (size_t)d * 4) : s.red;
#pragma unroll
    for (int j = 0; j < R; j++) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            if (F16W) {
                // same element order as wave_dot_f16 (gten_dev.h): segments ascending, 8 halves each
                const int e = c * 512 + lane * 8;
                if (e < d) {
                    const float4 a0 = *(const float4*)(s.row + e), a1 = *(const float4*)(s.row + e + 4);
                    const float fa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const unsigned u[4] = {wq[j][c].x, wq[j][c].y, wq[j][c].z, wq[j][c].w};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        acc += h2f((uint16_t)(u[i] & 0xffffu)) * fa[2 * i];
                        acc += h2f((uint16_t)(u[i] >> 16)) * fa[2 * i + 1];
                    }
                }
                continue;
            }
            const int isum = (WT == GTEN_Q4) ? dot_q8_q4_block(av[c], asum[c], wq[j][c])
                                             : dot_q8_q8_block(av[c], wq[j][c], wq1[j][c]);
            acc += (float)isum * (ad[c] * h2f(wd[j][c]));
        }
        acc = wave_sum(acc);
        if (EPI == EPI_SILUMUL) {
            if (lane == 0) res[(wid >> 2) * 32 + (wid & 3) * R + j] = acc;
        } else {
            if (lane == 0 && r0 + j < total) a.out[r0 + j] = acc;
            if (a.best_val && r0 + j < total && acc > best) { best = acc; best_i = r0 + j; }   // strict >: first maximum wins
        }
    }
    if (EPI == EPI_RAW && a.best_val && lane == 0) {
        a.best_val[blockIdx.x * NW + wid] = best;
        a.best_idx[blockIdx.x * NW + wid] = best_i;
    }
    if (EPI == EPI_SILUMUL) {
        // ---- 5. silu(write(gate)) * write(up), written as Q8 (gten/modules.cpp:238-247), once per slice
        __syncthreads();
        if (wid == 0) {
            const int e = lane & 31;
            float g = act_round32(res[e], F16W);                    // gate projection written in the activation dtype
            g = act_round32(g / (1.0f + expf(-g)), F16W);           // silu in place
            const float u = act_round32(res[32 + e], F16W);         // up projection written
            const float v = g * u;                                  // mul in place, then written:
            if (F16W) {
                if (lane < 32) a.act_f[(size_t)blockIdx.x * 32 + e] = h2f(f2h(v));
                return;
            }
            const Q8Scale sc = q8_scale_from_absmax(max32(fabsf(v)));
            const int q = q8_round(v, sc.scale);
            const int qs = sum32_i(q);
            if (lane < 32) {
                a.act_q[(size_t)blockIdx.x * 32 + e] = (int8_t)q;
                if (e == 0) { a.act_d[blockIdx.x] = sc.ddeq; a.act_sum[blockIdx.x] = qs; }
            }
        }
    }
}

// ------------------------------------------- W.x kernel, several sequences
//
// Multi-sequence decode: S independent sequences advance by one token per
// step and SHARE every weight pass -- the weights are streamed once and each row is dotted with
// S staged activation vectors (left in HBM by the EPI_STAGE launches above, ActQ8 layout per
// sequence, or an f32 row for f16).  Per sequence the arithmetic, its order and therefore the
// result are exactly those of the single-sequence kernel (tested bit for bit).
template <int WT, int NCH, int R, int S, int EPI, int NT>
__global__ __launch_bounds__(NT) void k_dec_gemvm(const unsigned long long h0, const unsigned long long h1, const unsigned long long h2,
                                                  const unsigned long long h3, const unsigned long long h4, const unsigned long long h5,
                                                  const unsigned long long h6, const Gemv8Args a)
{
    // hot arguments (GemvHot, PRO_ACTQ8 form): staged inputs, first matrix, sizes
    const int8_t* act_q = from_word<int8_t>(h0);
    const float* act_f = from_word<float>(h0);
    const float* act_d = from_word<float>(h1);
    const int* act_sum = from_word<int>(h2);
    const uint8_t* qs0 = from_word<uint8_t>(h3);
    const uint16_t* ds0 = from_word<uint16_t>(h4);
    const int hot_d_in = (int)(unsigned)(h5 & 0xffffffffull), hot_rows0 = (int)(unsigned)(h5 >> 32);
    (void)h6;
    constexpr int NW = NT / 64;
    constexpr bool F16W = (WT == GTEN_F16);
    static_assert(EPI != EPI_SILUMUL || (NT == 512 && S <= 8), "FFN slice epilogue: 8 waves, one per sequence");
    const int d = hot_d_in, nb = d >> 5;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* res = (float*)g_smem;                  // EPI_SILUMUL: [S][64]

    const int rows0 = hot_rows0, rows1 = a.n_mats > 1 ? a.rows[1] : 0, rows2 = a.n_mats > 2 ? a.rows[2] : 0;
    const int total = rows0 + rows1 + rows2;
    const int r0 = (EPI == EPI_SILUMUL) ? (wid >> 2) * rows0 + blockIdx.x * 32 + (wid & 3) * R
                                        : (blockIdx.x * NW + wid) * R;
    // f16 weights with 8 rows per wave (gate|up) would hold 8 x 4 x 16 bytes of weights beside R x S accumulators: beyond
    // the register file (round 1: 270 scratch accesses at 8 sequences).  Those launches take their rows in two batches
    // of 4 -- the second batch is requested once the first has been consumed.
    constexpr int RB = (F16W && R == 8) ? (S >= 8 ? 2 : 4) : R;
    uint4 wq[RB][NCH], wq1[RB][NCH];
    uint16_t wd[RB][NCH];
    // (one matrix per wave: the row counts of concatenated matrices are multiples of R -- see k_dec_gemv8)
    int lr0 = r0, rows_m = rows0;
    const uint8_t* qbase = qs0;
    const uint16_t* dbase = ds0;
    if (lr0 >= rows0 && rows1 > 0) {
        lr0 -= rows0; qbase = a.qs[1]; dbase = a.ds[1]; rows_m = rows1;
        if (lr0 >= rows1 && rows2 > 0) { lr0 -= rows1; qbase = a.qs[2]; dbase = a.ds[2]; rows_m = rows2; }
    }
    auto request_rows = [&](int jb) {
#pragma unroll
        for (int j = 0; j < RB; j++) {
            const int lr = min(lr0 + jb + j, rows_m - 1);
            const uint16_t* drow = dbase + (size_t)lr * nb;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int b = (c * 64 + lane < nb) ? c * 64 + lane : 0;
                if (F16W) {
                    const int e = (c * 512 + lane * 8 < d) ? c * 512 + lane * 8 : 0;
                    wq[j][c] = *(const uint4*)((const uint16_t*)qbase + (size_t)lr * d + e);
                    wq1[j][c] = make_uint4(0, 0, 0, 0);
                    wd[j][c] = 0;
                    continue;
                }
                if (WT == GTEN_Q4) {
                    wq[j][c] = ((const uint4*)(qbase + (size_t)lr * nb * 16))[b];
                } else {
                    const uint4* q0 = (const uint4*)(qbase + (size_t)lr * nb * 32);
                    wq[j][c] = q0[b];
                    wq1[j][c] = q0[nb + b];
                }
                wd[j][c] = drow[b];
            }
        }
    };
    request_rows(0);

    // ---- the S staged input vectors: HBM -> LDS once per workgroup (every wave needs all of them)
    //      layout: [S][d] quants | [S][nb] deltas | [S][nb] sums   (f16: [S][d] f32 values)
    // (decided at compile time from the row capacity NCH x 512 >= d, so that the reads below are LDS or global reads, not FLAT)
    constexpr bool lds_f = F16W && ((size_t)S * NCH * 512 * 4 <= GEMVM_F16_LDS_LIMIT);
    int8_t* lq = (int8_t*)(g_smem + (EPI == EPI_SILUMUL ? (size_t)S * 64 * 4 : 0));
    float* ld_ = (float*)(lq + (size_t)S * d);
    int* lsum = (int*)(ld_ + (size_t)S * nb);
    float* lf = (float*)lq;
    if (F16W) {
        if (lds_f) {
            for (int i = threadIdx.x * 4; i < S * d; i += NT * 4) *(float4*)(lf + i) = *(const float4*)(act_f + i);
        }
    } else {
        for (int i = threadIdx.x * 16; i < S * d; i += NT * 16) *(uint4*)(lq + i) = *(const uint4*)(act_q + i);
        for (int i = threadIdx.x; i < S * nb; i += NT) { ld_[i] = act_d[i]; lsum[i] = act_sum[i]; }
    }
    __syncthreads();


    float best[S];
    int best_i[S];
#pragma unroll
    for (int q = 0; q < S; q++) { best[q] = -INFINITY; best_i[q] = 0x7fffffff; }
#pragma unroll
    for (int jb = 0; jb < R; jb += RB) {
    if (jb > 0) request_rows(jb);
    float acc[RB][S];                           // a batch's sums are reduced and stored before the next batch starts
#pragma unroll
    for (int j = 0; j < RB; j++)
#pragma unroll
        for (int q = 0; q < S; q++) acc[j][q] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (F16W) {
            const int e = c * 512 + lane * 8;
            if (e < d) {
#pragma unroll
                for (int q = 0; q < S; q++) {
                    float4 a0, a1;
                    if constexpr (lds_f) { const float* row = lf + (size_t)q * d + e; a0 = *(const float4*)row; a1 = *(const float4*)(row + 4); }
                    else { const float* row = act_f + (size_t)q * d + e; a0 = *(const float4*)row; a1 = *(const float4*)(row + 4); }
                    const float fa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                    for (int j = 0; j < RB; j++) {
                        const unsigned u[4] = {wq[j][c].x, wq[j][c].y, wq[j][c].z, wq[j][c].w};
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            acc[j][q] += h2f((uint16_t)(u[i] & 0xffffu)) * fa[2 * i];
                            acc[j][q] += h2f((uint16_t)(u[i] >> 16)) * fa[2 * i + 1];
                        }
                    }
                    // (many sequences: keep hipcc from hoisting every sequence's LDS reads ahead of the arithmetic --
                    // 256 live values at 8 sequences, i.e. spills)
                    if (S >= 8) __builtin_amdgcn_sched_barrier(0);
                }
            }
            continue;
        }
        const int b = c * 64 + lane;
        const bool in = b < nb;
        const int bs = in ? b : 0;
        // this lane's block of all S sequences, then every weight row against them: the weight
        // block's nibbles are split once per row, not once per (row, sequence)
        int av[S][8], asum[S];
        float ad[S];
#pragma unroll
        for (int q = 0; q < S; q++) {
            const int4* ap = (const int4*)(lq + (size_t)q * d + (size_t)bs * 32);
            const int4 a0 = ap[0], a1 = ap[1];
            av[q][0] = a0.x; av[q][1] = a0.y; av[q][2] = a0.z; av[q][3] = a0.w;
            av[q][4] = a1.x; av[q][5] = a1.y; av[q][6] = a1.z; av[q][7] = a1.w;
            ad[q] = in ? ld_[q * nb + bs] : 0.f;
            asum[q] = in ? lsum[q * nb + bs] : 0;
        }
#pragma unroll
        for (int j = 0; j < RB; j++) {
            const float dw = h2f(wd[j][c]);
            if (WT == GTEN_Q4) {
                const Q4Unpacked u = q4_unpack(wq[j][c]);
#pragma unroll
                for (int q = 0; q < S; q++) acc[j][q] += (float)dot_q8_q4_unpacked(av[q], asum[q], u) * (ad[q] * dw);
            } else {
#pragma unroll
                for (int q = 0; q < S; q++) acc[j][q] += (float)dot_q8_q8_block(av[q], wq[j][c], wq1[j][c]) * (ad[q] * dw);
            }
        }
    }
#pragma unroll
    for (int jj = 0; jj < RB; jj++) {
        const int j = jb + jj;
#pragma unroll
        for (int q = 0; q < S; q++) {
            const float v = wave_sum(acc[jj][q]);
            if (EPI == EPI_SILUMUL) {
                if (lane == 0) res[q * 64 + (wid >> 2) * 32 + (wid & 3) * R + j] = v;
            } else {
                if (lane == 0 && r0 + j < total) a.out[(size_t)q * a.raw_stride + r0 + j] = v;
                if (a.best_val && r0 + j < total && v > best[q]) { best[q] = v; best_i[q] = r0 + j; }
            }
        }
    }
    }
    if (EPI == EPI_RAW && a.best_val && lane == 0) {
#pragma unroll
        for (int q = 0; q < S; q++) {
            a.best_val[(size_t)q * a.best_stride + blockIdx.x * NW + wid] = best[q];
            a.best_idx[(size_t)q * a.best_stride + blockIdx.x * NW + wid] = best_i[q];
        }
    }
    if (EPI == EPI_SILUMUL) {
        // silu(write(gate)) * write(up), written in the activation dtype, one wave per sequence
        __syncthreads();
        if (wid < S) {
            const int q = wid, e = lane & 31;
            const int nbf = rows0 >> 5;
            float g = act_round32(res[q * 64 + e], F16W);
            g = act_round32(g / (1.0f + expf(-g)), F16W);
            const float u = act_round32(res[q * 64 + 32 + e], F16W);
            const float v = g * u;
            if (F16W) {
                if (lane < 32) a.out_f[(size_t)q * rows0 + (size_t)blockIdx.x * 32 + e] = h2f(f2h(v));
            } else {
                const Q8Scale sc = q8_scale_from_absmax(max32(fabsf(v)));
                const int qv = q8_round(v, sc.scale);
                const int qs = sum32_i(qv);
                if (lane < 32) {
                    a.out_q[(size_t)q * rows0 + (size_t)blockIdx.x * 32 + e] = (int8_t)qv;
                    if (e == 0) { a.out_d[(size_t)q * nbf + blockIdx.x] = sc.ddeq; a.out_sum[(size_t)q * nbf + blockIdx.x] = qs; }
                }
            }
        }
    }
}
