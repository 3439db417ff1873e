// gten_decode.hip -- single-token decode fast path.
//
// Same arithmetic, same rounding points and the same bytes in the K/V caches as
// running the ten operators of gten_ops.hip one by one for start_pos = n-1
// (gten/modules.cpp:193-254 order), but organised for a launch-bound batch-1
// decode on MI355X:
//
//   * 6 launches per transformer block instead of 16, every one a wide grid;
//   * every W.x kernel writes its raw f32 dot products; the element-wise chain
//     that follows in the reference (write in activation dtype -> residual add
//     -> RMSNorm -> write ...) is recomputed by each workgroup of the NEXT
//     kernel in its prologue, entirely on chip (2048 elements, in registers),
//     so no kernel waits on a 1-workgroup element-wise launch; the one long
//     chain (silu(gate)*up over 5632 elements) runs once per 32-wide slice in
//     the gate/up kernel's EPILOGUE and is handed to the down projection staged;
//   * the step's position n comes from device memory, so one captured
//     hipGraph replays for every n (no per-step host work beyond one launch);
//   * attention is split over (head, 256-position chunk); probabilities are
//     still rounded to the activation dtype block by block with the GLOBAL max
//     and sum, as the reference does when it stores a probability row
//     (gten/ops.h:972-997), by separating the score pass from the p.V pass.
//
// Per block:  qkv -> attn_score -> attn_pv -> o -> gateup -> down
// then:       head (final norm + lm_head) -> argmax
#include "gten_dev.h"
#include "gten_rt.h"

#include <vector>
#include <algorithm>

using namespace gtd;

extern __shared__ __attribute__((aligned(16))) uint8_t g_smem[];

#define DEC_CHUNK 256            // attention positions per workgroup
#define DEC_MAX_LANES 4           // lanes of up to 64 sequences in one decoder (256 sequences)
#define DEC_ATT_MAXCH 8          // chunk partials / statistics a consumer requests up front (2048 positions)
#define GEMVM_F16_LDS_LIMIT 65536  // multi-sequence f16 inputs are staged in LDS up to this many bytes (8 sequences x 2048 x f32; the launchers raise
                                   // the kernel's dynamic LDS limit past the 64 KiB default where needed)

// launches of a decode step can be restricted to one kernel family (gten_hip_decoder_time_family)
static int g_only_family = -1;
#define DEC_LAUNCH(tag, kernel, grid, block, smem, ...)                                        \
    do {                                                                                       \
        if (g_only_family < 0 || g_only_family == (tag)) GTR_LAUNCH(tag, kernel, grid, block, smem, __VA_ARGS__); \
    } while (0)

// a launch whose kernel takes seven preloadable 64-bit words ahead of its argument struct (hot arguments, below)
#define DEC_LAUNCH_HOT(tag, kernel, grid, block, smem, hw, a) \
    DEC_LAUNCH(tag, kernel, grid, block, smem, (hw).w[0], (hw).w[1], (hw).w[2], (hw).w[3], (hw).w[4], (hw).w[5], (hw).w[6], a)

struct DecStep {
    int n;                        // context length of this step; the new row is n-1
    int advance;                  // bit 0: the argmax kernel bumps n afterwards (free-running replay);
                                  // bit 1: ... and stores its argmax as the NEXT input token (greedy generation without the host)
    int stop;                     // > 0: n is not bumped past it -- a slot whose run ends inside a slice of shared steps repeats
                                  // its last step (the same row, the same bytes) instead of cutting the slice short for everybody
};

// ---------------------------------------------------------------- LDS stage
//
// Every prologue leaves the W.x input vector staged in LDS: ActQ8 form for Q8
// activations, an f32 row (of exact fp16 values) for f16 activations.

struct ActStage {
    float* row;                   // d floats
    float* red;                   // 16 floats
    ActQ8 q8;                     // valid when adtype == Q8
};

__device__ __forceinline__ ActStage carve_stage(int d)
{
    ActStage s;
    s.red = (float*)g_smem;
    s.row = (float*)(g_smem + 64);
    s.q8 = actq8_carve(g_smem + 64 + (size_t)d * 4, d >> 5);
    return s;
}
static size_t stage_bytes(int d) { return 64 + (size_t)d * 4 + (size_t)(d >> 5) * 40; }

enum { PRO_EMBED = 0, PRO_RESID = 1, PRO_ATT = 2, PRO_ATTW = 3, PRO_ACTQ8 = 4 };
// PRO_ATT: the chunk partials of the two-pass attention are final, the prologue adds them; PRO_ATTW: the one-pass
// attention (k_dec_attn_one64) left chunk-local partials and statistics, the prologue joins them with their weights
enum { EPI_RAW = 0, EPI_SILUMUL = 1, EPI_STAGE = 2, EPI_STAGE_FRAG = 3 };

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
__global__ __launch_bounds__(512) void k_dec_mmvh(const uint16_t* __restrict__ a_ah, const void* __restrict__ a_w0, float* __restrict__ a_out,
                                                  const int a_d_in, const int a_d_out0, const int a_out_cols, const int a_S, const int a_n_mats,
                                                  const MmvRest rest)
{
    // (more than four row tiles: activation fragments two blocks ahead instead of four -- registers)
    constexpr int SP = 16 * RT, FR = 16 * FT, CB = (RT > 4) ? ((FT > 2) ? 1 : 2) : 4;
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
    constexpr int SP = 16 * RT, CBK = 4;
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

// ------------------------------------------------------------- attention

struct AttnArgs {
    const DecStep* step;
    const float* qkv_raw;         // [E | KV | KV] raw projections of the new row
    uint8_t* kcache; uint8_t* vcache; size_t kv_pitch;
    float* scores;                // [n_heads][max_ctx]
    float* stats;                 // [n_heads][n_chunks][2] (max, sum of exp)
    float* att_part;              // [n_heads][n_chunks][d_head]
    const float2* rope;
    const float2* rope_now;       // [seq][d_head / 2]: the rotation of each sequence's CURRENT position, left by the step's first
                                  // launch (PRO_EMBED) so that the score kernels can request it without knowing the position
    int adtype, n_heads, n_kv, d_head, max_ctx, n_chunks, n_embd;
    int grp_shift1;               // log2(n_heads / n_kv) + 1 when that ratio is a power of two, else 0 (set by the launchers)
    // multi-sequence decode: blockIdx.z = sequence; its caches come from a device table
    // [seq][layer][k|v], its scratch rows lie `*_stride` elements apart
    const void* const* kv_tab; int layer, n_layers;
    int qkv_stride, scores_stride, stats_stride, part_stride;
    int qkv_plane;                // grouped kernels: floats to the second K-split plane of qkv_raw (0: a single plane)
};

// The cache pointers may come out of the device table (multi-sequence), so hipcc only knows them as generic pointers
// and would use FLAT loads -- which count on the LDS counter as well and force full vmcnt(0) waits (FLAT may return
// out of order).  They always point to device memory: say so.
typedef const unsigned __attribute__((address_space(1)))* gmem_u32;
__device__ __forceinline__ gmem_u32 as_global(const void* p) { return (gmem_u32)(uintptr_t)p; }

// per-sequence view of the arguments (identity for single-sequence launches)
__device__ __forceinline__ AttnArgs attn_for_seq(const AttnArgs& a, int seq)
{
    AttnArgs t = a;
    t.step = a.step + seq;
    t.qkv_raw = a.qkv_raw + (size_t)seq * a.qkv_stride;
    t.scores = a.scores + (size_t)seq * a.scores_stride;
    t.stats = a.stats + (size_t)seq * a.stats_stride;
    t.att_part = a.att_part + (size_t)seq * a.part_stride;
    t.rope_now = a.rope_now + (size_t)seq * (a.d_head >> 1);
    if (a.kv_tab) {
        t.kcache = (uint8_t*)a.kv_tab[((size_t)seq * a.n_layers + a.layer) * 2];
        t.vcache = (uint8_t*)a.kv_tab[((size_t)seq * a.n_layers + a.layer) * 2 + 1];
    }
    return t;
}

// write(raw) -> rope -> write, for one head vector of d_head (32 or 64) elements
// held by lanes [0, d_head) of wave 0; returns the final f32 value (exact storage
// value) and, for Q8, leaves quants/deltas in qi8/qd/qd16.  The rotate-half
// partner (j, j + d_head/2) lives d_head/2 lanes away: one xor-shuffle.
// gten/modules.cpp:196-201 + gten/ops.h:714-755
// (cs = the rotation of this lane's pair, rope[pos * d_head/2 + (t & (d_head/2 - 1))], loaded by the caller)
__device__ __forceinline__ float head_prep_cs(float raw, bool act, bool do_rope, const float2 cs, int d_head, int adtype,
                                              int8_t* qi8, float* qd, uint16_t* qd16)
{
    const int t = threadIdx.x & 63;
    float v = act ? raw : 0.f;
    // Linear output written in the activation dtype
    if (adtype == GTEN_Q8) {
        const Q8Scale sc = q8_scale_from_absmax(nn_max32(fabsf(v)));
        v = (float)q8_round(v, sc.scale) * sc.ddeq;
    } else {
        v = h2f(f2h(v));
    }
    if (do_rope) {
        const int half = d_head >> 1;
        const float other = __shfl_xor(v, half, 64);
        const bool lo = (t & half) == 0;
        const float x0 = lo ? v : other, x1 = lo ? other : v;
        v = lo ? (x0 * cs.x - x1 * cs.y) : (x0 * cs.y + x1 * cs.x);
        if (!act) v = 0.f;
    }
    if (adtype == GTEN_Q8) {
        const Q8Scale sc = q8_scale_from_absmax(nn_max32(fabsf(v)));
        const int qv = q8_round(v, sc.scale);
        if (act) {
            qi8[t] = (int8_t)qv;
            if ((t & 31) == 0) { qd[t >> 5] = sc.ddeq; qd16[t >> 5] = sc.d16; }
        }
        v = (float)qv * sc.ddeq;
    } else {
        v = h2f(f2h(v));
    }
    return v;
}

__device__ __forceinline__ float head_prep(float raw, bool act, bool do_rope, int pos, int d_head, int adtype,
                                           const float2* __restrict__ rope, int8_t* qi8, float* qd, uint16_t* qd16)
{
    const int half = d_head >> 1;
    float2 cs = make_float2(1.f, 0.f);
    if (do_rope) cs = rope[(size_t)pos * half + ((threadIdx.x & 63) & (half - 1))];
    return head_prep_cs(raw, act, do_rope, cs, d_head, adtype, qi8, qd, qd16);
}

// pass 1: q.k scores of one head over one 256-position chunk, chunk max and sum
// of exponentials; also quantizes + RoPEs the new q/k/v rows and appends k, v to
// the caches (one designated workgroup per kv head).  gten/ops.h:930-970
__global__ __launch_bounds__(256) void k_dec_attn_score(const AttnArgs a)
{
    const int n = a.step->n, pos = n - 1;
    const int h = blockIdx.x, chunk = blockIdx.y, c0 = chunk * DEC_CHUNK;
    if (c0 >= n) return;
    const int dh = a.d_head, nblk = dh >> 5;
    const int grp = a.n_heads / a.n_kv, g = h / grp;
    const int kv_dim = a.n_kv * dh;
    const size_t head_bytes = (a.adtype == GTEN_Q8) ? (size_t)nblk * GTEN_Q8_BYTES : (size_t)dh * 2;

    float* red = (float*)g_smem;                 // 16
    float* qf = red + 16 + dh;                   // dh   (f16 mode: q values; Q8: unused)
    float* kf = qf + dh;                         // dh   new k row values
    float* qd = kf + dh;                         // 8
    float* kd = qd + 8;                          // 8
    uint16_t* d16 = (uint16_t*)(kd + 8);         // 16 halves
    int8_t* qi8 = (int8_t*)(d16 + 16);           // dh
    int8_t* ki8 = qi8 + dh;                      // dh
    int8_t* vi8 = ki8 + dh;                      // dh

    const bool has_new = (pos >= c0) && (pos < c0 + DEC_CHUNK);
    const bool writer = has_new && (h % grp == 0);
    // q (and, where needed, the new k / v rows) -- wave 0 only, d_head <= 64 lanes
    if (threadIdx.x < 64) {
        const int t = threadIdx.x;
        const bool act = t < dh;
        float v = head_prep(act ? a.qkv_raw[h * dh + t] : 0.f, act, true, pos, dh, a.adtype, a.rope, qi8, qd, d16);
        if (act) qf[t] = v;
        if (has_new) {
            v = head_prep(act ? a.qkv_raw[a.n_embd + g * dh + t] : 0.f, act, true, pos, dh, a.adtype, a.rope, ki8, kd, d16 + 4);
            if (act) kf[t] = v;
            if (writer && act) {
                uint8_t* krow = a.kcache + (size_t)pos * a.kv_pitch + (size_t)g * head_bytes;
                if (a.adtype == GTEN_Q8) {
                    uint8_t* blk = krow + (size_t)(t >> 5) * GTEN_Q8_BYTES;
                    blk[2 + (t & 31)] = (uint8_t)ki8[t];
                    if ((t & 31) == 0) *(uint16_t*)blk = d16[4 + (t >> 5)];
                } else {
                    ((uint16_t*)krow)[t] = f2h(v);
                }
            }
            if (writer) {
                v = head_prep(act ? a.qkv_raw[a.n_embd + kv_dim + g * dh + t] : 0.f, act, false, pos, dh, a.adtype, a.rope, vi8, kd + 4, d16 + 8);
                if (act) {
                    uint8_t* vrow = a.vcache + (size_t)pos * a.kv_pitch + (size_t)g * head_bytes;
                    if (a.adtype == GTEN_Q8) {
                        uint8_t* blk = vrow + (size_t)(t >> 5) * GTEN_Q8_BYTES;
                        blk[2 + (t & 31)] = (uint8_t)vi8[t];
                        if ((t & 31) == 0) *(uint16_t*)blk = d16[8 + (t >> 5)];
                    } else {
                        ((uint16_t*)vrow)[t] = f2h(v);
                    }
                }
            }
        }
    }
    __syncthreads();

    const float scale = 1.0f / sqrtf((float)dh);
    const int c = c0 + threadIdx.x;
    float sc = -INFINITY;
    if (c < n) {
        float acc = 0.f;
        if (a.adtype == GTEN_Q8) {
            const int* qi = (const int*)qi8;
            if (c == pos) {
                const int* ki = (const int*)ki8;
                for (int b = 0; b < nblk; b++) {
                    int isum = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) isum = dot4(qi[b * 8 + j], ki[b * 8 + j], isum);
                    acc += (float)isum * (qd[b] * kd[b]);
                }
            } else {
                const uint8_t* kp = a.kcache + (size_t)c * a.kv_pitch + (size_t)g * head_bytes;
                for (int b = 0; b < nblk; b++) {
                    const uint16_t* kw = (const uint16_t*)(kp + (size_t)b * GTEN_Q8_BYTES);
                    int isum = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int kv4 = (int)((unsigned)kw[1 + 2 * j] | ((unsigned)kw[2 + 2 * j] << 16));
                        isum = dot4(qi[b * 8 + j], kv4, isum);
                    }
                    acc += (float)isum * (qd[b] * h2f(kw[0]));
                }
            }
        } else {
            if (c == pos) {
                for (int e = 0; e < dh; e++) acc += qf[e] * kf[e];
            } else {
                const uint16_t* k16 = (const uint16_t*)(a.kcache + (size_t)c * a.kv_pitch + (size_t)g * head_bytes);
                for (int e = 0; e < dh; e++) acc += qf[e] * h2f(k16[e]);
            }
        }
        sc = acc * scale;
        a.scores[(size_t)h * a.max_ctx + c] = sc;
    }
    const float mx = block_max(sc, red);
    const float ex = (c < n) ? expf(sc - mx) : 0.f;
    const float sm = block_sum(ex, red);
    if (threadIdx.x == 0) {
        a.stats[((size_t)h * a.n_chunks + chunk) * 2 + 0] = mx;
        a.stats[((size_t)h * a.n_chunks + chunk) * 2 + 1] = sm;
    }
}

// pass 2: probabilities with the global max / sum, rounded to the activation
// dtype in 32-blocks along the context (partial tail at n), times V.
// gten/ops.h:972-997, 1046-1089
__global__ __launch_bounds__(256) void k_dec_attn_pv(const AttnArgs a)
{
    const int n = a.step->n;
    const int h = blockIdx.x, chunk = blockIdx.y, c0 = chunk * DEC_CHUNK;
    if (c0 >= n) return;
    const int dh = a.d_head, nblk = dh >> 5;
    const int grp = a.n_heads / a.n_kv, g = h / grp;
    const size_t head_bytes = (a.adtype == GTEN_Q8) ? (size_t)nblk * GTEN_Q8_BYTES : (size_t)dh * 2;
    const int nch = (n + DEC_CHUNK - 1) / DEC_CHUNK;

    float* p = (float*)g_smem;                   // 256
    float* part = p + DEC_CHUNK;                 // 256

    float M = -INFINITY;
    for (int j = 0; j < nch; j++) M = fmaxf(M, a.stats[((size_t)h * a.n_chunks + j) * 2]);
    float S = 0.f;
    for (int j = 0; j < nch; j++)
        S += a.stats[((size_t)h * a.n_chunks + j) * 2 + 1] * expf(a.stats[((size_t)h * a.n_chunks + j) * 2] - M);

    const int c = c0 + threadIdx.x;
    const int len = min(DEC_CHUNK, n - c0);
    p[threadIdx.x] = (c < n) ? expf(a.scores[(size_t)h * a.max_ctx + c] - M) / S : 0.f;
    round_row_inplace(p, a.adtype, len);         // thread t only touches p[t]: no barrier needed before
    __syncthreads();

    const int ngrp = blockDim.x / dh;
    const int e = threadIdx.x % dh, cg = threadIdx.x / dh;
    float acc = 0.f;
    if (cg < ngrp) {
        const uint8_t* vbase = a.vcache + (size_t)g * head_bytes;
        for (int cl = cg; cl < len; cl += ngrp) acc += p[cl] * load_elem(vbase + (size_t)(c0 + cl) * a.kv_pitch, a.adtype, e);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < dh) {
        float o = 0.f;
        for (int gi = 0; gi < ngrp; gi++) o += part[gi * dh + threadIdx.x];
        a.att_part[((size_t)h * a.n_chunks + chunk) * dh + threadIdx.x] = o;
    }
}

// ---- d_head = 64 specialisations of the two attention passes (TinyLlama's shape)
//
// Same arithmetic as the generic kernels above; what changes is how the cache
// rows travel: every thread requests all of its K (or V) bytes with plain
// dword loads up front (a kv-head slice is 68 bytes = 17 dwords in Q8, 128
// bytes in f16, always 4-byte aligned), so a pass costs one memory latency
// instead of one per cached row.

// Hot arguments of the two single-sequence attention launches (see GemvHot): seven preloadable 64-bit words.
//   score: qkv_raw | rope_now | kcache | step | kv_pitch, max_ctx | n_embd, n_heads + (n_kv << 16) | scores
//   p.V:   scores  | stats    | vcache | step | kv_pitch, max_ctx | n_chunks, n_heads + (n_kv << 16) | att_part
// MULTI launches (blockIdx.z = sequence) take everything from the struct through attn_for_seq.
struct AttnHotWords { unsigned long long w[7]; };

template <int ADT, bool MULTI>
__global__ __launch_bounds__(256) void k_dec_attn_score64(const unsigned long long h0, const unsigned long long h1, const unsigned long long h2,
                                                         const unsigned long long h3, const unsigned long long h4, const unsigned long long h5,
                                                         const unsigned long long h6, const AttnArgs a0)
{
    AttnArgs a = MULTI ? attn_for_seq(a0, blockIdx.z) : a0;
    if (!MULTI) {
        a.qkv_raw = from_word<float>(h0); a.rope_now = from_word<float2>(h1); a.kcache = (uint8_t*)from_word<uint8_t>(h2);
        a.step = from_word<DecStep>(h3); a.kv_pitch = (size_t)(unsigned)(h4 & 0xffffffffull); a.max_ctx = (int)(h4 >> 32);
        a.n_embd = (int)(unsigned)(h5 & 0xffffffffull); a.n_heads = (int)((h5 >> 32) & 0xffu); a.n_kv = (int)((h5 >> 40) & 0xffu); a.grp_shift1 = (int)(h5 >> 48);
        a.scores = (float*)from_word<float>(h6);
    }
    constexpr int dh = 64, nblk = 2;
    constexpr int NW = (ADT == GTEN_Q8) ? 17 : 32;     // dwords per kv-head slice
    // grid = (chunk, head, sequence): consecutive workgroup ids -- which the dispatcher deals round-robin to the 8 XCDs --
    // are the chunks of ONE head, so with 8 chunks every XCD reads its own eighth of the K / V history once instead of
    // every XCD fetching all of it (PMC: 4.8 MB -> per launch before the swap, against 0.56 MB of cache)
    const int h = blockIdx.y, chunk = blockIdx.x, c0 = chunk * DEC_CHUNK;
    // (heads per kv head: a shift when it is a power of two -- two integer divisions ahead of the first request otherwise)
    const int grp = a.grp_shift1 ? (1 << (a.grp_shift1 - 1)) : a.n_heads / a.n_kv, g = a.grp_shift1 ? (h >> (a.grp_shift1 - 1)) : h / grp;
    const int kv_dim = a.n_kv * dh;
    const size_t head_bytes = (ADT == GTEN_Q8) ? (size_t)nblk * GTEN_Q8_BYTES : (size_t)dh * 2;

    float* red = (float*)g_smem;                 // 16
    float* qf = red + 16 + dh;                   // dh
    float* kf = qf + dh;                         // dh
    float* qd = kf + dh;                         // 8
    float* kd = qd + 8;                          // 8
    uint16_t* d16 = (uint16_t*)(kd + 8);         // 16 halves
    int8_t* qi8 = (int8_t*)(d16 + 16);           // dh
    int8_t* ki8 = qi8 + dh;                      // dh
    int8_t* vi8 = ki8 + dh;                      // dh

    // ---- everything is requested before the context length is known: this thread's cached K row (rows past the
    //      context are readable -- the caches span max_ctx -- and unused; the row AT the new position is being
    //      written by this very launch and is not used either: that score comes from the new k row on chip), the
    //      raw projection this wave turns into a head vector (wave 0: q, 1: new k row, 2: new v row), its rotation
    const int c = c0 + threadIdx.x;
    const int t = threadIdx.x & 63, pw = threadIdx.x >> 6;
    const int roff = (pw == 1) ? a.n_embd + g * dh : (pw == 2) ? a.n_embd + kv_dim + g * dh : h * dh;
    float raw = a.qkv_raw[roff + t];
    if (MULTI) {
        // second K-split plane of the projections (wide path, k_dec_mmv): requested unconditionally (plane 0: the same word)
        const float raw2 = a.qkv_raw[a.qkv_plane + roff + t];
        raw += a.qkv_plane ? raw2 : 0.f;
    }
    const float2 rot = a.rope_now[t & 31];
    __builtin_amdgcn_sched_barrier(0);            // these two come back first (vmcnt is in order): the head vectors are built while the K rows fly
    const int cs = min(c, a.max_ctx - 1);
    const gmem_u32 kp = as_global(a.kcache + (size_t)g * head_bytes) + (unsigned)cs * (unsigned)(a.kv_pitch >> 2);
    unsigned kw[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) kw[j] = kp[j];
    __builtin_amdgcn_sched_barrier(0);
    const int n = a.step->n, pos = n - 1;
    if (c0 >= n) return;

    const bool has_new = (pos >= c0) && (pos < c0 + DEC_CHUNK);
    const bool writer = has_new && (h == g * grp);
    // the three new head vectors are independent: one wave each (0: q | 1: new k row | 2: new v row), one copy of
    // the code (the k / v waves also run where their row is not needed: it only lands in this workgroup's scratch)
    if (pw < 3) {
        int8_t* dq = (pw == 0) ? qi8 : (pw == 1) ? ki8 : vi8;
        float* dd = (pw == 0) ? qd : (pw == 1) ? kd : kd + 4;
        const float v = head_prep_cs(raw, true, pw != 2, rot, dh, ADT, dq, dd, d16 + 4 * pw);
        if (pw == 0) qf[t] = v;
        if (pw == 1) kf[t] = v;
        if (pw >= 1 && writer) {
            uint8_t* row = ((pw == 1) ? a.kcache : a.vcache) + (size_t)pos * a.kv_pitch + (size_t)g * head_bytes;
            if (ADT == GTEN_Q8) {
                uint8_t* blk = row + (size_t)(t >> 5) * GTEN_Q8_BYTES;
                store_global<uint8_t>(blk + 2 + (t & 31), (uint8_t)dq[t]);
                if ((t & 31) == 0) store_global<uint16_t>(blk, d16[4 * pw + (t >> 5)]);
            } else {
                store_global<uint16_t>((uint16_t*)row + t, f2h(v));
            }
        }
    }
    __syncthreads();

    // every lane scores its cached row (the lane AT the new position holds unused bytes there); the chunk that
    // contains the new position then scores the new k row from the chip -- uniform control flow, same arithmetic
    const float scale = 1.0f / sqrtf((float)dh);
    float acc = 0.f;
    if (ADT == GTEN_Q8) {
        const int* qi = (const int*)qi8;
        // slice bytes: [d0 | q0 x32 | d1 | q1 x32]; q0 straddles dwords by 2 bytes
        int isum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) isum = dot4(qi[j], (int)__builtin_amdgcn_alignbit(kw[j + 1], kw[j], 16), isum);
        acc += (float)isum * (qd[0] * h2f((uint16_t)(kw[0] & 0xffffu)));
        isum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) isum = dot4(qi[8 + j], (int)kw[9 + j], isum);
        acc += (float)isum * (qd[1] * h2f((uint16_t)(kw[8] >> 16)));
    } else {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            acc += qf[2 * j] * h2f((uint16_t)(kw[j] & 0xffffu));
            acc += qf[2 * j + 1] * h2f((uint16_t)(kw[j] >> 16));
        }
    }
    if (has_new) {
        float accn = 0.f;
        if (ADT == GTEN_Q8) {
            const int* qi = (const int*)qi8;
            const int* ki = (const int*)ki8;
#pragma unroll
            for (int b = 0; b < nblk; b++) {
                int isum = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) isum = dot4(qi[b * 8 + j], ki[b * 8 + j], isum);
                accn += (float)isum * (qd[b] * kd[b]);
            }
        } else {
            for (int e = 0; e < dh; e++) accn += qf[e] * kf[e];
        }
        if (c == pos) acc = accn;
    }
    float sc = -INFINITY;
    if (c < n) {
        sc = acc * scale;
        a.scores[(size_t)h * a.max_ctx + c] = sc;
    }
    const float mx = block_max_n<4>(sc, red);               // red: first use; the sum takes its own words
    const float ex = (c < n) ? expf(sc - mx) : 0.f;
    const float sm = block_sum_n<4>(ex, red + 4);
    if (threadIdx.x == 0) {
        a.stats[((size_t)h * a.n_chunks + chunk) * 2 + 0] = mx;
        a.stats[((size_t)h * a.n_chunks + chunk) * 2 + 1] = sm;
    }
}

// Softmax statistics of a head for the p.V passes: global max M and S = sum_j l_j * exp(m_j - M), chunks in order.
// Up to DEC_ATT_MAXCH chunks the (max, sum) pairs are ONE load per lane (lane j & 7 holds chunk j; requested by the
// caller at kernel entry as `st`), the eight exponentials run in eight lanes at once, and the sum is taken in chunk
// order through readlanes -- the same values and the same order as the sequential loop it replaces (x + 0 == x).
__device__ __forceinline__ void softmax_stats8(const float2 st, int nch, float& M, float& S)
{
    const int j = threadIdx.x & 7;
    float m = (j < nch) ? st.x : -INFINITY;
    m = quad_max(m);
    M = fmaxf(m, dpp_mov<0x141>(m));                         // row_half_mirror: all 8 lanes of the group
    const float t = (j < nch) ? st.y * expf(st.x - M) : 0.f;
    S = 0.f;
#pragma unroll
    for (int q = 0; q < DEC_ATT_MAXCH; q++) S += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), q));
}

template <int ADT, bool MULTI>
__global__ __launch_bounds__(256) void k_dec_attn_pv64(const unsigned long long h0, const unsigned long long h1, const unsigned long long h2,
                                                      const unsigned long long h3, const unsigned long long h4, const unsigned long long h5,
                                                      const unsigned long long h6, const AttnArgs a0)
{
    AttnArgs a = MULTI ? attn_for_seq(a0, blockIdx.z) : a0;
    if (!MULTI) {
        a.scores = (float*)from_word<float>(h0); a.stats = (float*)from_word<float>(h1); a.vcache = (uint8_t*)from_word<uint8_t>(h2);
        a.step = from_word<DecStep>(h3); a.kv_pitch = (size_t)(unsigned)(h4 & 0xffffffffull); a.max_ctx = (int)(h4 >> 32);
        a.n_chunks = (int)(unsigned)(h5 & 0xffffffffull); a.n_heads = (int)((h5 >> 32) & 0xffu); a.n_kv = (int)((h5 >> 40) & 0xffu); a.grp_shift1 = (int)(h5 >> 48);
        a.att_part = (float*)from_word<float>(h6);
    }
    constexpr int dh = 64;
    constexpr int NW = (ADT == GTEN_Q8) ? 17 : 32;     // dwords per kv-head slice
    // grid = (chunk, head, sequence): consecutive workgroup ids -- which the dispatcher deals round-robin to the 8 XCDs --
    // are the chunks of ONE head, so with 8 chunks every XCD reads its own eighth of the K / V history once instead of
    // every XCD fetching all of it (PMC: 4.8 MB -> per launch before the swap, against 0.56 MB of cache)
    const int h = blockIdx.y, chunk = blockIdx.x, c0 = chunk * DEC_CHUNK;
    // (heads per kv head: a shift when it is a power of two -- two integer divisions ahead of the first request otherwise)
    const int grp = a.grp_shift1 ? (1 << (a.grp_shift1 - 1)) : a.n_heads / a.n_kv, g = a.grp_shift1 ? (h >> (a.grp_shift1 - 1)) : h / grp;
    const size_t head_bytes = (ADT == GTEN_Q8) ? (size_t)2 * GTEN_Q8_BYTES : (size_t)dh * 2;

    float* p = (float*)g_smem;                   // 256
    float* part = p + DEC_CHUNK;                 // 256
    unsigned* vl = (unsigned*)(part + DEC_CHUNK);// DEC_CHUNK * NW dwords: the chunk's V slices, row-major

    // ---- everything this workgroup reads is requested before the context length is even known: this thread's
    //      score, the head's chunk statistics (the stats array has DEC_ATT_MAXCH chunks of slack), then the whole
    //      V chunk: dword idx -> (row idx / NW, word idx % NW).  Rows past the context are readable (the caches
    //      span max_ctx) and never used.
    const int c = c0 + threadIdx.x;
    const float sc_raw = a.scores[(size_t)h * a.max_ctx + min(c, a.max_ctx - 1)];
    const float2 st = ((const float2*)a.stats)[(size_t)h * a.n_chunks + (threadIdx.x & 7)];
    __builtin_amdgcn_sched_barrier(0);            // these two come back first (vmcnt is in order): the softmax math starts on them
    unsigned vw[NW];
    {
        // idx = t + 256 k -> (row, word) = (idx / NW, idx % NW), stepped without a division: 256 = (256 / NW) NW + 256 % NW
        int row = (int)threadIdx.x / NW, w = (int)threadIdx.x % NW;
        const gmem_u32 vbase = as_global(a.vcache + (size_t)g * head_bytes);
        const unsigned pitch_w = (unsigned)(a.kv_pitch >> 2);   // rows are 4-byte aligned (68 / 128-byte head slices);
        const int last = a.max_ctx - 1 - c0;                    // a cache is far below 4 GiB: 32-bit word offsets
#pragma unroll
        for (int k = 0; k < NW; k++) {
            vw[k] = vbase[(unsigned)(c0 + min(row, last)) * pitch_w + (unsigned)w];
            row += 256 / NW; w += 256 % NW;
            if (w >= NW) { w -= NW; row++; }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int n = a.step->n;
    if (c0 >= n) return;
    const int nch = (n + DEC_CHUNK - 1) / DEC_CHUNK;
    const int len = min(DEC_CHUNK, n - c0);

    float M, S;
    softmax_stats8(st, nch, M, S);                // decoder_create: n_chunks <= DEC_ATT_MAXCH

    // probabilities, rounded to the activation dtype in registers (a Q8 block = 32 consecutive lanes; the partial
    // tail block takes its absmax over the live positions, zeros beyond: round_row_inplace's rule)
    float pr = (c < n) ? expf(sc_raw - M) / S : 0.f;
    if (ADT == GTEN_Q8) {
        const Q8Scale s8 = q8_scale_from_absmax(max32(fabsf(pr)));
        if (c < n) pr = (float)q8_round(pr, s8.scale) * s8.ddeq;
    } else {
        pr = h2f(f2h(pr));
    }
    p[threadIdx.x] = pr;
#pragma unroll
    for (int k = 0; k < NW; k++) vl[threadIdx.x + k * 256] = vw[k];
    __syncthreads();

    const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const uint8_t* vb = (const uint8_t*)vl;
    float acc = 0.f;
    if (ADT == GTEN_Q8) {
        const int qoff = (e < 32) ? 2 + e : 36 + (e - 32), doff = (e < 32) ? 0 : 34;
        if (len == DEC_CHUNK) {
            // a full chunk (every chunk but the last): eight terms per round, the LDS reads of round r + 1 issued
            // ahead of the arithmetic of round r (software pipeline) -- same terms, same order
            float pp[2][8];
            int qv[2][8];
            unsigned dv[2][8];
            auto fetch = [&](int r, int slot) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint8_t* row = vb + (size_t)(cg + 4 * (8 * r + u)) * 68;
                    pp[slot][u] = p[cg + 4 * (8 * r + u)];
                    qv[slot][u] = (int)(int8_t)row[qoff];
                    dv[slot][u] = *(const uint16_t*)(row + doff);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int r = 0; r < DEC_CHUNK / 32; r++) {
                if (r + 1 < DEC_CHUNK / 32) fetch(r + 1, (r + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; u++) acc += pp[r & 1][u] * ((float)qv[r & 1][u] * h2f((uint16_t)dv[r & 1][u]));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 8
            for (int cl = cg; cl < len; cl += 4) {
                const uint8_t* row = vb + (size_t)cl * 68;
                acc += p[cl] * ((float)(int8_t)row[qoff] * h2f(*(const uint16_t*)(row + doff)));
            }
        }
    } else {
        for (int cl = cg; cl < len; cl += 4) acc += p[cl] * h2f(((const uint16_t*)(vb + (size_t)cl * 128))[e]);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < dh) {
        float o = 0.f;
        for (int gi = 0; gi < 4; gi++) o += part[gi * dh + threadIdx.x];
        a.att_part[((size_t)h * a.n_chunks + chunk) * dh + threadIdx.x] = o;
    }
}

// ---- both passes in ONE launch with CHUNK-LOCAL softmax statistics (d_head 64; single-sequence decode and the
//      2 / 4-sequence GEMV path)
//
// The two launches above are separated only by the statistics of the whole row.  (Exchanging them inside one launch
// was built and measured in round 1: store -> drain -> atomic -> poll -> reload is ~3 dependent L2 round trips, 12.7 us
// against 4.9 + 6.4 us for the two launches.)  Here nothing is exchanged: a (head, chunk) workgroup normalises its
// probabilities by its OWN maximum m_c and sum l_c, rounds them to the activation dtype (gten/ops.h:972-997 -- Q8
// blocks of 32 along the context, partial tail block) and leaves o_c = p_c . V_c plus (m_c, l_c); the consumer (the o
// projection's prologue, PRO_ATTW) joins the chunks: out = sum_c w_c o_c, w_c = l_c exp(m_c - M) / sum_j l_j exp(m_j - M).
//   * one chunk (n <= 256): m_c = M, l_c = S, w_0 = 1 -- the bytes of the two-pass kernels and of the operator path;
//   * several chunks: a probability row is rounded against its chunk's scale instead of the row's.  The Q8 quants are
//     scale-free (q = round(p 127 / absmax)), so what moves is the fp16 rounding of the block delta (and for f16
//     activations the fp16 rounding of p itself): a relative 2^-11 per block, the size of the rounding the reference
//     itself applies at that point (deviation 4; inside the f16 / q8 / q4 bands, tests).
// The K rows and the V chunk are both requested at kernel entry, so the launch costs one memory latency.
template <int ADT, bool MULTI>
__global__ __launch_bounds__(256) void k_dec_attn_one64(const unsigned long long h0, const unsigned long long h1, const unsigned long long h2,
                                                       const unsigned long long h3, const unsigned long long h4, const unsigned long long h5,
                                                       const unsigned long long h6, const AttnArgs a0)
{
    AttnArgs a = MULTI ? attn_for_seq(a0, blockIdx.z) : a0;
    if (!MULTI) {
        // hot words: qkv_raw | rope_now | kcache | step | kv_pitch, max_ctx | n_embd, n_heads + (n_kv << 8) + (grp_shift1 << 16) | vcache
        a.qkv_raw = from_word<float>(h0); a.rope_now = from_word<float2>(h1); a.kcache = (uint8_t*)from_word<uint8_t>(h2);
        a.step = from_word<DecStep>(h3); a.kv_pitch = (size_t)(unsigned)(h4 & 0xffffffffull); a.max_ctx = (int)(h4 >> 32);
        a.n_embd = (int)(unsigned)(h5 & 0xffffffffull); a.n_heads = (int)((h5 >> 32) & 0xffu); a.n_kv = (int)((h5 >> 40) & 0xffu); a.grp_shift1 = (int)(h5 >> 48);
        a.vcache = (uint8_t*)from_word<uint8_t>(h6);
    }
    constexpr int dh = 64, nblk = 2;
    constexpr int NW = (ADT == GTEN_Q8) ? 17 : 32;     // dwords per kv-head slice
    // grid = (chunk, head, sequence): the chunks of one head go to different XCDs (see k_dec_attn_score64)
    const int h = blockIdx.y, chunk = blockIdx.x, c0 = chunk * DEC_CHUNK;
    const int grp = a.grp_shift1 ? (1 << (a.grp_shift1 - 1)) : a.n_heads / a.n_kv, g = a.grp_shift1 ? (h >> (a.grp_shift1 - 1)) : h / grp;
    const int kv_dim = a.n_kv * dh;
    const size_t head_bytes = (ADT == GTEN_Q8) ? (size_t)nblk * GTEN_Q8_BYTES : (size_t)dh * 2;

    float* red = (float*)g_smem;                 // 16
    float* qf = red + 16 + dh;                   // dh
    float* kf = qf + dh;                         // dh
    float* qd = kf + dh;                         // 8
    float* kd = qd + 8;                          // 8
    uint16_t* d16 = (uint16_t*)(kd + 8);         // 16 halves
    int8_t* qi8 = (int8_t*)(d16 + 16);           // dh
    int8_t* ki8 = qi8 + dh;                      // dh
    int8_t* vi8 = ki8 + dh;                      // dh
    float* p = (float*)(g_smem + 1152);          // 256 (the head-vector scratch above ends at byte 1120)
    float* part = p + DEC_CHUNK;                 // 256
    unsigned* vl = (unsigned*)(part + DEC_CHUNK);// DEC_CHUNK * NW dwords: the chunk's V slices, row-major

    // ---- every request before the context length is known (k_dec_attn_score64 / k_dec_attn_pv64 explain why each is
    //      safe): the raw projection this wave turns into a head vector, its rotation, this thread's cached K row,
    //      the whole V chunk.  The V row AT the new position is being written by this very launch: that term comes
    //      from the new v row on chip (below).
    const int c = c0 + threadIdx.x;
    const int t = threadIdx.x & 63, pw = threadIdx.x >> 6;
    const int roff = (pw == 1) ? a.n_embd + g * dh : (pw == 2) ? a.n_embd + kv_dim + g * dh : h * dh;
    float raw = a.qkv_raw[roff + t];
    if (MULTI) {
        const float raw2 = a.qkv_raw[a.qkv_plane + roff + t];
        raw += a.qkv_plane ? raw2 : 0.f;
    }
    const float2 rot = a.rope_now[t & 31];
    __builtin_amdgcn_sched_barrier(0);
    const int cs = min(c, a.max_ctx - 1);
    const unsigned pitch_w = (unsigned)(a.kv_pitch >> 2);
    const gmem_u32 kp = as_global(a.kcache + (size_t)g * head_bytes) + (unsigned)cs * pitch_w;
    // (round 4: a row's slice as 16-byte requests -- four and one dword for a Q8 slice of 68 bytes, eight for an f16 one.  The
    //  texture path spends its time per wave INSTRUCTION and per line touched, not per byte: 17 dword requests per thread and
    //  matrix were ~1.3 us of the launch; the slice is 4-byte aligned only, which gfx9 global loads take)
    unsigned kw[NW];
    unsigned vw[NW];
    {
        const gmem_u32 vp = as_global(a.vcache + (size_t)g * head_bytes) + (unsigned)cs * pitch_w;      // THIS thread's V row too
        typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
        for (int j = 0; j + 4 <= NW; j += 4) {
            const u4u kq = *(const __attribute__((address_space(1))) u4u*)(kp + j);
            const u4u vq = *(const __attribute__((address_space(1))) u4u*)(vp + j);
            kw[j] = kq.x; kw[j + 1] = kq.y; kw[j + 2] = kq.z; kw[j + 3] = kq.w;
            vw[j] = vq.x; vw[j + 1] = vq.y; vw[j + 2] = vq.z; vw[j + 3] = vq.w;
        }
#pragma unroll
        for (int j = NW & ~3; j < NW; j++) { kw[j] = kp[j]; vw[j] = vp[j]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int n = a.step->n, pos = n - 1;
    if (c0 >= n) return;
    const int len = min(DEC_CHUNK, n - c0);

    const bool has_new = (pos >= c0) && (pos < c0 + DEC_CHUNK);
    const bool writer = has_new && (h == g * grp);
    float vnew = 0.f;                             // wave 2: the new v row's element t (exact storage value)
    if (pw < 3) {
        int8_t* dq = (pw == 0) ? qi8 : (pw == 1) ? ki8 : vi8;
        float* dd = (pw == 0) ? qd : (pw == 1) ? kd : kd + 4;
        const float v = head_prep_cs(raw, true, pw != 2, rot, dh, ADT, dq, dd, d16 + 4 * pw);
        if (pw == 0) qf[t] = v;
        if (pw == 1) kf[t] = v;
        if (pw == 2) vnew = v;
        if (pw >= 1 && writer) {
            uint8_t* row = ((pw == 1) ? a.kcache : a.vcache) + (size_t)pos * a.kv_pitch + (size_t)g * head_bytes;
            if (ADT == GTEN_Q8) {
                uint8_t* blk = row + (size_t)(t >> 5) * GTEN_Q8_BYTES;
                store_global<uint8_t>(blk + 2 + (t & 31), (uint8_t)dq[t]);
                if ((t & 31) == 0) store_global<uint16_t>(blk, d16[4 * pw + (t >> 5)]);
            } else {
                store_global<uint16_t>((uint16_t*)row + t, f2h(v));
            }
        }
    }
    // the V chunk goes to LDS now (its requests were issued after the K rows: by the time the scores are done it is
    // there); the new position's slice is patched from the chip below
#pragma unroll
    for (int k = 0; k < NW; k++) vl[threadIdx.x * NW + k] = vw[k];            // row-major, as before (NW odd / a row per bank group: no conflict)
    __syncthreads();

    // ---- scores (k_dec_attn_score64's arithmetic)
    const float scale = 1.0f / sqrtf((float)dh);
    float acc = 0.f;
    if (ADT == GTEN_Q8) {
        const int* qi = (const int*)qi8;
        int isum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) isum = dot4(qi[j], (int)__builtin_amdgcn_alignbit(kw[j + 1], kw[j], 16), isum);
        acc += (float)isum * (qd[0] * h2f((uint16_t)(kw[0] & 0xffffu)));
        isum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) isum = dot4(qi[8 + j], (int)kw[9 + j], isum);
        acc += (float)isum * (qd[1] * h2f((uint16_t)(kw[8] >> 16)));
    } else {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            acc += qf[2 * j] * h2f((uint16_t)(kw[j] & 0xffffu));
            acc += qf[2 * j + 1] * h2f((uint16_t)(kw[j] >> 16));
        }
    }
    if (has_new) {
        float accn = 0.f;
        if (ADT == GTEN_Q8) {
            const int* qi = (const int*)qi8;
            const int* ki = (const int*)ki8;
#pragma unroll
            for (int b = 0; b < nblk; b++) {
                int isum = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) isum = dot4(qi[b * 8 + j], ki[b * 8 + j], isum);
                accn += (float)isum * (qd[b] * kd[b]);
            }
        } else {
            for (int e = 0; e < dh; e++) accn += qf[e] * kf[e];
        }
        if (c == pos) acc = accn;
        // the new position's V slice, from the chip: the bytes the writer workgroup stores (every workgroup of the
        // kv group computes the same ones)
        if (pw == 2) {
            uint8_t* vrow = (uint8_t*)vl + (size_t)(pos - c0) * (NW * 4);
            if (ADT == GTEN_Q8) {
                vrow[(t >> 5) * GTEN_Q8_BYTES + 2 + (t & 31)] = (uint8_t)vi8[t];
                if ((t & 31) == 0) *(uint16_t*)(vrow + (t >> 5) * GTEN_Q8_BYTES) = d16[8 + (t >> 5)];
            } else {
                ((uint16_t*)vrow)[t] = f2h(vnew);
            }
        }
    }
    const float sc = (c < n) ? acc * scale : -INFINITY;
    const float mx = block_max_n<4>(sc, red);               // red: first use; the sum takes its own words
    const float ex = (c < n) ? expf(sc - mx) : 0.f;
    const float sm = block_sum_n<4>(ex, red + 4);

    // ---- probabilities against the chunk's own statistics, rounded to the activation dtype in registers
    float pr = (c < n) ? ex / sm : 0.f;
    if (ADT == GTEN_Q8) {
        const Q8Scale s8 = q8_scale_from_absmax(max32(fabsf(pr)));
        if (c < n) pr = (float)q8_round(pr, s8.scale) * s8.ddeq;
    } else {
        pr = h2f(f2h(pr));
    }
    p[threadIdx.x] = pr;
    __syncthreads();                                         // p, and the patched V slice

    // ---- p . V (k_dec_attn_pv64's arithmetic)
    const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const uint8_t* vb = (const uint8_t*)vl;
    float o = 0.f;
    if (ADT == GTEN_Q8) {
        const int qoff = (e < 32) ? 2 + e : 36 + (e - 32), doff = (e < 32) ? 0 : 34;
        if (len == DEC_CHUNK) {
            float pp[2][8];
            int qv[2][8];
            unsigned dv[2][8];
            auto fetch = [&](int r, int slot) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint8_t* row = vb + (size_t)(cg + 4 * (8 * r + u)) * 68;
                    pp[slot][u] = p[cg + 4 * (8 * r + u)];
                    qv[slot][u] = (int)(int8_t)row[qoff];
                    dv[slot][u] = *(const uint16_t*)(row + doff);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int r = 0; r < DEC_CHUNK / 32; r++) {
                if (r + 1 < DEC_CHUNK / 32) fetch(r + 1, (r + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; u++) o += pp[r & 1][u] * ((float)qv[r & 1][u] * h2f((uint16_t)dv[r & 1][u]));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 8
            for (int cl = cg; cl < len; cl += 4) {
                const uint8_t* row = vb + (size_t)cl * 68;
                o += p[cl] * ((float)(int8_t)row[qoff] * h2f(*(const uint16_t*)(row + doff)));
            }
        }
    } else {
        for (int cl = cg; cl < len; cl += 4) o += p[cl] * h2f(((const uint16_t*)(vb + (size_t)cl * 128))[e]);
    }
    part[threadIdx.x] = o;
    __syncthreads();
    if (threadIdx.x < dh) {
        float r = 0.f;
        for (int gi = 0; gi < 4; gi++) r += part[gi * dh + threadIdx.x];
        a.att_part[((size_t)h * a.n_chunks + chunk) * dh + threadIdx.x] = r;
    }
    if (threadIdx.x == 64) {
        a.stats[((size_t)h * a.n_chunks + chunk) * 2 + 0] = mx;
        a.stats[((size_t)h * a.n_chunks + chunk) * 2 + 1] = sm;
    }
}

// ---- the two passes for MANY sequences: one workgroup per (kv head, chunk, sequence) serves all the query heads of
// the group (8 for TinyLlama), so a K / V chunk is read once instead of once per query head and the launch has
// 8x fewer workgroups -- at 32 sequences the per-head kernels above spend 29 + 44 us per block on 8192 small
// workgroups.  Q8 activations, d_head 64, <= 8 heads per group.  Per (head, position) the arithmetic and every
// reduction order are those of k_dec_attn_score64 / k_dec_attn_pv64: byte-identical scores, statistics and outputs.
#define DEC_MAXGRP 8

template <int GRP, int ADT>
__global__ __launch_bounds__(256) void k_dec_attn_score_g(const AttnArgs a0)
{
    // grid = (sequence, chunk, kv head): the workgroups of a SHORT context's few live chunks are consecutive
    // sequence indices, i.e. spread over all XCDs (with the kv head in x they were 4 of every 32 workgroups: half
    // the chip idle at n <= 256)
    constexpr int dh = 64, nblk = 2, NW = (ADT == GTEN_Q8) ? 17 : 32;     // dwords per kv-head slice (Q8 blocks | f16)
    const int g = blockIdx.z, chunk = blockIdx.y, c0 = chunk * DEC_CHUNK;
    const AttnArgs a = attn_for_seq(a0, blockIdx.x);              // (the cache pointers and the position are requested together)
    const int kv_dim = a.n_kv * dh;
    const size_t head_bytes = (ADT == GTEN_Q8) ? (size_t)nblk * GTEN_Q8_BYTES : (size_t)dh * 2;

    float* red = (float*)g_smem;                                  // [2][4][GRP]: maxima, then sums
    float* qd = red + 8 * GRP;                                    // [GRP][2] (+ pad to 4)
    float* kd = qd + 4 * GRP;                                     // 8: new k deltas, new v deltas
    float* qf = kd + 8;                                           // scratch f32 row of head_prep (unused values)
    uint16_t* d16 = (uint16_t*)(qf + dh);                         // [GRP + 2][4] halves
    int8_t* qi8 = (int8_t*)(d16 + 4 * (GRP + 2));                 // [GRP][64]
    int8_t* ki8 = qi8 + GRP * dh;                                 // 64
    int8_t* vi8 = ki8 + dh;                                       // 64
    float* qfa = (float*)(((uintptr_t)(vi8 + dh) + 15) & ~(uintptr_t)15);   // f16 activations: [GRP][64] q values, then the new k row [64]
    float* kfa = qfa + GRP * dh;

    // ---- requests, none of which needs the context length: the raw projections this wave turns into head vectors
    //      (wave w: query heads w, w + 4; wave 0 also the new k row, wave 1 the new v row), the rotation of the
    //      current position (left by the step's first launch), then this thread's cached K row (rows past the
    //      context are readable and unused; the row AT the new position is taken from the chip instead)
    const int t = threadIdx.x & 63, pw = threadIdx.x >> 6;
    constexpr int NJ = (GRP + 3) / 4;
    float qraw[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) qraw[jj] = a.qkv_raw[(g * GRP + min(pw + 4 * jj, GRP - 1)) * dh + t];
    float kvraw = a.qkv_raw[a.n_embd + ((pw & 1) ? kv_dim : 0) + g * dh + t];
    {
        // second K-split plane of the projections (k_dec_mmv): requested unconditionally (plane 0: the same words again)
        float qraw2[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw2[jj] = a.qkv_raw[a.qkv_plane + (g * GRP + min(pw + 4 * jj, GRP - 1)) * dh + t];
        const float kvraw2 = a.qkv_raw[a.qkv_plane + a.n_embd + ((pw & 1) ? kv_dim : 0) + g * dh + t];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw[jj] += a.qkv_plane ? qraw2[jj] : 0.f;
        kvraw += a.qkv_plane ? kvraw2 : 0.f;
    }
    const float2 rot = a.rope_now[t & 31];
    __builtin_amdgcn_sched_barrier(0);
    const int c = c0 + threadIdx.x;
    const int cs = min(c, a.max_ctx - 1);
    const gmem_u32 kp = as_global(a.kcache + (size_t)g * head_bytes) + (unsigned)cs * (unsigned)(a.kv_pitch >> 2);
    unsigned kw[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) kw[j] = kp[j];
    __builtin_amdgcn_sched_barrier(0);
    const int n = a.step->n, pos = n - 1;
    if (c0 >= n) return;

    // ---- head vectors
    const bool has_new = (pos >= c0) && (pos < c0 + DEC_CHUNK);
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
        const int j = pw + 4 * jj;
        if (j < GRP) {
            const float v = head_prep_cs(qraw[jj], true, true, rot, dh, ADT, qi8 + j * dh, qd + 2 * j, d16 + 4 * j);
            if (ADT != GTEN_Q8) qfa[j * dh + t] = v;
        }
    }
    if (pw < 2 && has_new) {
        int8_t* dq = pw ? vi8 : ki8;
        const float v = head_prep_cs(kvraw, true, pw == 0, rot, dh, ADT, dq, kd + 4 * pw, d16 + 4 * (GRP + pw));
        uint8_t* row = (pw ? a.vcache : a.kcache) + (size_t)pos * a.kv_pitch + (size_t)g * head_bytes;
        if (ADT == GTEN_Q8) {
            uint8_t* blk = row + (size_t)(t >> 5) * GTEN_Q8_BYTES;
            store_global<uint8_t>(blk + 2 + (t & 31), (uint8_t)dq[t]);
            if ((t & 31) == 0) store_global<uint16_t>(blk, d16[4 * (GRP + pw) + (t >> 5)]);
        } else {
            if (pw == 0) kfa[t] = v;
            store_global<uint16_t>((uint16_t*)row + t, f2h(v));
        }
    }
    __syncthreads();

    // ---- this position against every head of the group: every lane scores its cached row (the lane AT the new
    //      position holds unused bytes there); the chunk that contains the new position then scores the new k row
    //      from the chip -- uniform control flow, same arithmetic
    const float scale = 1.0f / sqrtf((float)dh);
    float sc[GRP];
    if (ADT == GTEN_Q8) {
        const float kd0 = h2f((uint16_t)(kw[0] & 0xffffu)), kd1 = h2f((uint16_t)(kw[8] >> 16));
        int kq[16];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            kq[j] = (int)__builtin_amdgcn_alignbit(kw[j + 1], kw[j], 16);
            kq[8 + j] = (int)kw[9 + j];
        }
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int* qi = (const int*)(qi8 + j * dh);
            float acc = 0.f;
            int isum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) isum = dot4(qi[k], kq[k], isum);
            acc += (float)isum * (qd[2 * j] * kd0);
            isum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) isum = dot4(qi[8 + k], kq[8 + k], isum);
            acc += (float)isum * (qd[2 * j + 1] * kd1);
            sc[j] = acc * scale;
        }
    } else {
        // f16: the elements in order, as k_dec_attn_score64 adds them; a K element is converted once for all heads
        float acc[GRP];
#pragma unroll
        for (int j = 0; j < GRP; j++) acc[j] = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; k++) {
            const float k0 = h2f((uint16_t)(kw[k] & 0xffffu)), k1 = h2f((uint16_t)(kw[k] >> 16));
#pragma unroll
            for (int j = 0; j < GRP; j++) {
                const float2 q2 = *(const float2*)(qfa + j * dh + 2 * k);
                acc[j] += q2.x * k0;
                acc[j] += q2.y * k1;
            }
        }
#pragma unroll
        for (int j = 0; j < GRP; j++) sc[j] = acc[j] * scale;
    }
    if (has_new) {
        const int* ki = (const int*)ki8;
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            float acc = 0.f;
            if (ADT == GTEN_Q8) {
                const int* qi = (const int*)(qi8 + j * dh);
                int isum = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) isum = dot4(qi[k], ki[k], isum);
                acc += (float)isum * (qd[2 * j] * kd[0]);
                isum = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) isum = dot4(qi[8 + k], ki[8 + k], isum);
                acc += (float)isum * (qd[2 * j + 1] * kd[1]);
            } else {
                for (int e = 0; e < dh; e++) acc += qfa[j * dh + e] * kfa[e];
            }
            if (c == pos) sc[j] = acc * scale;
        }
    }
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        if (c < n) a.scores[(size_t)(g * GRP + j) * a.max_ctx + c] = sc[j];
        else sc[j] = -INFINITY;
    }
    // ---- chunk maximum and sum of exponentials per head (block_max / block_sum, all heads per barrier pair)
    float mx[GRP];
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        const float m = wave_max_dpp(sc[j]);
        if (t == 0) red[pw * GRP + j] = m;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        float m = red[j];
        for (int w = 1; w < 4; w++) m = fmaxf(m, red[w * GRP + j]);
        mx[j] = m;
    }
    float* reds = red + 4 * GRP;                                  // the sums take their own words: no barrier between the two
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        const float ex = (c < n) ? expf(sc[j] - mx[j]) : 0.f;
        const float sw = wave_sum(ex);
        if (t == 0) reds[pw * GRP + j] = sw;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            float sm = 0.f;
            for (int w = 0; w < 4; w++) sm += reds[w * GRP + j];
            a.stats[((size_t)(g * GRP + j) * a.n_chunks + chunk) * 2 + 0] = mx[j];
            a.stats[((size_t)(g * GRP + j) * a.n_chunks + chunk) * 2 + 1] = sm;
        }
    }
}

// ---- f16 scores of MANY sequences on the matrix cores (16 sequences and up; f16 activations).  An f16 score costs two f32
// VALU operations per (head, position, element) in the scalar order -- 67 us per launch at 64 sequences, the bound of the
// f16 wide path.  Here a workgroup (kv head, chunk, sequence) forms Q (the group's heads, padded to 16 rows) x K^T (256
// positions) with two v_mfma_f32_16x16x32_f16 per 16 positions: f16 products, exact in f32, added inside the matrix core.
// The core's order of additions is not the scalar loop's: scores agree with k_dec_attn_score64 to f32 summation-order
// noise (the wide path's tolerance: model band), statistics and cache rows are formed the same way.
typedef _Float16 att_h8 __attribute__((ext_vector_type(8)));
typedef float att_f4 __attribute__((ext_vector_type(4)));
typedef float att_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float row16_max_f(float v)           // maximum over the 16 lanes of a row; every lane gets it
{
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    return fmaxf(v, dpp_mov<0x140>(v));
}
__device__ __forceinline__ float row16_sum_f(float v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

template <int GRP>
__global__ __launch_bounds__(256) void k_dec_attn_score_gm_f16(const AttnArgs a0)
{
    constexpr int dh = 64;
    const int g = blockIdx.z, chunk = blockIdx.y, c0 = chunk * DEC_CHUNK;
    const AttnArgs a = attn_for_seq(a0, blockIdx.x);
    const int kv_dim = a.n_kv * dh;
    const size_t head_bytes = (size_t)dh * 2;

    float* red = (float*)g_smem;                                  // [2][4][16]: per wave and head, maxima then sums
    float* mxs = red + 128;                                       // [16]
    uint16_t* d16 = (uint16_t*)(mxs + 16);                        // scratch of head_prep (unused for f16)
    float* qd = (float*)(d16 + 64);                               // scratch
    int8_t* qi8 = (int8_t*)(qd + 32);                             // scratch [64]
    uint16_t* qh = (uint16_t*)(((uintptr_t)(qi8 + 64) + 15) & ~(uintptr_t)15);   // [16][64] f16: the group's q vectors, zero rows beyond
    uint16_t* kh = qh + 16 * dh;                                  // [64] f16: the new k row

    // ---- requests, none of which needs the context length
    const int t = threadIdx.x & 63, pw = threadIdx.x >> 6, lc = t & 15, lq = t >> 4;
    constexpr int NJ = (GRP + 3) / 4;
    float qraw[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) qraw[jj] = a.qkv_raw[(g * GRP + min(pw + 4 * jj, GRP - 1)) * dh + t];
    float kvraw = a.qkv_raw[a.n_embd + ((pw & 1) ? kv_dim : 0) + g * dh + t];
    {
        float qraw2[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw2[jj] = a.qkv_raw[a.qkv_plane + (g * GRP + min(pw + 4 * jj, GRP - 1)) * dh + t];
        const float kvraw2 = a.qkv_raw[a.qkv_plane + a.n_embd + ((pw & 1) ? kv_dim : 0) + g * dh + t];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw[jj] += a.qkv_plane ? qraw2[jj] : 0.f;
        kvraw += a.qkv_plane ? kvraw2 : 0.f;
    }
    const float2 rot = a.rope_now[t & 31];
    __builtin_amdgcn_sched_barrier(0);
    // K fragments: wave pw owns positions c0 + 64 pw + 16 tt + lc (tt = 0..3); lane (lc, lq) reads elements 32 s + 8 lq ..
    uint4 kb[4][2];
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
        const int cs = min(c0 + 64 * pw + 16 * tt + lc, a.max_ctx - 1);
        const gmem_u32 kp = as_global(a.kcache + (size_t)g * head_bytes) + (unsigned)cs * (unsigned)(a.kv_pitch >> 2) + 4 * lq;
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
            const gmem_u32 kq = kp + 16 * s2;
            kb[tt][s2] = make_uint4(kq[0], kq[1], kq[2], kq[3]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int n = a.step->n, pos = n - 1;
    if (c0 >= n) return;

    // ---- head vectors (f16), rows beyond the group zeroed
    const bool has_new = (pos >= c0) && (pos < c0 + DEC_CHUNK);
    for (int i = threadIdx.x; i < (16 - GRP) * dh; i += 256) qh[GRP * dh + i] = 0;
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
        const int j = pw + 4 * jj;
        if (j < GRP) qh[j * dh + t] = f2h(head_prep_cs(qraw[jj], true, true, rot, dh, GTEN_F16, qi8, qd, d16));
    }
    if (pw < 2 && has_new) {
        const float v = head_prep_cs(kvraw, true, pw == 0, rot, dh, GTEN_F16, qi8, qd, d16);
        uint8_t* row = (pw ? a.vcache : a.kcache) + (size_t)pos * a.kv_pitch + (size_t)g * head_bytes;
        if (pw == 0) kh[t] = f2h(v);
        store_global<uint16_t>((uint16_t*)row + t, f2h(v));
    }
    __syncthreads();

    // ---- scores: rows = heads (this lane's outputs: heads 4 lq + i), columns = positions
    att_h8 qa[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) qa[s2] = *(const att_h8*)(qh + lc * dh + 32 * s2 + 8 * lq);
    float sc[4][4];
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
        att_f4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[0], __builtin_bit_cast(att_h8, kb[tt][0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[1], __builtin_bit_cast(att_h8, kb[tt][1]), acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) sc[tt][i] = acc[i] * 0.125f;
    }
    if (has_new) {
        // the new position's row is not in the cache yet for the other workgroups' view: score it from the chip (every column
        // of this product is the new k row; the lane that owns the position keeps it)
        att_h8 kn[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) kn[s2] = *(const att_h8*)(kh + 32 * s2 + 8 * lq);
        att_f4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[0], kn[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[1], kn[1], acc, 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; tt++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (c0 + 64 * pw + 16 * tt + lc == pos) sc[tt][i] = acc[i] * 0.125f;
    }
    const bool live = 4 * lq < GRP;                               // lanes whose rows are real heads
    float hm[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float m = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            const int c = c0 + 64 * pw + 16 * tt + lc;
            if (c < n) {
                if (live && 4 * lq + i < GRP) a.scores[(size_t)(g * GRP + 4 * lq + i) * a.max_ctx + c] = sc[tt][i];
                m = fmaxf(m, sc[tt][i]);
            } else {
                sc[tt][i] = -INFINITY;
            }
        }
        hm[i] = row16_max_f(m);
        if (lc == 0) red[pw * 16 + 4 * lq + i] = hm[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int hh = 4 * lq + i;
        hm[i] = fmaxf(fmaxf(red[hh], red[16 + hh]), fmaxf(red[32 + hh], red[48 + hh]));
    }
    float* reds = red + 64;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float e = 0.f;
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            const int c = c0 + 64 * pw + 16 * tt + lc;
            e += (c < n) ? expf(sc[tt][i] - hm[i]) : 0.f;
        }
        e = row16_sum_f(e);
        if (lc == 0) reds[pw * 16 + 4 * lq + i] = e;
    }
    __syncthreads();
    if (threadIdx.x < GRP) {
        const int hh = threadIdx.x;
        float sm = 0.f;
        for (int w = 0; w < 4; w++) sm += reds[w * 16 + hh];
        const float mxh = fmaxf(fmaxf(red[hh], red[16 + hh]), fmaxf(red[32 + hh], red[48 + hh]));
        a.stats[((size_t)(g * GRP + hh) * a.n_chunks + chunk) * 2 + 0] = mxh;
        a.stats[((size_t)(g * GRP + hh) * a.n_chunks + chunk) * 2 + 1] = sm;
    }
}

// EXACT: every p.V term is rounded as the reference rounds it (multiply, then add: k_dec_attn_pv64's bytes) -- the
// 8-sequence path, whose sequences are bit-identical to single-sequence decode.  The 16-64-sequence path already adds its
// W.x block sums in another order (k_dec_mmv), and there the kernel is bound by exactly these two VALU operations per
// (head, position, element): EXACT = false fuses them (one rounding instead of two: closer to the exact sum, not further).
template <int GRP, bool EXACT, int ADT>
__global__ __launch_bounds__(256) void k_dec_attn_pv_g(const AttnArgs a0)
{
    constexpr int dh = 64, NW = (ADT == GTEN_Q8) ? 17 : 32;           // dwords per kv-head slice (Q8 blocks | f16)
    const int g = blockIdx.z, chunk = blockIdx.y, c0 = chunk * DEC_CHUNK;
    const AttnArgs a = attn_for_seq(a0, blockIdx.x);
    const size_t head_bytes = (ADT == GTEN_Q8) ? (size_t)2 * GTEN_Q8_BYTES : (size_t)dh * 2;

    constexpr int GP = (GRP + 1) / 2;                             // head pairs: the p.V terms of two heads are ONE packed f32 operation
    float* p = (float*)g_smem;                                    // [GP][4][64][2]: position c of heads 2 jj, 2 jj + 1 at [jj][c & 3][c >> 2][.]
    float* part = p;                                              // [GP][4][2][64]: OVER p -- wave cg reads only p[.][cg][.][.] and later writes
                                                                  // only part[.][cg][.][lane], the same words: 8 KB of LDS less per
                                                                  // workgroup, i.e. 6 instead of 4 resident workgroups per CU
    unsigned* vl = (unsigned*)(p + 2 * GP * DEC_CHUNK);           // DEC_CHUNK * NW dwords: the chunk's V slices, row-major
    float* ms = (float*)(vl + DEC_CHUNK * NW);                    // [GRP][2]: the row maximum and sum of each head
    float* tl = ms + 16;                                          // [GRP][8]: the heads' chunk terms l_j exp(m_j - M) (16-byte aligned)

    // ---- requests, none of which needs the context length: the chunk statistics (lane 8 j + q of wave 0: head j,
    //      chunk q; the stats array has DEC_ATT_MAXCH chunks of slack), this position's score under every head, then
    //      the whole V chunk: dword idx -> (row idx / NW, word idx % NW), rows past the context readable and unused
    const int c = c0 + threadIdx.x;
    const int sj = min((int)threadIdx.x >> 3, GRP - 1), sq = threadIdx.x & 7;
    const float2 st = ((const float2*)a.stats)[(size_t)(g * GRP + sj) * a.n_chunks + sq];
    float scv[GRP];
#pragma unroll
    for (int j = 0; j < GRP; j++) scv[j] = a.scores[(size_t)(g * GRP + j) * a.max_ctx + min(c, a.max_ctx - 1)];
    __builtin_amdgcn_sched_barrier(0);
    unsigned vw[NW];
    {
        int row = (int)threadIdx.x / NW, w = (int)threadIdx.x % NW;
        const gmem_u32 vbase = as_global(a.vcache + (size_t)g * head_bytes);
        const unsigned pitch_w = (unsigned)(a.kv_pitch >> 2);
        const int last = a.max_ctx - 1 - c0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            vw[k] = vbase[(unsigned)(c0 + min(row, last)) * pitch_w + (unsigned)w];
            row += 256 / NW; w += 256 % NW;
            if (w >= NW) { w -= NW; row++; }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int n = a.step->n;
    if (c0 >= n) return;
    const int nch = (n + DEC_CHUNK - 1) / DEC_CHUNK;
    const int len = min(DEC_CHUNK, n - c0);

    // ---- row maximum and sum of every head from the chunk statistics, once per workgroup: the exponentials of all
    //      (head, chunk) pairs at once in wave 0, each head's terms then added in chunk order by one lane -- the
    //      values and the order of the sequential loop (x + 0 == x)
    {
        if (threadIdx.x < 64) {
            float m = (sq < nch) ? st.x : -INFINITY;
            m = quad_max(m);
            const float M = fmaxf(m, dpp_mov<0x141>(m));
            if (threadIdx.x < 8 * GRP) tl[threadIdx.x] = (sq < nch) ? st.y * expf(st.x - M) : 0.f;
            const float Mj = __shfl(M, (threadIdx.x & 7) * 8, 64);   // all 64 lanes take part: a shuffle reads live lanes only
            if (threadIdx.x < GRP) {
                // (same wave: the LDS writes above are ordered before these reads)
                const float4 t0 = *(const float4*)(tl + threadIdx.x * 8), t1 = *(const float4*)(tl + threadIdx.x * 8 + 4);
                float S = 0.f;
                S += t0.x; S += t0.y; S += t0.z; S += t0.w; S += t1.x; S += t1.y; S += t1.z; S += t1.w;
                ms[threadIdx.x * 2] = Mj; ms[threadIdx.x * 2 + 1] = S;
            }
        }
    }
    __syncthreads();
    // ---- probabilities of every head of the group, rounded to the activation dtype along the context (the Q8
    //      block of position c = the 32 lanes around this thread: round_row_inplace, in registers)
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        const float x = (c < n) ? expf(scv[j] - ms[2 * j]) / ms[2 * j + 1] : 0.f;
        float pr;
        if (ADT == GTEN_Q8) {
            const Q8Scale qs = q8_scale_from_absmax(max32(fabsf(x)));
            pr = (c < n) ? (float)q8_round(x, qs.scale) * qs.ddeq : 0.f;
        } else {
            pr = h2f(f2h(x));
        }
        p[((((j >> 1) * 4 + (threadIdx.x & 3)) * 64 + (threadIdx.x >> 2)) << 1) + (j & 1)] = pr;
    }
    if (GRP & 1) p[((((GRP >> 1) * 4 + (threadIdx.x & 3)) * 64 + (threadIdx.x >> 2)) << 1) + 1] = 0.f;
#pragma unroll
    for (int k = 0; k < NW; k++) vl[threadIdx.x + k * 256] = vw[k];
    __syncthreads();

    // ---- p.V: a V element is dequantized once and feeds all heads; four positions of this thread's stride-4
    //      sequence per step (their probabilities are one 16-byte LDS read per head)
    const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const uint8_t* vb = (const uint8_t*)vl;
    const int qoff = (e < 32) ? 2 + e : 36 + (e - 32), doff = (e < 32) ? 0 : 34;
    // (multiply and add of two heads in one v_pk_mul_f32 / v_pk_add_f32, or one v_pk_fma_f32: per head the same operations,
    //  the same roundings, the same order)
    att_f2 acc[GP];
#pragma unroll
    for (int jj = 0; jj < GP; jj++) acc[jj] = att_f2{0.f, 0.f};
    for (int i = 0; cg + 4 * i < len; i += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            // positions past the context's end are rows of the chunk all the same (i + u <= 63) and meet p = 0
            const uint8_t* row = vb + (size_t)(cg + 4 * (i + u)) * (NW * 4);
            v[u] = (ADT == GTEN_Q8) ? (float)(int8_t)row[qoff] * h2f(*(const uint16_t*)(row + doff)) : h2f(((const uint16_t*)row)[e]);
        }
#pragma unroll
        for (int jj = 0; jj < GP; jj++) {
            const float* pj = p + (((jj * 4 + cg) * 64 + i) << 1);
            const float4 pa = *(const float4*)pj, pb = *(const float4*)(pj + 4);
            const att_f2 p0{pa.x, pa.y}, p1{pa.z, pa.w}, p2{pb.x, pb.y}, p3{pb.z, pb.w};
            if (EXACT) {
                acc[jj] += p0 * att_f2{v[0], v[0]};
                acc[jj] += p1 * att_f2{v[1], v[1]};
                acc[jj] += p2 * att_f2{v[2], v[2]};
                acc[jj] += p3 * att_f2{v[3], v[3]};
            } else {
                acc[jj] = __builtin_elementwise_fma(p0, att_f2{v[0], v[0]}, acc[jj]);
                acc[jj] = __builtin_elementwise_fma(p1, att_f2{v[1], v[1]}, acc[jj]);
                acc[jj] = __builtin_elementwise_fma(p2, att_f2{v[2], v[2]}, acc[jj]);
                acc[jj] = __builtin_elementwise_fma(p3, att_f2{v[3], v[3]}, acc[jj]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < GRP; j++) part[(((j >> 1) * 4 + cg) * 2 + (j & 1)) * 64 + e] = (j & 1) ? acc[j >> 1].y : acc[j >> 1].x;
    __syncthreads();
    for (int idx = threadIdx.x; idx < GRP * dh; idx += 256) {
        const int j = idx >> 6, ee = idx & 63;
        float o = 0.f;
        for (int gi = 0; gi < 4; gi++) o += part[(((j >> 1) * 4 + gi) * 2 + (j & 1)) * 64 + ee];
        a.att_part[((size_t)(g * GRP + j) * a.n_chunks + chunk) * dh + ee] = o;
    }
}

// ---- both passes of the grouped pair in ONE launch with chunk-local statistics: k_dec_attn_one64's scheme (see there)
// for a whole kv group.  A (sequence, chunk, kv head) workgroup requests its K rows AND its V chunk at kernel entry,
// scores every head of the group, normalises against the chunk's own maxima / sums, rounds the probabilities to the
// activation dtype and leaves p_c . V_c with (m_c, l_c) for the consumer's PRO_ATTW join -- no score round trip through
// HBM (16.8 MB written and read back per block at 64 sequences), no second launch, the statistics formed once.
// Per (head, position) the arithmetic and every reduction order are those of k_dec_attn_one64 (EXACT: byte-identical
// outputs and statistics, the 8-sequence path; EXACT = false fuses each p.V multiply-add, the 16-64-sequence path).
template <int GRP, bool EXACT, int ADT>
__global__ __launch_bounds__(256) void k_dec_attn_one_g(const AttnArgs a0)
{
    constexpr int dh = 64, nblk = 2, NW = (ADT == GTEN_Q8) ? 17 : 32;
    constexpr int GP = (GRP + 1) / 2;
    const int g = blockIdx.z, chunk = blockIdx.y, c0 = chunk * DEC_CHUNK;
    const AttnArgs a = attn_for_seq(a0, blockIdx.x);
    const int kv_dim = a.n_kv * dh;
    const size_t head_bytes = (ADT == GTEN_Q8) ? (size_t)nblk * GTEN_Q8_BYTES : (size_t)dh * 2;

    float* red = (float*)g_smem;                                  // [2][4][GRP]: maxima, then sums
    float* qd = red + 8 * GRP;                                    // [GRP][2] (+ pad to 4)
    float* kd = qd + 4 * GRP;                                     // 8: new k deltas, new v deltas
    float* qf = kd + 8;                                           // scratch f32 row of head_prep (unused values)
    uint16_t* d16 = (uint16_t*)(qf + dh);                         // [GRP + 2][4] halves
    int8_t* qi8 = (int8_t*)(d16 + 4 * (GRP + 2));                 // [GRP][64]
    int8_t* ki8 = qi8 + GRP * dh;                                 // 64
    int8_t* vi8 = ki8 + dh;                                       // 64
    float* qfa = (float*)(((uintptr_t)(vi8 + dh) + 15) & ~(uintptr_t)15);   // f16 activations: [GRP][64] q values, then the new k row [64]
    float* kfa = qfa + GRP * dh;
    float* p = (ADT == GTEN_Q8) ? qfa : kfa + dh;                 // [GP][4][64][2]: position c of heads 2 jj, 2 jj + 1 at [jj][c & 3][c >> 2][.]
    float* part = p;                                              // OVER p (see k_dec_attn_pv_g)
    unsigned* vl = (unsigned*)(p + 2 * GP * DEC_CHUNK);           // DEC_CHUNK * NW dwords: the chunk's V slices, row-major

    // ---- requests, none of which needs the context length
    const int t = threadIdx.x & 63, pw = threadIdx.x >> 6;
    constexpr int NJ = (GRP + 3) / 4;
    float qraw[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) qraw[jj] = a.qkv_raw[(g * GRP + min(pw + 4 * jj, GRP - 1)) * dh + t];
    float kvraw = a.qkv_raw[a.n_embd + ((pw & 1) ? kv_dim : 0) + g * dh + t];
    {
        float qraw2[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw2[jj] = a.qkv_raw[a.qkv_plane + (g * GRP + min(pw + 4 * jj, GRP - 1)) * dh + t];
        const float kvraw2 = a.qkv_raw[a.qkv_plane + a.n_embd + ((pw & 1) ? kv_dim : 0) + g * dh + t];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw[jj] += a.qkv_plane ? qraw2[jj] : 0.f;
        kvraw += a.qkv_plane ? kvraw2 : 0.f;
    }
    const float2 rot = a.rope_now[t & 31];
    __builtin_amdgcn_sched_barrier(0);
    const int c = c0 + threadIdx.x;
    const int cs = min(c, a.max_ctx - 1);
    const unsigned pitch_w = (unsigned)(a.kv_pitch >> 2);
    const gmem_u32 kp = as_global(a.kcache + (size_t)g * head_bytes) + (unsigned)cs * pitch_w;
    unsigned kw[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) kw[j] = kp[j];
    unsigned vw[NW];
    {
        int row = (int)threadIdx.x / NW, w = (int)threadIdx.x % NW;
        const gmem_u32 vbase = as_global(a.vcache + (size_t)g * head_bytes);
        const int last = a.max_ctx - 1 - c0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            vw[k] = vbase[(unsigned)(c0 + min(row, last)) * pitch_w + (unsigned)w];
            row += 256 / NW; w += 256 % NW;
            if (w >= NW) { w -= NW; row++; }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int n = a.step->n, pos = n - 1;
    if (c0 >= n) return;
    const int len = min(DEC_CHUNK, n - c0);

    // ---- head vectors
    const bool has_new = (pos >= c0) && (pos < c0 + DEC_CHUNK);
    float vnew = 0.f;
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
        const int j = pw + 4 * jj;
        if (j < GRP) {
            const float v = head_prep_cs(qraw[jj], true, true, rot, dh, ADT, qi8 + j * dh, qd + 2 * j, d16 + 4 * j);
            if (ADT != GTEN_Q8) qfa[j * dh + t] = v;
        }
    }
    if (pw < 2 && has_new) {
        int8_t* dq = pw ? vi8 : ki8;
        const float v = head_prep_cs(kvraw, true, pw == 0, rot, dh, ADT, dq, kd + 4 * pw, d16 + 4 * (GRP + pw));
        uint8_t* row = (pw ? a.vcache : a.kcache) + (size_t)pos * a.kv_pitch + (size_t)g * head_bytes;
        if (ADT == GTEN_Q8) {
            uint8_t* blk = row + (size_t)(t >> 5) * GTEN_Q8_BYTES;
            store_global<uint8_t>(blk + 2 + (t & 31), (uint8_t)dq[t]);
            if ((t & 31) == 0) store_global<uint16_t>(blk, d16[4 * (GRP + pw) + (t >> 5)]);
        } else {
            if (pw == 0) kfa[t] = v;
            if (pw == 1) vnew = v;
            store_global<uint16_t>((uint16_t*)row + t, f2h(v));
        }
    }
#pragma unroll
    for (int k = 0; k < NW; k++) vl[threadIdx.x + k * 256] = vw[k];
    __syncthreads();

    // ---- this position against every head of the group (k_dec_attn_score_g's arithmetic)
    const float scale = 1.0f / sqrtf((float)dh);
    float sc[GRP];
    if (ADT == GTEN_Q8) {
        const float kd0 = h2f((uint16_t)(kw[0] & 0xffffu)), kd1 = h2f((uint16_t)(kw[8] >> 16));
        int kq[16];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            kq[j] = (int)__builtin_amdgcn_alignbit(kw[j + 1], kw[j], 16);
            kq[8 + j] = (int)kw[9 + j];
        }
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            const int* qi = (const int*)(qi8 + j * dh);
            float acc = 0.f;
            int isum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) isum = dot4(qi[k], kq[k], isum);
            acc += (float)isum * (qd[2 * j] * kd0);
            isum = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) isum = dot4(qi[8 + k], kq[8 + k], isum);
            acc += (float)isum * (qd[2 * j + 1] * kd1);
            sc[j] = acc * scale;
        }
    } else {
        float acc[GRP];
#pragma unroll
        for (int j = 0; j < GRP; j++) acc[j] = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; k++) {
            const float k0 = h2f((uint16_t)(kw[k] & 0xffffu)), k1 = h2f((uint16_t)(kw[k] >> 16));
#pragma unroll
            for (int j = 0; j < GRP; j++) {
                const float2 q2 = *(const float2*)(qfa + j * dh + 2 * k);
                acc[j] += q2.x * k0;
                acc[j] += q2.y * k1;
            }
        }
#pragma unroll
        for (int j = 0; j < GRP; j++) sc[j] = acc[j] * scale;
    }
    if (has_new) {
        const int* ki = (const int*)ki8;
#pragma unroll
        for (int j = 0; j < GRP; j++) {
            float acc = 0.f;
            if (ADT == GTEN_Q8) {
                const int* qi = (const int*)(qi8 + j * dh);
                int isum = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) isum = dot4(qi[k], ki[k], isum);
                acc += (float)isum * (qd[2 * j] * kd[0]);
                isum = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) isum = dot4(qi[8 + k], ki[8 + k], isum);
                acc += (float)isum * (qd[2 * j + 1] * kd[1]);
            } else {
                for (int e = 0; e < dh; e++) acc += qfa[j * dh + e] * kfa[e];
            }
            if (c == pos) sc[j] = acc * scale;
        }
        // the new position's V slice comes from the chip (the cache row is being written by this very launch)
        if (pw == 1) {
            uint8_t* vrow = (uint8_t*)vl + (size_t)(pos - c0) * (NW * 4);
            if (ADT == GTEN_Q8) {
                vrow[(t >> 5) * GTEN_Q8_BYTES + 2 + (t & 31)] = (uint8_t)vi8[t];
                if ((t & 31) == 0) *(uint16_t*)(vrow + (t >> 5) * GTEN_Q8_BYTES) = d16[4 * (GRP + 1) + (t >> 5)];
            } else {
                ((uint16_t*)vrow)[t] = f2h(vnew);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < GRP; j++)
        if (c >= n) sc[j] = -INFINITY;
    // ---- chunk maximum and sum of exponentials per head (all heads per barrier pair)
    float mx[GRP], ex[GRP], sm[GRP];
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        const float m = wave_max_dpp(sc[j]);
        if (t == 0) red[pw * GRP + j] = m;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        float m = red[j];
        for (int w = 1; w < 4; w++) m = fmaxf(m, red[w * GRP + j]);
        mx[j] = m;
    }
    float* reds = red + 4 * GRP;                                  // the sums take their own words: no barrier between the two
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        ex[j] = (c < n) ? expf(sc[j] - mx[j]) : 0.f;
        const float sw = wave_sum(ex[j]);
        if (t == 0) reds[pw * GRP + j] = sw;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        float s_ = 0.f;
        for (int w = 0; w < 4; w++) s_ += reds[w * GRP + j];
        sm[j] = s_;
    }
    if (threadIdx.x < GRP) {
        // (a register array indexed by the thread: selected by a chain of compares, GRP <= 8)
        float mj = mx[0], sj = sm[0];
#pragma unroll
        for (int j = 1; j < GRP; j++) { mj = ((int)threadIdx.x == j) ? mx[j] : mj; sj = ((int)threadIdx.x == j) ? sm[j] : sj; }
        a.stats[((size_t)(g * GRP + threadIdx.x) * a.n_chunks + chunk) * 2 + 0] = mj;
        a.stats[((size_t)(g * GRP + threadIdx.x) * a.n_chunks + chunk) * 2 + 1] = sj;
    }
    // ---- probabilities against the chunk's own statistics, rounded to the activation dtype (Q8 block = the 32 lanes
    //      around this thread)
#pragma unroll
    for (int j = 0; j < GRP; j++) {
        const float x = (c < n) ? ex[j] / sm[j] : 0.f;
        float pr;
        if (ADT == GTEN_Q8) {
            const Q8Scale qs = q8_scale_from_absmax(max32(fabsf(x)));
            pr = (c < n) ? (float)q8_round(x, qs.scale) * qs.ddeq : 0.f;
        } else {
            pr = h2f(f2h(x));
        }
        p[((((j >> 1) * 4 + (threadIdx.x & 3)) * 64 + (threadIdx.x >> 2)) << 1) + (j & 1)] = pr;
    }
    if (GRP & 1) p[((((GRP >> 1) * 4 + (threadIdx.x & 3)) * 64 + (threadIdx.x >> 2)) << 1) + 1] = 0.f;
    __syncthreads();

    // ---- p.V (k_dec_attn_pv_g's arithmetic)
    const int e = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const uint8_t* vb = (const uint8_t*)vl;
    const int qoff = (e < 32) ? 2 + e : 36 + (e - 32), doff = (e < 32) ? 0 : 34;
    att_f2 acc[GP];
#pragma unroll
    for (int jj = 0; jj < GP; jj++) acc[jj] = att_f2{0.f, 0.f};
    for (int i = 0; cg + 4 * i < len; i += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint8_t* row = vb + (size_t)(cg + 4 * (i + u)) * (NW * 4);
            v[u] = (ADT == GTEN_Q8) ? (float)(int8_t)row[qoff] * h2f(*(const uint16_t*)(row + doff)) : h2f(((const uint16_t*)row)[e]);
        }
#pragma unroll
        for (int jj = 0; jj < GP; jj++) {
            const float* pj = p + (((jj * 4 + cg) * 64 + i) << 1);
            const float4 pa = *(const float4*)pj, pb = *(const float4*)(pj + 4);
            const att_f2 p0{pa.x, pa.y}, p1{pa.z, pa.w}, p2{pb.x, pb.y}, p3{pb.z, pb.w};
            if (EXACT) {
                acc[jj] += p0 * att_f2{v[0], v[0]};
                acc[jj] += p1 * att_f2{v[1], v[1]};
                acc[jj] += p2 * att_f2{v[2], v[2]};
                acc[jj] += p3 * att_f2{v[3], v[3]};
            } else {
                acc[jj] = __builtin_elementwise_fma(p0, att_f2{v[0], v[0]}, acc[jj]);
                acc[jj] = __builtin_elementwise_fma(p1, att_f2{v[1], v[1]}, acc[jj]);
                acc[jj] = __builtin_elementwise_fma(p2, att_f2{v[2], v[2]}, acc[jj]);
                acc[jj] = __builtin_elementwise_fma(p3, att_f2{v[3], v[3]}, acc[jj]);
            }
        }
    }
    // (part lies over p: wave cg has read only p[.][cg][.][.] and writes only part[.][cg][.][lane], the same words)
#pragma unroll
    for (int j = 0; j < GRP; j++) part[(((j >> 1) * 4 + cg) * 2 + (j & 1)) * 64 + e] = (j & 1) ? acc[j >> 1].y : acc[j >> 1].x;
    __syncthreads();
    for (int idx = threadIdx.x; idx < GRP * dh; idx += 256) {
        const int j = idx >> 6, ee = idx & 63;
        float o = 0.f;
        for (int gi = 0; gi < 4; gi++) o += part[(((j >> 1) * 4 + gi) * 2 + (j & 1)) * 64 + ee];
        a.att_part[((size_t)(g * GRP + j) * a.n_chunks + chunk) * dh + ee] = o;
    }
}

// ---- 16-64 sequences, Q8 activations: the whole attention of a (sequence, chunk, kv head) in ONE launch ON THE MATRIX CORES.
//
// The VALU pair above spends ~1900 instructions per thread on a chunk (scores 8 heads x 17 dot4, exponentials and Q8
// roundings per (head, position), 2 VALU operations per p.V term); merged as it stands it lost to the pair (occupancy,
// DESIGN.md 3.3).  Here the two contractions are matrix instructions and a workgroup needs ~1000 instructions per thread:
//   * the K and the V chunk are requested at entry as COALESCED dwords exactly as they lie in the cache (256 positions x
//     17 dwords each, 17 requests per thread and matrix); K is parked in LDS at once, V stays in registers until the scores
//     are done (its arrival hides behind them) and then takes K's place;
//   * scores: v_mfma_i32_16x16x32_i8, A = the group's head vectors (rows 8..15 zero), B = 16 positions of K; one
//     instruction per (16 positions, quant block) gives the exact integer block dots, scaled dq dk as the scalar code;
//   * chunk-local softmax (k_dec_attn_one64's scheme, hardware exponential), probabilities rounded to Q8 blocks of 32
//     along the context exactly as the reference stores them;
//   * p.V: v_mfma_f32_16x16x32_f16 with A = f16(p_q8 * dv[position]) -- the V row's block delta folded into the
//     probability, one fp16 rounding -- and B = the V quants as exact f16 integers: 8 matrix instructions per wave.
// The consumer joins the chunks with PRO_ATTW.  Numerics: the wide path's (model band; tests/test_multiseq_oracle_gpu.py
// holds every slot to the oracle and to the reference's goldens), not the byte-exact 8-sequence path's.
typedef int att_v4i __attribute__((ext_vector_type(4)));

// (Measured and not kept: the same kernel PERSISTENT -- at most 4 workgroups per CU walking the items, the next item's K / V
// chunk requested into registers while the current one is computed, so that loading and computing overlap instead of
// alternating in two rounds: 64 us against 37 us per launch at 64 sequences.  The 40 prefetch registers pushed the
// compute phase into scratch, and a wait for a scratch reload is a wait for every older request -- the prefetch itself.)
template <int GRP>
__global__ __launch_bounds__(256) void k_dec_attn_mm_g(const AttnArgs a0, const int n_seq)
{
    constexpr int dh = 64, NW = 17, PP = 264;                     // dwords per cached kv-head slice; halfs per probability row
    // 1-D grid, id = ((sc / 8) * n_kv + g) * 8 + sc % 8 with sc = chunk * n_seq + seq: the kv heads of one (sequence, chunk)
    // -- whose 68-byte slices share the 128-byte lines of a 272-byte cache row -- are 8 ids apart, i.e. dispatched together
    // on ONE XCD (ids go round-robin over the 8 XCDs), so that its L2 fetches each line once (PMC: 95 MB per launch at 64
    // sequences against 71 MB of K / V with the kv head in the slowest grid dimension); short contexts still spread
    // their few live chunks over all XCDs.
    const int sc_lo = blockIdx.x & 7, t1 = blockIdx.x >> 3, g = t1 % a0.n_kv, sci = (t1 / a0.n_kv) * 8 + sc_lo;
    const int chunk = sci / n_seq, c0 = chunk * DEC_CHUNK;
    const AttnArgs a = attn_for_seq(a0, sci - chunk * n_seq);
    const int kv_dim = a.n_kv * dh;
    constexpr size_t head_bytes = 2 * GTEN_Q8_BYTES;

    unsigned* kl = (unsigned*)g_smem;                             // [256][17]: the chunk's K slices as they lie in the cache; dead after
    _Float16* pl = (_Float16*)g_smem;                             // the scores: the probability rows [2 halves][8 heads][PP] lie over them
    unsigned* vl = kl + DEC_CHUNK * NW;                           // [256][17]
    int8_t* qi8 = (int8_t*)(vl + DEC_CHUNK * NW);                 // [16][64], rows GRP..15 zero
    float* qd = (float*)(qi8 + 16 * dh);                          // [16][2]
    float* kd = qd + 32;                                          // 8: new k deltas, new v deltas (head_prep scratch)
    uint16_t* d16 = (uint16_t*)(kd + 8);                          // [GRP + 2][4] halves
    int8_t* ki8 = (int8_t*)(d16 + 4 * (DEC_MAXGRP + 2));          // 64
    int8_t* vi8 = ki8 + dh;                                       // 64
    float* red = (float*)(vi8 + dh);                              // [2][4][16]: maxima, then sums
    unsigned* vnew = (unsigned*)(red + 128);                      // 17 dwords: the new position's V slice in cache layout

    // ---- requests, none of which needs the context length: raw projections, rotation, then the K and the V chunk as
    //      coalesced dwords (dword idx -> (row idx / 17, word idx % 17); rows past the context are readable and masked below)
    const int t = threadIdx.x & 63, pw = threadIdx.x >> 6, lc = t & 15, lq = t >> 4;
    constexpr int NJ = (GRP + 3) / 4;
    float qraw[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) qraw[jj] = a.qkv_raw[(g * GRP + min(pw + 4 * jj, GRP - 1)) * dh + t];
    float kvraw = a.qkv_raw[a.n_embd + ((pw & 1) ? kv_dim : 0) + g * dh + t];
    {
        float qraw2[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw2[jj] = a.qkv_raw[a.qkv_plane + (g * GRP + min(pw + 4 * jj, GRP - 1)) * dh + t];
        const float kvraw2 = a.qkv_raw[a.qkv_plane + a.n_embd + ((pw & 1) ? kv_dim : 0) + g * dh + t];
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw[jj] += a.qkv_plane ? qraw2[jj] : 0.f;
        kvraw += a.qkv_plane ? kvraw2 : 0.f;
    }
    const float2 rot = a.rope_now[t & 31];
    __builtin_amdgcn_sched_barrier(0);
    unsigned kw[NW], vw[NW];
    {
        int row = (int)threadIdx.x / NW, w = (int)threadIdx.x % NW;
        const gmem_u32 kbase = as_global(a.kcache + (size_t)g * head_bytes), vbase = as_global(a.vcache + (size_t)g * head_bytes);
        const unsigned pitch_w = (unsigned)(a.kv_pitch >> 2);
        const int last = a.max_ctx - 1 - c0;
        // (all of K first, then all of V: requests return in order, and the scores must not wait for the V chunk)
        unsigned off[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) {
            off[k] = (unsigned)(c0 + min(row, last)) * pitch_w + (unsigned)w;
            kw[k] = kbase[off[k]];
            row += 256 / NW; w += 256 % NW;
            if (w >= NW) { w -= NW; row++; }
        }
#pragma unroll
        for (int k = 0; k < NW; k++) vw[k] = vbase[off[k]];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int n = a.step->n, pos = n - 1;
    if (c0 >= n) return;

    // ---- head vectors (k_dec_attn_score_g's), rows GRP..15 of the A operand zeroed
    const bool has_new = (pos >= c0) && (pos < c0 + DEC_CHUNK);
    for (int i = threadIdx.x; i < (16 - GRP) * dh / 4; i += 256) ((int*)(qi8 + GRP * dh))[i] = 0;
    if (threadIdx.x < 2 * (16 - GRP)) qd[2 * GRP + threadIdx.x] = 0.f;
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
        const int j = pw + 4 * jj;
        if (j < GRP) head_prep_cs(qraw[jj], true, true, rot, dh, GTEN_Q8, qi8 + j * dh, qd + 2 * j, d16 + 4 * j);
    }
    if (pw < 2 && has_new) {
        int8_t* dq = pw ? vi8 : ki8;
        head_prep_cs(kvraw, true, pw == 0, rot, dh, GTEN_Q8, dq, kd + 4 * pw, d16 + 4 * (GRP + pw));
        uint8_t* row = (pw ? a.vcache : a.kcache) + (size_t)pos * a.kv_pitch + (size_t)g * head_bytes;
        uint8_t* blk = row + (size_t)(t >> 5) * GTEN_Q8_BYTES;
        store_global<uint8_t>(blk + 2 + (t & 31), (uint8_t)dq[t]);
        if ((t & 31) == 0) store_global<uint16_t>(blk, d16[4 * (GRP + pw) + (t >> 5)]);
    }
    // the K chunk goes to LDS now; the V chunk stays in its registers, in flight, until the scores are done
#pragma unroll
    for (int k = 0; k < NW; k++) kl[threadIdx.x + k * 256] = kw[k];
    __syncthreads();
    if (pw < 2 && has_new) {
        // the new position's K / V slice comes from the chip (the cache row is being written by this very launch): K patched
        // in place, V assembled in cache layout for the store below
        uint8_t* row = pw ? (uint8_t*)vnew : (uint8_t*)kl + (size_t)(pos - c0) * (NW * 4);
        const int8_t* dq = pw ? vi8 : ki8;
        row[(t >> 5) * GTEN_Q8_BYTES + 2 + (t & 31)] = (uint8_t)dq[t];
        if ((t & 31) == 0) *(uint16_t*)(row + (t >> 5) * GTEN_Q8_BYTES) = d16[4 * (GRP + pw) + (t >> 5)];
    }
    if (has_new) __syncthreads();                                 // (uniform per workgroup)

    // ---- scores on the matrix cores: rows = heads, columns = positions 64 pw + 16 tt + lc.  A K slice is
    //      [d0 | q0 x32 | d1 | q1 x32]: block 0's quants straddle the dwords by two bytes, block 1's are aligned.
    //      The group's 8 heads fill rows 0..7 of the 16-row tile, so the results sit in lanes 0..31 (lq < 2) only;
    //      v_permlane32_swap hands tiles 2, 3 to lanes 32..63: afterwards lane (lc, lq) owns heads 4 (lq & 1) + i
    //      and tiles 2 (lq >> 1) + u -- 8 (head, position) pairs per lane, every lane busy, and the lane's two
    //      positions under a head are exactly one Q8 block of the probability row.
    long qa[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) qa[s2] = *(const long*)(qi8 + lc * dh + 32 * s2 + 8 * lq);
    att_v4i i0[4], i1[4];
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
        const unsigned* krow = kl + (64 * pw + 16 * tt + lc) * NW;
        const unsigned w0 = krow[2 * lq], w1 = krow[2 * lq + 1], w2 = krow[2 * lq + 2];
        const unsigned x0 = krow[9 + 2 * lq], x1 = krow[10 + 2 * lq];
        const long kb0 = (long)(((unsigned long)__builtin_amdgcn_alignbit(w2, w1, 16) << 32) | __builtin_amdgcn_alignbit(w1, w0, 16));
        const long kb1 = (long)(((unsigned long)x1 << 32) | x0);
        const att_v4i z = {0, 0, 0, 0};
        i0[tt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(qa[0], kb0, z, 0, 0, 0);
        i1[tt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(qa[1], kb1, z, 0, 0, 0);
    }
    const int hq = lq & 1, tsel = lq >> 1;
    float qdl[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++) { qdl[i][0] = qd[2 * (4 * hq + i)]; qdl[i][1] = qd[2 * (4 * hq + i) + 1]; }
    float sc[2][4];                                               // scores, later their exponentials: [tile 2 tsel + u][head 4 hq + i]
    int pl_[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        pl_[u] = 64 * pw + 16 * (2 * tsel + u) + lc;
        const unsigned kd0w = kl[pl_[u] * NW], kd1w = kl[pl_[u] * NW + 8];
        const float kd0 = h2f((uint16_t)(kd0w & 0xffffu)), kd1 = h2f((uint16_t)(kd1w >> 16));
        const bool live = c0 + pl_[u] < n;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // (first operand: lanes 32..63 receive the second operand's lanes 0..31)
            const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)i0[u][i], (unsigned)i0[2 + u][i], false, false);
            const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)i1[u][i], (unsigned)i1[2 + u][i], false, false);
            float acc = 0.f;
            acc += (float)(int)r0[0] * (qdl[i][0] * kd0);
            acc += (float)(int)r1[0] * (qdl[i][1] * kd1);
            sc[u][i] = live ? acc * 0.125f : -INFINITY;           // 1 / sqrt(64)
        }
    }
    // ---- chunk maximum and sum of exponentials per head: 8 partials per head (4 waves x 2 lane halves)
    float M[4], L[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float m = row16_max_f(fmaxf(sc[0][i], sc[1][i]));
        if (lc == 0) red[(2 * pw + tsel) * 8 + 4 * hq + i] = m;
    }
    __syncthreads();                                              // (every wave is done with the K rows: pl may be written from here on)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int hh = 4 * hq + i;
        float m = red[hh];
#pragma unroll
        for (int q = 1; q < 8; q++) m = fmaxf(m, red[q * 8 + hh]);
        M[i] = m;
    }
    float* reds = red + 64;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        sc[0][i] = __expf(sc[0][i] - M[i]);                       // exp(-inf) = 0 for masked positions
        sc[1][i] = __expf(sc[1][i] - M[i]);
        const float e = row16_sum_f(sc[0][i] + sc[1][i]);
        if (lc == 0) reds[(2 * pw + tsel) * 8 + 4 * hq + i] = e;
    }
    {
        // the V chunk lands in LDS (its requests have been in flight since kernel entry); the new position's slice from the chip
        int row = (int)threadIdx.x / NW, w = (int)threadIdx.x % NW;
        const int newrow = has_new ? pos - c0 : -1;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            vl[threadIdx.x + k * 256] = (row == newrow) ? vnew[w] : vw[k];
            row += 256 / NW; w += 256 % NW;
            if (w >= NW) { w -= NW; row++; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int hh = 4 * hq + i;
        float l = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) l += reds[q * 8 + hh];
        L[i] = l;
    }
    if (pw == 0 && tsel == 0 && lc == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (4 * hq + i < GRP) {
                a.stats[((size_t)(g * GRP + 4 * hq + i) * a.n_chunks + chunk) * 2 + 0] = M[i];
                a.stats[((size_t)(g * GRP + 4 * hq + i) * a.n_chunks + chunk) * 2 + 1] = L[i];
            }
    }
    // ---- probabilities: this lane's two positions under a head are one Q8 block of 32 along the context (with the other 15
    //      lanes of its row); then the V row's block delta folded in and rounded to f16: the A operand of p.V,
    //      [half][head][position]
    float dvl[2][2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        dvl[u][0] = h2f((uint16_t)(vl[pl_[u] * NW] & 0xffffu)); dvl[u][1] = h2f((uint16_t)(vl[pl_[u] * NW + 8] >> 16));
        live[u] = c0 + pl_[u] < n;                               // (rows past the context hold arbitrary deltas: keep them out)
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float rL = recip_rn(L[i]);
        const float p0 = sc[0][i] * rL, p1 = sc[1][i] * rL;
        const Q8Scale qs = q8_scale_from_absmax(row16_max_f(fmaxf(p0, p1)));
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const float pq = (float)q8_round(u ? p1 : p0, qs.scale) * qs.ddeq;
            pl[(0 * 8 + 4 * hq + i) * PP + pl_[u]] = f2hv(live[u] ? pq * dvl[u][0] : 0.f);
            pl[(1 * 8 + 4 * hq + i) * PP + pl_[u]] = f2hv(live[u] ? pq * dvl[u][1] : 0.f);
        }
    }
    __syncthreads();

    // ---- p.V on the matrix cores: wave pw owns elements 16 pw .. 16 pw + 15 (block half pw >> 1), 32 positions per instruction
    att_f4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        const _Float16* prow = pl + ((pw >> 1) * 8 + (lc & 7)) * PP + 8 * lq;
        // element e of a slice sits at byte 2 + e (block 0) or 4 + e (block 1: behind the second delta)
        const uint8_t* vcol = (const uint8_t*)vl + (size_t)(8 * lq) * (NW * 4) + 16 * pw + lc + ((pw >> 1) ? 4 : 2);
#pragma unroll
        for (int ks = 0; ks < DEC_CHUNK / 32; ks++) {
            const att_h8 av = *(const att_h8*)(prow + 32 * ks);
            unsigned hb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned b0 = vcol[(size_t)(32 * ks + 2 * j) * (NW * 4)], b1 = vcol[(size_t)(32 * ks + 2 * j + 1) * (NW * 4)];
                // int8 -> exact f16: 0x6400 | (b ^ 0x80) is 1024 + (b + 128); minus 1152
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const unsigned u = (b0 | (b1 << 16)) ^ 0x64806480u;
                h2 hv = __builtin_bit_cast(h2, u) - (h2){(_Float16)1152.0f, (_Float16)1152.0f};
                hb[j] = __builtin_bit_cast(unsigned, hv);
            }
            att_h8 bv;
            __builtin_memcpy(&bv, hb, 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (4 * lq + i < GRP) a.att_part[((size_t)(g * GRP + 4 * lq + i) * a.n_chunks + chunk) * dh + 16 * pw + lc] = acc[i];
}

// greedy argmax, strict '>' so the first maximum wins (tinyllama.cpp:416-424).
// Works on (value, index) candidates: either the logits themselves (idx == null)
// or the per-wave winners the lm_head kernel left behind.
__global__ __launch_bounds__(1024) void k_dec_argmax(const float* __restrict__ vals0, const int* __restrict__ idxs0, int count,
                                                     DecStep* step0, int32_t* __restrict__ result0, int cand_stride, int result_stride,
                                                     int32_t* __restrict__ tokens0, int tok_stride)
{
    // one workgroup per sequence
    const float* vals = vals0 + (size_t)blockIdx.x * cand_stride;
    const int* idxs = idxs0 ? idxs0 + (size_t)blockIdx.x * cand_stride : nullptr;
    DecStep* step = step0 + blockIdx.x;
    int32_t* result = result0 + (size_t)blockIdx.x * result_stride;
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const float v = vals[i];
        const int vi = idxs ? idxs[i] : i;
        if (v > best || (v == best && vi < idx)) { best = v; idx = vi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { bv[wid] = best; bi[wid] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        if (idx == 0x7fffffff) idx = 0;
        const int n = step->n;
        result[n] = idx;                       // argmax of the step that computed row n-1
        const int adv = step->advance;
        if (adv & 2) tokens0[(size_t)blockIdx.x * tok_stride + n] = idx;   // greedy generation: the next step embeds it (tinyllama.cpp:426)
        const int stop = step->stop;
        if ((adv & 1) && (stop <= 0 || n < stop)) step->n = n + 1;          // free-running replay: the next launch decodes row n
    }
}

// --------------------------------------------------------------- host side

using namespace gtr;

struct gten_hip_decoder {
    gten_hip_decoder_desc d;
    std::vector<gten_hip_layer_ptrs> layers;
    DecStep* step = nullptr;
    int32_t* tokens = nullptr;     // [max_ctx + 1] teacher-forcing / prompt ids
    int32_t* result = nullptr;     // [max_ctx + 2] argmax per step, indexed by n
    int32_t* ids_stage = nullptr;  // [n_seq][64]: gten_hip_decoder_slot_ids_all's gather (made on first use)
    float *qkv_raw = nullptr, *proj_raw = nullptr, *down_raw = nullptr;
    float *scores = nullptr, *stats = nullptr, *att_part = nullptr;
    uint8_t *xbuf = nullptr, *hbuf = nullptr;
    float* act_f = nullptr;        // FFN activation staged by the gate/up epilogue, f16 configurations
    int8_t* act_q = nullptr;       // FFN activation staged by the gate/up epilogue (ActQ8 layout)
    float* act_d = nullptr;
    int* act_sum = nullptr;
    float* best_val = nullptr;     // lm_head per-wave winners
    int* best_idx = nullptr;
    int n_best = 0;
    std::vector<DecStep> slots;    // continuous batching (slot_start / slot_park / run): host view of every slot's step word
    int dev_n = -1;                // value of step->n on the device after the queued work (-1: unknown)
    std::vector<int> dev_ns;       // the same per sequence after a ragged step (empty: uniform, see dev_n)
    int only_family = -1;          // >= 0: enqueue only the launches of this kernel family (timing replays)
    // ---- multi-sequence decode (n_seq > 1): per-sequence rows of every scratch buffer above, plus
    int n_seq = 1;
    const void** kv_tab = nullptr; // device: [n_seq][n_layers][k|v]
    std::vector<const void*> kv_real;   // the same on the host (slot_park points a slot's entries at the dummy caches, slot_start back)
    std::vector<char> kv_parked;        // per sequence: entries currently redirected
    void* dummy_kv = nullptr;           // one K and one V cache nobody reads meaningfully: where a PARKED slot's idle steps write
    int8_t* stg_q = nullptr;       // staged n_embd-wide input of the next W.x: [n_seq] ActQ8 / f32 rows
    float* stg_d = nullptr;
    int* stg_sum = nullptr;
    float* stg_f = nullptr;
    float* logits_m = nullptr;     // [n_seq][n_vocab]
    float* gu_raw = nullptr;       // [n_seq][2 n_ffn] raw gate | up rows (wide decode, n_seq >= 16)
    int n_chunks = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph_k = nullptr;       // DEC_GRAPH_STEPS consecutive steps in one graph (the position lives on the device and
    hipGraphExec_t exec_k = nullptr;    // the last kernel of a step advances it): one replay per DEC_GRAPH_STEPS tokens
    // continuous batching (gten_hip_decoder_run): the same two graphs per SUBSET of lanes -- a lane whose slots are all parked is
    // left out of the step (its launch chain costs what a full lane costs however few slots are live); index = lane mask
    hipGraph_t graph_m[1 << DEC_MAX_LANES] = {}, graph_km[1 << DEC_MAX_LANES] = {};
    hipGraphExec_t exec_m[1 << DEC_MAX_LANES] = {}, exec_km[1 << DEC_MAX_LANES] = {};
    unsigned lane_mask = 0;             // lanes the NEXT enqueue takes (0: all)
    int last_run_lanes = 0;             // lanes the last gten_hip_decoder_run took
    const float2* rope = nullptr;
    float2* rope_now = nullptr;       // [n_seq][d_head / 2], see Gemv8Args
    // ---- more than 64 sequences: LANES.  The step of a wide decoder is a chain of ~180 dependent launches that leaves most
    // of the chip idle between and inside them; a decoder of 128 / 192 / 256 sequences runs 2 / 3 / 4 such chains -- one per
    // lane of 64 sequences, each on the rows of every buffer that belong to its sequences -- as parallel branches of ONE
    // graph (fork behind the previous replay, join at the end), so the chains fill each other's gaps.  Per sequence the
    // kernels, their arguments and therefore the results are those of a 64-sequence decoder.
    bool exact = false;               // gten_hip_set_decode_exact at creation
    bool persist_on = false;          // gten_hip_set_decode_persistent at creation: the step as ONE persistent launch (gten_decode_persist.h)
    struct PersistState* persist = nullptr;
    int lanes = 1;
    hipStream_t lane_stream[DEC_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};    // capture / eager side streams of lanes 1..
    hipEvent_t lane_fork = nullptr, lane_join[DEC_MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
};

// the rows of every per-sequence buffer that belong to one lane (lane 0 of a single-lane decoder: the buffers themselves)
struct LaneBufs {
    int n_seq;
    DecStep* step; int32_t* tokens; int32_t* result;
    float *qkv_raw, *proj_raw, *down_raw, *scores, *stats, *att_part;
    uint8_t *xbuf, *hbuf;
    float* act_f; int8_t* act_q; float* act_d; int* act_sum;
    const void** kv_tab;
    int8_t* stg_q; float* stg_d; int* stg_sum; float* stg_f;
    float* logits_m; float* gu_raw; float2* rope_now;
};
static LaneBufs lane_bufs(const gten_hip_decoder* dc, int lane)
{
    const gten_hip_decoder_desc& d = dc->d;
    const size_t SL = (size_t)(dc->n_seq / dc->lanes), o = (size_t)lane * SL;      // sequences per lane, first sequence of this lane
    const size_t E = (size_t)d.n_embd, F = (size_t)d.n_ffn, dh = E / d.n_heads, KV = dh * d.n_kv_heads, V = (size_t)d.n_vocab;
    const size_t planes = dc->n_seq >= 16 ? 2 : 1, H = (size_t)d.n_heads, C = (size_t)dc->n_chunks;
    LaneBufs b;
    b.n_seq = (int)SL;
    b.step = dc->step + o; b.tokens = dc->tokens + o * (d.max_ctx + 1); b.result = dc->result + o * (d.max_ctx + 2);
    b.qkv_raw = dc->qkv_raw + o * planes * (E + 2 * KV); b.proj_raw = dc->proj_raw + o * planes * E; b.down_raw = dc->down_raw + o * planes * E;
    b.scores = dc->scores + o * H * d.max_ctx; b.stats = dc->stats + o * H * C * 2; b.att_part = dc->att_part + o * H * C * dh;
    b.xbuf = dc->xbuf + o * E * 4; b.hbuf = dc->hbuf + o * E * 4;
    b.act_f = dc->act_f + o * F; b.act_q = dc->act_q + o * 2 * F; b.act_d = dc->act_d + o * (F / 32); b.act_sum = dc->act_sum + o * (F / 32);
    b.kv_tab = dc->kv_tab ? dc->kv_tab + o * d.n_layers * 2 : nullptr;
    b.stg_q = dc->stg_q ? dc->stg_q + o * 2 * E : nullptr; b.stg_d = dc->stg_d ? dc->stg_d + o * (E / 32) : nullptr;
    b.stg_sum = dc->stg_sum ? dc->stg_sum + o * (E / 32) : nullptr; b.stg_f = dc->stg_f ? dc->stg_f + o * E : nullptr;
    b.logits_m = dc->logits_m ? dc->logits_m + o * V : nullptr; b.gu_raw = dc->gu_raw ? dc->gu_raw + o * planes * 2 * F : nullptr;
    b.rope_now = dc->rope_now + o * (dh / 2);
    return b;
}

// gten_hip_set_decode_exact (include/gten_hip.h): the exact forms of the decode step -- row-global rounding points of the
// attention probabilities (two launches), exact p.V terms, integer block sums in the wide W.x (k_dec_mmv) -- instead of
// the fast ones.  A decoder keeps the choice it was created with (its graphs are captured once); g_exact_now is that
// choice while one of its steps is being enqueued.
static bool g_decode_exact = false;
static bool g_exact_now = false;
// off by default: on the bench's serving queue (1024 prompts through two lanes of 128 slots) only 3 % of the lane-steps have an
// empty lane, and the extra graphs cost more than that returns: 33.5 k against 37.0 k new ids/s, A/B on one box
static bool g_lane_skip = false;         // gten_hip_decoder_run leaves lanes without a live slot out of the step (gten_hip_set_lane_skip)
extern "C" int gten_hip_set_decode_exact(int on)
{
    g_decode_exact = on != 0;
    return 0;
}

// many sequences, Q8 activations, 64-wide heads, 8 (or 4, 2, 1) query heads per kv head: grouped kernels
static bool attention_grouped_ok(const AttnArgs& t, int n_seq)
{
    const int grp = t.n_heads / t.n_kv;
    const bool off = false;
    const int min_seq = 8;                                       // measured (q4, ctx 2048): 8 sequences +6 %, 4 and 2 slower
    if (t.adtype == GTEN_F16 && n_seq < 16) return false;        // f16 below 16 sequences: the per-head kernels (one launch, k_dec_attn_one64)
    return !off && n_seq >= min_seq && (t.adtype == GTEN_Q8 || t.adtype == GTEN_F16) && t.d_head == 64 && (grp == 8 || grp == 4 || grp == 2 || grp == 1);
}

static int grp_shift1_of(int n_heads, int n_kv)
{
    const int grp = n_heads / n_kv;
    return (grp > 0 && (grp & (grp - 1)) == 0) ? __builtin_ctz(grp) + 1 : 0;
}

static bool attention_one_pass(int d_head);

// which grouped launches run in the one-pass form (k_dec_attn_one_g; the consumer then joins with PRO_ATTW): every
// Q8 configuration; f16 only below 16 sequences is per-head anyway, from 16 up its scores stay on the matrix cores
// (k_dec_attn_score_gm_f16) in the two-launch form
// 16-64 sequences with Q8 activations: k_dec_attn_mm_g (one launch, matrix cores).  The exact forms
// (gten_hip_set_decode_exact) keep the VALU pair: exact p.V terms, row-global rounding points; tests compare.
static bool grouped_mm(const AttnArgs& t, int n_seq)
{
    return n_seq >= 16 && t.adtype == GTEN_Q8 && t.d_head == 64 && !g_exact_now;
}

static bool grouped_one_pass(const AttnArgs& t, int n_seq)
{
    if (grouped_mm(t, n_seq)) return true;
    // 8 sequences: one launch (per sequence the bytes of single-sequence decode).  From 16 sequences up the two
    // launches measured FASTER than the merged kernel (64 sequences, ctx 2048: 22.3 + 25.8 us against 62.4 us per
    // block -- the merged workgroup holds K rows, V chunk and every head's scores at once: 104 VGPRs, 27 KB of LDS,
    // 4 workgroups per CU through five barriers each), so the wide path keeps the two-launch pair.
    return attention_one_pass(t.d_head) && t.adtype == GTEN_Q8 && n_seq <= 8;
}

template <int GRP, int ADT>
static int launch_attention_g(const AttnArgs& t, int n_seq)
{
    constexpr size_t NW = (ADT == GTEN_Q8) ? 17 : 32;
    const dim3 grid(n_seq, t.n_chunks, t.n_kv);
    if constexpr (ADT == GTEN_Q8) if (grouped_mm(t, n_seq)) {
        const size_t smem = (size_t)2 * DEC_CHUNK * 68 + 16 * 64 + 32 * 4 + 8 * 4 + (size_t)4 * (DEC_MAXGRP + 2) * 2 + 128 + 128 * 4 + 80 + 64;
        GTR_REQUIRE((n_seq * t.n_chunks) % 8 == 0, "decoder: %d sequences x %d chunks is not a multiple of 8", n_seq, t.n_chunks);
        DEC_LAUNCH(KT_DEC_ATTN_SCORE, (k_dec_attn_mm_g<GRP>), dim3(n_seq * t.n_chunks * t.n_kv), dim3(256), smem, t, n_seq);
        return 0;
    }
    if constexpr (ADT == GTEN_Q8) if (grouped_one_pass(t, n_seq)) {
        constexpr size_t GP = (GRP + 1) / 2;
        const size_t smem = (size_t)(8 * GRP + 4 * GRP + 8 + 64) * 4 + (size_t)4 * (GRP + 2) * 2 + (size_t)(GRP + 2) * 64 + 16 +
                            (ADT == GTEN_Q8 ? 0 : (size_t)(GRP + 1) * 64 * 4) + 2 * GP * DEC_CHUNK * 4 + (size_t)DEC_CHUNK * NW * 4;
        DEC_LAUNCH(KT_DEC_ATTN_SCORE, (k_dec_attn_one_g<GRP, true, ADT>), grid, dim3(256), smem, t);      // (<= 8 sequences: exact p.V terms)
        return 0;
    }
    const size_t smem1 = (size_t)(8 * GRP + 4 * GRP + 8 + 64) * 4 + (size_t)4 * (GRP + 2) * 2 + (size_t)(GRP + 2) * 64 + 64 +
                         (ADT == GTEN_Q8 ? 0 : (size_t)(GRP + 1) * 64 * 4 + 16);
    const size_t smem2 = (size_t)((GRP + 1) / 2 * 2) * DEC_CHUNK * 4 + (size_t)DEC_CHUNK * NW * 4 + (size_t)(16 + 8 * GRP) * 4;
    // exact p.V terms up to 8 sequences (bit-identical to single-sequence decode) or on request (gten_hip_set_decode_exact)
    const bool exact = n_seq <= 8 || g_exact_now;
    if (smem2 > 64 * 1024) {
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_attn_pv_g<GRP, true, ADT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_attn_pv_g<GRP, false, ADT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
    }
    if (ADT == GTEN_F16 && n_seq >= 16) {
        // 16 sequences and up: the scores of a group as a matrix product on the matrix cores (k_dec_attn_score_gm_f16)
        const size_t smem_m = (size_t)(128 + 16) * 4 + 128 + 128 + 64 + 16 + (size_t)17 * 64 * 2 + 64;
        DEC_LAUNCH(KT_DEC_ATTN_SCORE, (k_dec_attn_score_gm_f16<GRP>), grid, dim3(256), smem_m, t);
    } else if (ADT == GTEN_F16) {
        // f16 scores are 2 VALU operations per (head, position, element) however the heads are grouped (f32 products added in
        // element order): the per-head kernel spreads them over 8x the workgroups and measured faster (16 / 32 sequences:
        // 20.7 / 36.1 us against 30.2 / 53.8) -- same scores, statistics and cache rows, so p.V below can still be grouped
        AttnArgs t1 = t;
        t1.grp_shift1 = grp_shift1_of(t.n_heads, t.n_kv);
        const AttnHotWords none{{0, 0, 0, 0, 0, 0, 0}};
        const size_t smem_s = (size_t)(16 + 3 * 64 + 16) * 4 + 32 + (size_t)3 * 64 + 64;
        DEC_LAUNCH_HOT(KT_DEC_ATTN_SCORE, (k_dec_attn_score64<GTEN_F16, true>), dim3(t.n_chunks, t.n_heads, n_seq), dim3(256), smem_s, none, t1);
    } else {
        DEC_LAUNCH(KT_DEC_ATTN_SCORE, (k_dec_attn_score_g<GRP, ADT>), grid, dim3(256), smem1, t);
    }
    if (exact) DEC_LAUNCH(KT_DEC_ATTN_PV, (k_dec_attn_pv_g<GRP, true, ADT>), grid, dim3(256), smem2, t);
    else DEC_LAUNCH(KT_DEC_ATTN_PV, (k_dec_attn_pv_g<GRP, false, ADT>), grid, dim3(256), smem2, t);
    return 0;
}

static int launch_attention_grouped(const AttnArgs& t, int n_seq)
{
    if (t.adtype == GTEN_Q8) {
        switch (t.n_heads / t.n_kv) {
        case 8: return launch_attention_g<8, GTEN_Q8>(t, n_seq);
        case 4: return launch_attention_g<4, GTEN_Q8>(t, n_seq);
        case 2: return launch_attention_g<2, GTEN_Q8>(t, n_seq);
        default: return launch_attention_g<1, GTEN_Q8>(t, n_seq);
        }
    }
    switch (t.n_heads / t.n_kv) {
    case 8: return launch_attention_g<8, GTEN_F16>(t, n_seq);
    case 4: return launch_attention_g<4, GTEN_F16>(t, n_seq);
    case 2: return launch_attention_g<2, GTEN_F16>(t, n_seq);
    default: return launch_attention_g<1, GTEN_F16>(t, n_seq);
    }
}

// One launch with chunk-local statistics (k_dec_attn_one64) wherever the per-head 64-wide kernels run; the consumer
// must then join the chunks with PRO_ATTW.  The exact forms (gten_hip_set_decode_exact) keep the two launches, whose
// probabilities are rounded against the statistics of the whole row exactly as the reference stores them (the contexts
// beyond one chunk then match the operator path's rounding points; tests compare both).
static bool attention_one_pass(int d_head)
{
    return d_head == 64 && !g_exact_now;
}

static int launch_attention(const AttnArgs& t0, dim3 agrid, size_t smem1)
{
    AttnArgs t = t0;
    t.grp_shift1 = grp_shift1_of(t.n_heads, t.n_kv);
    if (attention_one_pass(t.d_head)) {
        const size_t nw = (t.adtype == GTEN_Q8) ? 17 : 32;
        const size_t smem = 1152 + (size_t)2 * DEC_CHUNK * 4 + (size_t)DEC_CHUNK * nw * 4;
        const unsigned long long geo = (unsigned long long)(unsigned)t.kv_pitch | ((unsigned long long)(unsigned)t.max_ctx << 32);
        const unsigned long long heads = ((unsigned long long)(unsigned)t.n_heads << 32) | ((unsigned long long)(unsigned)t.n_kv << 40) |
                                         ((unsigned long long)(unsigned)t.grp_shift1 << 48);
        const AttnHotWords hw{{(unsigned long long)(uintptr_t)t.qkv_raw, (unsigned long long)(uintptr_t)t.rope_now, (unsigned long long)(uintptr_t)t.kcache,
                               (unsigned long long)(uintptr_t)t.step, geo, (unsigned long long)(unsigned)t.n_embd | heads, (unsigned long long)(uintptr_t)t.vcache}};
        const bool multi = agrid.z > 1 || t.kv_tab != nullptr;
        const dim3 g64(agrid.y, agrid.x, agrid.z);     // (chunk, head, sequence)
        if (t.adtype == GTEN_Q8) {
            if (multi) DEC_LAUNCH_HOT(KT_DEC_ATTN_SCORE, (k_dec_attn_one64<GTEN_Q8, true>), g64, dim3(256), smem, hw, t);
            else DEC_LAUNCH_HOT(KT_DEC_ATTN_SCORE, (k_dec_attn_one64<GTEN_Q8, false>), g64, dim3(256), smem, hw, t);
        } else {
            if (multi) DEC_LAUNCH_HOT(KT_DEC_ATTN_SCORE, (k_dec_attn_one64<GTEN_F16, true>), g64, dim3(256), smem, hw, t);
            else DEC_LAUNCH_HOT(KT_DEC_ATTN_SCORE, (k_dec_attn_one64<GTEN_F16, false>), g64, dim3(256), smem, hw, t);
        }
        return 0;
    }
    if (t.d_head == 64) {
        const size_t nw = (t.adtype == GTEN_Q8) ? 17 : 32;
        const size_t smem2 = (size_t)2 * DEC_CHUNK * 4 + (size_t)DEC_CHUNK * nw * 4;
        // single-sequence launches hand the words the first requests are formed from as preloadable scalars
        const unsigned long long geo = (unsigned long long)(unsigned)t.kv_pitch | ((unsigned long long)(unsigned)t.max_ctx << 32);
        const unsigned long long heads = ((unsigned long long)(unsigned)t.n_heads << 32) | ((unsigned long long)(unsigned)t.n_kv << 40) |
                                         ((unsigned long long)(unsigned)t.grp_shift1 << 48);
        const AttnHotWords hs{{(unsigned long long)(uintptr_t)t.qkv_raw, (unsigned long long)(uintptr_t)t.rope_now, (unsigned long long)(uintptr_t)t.kcache,
                               (unsigned long long)(uintptr_t)t.step, geo, (unsigned long long)(unsigned)t.n_embd | heads, (unsigned long long)(uintptr_t)t.scores}};
        const AttnHotWords hp{{(unsigned long long)(uintptr_t)t.scores, (unsigned long long)(uintptr_t)t.stats, (unsigned long long)(uintptr_t)t.vcache,
                               (unsigned long long)(uintptr_t)t.step, geo, (unsigned long long)(unsigned)t.n_chunks | heads, (unsigned long long)(uintptr_t)t.att_part}};
        const bool multi = agrid.z > 1 || t.kv_tab != nullptr;
        const dim3 g64(agrid.y, agrid.x, agrid.z);     // (chunk, head, sequence): see k_dec_attn_score64
#define ATT_LAUNCH2(ADT, MULTI)                                                                                                          \
        do {                                                                                                                             \
            DEC_LAUNCH_HOT(KT_DEC_ATTN_SCORE, (k_dec_attn_score64<ADT, MULTI>), g64, dim3(256), smem1, hs, t);                           \
            DEC_LAUNCH_HOT(KT_DEC_ATTN_PV, (k_dec_attn_pv64<ADT, MULTI>), g64, dim3(256), smem2, hp, t);                                 \
        } while (0)
        if (t.adtype == GTEN_Q8) { if (multi) ATT_LAUNCH2(GTEN_Q8, true); else ATT_LAUNCH2(GTEN_Q8, false); }
        else { if (multi) ATT_LAUNCH2(GTEN_F16, true); else ATT_LAUNCH2(GTEN_F16, false); }
#undef ATT_LAUNCH2
        return 0;
    }
    DEC_LAUNCH(KT_DEC_ATTN_SCORE, k_dec_attn_score, agrid, dim3(256), smem1, t);
    DEC_LAUNCH(KT_DEC_ATTN_PV, k_dec_attn_pv, agrid, dim3(256), (size_t)2 * DEC_CHUNK * 4, t);
    return 0;
}

// the preloadable words of a k_dec_gemv8 launch (GemvHot)
template <int WT, int PRO>
static GemvHotWords hot_of(const Gemv8Args& a)
{
    GemvHotWords hw;
    GemvHot& h = hw.h;
    if (PRO == PRO_RESID) { h.p0 = a.res_raw; h.p1 = a.res_a; h.p2 = a.norm_w; }
    else if (PRO == PRO_EMBED) { h.p0 = a.table; h.p1 = a.tokens; h.p2 = a.norm_w; }
    else if (PRO == PRO_ATT || PRO == PRO_ATTW) {
        h.p0 = a.att_part; h.p1 = (const void*)(uintptr_t)((unsigned)a.d_head_shift | ((unsigned)a.n_chunks << 8));
        h.p2 = (PRO == PRO_ATTW) ? a.att_stats : nullptr;
    }
    else { h.p0 = (WT == GTEN_F16) ? (const void*)a.act_f : (const void*)a.act_q; h.p1 = a.act_d; h.p2 = a.act_sum; }
    h.qs0 = a.qs[0]; h.ds0 = a.ds[0]; h.d_in = a.d_in; h.rows0 = a.rows[0]; h.step = a.step;
    return hw;
}

// ... and in k_dec_gemv8's own layout (see the kernel): matrices 1 and 2 and all row counts ride along
template <int WT, int PRO>
static GemvHotWords hot_of_gemv8(const Gemv8Args& a)
{
    GemvHotWords hw = hot_of<WT, PRO>(a);
    const unsigned long long r1 = a.n_mats > 1 ? (unsigned)a.rows[1] : 0u, r2 = a.n_mats > 2 ? (unsigned)a.rows[2] : 0u;
    hw.w[4] = (unsigned long long)(uintptr_t)(a.n_mats > 1 ? a.qs[1] : nullptr);
    hw.w[5] = (unsigned long long)(unsigned)a.d_in | ((unsigned long long)(unsigned)a.rows[0] << 16) | (r1 << 32) | (r2 << 48);
    hw.w[6] = (PRO == PRO_RESID) ? (unsigned long long)(uintptr_t)(a.n_mats > 2 ? a.qs[2] : nullptr) : (unsigned long long)(uintptr_t)a.step;
    return hw;
}

template <int WT, int PRO, int NCH, int R, int NT, int NM = 0>
static int launch_gemv8(int tag, const Gemv8Args& a, int total_rows)
{
    const int rows_per_wg = (NT / 64) * R;
    const dim3 grid((total_rows + rows_per_wg - 1) / rows_per_wg), block(NT);
    for (int k = 0; k + 1 < a.n_mats; k++) GTR_REQUIRE(a.rows[k] % R == 0, "decoder: concatenated matrices must hold a multiple of %d rows", R);
    const GemvHotWords hw = hot_of_gemv8<WT, PRO>(a);
    DEC_LAUNCH_HOT(tag, (k_dec_gemv8<WT, PRO, NCH, R, EPI_RAW, NT, NM>), grid, block, stage_bytes((PRO == PRO_ACTQ8 && WT != GTEN_F16) ? 32 : a.d_in), hw, a);
    return 0;
}

// gate + up projections with the silu*up chain in the epilogue: one workgroup per 32-wide FFN slice
template <int WT>
static int launch_gateup8(const Gemv8Args& a, int n_ffn)
{
    const dim3 grid(n_ffn / 32), block(512);
    GTR_REQUIRE(n_ffn % 32 == 0, "decoder: n_ffn %d is not a multiple of 32", n_ffn);
    const GemvHotWords hw = hot_of_gemv8<WT, PRO_RESID>(a);
    DEC_LAUNCH_HOT(KT_DEC_GEMV_GATEUP, (k_dec_gemv8<WT, PRO_RESID, (WT == GTEN_F16 ? 4 : 1), 8, EPI_SILUMUL, 512>), grid, block, stage_bytes(a.d_in), hw, a);
    return 0;
}

static void set_mat(Gemv8Args& a, int k, const void* w, int wdtype, int rows, int cols)
{
    const size_t nb = (size_t)cols / 32;
    a.qs[k] = (const uint8_t*)w;
    a.ds[k] = (wdtype == GTEN_F16) ? nullptr
                                   : (const uint16_t*)((const uint8_t*)w + (size_t)rows * nb * (wdtype == GTEN_Q4 ? 16 : 32));
    a.rows[k] = rows;
}

// The register-prologue kernels, all three configurations.  NCH = lane passes over a weight
// row: 64 quant blocks (2048 elements) per pass for Q8/Q4, one 512-element segment per pass for f16.
template <int WT>
static int enqueue_step_q8act(gten_hip_decoder* dc)
{
    const gten_hip_decoder_desc& d = dc->d;
    const int E = d.n_embd, F = d.n_ffn, dh = E / d.n_heads, KV = dh * d.n_kv_heads;
    const size_t kv_pitch = gten_hip_row_bytes(d.adtype, KV);
    constexpr bool F16W = (WT == GTEN_F16);
    constexpr int NE = F16W ? 4 : 1;              // passes over an n_embd-wide row (<= 2048)
    constexpr int NF = F16W ? 11 : 3;             // passes over an n_ffn-wide row (<= 5632 / 6144)
    const bool wideF = F16W ? (F > 2048) : (F > 2048);
    float* xbuf = (float*)dc->xbuf;
    float* hbuf = (float*)dc->hbuf;
    int rc;
    for (int l = 0; l < d.n_layers; l++) {
        const gten_hip_layer_ptrs& L = dc->layers[l];
        Gemv8Args a{};
        a.step = dc->step; a.d_in = E; a.n_mats = 3;
        set_mat(a, 0, L.wq, WT, E, E); set_mat(a, 1, L.wk, WT, KV, E); set_mat(a, 2, L.wv, WT, KV, E);
        a.out = dc->qkv_raw; a.norm_w = (const uint16_t*)L.attn_norm; a.x_out = xbuf;
        if (l == 0) {
            a.table = d.embed; a.rope = dc->rope; a.rope_now = dc->rope_now; a.rope_half = dh / 2; a.n_vocab = d.n_vocab; a.tokens = dc->tokens;
            rc = launch_gemv8<WT, PRO_EMBED, NE, 2, 512>(KT_DEC_GEMV_QKV, a, E + 2 * KV);
        } else {
            a.res_a = hbuf; a.res_raw = dc->down_raw;
            rc = launch_gemv8<WT, PRO_RESID, NE, 2, 512>(KT_DEC_GEMV_QKV, a, E + 2 * KV);
        }
        if (rc) return rc;
        AttnArgs t{};
        t.step = dc->step; t.qkv_raw = dc->qkv_raw; t.kcache = (uint8_t*)L.kcache; t.vcache = (uint8_t*)L.vcache;
        t.kv_pitch = kv_pitch; t.scores = dc->scores; t.stats = dc->stats; t.att_part = dc->att_part; t.rope = dc->rope; t.rope_now = dc->rope_now;
        t.adtype = d.adtype; t.n_heads = d.n_heads; t.n_kv = d.n_kv_heads; t.d_head = dh; t.max_ctx = d.max_ctx;
        t.n_chunks = dc->n_chunks; t.n_embd = E;
        const dim3 agrid(d.n_heads, dc->n_chunks);
        const size_t smem1 = (size_t)(16 + 3 * dh + 16) * 4 + 32 + (size_t)3 * dh + 64;
        if (int arc = launch_attention(t, agrid, smem1)) return arc;
        Gemv8Args o{};
        o.step = dc->step; o.d_in = E; o.n_mats = 1; set_mat(o, 0, L.wo, WT, E, E); o.out = dc->proj_raw;
        o.att_part = dc->att_part; o.d_head = dh; o.d_head_shift = __builtin_ctz(dh); o.n_chunks = dc->n_chunks;
        o.att_stats = dc->stats;
        // rows per wave and threads per workgroup, measured (q4, step ms; 0.4976 with 2 rows on 512 threads everywhere but down's
        // 256): q|k|v 1 row 0.510, on 256 threads 0.500; o 1 row 0.4845 (3.90 against 4.22 us per launch: every CU gets a workgroup),
        // 4 rows 0.512, 1 row on 256 threads 0.504, 2 rows on 256 threads 0.489; down 1 row 0.494, 1 row on 512 threads 0.496;
        // lm_head 4 / 8 / 16 rows per wave 0.486 / 0.488 / 0.487 (then, with o at 1 row)
        rc = attention_one_pass(dh) ? launch_gemv8<WT, PRO_ATTW, NE, 1, 512, 1>(KT_DEC_GEMV_O, o, E)
                                    : launch_gemv8<WT, PRO_ATT, NE, 1, 512, 1>(KT_DEC_GEMV_O, o, E);
        if (rc) return rc;
        Gemv8Args gu{};
        gu.step = dc->step; gu.d_in = E; gu.n_mats = 2; set_mat(gu, 0, L.wgate, WT, F, E); set_mat(gu, 1, L.wup, WT, F, E);
        gu.res_a = xbuf; gu.res_raw = dc->proj_raw; gu.x_out = hbuf; gu.norm_w = (const uint16_t*)L.ffn_norm;
        gu.act_q = dc->act_q; gu.act_d = dc->act_d; gu.act_sum = dc->act_sum; gu.act_f = dc->act_f;
        if ((rc = launch_gateup8<WT>(gu, F))) return rc;
        Gemv8Args dn{};
        dn.step = dc->step; dn.d_in = F; dn.n_mats = 1; set_mat(dn, 0, L.wdown, WT, E, F); dn.out = dc->down_raw;
        dn.act_q = dc->act_q; dn.act_d = dc->act_d; dn.act_sum = dc->act_sum;
        dn.act_f = dc->act_f;
        rc = wideF ? launch_gemv8<WT, PRO_ACTQ8, NF, 2, 256, 1>(KT_DEC_GEMV_DOWN, dn, E)
                   : launch_gemv8<WT, PRO_ACTQ8, NE, 2, 256, 1>(KT_DEC_GEMV_DOWN, dn, E);
        if (rc) return rc;
    }
    Gemv8Args hd{};
    hd.step = dc->step; hd.d_in = E; hd.n_mats = 1; set_mat(hd, 0, d.lm_head, WT, d.n_vocab, E); hd.out = d.logits;
    hd.res_a = hbuf; hd.res_raw = dc->down_raw; hd.x_out = nullptr; hd.norm_w = (const uint16_t*)d.final_norm;
    hd.best_val = dc->best_val; hd.best_idx = dc->best_idx;
    if ((rc = launch_gemv8<WT, PRO_RESID, NE, F16W ? 4 : 8, 512, 1>(KT_DEC_GEMV_HEAD, hd, d.n_vocab))) return rc;
    DEC_LAUNCH(KT_DEC_ARGMAX, k_dec_argmax, dim3(1), dim3(1024), 0, (const float*)dc->best_val, (const int*)dc->best_idx,
               dc->n_best, dc->step, dc->result, 0, 0, dc->tokens, d.max_ctx + 1);
    return 0;
}

// ---- one decode step of n_seq sequences that share every weight pass
template <int WT, int PRO>
static int launch_stage(int tag, Gemv8Args a, int n_seq)
{
    const GemvHotWords hw = hot_of_gemv8<WT, PRO>(a);
    DEC_LAUNCH_HOT(tag, (k_dec_gemv8<WT, PRO, 1, 1, EPI_STAGE, 512>), dim3(n_seq), dim3(512), stage_bytes(a.d_in), hw, a);
    return 0;
}

template <int WT, int PRO>
static int launch_stage_frag(int tag, Gemv8Args a, int n_seq)
{
    a.frag_rt = (n_seq + 15) / 16;
    const GemvHotWords hw = hot_of_gemv8<WT, PRO>(a);
    DEC_LAUNCH_HOT(tag, (k_dec_gemv8<WT, PRO, 1, 1, EPI_STAGE_FRAG, 512>), dim3(n_seq), dim3(512), stage_bytes(a.d_in), hw, a);
    return 0;
}

static size_t gemvm_lds_bytes(int wt, int n_seq, int d, bool silumul)
{
    size_t b = silumul ? (size_t)n_seq * 64 * 4 : 0;
    if (wt == GTEN_F16) b += ((size_t)n_seq * ((d + 511) / 512 * 512) * 4 <= GEMVM_F16_LDS_LIMIT) ? (size_t)n_seq * d * 4 : 0;   // (k_dec_gemvm: lds_f)
    else b += (size_t)n_seq * (d + (size_t)(d / 32) * 8);
    return b + 16;
}

template <int WT, int NCH, int R, int S, int NT>
static int launch_gemvm(int tag, const Gemv8Args& a, int total_rows)
{
    const int rows_per_wg = (NT / 64) * R;
    for (int k = 0; k + 1 < a.n_mats; k++) GTR_REQUIRE(a.rows[k] % R == 0, "decoder: concatenated matrices must hold a multiple of %d rows", R);
    const GemvHotWords hw = hot_of<WT, PRO_ACTQ8>(a);
    const size_t lds = gemvm_lds_bytes(WT, S, a.d_in, false);
    static bool raised = false;                    // (per instantiation)
    if (lds > 64 * 1024 && !raised) {
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_gemvm<WT, NCH, R, S, EPI_RAW, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    DEC_LAUNCH_HOT(tag, (k_dec_gemvm<WT, NCH, R, S, EPI_RAW, NT>), dim3((total_rows + rows_per_wg - 1) / rows_per_wg), dim3(NT), lds, hw, a);
    return 0;
}

template <int WT, int S>
static int enqueue_step_multi(gten_hip_decoder* dc)
{
    const gten_hip_decoder_desc& d = dc->d;
    const int E = d.n_embd, F = d.n_ffn, dh = E / d.n_heads, KV = dh * d.n_kv_heads, V = d.n_vocab;
    const size_t kv_pitch = gten_hip_row_bytes(d.adtype, KV);
    constexpr bool F16W = (WT == GTEN_F16);
    constexpr int NE = F16W ? 4 : 1, NF = F16W ? 11 : 3;
    constexpr int RH = F16W ? 2 : 4;              // lm_head rows per wave
    const bool wideF = F > 2048;
    float* xbuf = (float*)dc->xbuf;
    float* hbuf = (float*)dc->hbuf;
    int rc;
    // stage descriptors shared by every launch
    Gemv8Args base{};
    base.step = dc->step; base.tok_stride = d.max_ctx + 1; base.part_stride = d.n_heads * dc->n_chunks * dh;
    base.best_stride = dc->n_best;
    for (int l = 0; l < d.n_layers; l++) {
        const gten_hip_layer_ptrs& L = dc->layers[l];
        // x (and its RMSNorm) per sequence -> stage
        Gemv8Args st = base;
        st.d_in = E; st.norm_w = (const uint16_t*)L.attn_norm; st.x_out = xbuf;
        st.act_q = dc->stg_q; st.act_d = dc->stg_d; st.act_sum = dc->stg_sum; st.act_f = dc->stg_f;
        if (l == 0) {
            st.table = d.embed; st.rope = dc->rope; st.rope_now = dc->rope_now; st.rope_half = dh / 2; st.n_vocab = V; st.tokens = dc->tokens;
            rc = launch_stage<WT, PRO_EMBED>(KT_DEC_STAGE, st, S);
        } else {
            st.res_a = hbuf; st.res_raw = dc->down_raw; st.raw_stride = E;
            rc = launch_stage<WT, PRO_RESID>(KT_DEC_STAGE, st, S);
        }
        if (rc) return rc;
        Gemv8Args a = base;
        a.d_in = E; a.n_mats = 3;
        set_mat(a, 0, L.wq, WT, E, E); set_mat(a, 1, L.wk, WT, KV, E); set_mat(a, 2, L.wv, WT, KV, E);
        a.out = dc->qkv_raw; a.raw_stride = E + 2 * KV;
        a.act_q = dc->stg_q; a.act_d = dc->stg_d; a.act_sum = dc->stg_sum; a.act_f = dc->stg_f;
        if ((rc = launch_gemvm<WT, NE, 2, S, 256>(KT_DEC_GEMV_QKV, a, E + 2 * KV))) return rc;
        // attention, one grid plane per sequence
        AttnArgs t{};
        t.step = dc->step; t.qkv_raw = dc->qkv_raw; t.kv_pitch = kv_pitch; t.scores = dc->scores; t.stats = dc->stats;
        t.att_part = dc->att_part; t.rope = dc->rope; t.rope_now = dc->rope_now;
        t.adtype = d.adtype; t.n_heads = d.n_heads; t.n_kv = d.n_kv_heads; t.d_head = dh; t.max_ctx = d.max_ctx;
        t.n_chunks = dc->n_chunks; t.n_embd = E;
        t.kv_tab = (const void* const*)dc->kv_tab; t.layer = l; t.n_layers = d.n_layers;
        t.qkv_stride = E + 2 * KV; t.scores_stride = d.n_heads * d.max_ctx; t.stats_stride = d.n_heads * dc->n_chunks * 2;
        t.part_stride = d.n_heads * dc->n_chunks * dh;
        const dim3 agrid(d.n_heads, dc->n_chunks, S);
        const size_t smem1 = (size_t)(16 + 3 * dh + 16) * 4 + 32 + (size_t)3 * dh + 64;
        const bool grouped = attention_grouped_ok(t, S);
        if ((rc = grouped ? launch_attention_grouped(t, S) : launch_attention(t, agrid, smem1))) return rc;
        // attention rows -> stage -> o projection
        Gemv8Args sa = base;
        sa.d_in = E; sa.att_part = dc->att_part; sa.d_head = dh; sa.d_head_shift = __builtin_ctz(dh); sa.n_chunks = dc->n_chunks;
        sa.att_stats = dc->stats; sa.stats_stride = d.n_heads * dc->n_chunks * 2;
        sa.act_q = dc->stg_q; sa.act_d = dc->stg_d; sa.act_sum = dc->stg_sum; sa.act_f = dc->stg_f;
        rc = (grouped ? grouped_one_pass(t, S) : attention_one_pass(dh)) ? launch_stage<WT, PRO_ATTW>(KT_DEC_STAGE, sa, S)
                                                                         : launch_stage<WT, PRO_ATT>(KT_DEC_STAGE, sa, S);
        if (rc) return rc;
        Gemv8Args o = base;
        o.d_in = E; o.n_mats = 1; set_mat(o, 0, L.wo, WT, E, E); o.out = dc->proj_raw; o.raw_stride = E;
        o.act_q = dc->stg_q; o.act_d = dc->stg_d; o.act_sum = dc->stg_sum; o.act_f = dc->stg_f;
        if ((rc = launch_gemvm<WT, NE, 2, S, 256>(KT_DEC_GEMV_O, o, E))) return rc;
        // h = x + proj (and its RMSNorm) -> stage -> gate/up with the silu*up chain in the epilogue
        Gemv8Args sh = base;
        sh.d_in = E; sh.res_a = xbuf; sh.res_raw = dc->proj_raw; sh.raw_stride = E; sh.x_out = hbuf;
        sh.norm_w = (const uint16_t*)L.ffn_norm;
        sh.act_q = dc->stg_q; sh.act_d = dc->stg_d; sh.act_sum = dc->stg_sum; sh.act_f = dc->stg_f;
        if ((rc = launch_stage<WT, PRO_RESID>(KT_DEC_STAGE, sh, S))) return rc;
        Gemv8Args gu = base;
        gu.d_in = E; gu.n_mats = 2; set_mat(gu, 0, L.wgate, WT, F, E); set_mat(gu, 1, L.wup, WT, F, E);
        gu.act_q = dc->stg_q; gu.act_d = dc->stg_d; gu.act_sum = dc->stg_sum; gu.act_f = dc->stg_f;
        gu.out_q = dc->act_q; gu.out_d = dc->act_d; gu.out_sum = dc->act_sum; gu.out_f = dc->act_f;
        const GemvHotWords ghw = hot_of<WT, PRO_ACTQ8>(gu);
        const size_t gu_lds = gemvm_lds_bytes(WT, S, E, true);
        if (gu_lds > 64 * 1024 && l == 0)
            GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_gemvm<WT, NE, 8, S, EPI_SILUMUL, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gu_lds));
        DEC_LAUNCH_HOT(KT_DEC_GEMV_GATEUP, (k_dec_gemvm<WT, NE, 8, S, EPI_SILUMUL, 512>), dim3(F / 32), dim3(512), gu_lds, ghw, gu);
        Gemv8Args dn = base;
        dn.d_in = F; dn.n_mats = 1; set_mat(dn, 0, L.wdown, WT, E, F); dn.out = dc->down_raw; dn.raw_stride = E;
        dn.act_q = dc->act_q; dn.act_d = dc->act_d; dn.act_sum = dc->act_sum; dn.act_f = dc->act_f;
        rc = wideF ? launch_gemvm<WT, NF, 2, S, 256>(KT_DEC_GEMV_DOWN, dn, E) : launch_gemvm<WT, NE, 2, S, 256>(KT_DEC_GEMV_DOWN, dn, E);
        if (rc) return rc;
    }
    Gemv8Args sf = base;
    sf.d_in = E; sf.res_a = hbuf; sf.res_raw = dc->down_raw; sf.raw_stride = E; sf.norm_w = (const uint16_t*)d.final_norm;
    sf.act_q = dc->stg_q; sf.act_d = dc->stg_d; sf.act_sum = dc->stg_sum; sf.act_f = dc->stg_f;
    if ((rc = launch_stage<WT, PRO_RESID>(KT_DEC_STAGE, sf, S))) return rc;
    Gemv8Args hd = base;
    hd.d_in = E; hd.n_mats = 1; set_mat(hd, 0, d.lm_head, WT, V, E); hd.out = dc->logits_m; hd.raw_stride = V;
    hd.act_q = dc->stg_q; hd.act_d = dc->stg_d; hd.act_sum = dc->stg_sum; hd.act_f = dc->stg_f;
    hd.best_val = dc->best_val; hd.best_idx = dc->best_idx;
    if ((rc = launch_gemvm<WT, NE, RH, S, 512>(KT_DEC_GEMV_HEAD, hd, V))) return rc;
    DEC_LAUNCH(KT_DEC_ARGMAX, k_dec_argmax, dim3(S), dim3(1024), 0, (const float*)dc->best_val, (const int*)dc->best_idx,
               dc->n_best, dc->step, dc->result, dc->n_best, d.max_ctx + 2, dc->tokens, d.max_ctx + 1);
    return 0;
}

static size_t mmv_lds_bytes(int wt, int rt, int ft, int d_in, int ks)
{
    const size_t nb = (size_t)d_in / 32 / (size_t)ks, sp = 16 * (size_t)rt, fr = 16 * (size_t)ft;
    return std::max(fr * nb * (wt == GTEN_Q4 ? 16 : 32), 8 * sp * 64) + fr * nb * 2 + nb * sp * 4;
}

// feature tiles per workgroup: wide outputs at short K take 4 (Q4, <= 32 rows) / 2 -- see k_dec_mmv
template <int WT, int RT>
static int mmv_feature_tiles(const MmvArgs& a)
{
    constexpr int FTW = (WT == GTEN_Q4 && RT <= 2) ? 4 : 2;
    const int cols = a.d_out[0] + (a.n_mats > 1 ? a.d_out[1] : 0) + (a.n_mats > 2 ? a.d_out[2] : 0);
    const int ppr = (a.d_in / 32) * (WT == GTEN_Q4 ? 16 : 32) / 16;
    bool ok = cols >= 4096 && ppr <= 32 * (MMV_MAXP / FTW);
    for (int k = 0; k + 1 < a.n_mats; k++) ok = ok && a.d_out[k] % (16 * FTW) == 0;
    return ok ? FTW : 1;
}

template <int WT, int RT>
static int launch_mmv_rt(int tag, const MmvArgs& a)
{
    constexpr int FTW = (WT == GTEN_Q4 && RT <= 2) ? 4 : 2;
    const int ft = mmv_feature_tiles<WT, RT>(a);
    const int ks = a.ks > 1 ? a.ks : 1;
    GTR_REQUIRE((a.d_in / 32) % (8 * ks) == 0, "decoder: %d K slices do not divide the %d quant blocks into eight waves", ks, a.d_in / 32);
    const size_t smem = mmv_lds_bytes(WT, RT, ft, a.d_in, ks);
    GTR_REQUIRE(smem <= 150 * 1024, "decoder: the slab and delta table of d_in %d x %d rows do not fit LDS", a.d_in, a.S);
    const int cols = a.d_out[0] + (a.n_mats > 1 ? a.d_out[1] : 0) + (a.n_mats > 2 ? a.d_out[2] : 0);
    const int nbw = a.d_in / 256 / ks;                          // quant blocks per wave
    const dim3 grid((cols + 16 * ft - 1) / (16 * ft), ks);
    // activation chunks: all of a wave's blocks at once when that is 8 or fewer, else elevens (5632 / 256 = 22);
    // more than 32 rows or several feature tiles: fours (registers)
    const MmvRest rest{a.w[1], a.w[2], a.d_out[1], a.d_out[2], a.plane};
#define MMV_ARGS a.aq, a.ad, a.w[0], a.out, a.d_in, a.d_out[0], a.out_cols, a.S, a.n_mats, rest
    if (ft > 1)
        DEC_LAUNCH(tag, (k_dec_mmv<WT, RT, 4, FTW>), grid, dim3(512), smem, MMV_ARGS);
    else if (RT > 2)
        DEC_LAUNCH(tag, (k_dec_mmv<WT, RT, 4, 1>), grid, dim3(512), smem, MMV_ARGS);
    else if (nbw % 11 == 0)
        DEC_LAUNCH(tag, (k_dec_mmv<WT, (RT > 2 ? 1 : RT), 11, 1>), grid, dim3(512), smem, MMV_ARGS);
    else if (nbw <= 4)
        DEC_LAUNCH(tag, (k_dec_mmv<WT, (RT > 2 ? 1 : RT), 4, 1>), grid, dim3(512), smem, MMV_ARGS);
    else
        DEC_LAUNCH(tag, (k_dec_mmv<WT, (RT > 2 ? 1 : RT), 8, 1>), grid, dim3(512), smem, MMV_ARGS);
#undef MMV_ARGS
    return 0;
}

// k_dec_mmvh (f16 fragments): four feature tiles per workgroup for the wide outputs (Q4; two for Q8), else one
struct MmvhArgs {
    const uint16_t* ah;
    const void* w[3]; int d_out[3]; int n_mats;
    float* out; int out_cols;
    int S, d_in, ks, plane;
};
template <int WT, int RT>
static int launch_mmvh_rt(int tag, const MmvhArgs& a)
{
    constexpr int FTW = (WT == GTEN_Q4 && RT <= 2) ? 4 : 2;
    const int cols = a.d_out[0] + (a.n_mats > 1 ? a.d_out[1] : 0) + (a.n_mats > 2 ? a.d_out[2] : 0);
    const int ppr = (a.d_in / 32) * (WT == GTEN_Q4 ? 16 : 32) / 16;
    // (eight row tiles, two feature tiles for the 2048 / 2560-wide projections too -- half the activation re-reads, half the
    //  workgroups: 50.0 k against 51.8 k tok/s at 128 sequences, 67.3 k = 67.5 k at 256: not kept)
    bool wide_out = cols >= 4096 && ppr <= 32 * (MMV_MAXP / FTW);
    for (int k = 0; k + 1 < a.n_mats; k++) wide_out = wide_out && a.d_out[k] % (16 * FTW) == 0;
    const int ft = wide_out ? FTW : 1;
    const int ks = a.ks > 1 ? a.ks : 1;
    GTR_REQUIRE((a.d_in / 32) % (8 * ks) == 0, "decoder: %d K slices do not divide the %d quant blocks into eight waves", ks, a.d_in / 32);
    const size_t nbs = (size_t)a.d_in / 32 / ks, fr = 16 * (size_t)ft;
    const size_t smem = std::max(fr * nbs * (WT == GTEN_Q4 ? 16 : 32), (size_t)8 * 16 * RT * 64) + fr * nbs * 2;
    GTR_REQUIRE(smem <= 150 * 1024, "decoder: the weight slab of d_in %d does not fit LDS", a.d_in);
    const dim3 grid((cols + 16 * ft - 1) / (16 * ft), ks);
    const MmvRest rest{a.w[1], a.w[2], a.d_out[1], a.d_out[2], a.plane};
    if (ft > 1)
        DEC_LAUNCH(tag, (k_dec_mmvh<WT, RT, FTW, false>), grid, dim3(512), smem, a.ah, a.w[0], a.out, a.d_in, a.d_out[0], a.out_cols, a.S, a.n_mats, rest);
    else
        DEC_LAUNCH(tag, (k_dec_mmvh<WT, RT, 1, false>), grid, dim3(512), smem, a.ah, a.w[0], a.out, a.d_in, a.d_out[0], a.out_cols, a.S, a.n_mats, rest);
    return 0;
}
// gate | up with the silu * mul chain in the epilogue: one workgroup per 32-wide FFN slice, fragments for the down projection
template <int WT>
static int launch_mmvh_silu(int tag, const uint16_t* ah, const void* wgate, const void* wup, int n_ffn, int d_in, int S, uint16_t* out_frag)
{
    constexpr int WQ = (WT == GTEN_F16) ? GTEN_Q8 : WT;
    GTR_REQUIRE(n_ffn % 32 == 0 && d_in % 256 == 0, "decoder: FFN %d x %d does not tile", n_ffn, d_in);
    // (row tiles of the KERNEL instance: five to eight tiles run the eight-tile instance, whose LDS layout and fragment stride
    //  are those of eight tiles -- and the staging launches must have written eight, so only full 128-row lanes take it)
    const size_t nbs = (size_t)d_in / 32, rt_s = (S + 15) / 16, rt = rt_s <= 4 ? rt_s : 8;
    GTR_REQUIRE(rt_s <= 4 || rt_s == 8, "decoder: a lane of %d rows (one to four row tiles, or exactly eight)", S);
    const size_t smem = std::max((size_t)64 * nbs * (WQ == GTEN_Q4 ? 16 : 32), (size_t)8 * 16 * rt * 64) + 64 * nbs * 2 + 4 * 16 * rt * 16 * 4;
    GTR_REQUIRE(smem <= 150 * 1024 && nbs * (WQ == GTEN_Q4 ? 1 : 2) <= (size_t)32 * (MMV_MAXP / 4), "decoder: the FFN slab of d_in %d does not fit", d_in);
    const MmvRest rest{wup, nullptr, n_ffn, 0, 0};
    const dim3 grid(n_ffn / 32, 1);
#define MMVH_S(RT_) DEC_LAUNCH(tag, (k_dec_mmvh<WQ, RT_, 4, true>), grid, dim3(512), smem, ah, wgate, (float*)out_frag, d_in, n_ffn, 0, S, 2, rest)
    switch ((int)rt) {
    case 1: MMVH_S(1); break;
    case 2: MMVH_S(2); break;
    case 3: MMVH_S(3); break;
    case 4: MMVH_S(4); break;
    default: MMVH_S(8); break;
    }
#undef MMVH_S
    return 0;
}

template <int WT>
static int launch_mmvh(int tag, const MmvhArgs& a)
{
    GTR_REQUIRE(a.d_in % 256 == 0 && a.S >= 1 && a.S <= 128, "decoder: skinny W.x wants d_in %% 256 == 0 and <= 128 rows");
    GTR_REQUIRE((a.S + 15) / 16 <= 4 || (a.S + 15) / 16 == 8, "decoder: a lane of %d rows (one to four row tiles, or exactly eight: the eight-tile "
                "instance reads the staging with a stride of eight tiles)", a.S);
    GTR_REQUIRE((size_t)16 * (a.d_in / 32) * (WT == GTEN_Q4 ? 16 : 32) <= (size_t)MMV_MAXP * 512 * 16, "decoder: d_in %d too long for the weight slab", a.d_in);
    for (int k = 0; k + 1 < a.n_mats; k++) GTR_REQUIRE(a.d_out[k] % 16 == 0, "decoder: concatenated matrices must be multiples of 16 wide");
    constexpr int WQ = (WT == GTEN_F16) ? GTEN_Q8 : WT;
    switch ((a.S + 15) / 16) {
    case 1: return launch_mmvh_rt<WQ, 1>(tag, a);
    case 2: return launch_mmvh_rt<WQ, 2>(tag, a);
    case 3: return launch_mmvh_rt<WQ, 3>(tag, a);
    case 4: return launch_mmvh_rt<WQ, 4>(tag, a);
    default: return launch_mmvh_rt<WQ, 8>(tag, a);       // 65 .. 128 rows: eight row tiles (rows past S are never stored)
    }
}
template <int WT>
static int mmvh_prepare()
{
    constexpr int FTA = (WT == GTEN_Q4) ? 4 : 2;        // <= 32 rows
#define MMVH_ATTR(RT_, FT_) GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_mmvh<WT, RT_, FT_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
    MMVH_ATTR(1, 1); MMVH_ATTR(2, 1); MMVH_ATTR(3, 1); MMVH_ATTR(4, 1);
    MMVH_ATTR(1, FTA); MMVH_ATTR(2, FTA); MMVH_ATTR(3, 2); MMVH_ATTR(4, 2);
    MMVH_ATTR(8, 1); MMVH_ATTR(8, 2);
#define MMVH_ATTR_S(RT_) GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_mmvh<WT, RT_, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
    MMVH_ATTR_S(1); MMVH_ATTR_S(2); MMVH_ATTR_S(3); MMVH_ATTR_S(4); MMVH_ATTR_S(8);
#undef MMVH_ATTR_S
#undef MMVH_ATTR
    return 0;
}
// the wide path's W.x form: f16 fragments with folded deltas (k_dec_mmvh) unless the exact forms are selected
static bool mmv_folded()
{
    return !g_exact_now;
}

// (before the first launch, outside any stream capture: the slab may need more than 64 KB of LDS)
template <int WT>
static int mmv_prepare()
{
    constexpr int FTA = (WT == GTEN_Q4) ? 4 : 2;        // <= 32 rows
#define MMV_ATTR(RT_, CB_, FT_) GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_mmv<WT, RT_, CB_, FT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
    MMV_ATTR(1, 8, 1); MMV_ATTR(2, 8, 1); MMV_ATTR(1, 11, 1); MMV_ATTR(2, 11, 1); MMV_ATTR(3, 4, 1); MMV_ATTR(4, 4, 1);
    MMV_ATTR(1, 4, 1); MMV_ATTR(2, 4, 1);
    MMV_ATTR(1, 4, FTA); MMV_ATTR(2, 4, FTA); MMV_ATTR(3, 4, 2); MMV_ATTR(4, 4, 2);
#undef MMV_ATTR
    return 0;
}

template <int RT>
static int launch_mmv_f16_rt(int tag, const MmvArgs& a)
{
    const int ks = a.ks > 1 ? a.ks : 1;
    GTR_REQUIRE(a.d_in % (256 * ks) == 0, "decoder: %d K slices x eight waves do not divide d_in %d into 32-element steps", ks, a.d_in);
    const int cols = a.d_out[0] + (a.n_mats > 1 ? a.d_out[1] : 0) + (a.n_mats > 2 ? a.d_out[2] : 0);
    const MmvRest rest{a.w[1], a.w[2], a.d_out[1], a.d_out[2], a.plane};
    DEC_LAUNCH(tag, (k_dec_mmv_f16<RT>), dim3((cols + 15) / 16, ks), dim3(512), (size_t)8 * 16 * RT * 16 * 4,
               (const uint16_t*)a.aq, a.w[0], a.out, a.d_in, a.d_out[0], a.out_cols, a.S, a.n_mats, rest);
    return 0;
}

template <int WT>
static int launch_mmv(int tag, const MmvArgs& a)
{
    GTR_REQUIRE(a.d_in % 256 == 0 && a.S >= 1 && a.S <= 64, "decoder: skinny W.x wants d_in %% 256 == 0 and <= 64 rows");
    if (WT == GTEN_F16) {
        for (int k = 0; k + 1 < a.n_mats; k++) GTR_REQUIRE(a.d_out[k] % 16 == 0, "decoder: concatenated matrices must be multiples of 16 wide");
        switch ((a.S + 15) / 16) {
        case 1: return launch_mmv_f16_rt<1>(tag, a);
        case 2: return launch_mmv_f16_rt<2>(tag, a);
        case 3: return launch_mmv_f16_rt<3>(tag, a);
        default: return launch_mmv_f16_rt<4>(tag, a);
        }
    }
    GTR_REQUIRE((size_t)16 * (a.d_in / 32) * (WT == GTEN_Q4 ? 16 : 32) <= (size_t)MMV_MAXP * 512 * 16 && (size_t)(a.d_in / 32) * 64 * 4 <= (size_t)MMV_MAXD * 512 * 16,
                "decoder: d_in %d too long for the weight slab / delta table", a.d_in);
    for (int k = 0; k + 1 < a.n_mats; k++) GTR_REQUIRE(a.d_out[k] % 16 == 0, "decoder: concatenated matrices must be multiples of 16 wide");
    constexpr int WQ = (WT == GTEN_F16) ? GTEN_Q8 : WT;        // (never reached for f16: keeps k_dec_mmv<f16> from being instantiated)
    switch ((a.S + 15) / 16) {
    case 1: return launch_mmv_rt<WQ, 1>(tag, a);
    case 2: return launch_mmv_rt<WQ, 2>(tag, a);
    case 3: return launch_mmv_rt<WQ, 3>(tag, a);
    default: return launch_mmv_rt<WQ, 4>(tag, a);
    }
}

// ---- n_seq >= 16: every W.x of the step runs on the matrix cores (k_dec_mmv, rows = sequences) fed from the
// staging launches in fragment order; attention stays one grid plane per sequence.  The linears then add their
// block sums in k_dec_mmv's order (eight K slices) instead of the GEMV wave tree, so a sequence's logits are not
// bit-for-bit those of the single-sequence decoder (tests: model band, graph == eager, run-to-run identical).
template <int WT>
static int enqueue_step_wide(gten_hip_decoder* dc, int lane)
{
    const LaneBufs b = lane_bufs(dc, lane);
    const gten_hip_decoder_desc& d = dc->d;
    const int S = b.n_seq;
    const int E = d.n_embd, F = d.n_ffn, dh = E / d.n_heads, KV = dh * d.n_kv_heads, V = d.n_vocab;
    const size_t kv_pitch = gten_hip_row_bytes(d.adtype, KV);
    float* xbuf = (float*)b.xbuf;
    float* hbuf = (float*)b.hbuf;
    int rc;
    bool grouped = false, grouped_known = false;
    Gemv8Args base{};
    base.step = b.step; base.tok_stride = d.max_ctx + 1; base.part_stride = d.n_heads * dc->n_chunks * dh;
    base.best_stride = dc->n_best;
    const bool folded = WT != GTEN_F16 && mmv_folded();
    base.frag_h16 = folded ? 1 : 0;
    auto mmh = [&](int tag, const int8_t* aq, float* out, int out_cols, int d_in, int ks, const void* w, int d_out,
                   const void* w1, int d1, const void* w2, int d2) -> int {
        MmvhArgs a{};
        a.ah = (const uint16_t*)aq; a.w[0] = w; a.w[1] = w1; a.w[2] = w2; a.d_out[0] = d_out; a.d_out[1] = d1; a.d_out[2] = d2;
        a.n_mats = w2 ? 3 : (w1 ? 2 : 1); a.out = out; a.out_cols = out_cols; a.S = S; a.d_in = d_in;
        a.ks = ks; a.plane = S * out_cols;
        return launch_mmvh<WT>(tag, a);
    };
    auto mm = [&](int tag, const int8_t* aq, const float* ad, const int*, float* out, int out_cols, int d_in, const void* w, int d_out,
                  const void* w1 = nullptr, int d1 = 0, const void* w2 = nullptr, int d2 = 0) -> int {
        if (folded) return mmh(tag, aq, out, out_cols, d_in, 1, w, d_out, w1, d1, w2, d2);
        MmvArgs a{};
        a.aq = aq; a.ad = ad; a.w[0] = w; a.w[1] = w1; a.w[2] = w2; a.d_out[0] = d_out; a.d_out[1] = d1; a.d_out[2] = d2;
        a.n_mats = w2 ? 3 : (w1 ? 2 : 1); a.out = out; a.out_cols = out_cols; a.S = S; a.d_in = d_in;
        return launch_mmv<WT>(tag, a);
    };
    // K split of the launches with few feature tiles (q|k|v, o, down): two workgroups per tile, two output planes that
    // the consumers add (staging launches: raw_plane; grouped score kernel: qkv_plane)
    const int ksplit = 2;
    auto mmk = [&](int tag, const int8_t* aq, const float* ad, float* out, int out_cols, int d_in, int ks, const void* w, int d_out,
                   const void* w1 = nullptr, int d1 = 0, const void* w2 = nullptr, int d2 = 0) -> int {
        if (folded) return mmh(tag, aq, out, out_cols, d_in, ks, w, d_out, w1, d1, w2, d2);
        MmvArgs a{};
        a.aq = aq; a.ad = ad; a.w[0] = w; a.w[1] = w1; a.w[2] = w2; a.d_out[0] = d_out; a.d_out[1] = d1; a.d_out[2] = d2;
        a.n_mats = w2 ? 3 : (w1 ? 2 : 1); a.out = out; a.out_cols = out_cols; a.S = S; a.d_in = d_in;
        a.ks = ks; a.plane = S * out_cols;
        return launch_mmv<WT>(tag, a);
    };
    auto ks_of = [&](int d_in) { return (ksplit > 1 && (d_in / 32) % (8 * ksplit) == 0) ? ksplit : 1; };
    for (int l = 0; l < d.n_layers; l++) {
        const gten_hip_layer_ptrs& L = dc->layers[l];
        Gemv8Args st = base;
        st.d_in = E; st.norm_w = (const uint16_t*)L.attn_norm; st.x_out = xbuf;
        st.act_q = b.stg_q; st.act_d = b.stg_d; st.act_sum = b.stg_sum; st.act_f = b.stg_f;
        if (l == 0) {
            st.table = d.embed; st.rope = dc->rope; st.rope_now = b.rope_now; st.rope_half = dh / 2; st.n_vocab = V; st.tokens = b.tokens;
            rc = launch_stage_frag<WT, PRO_EMBED>(KT_DEC_STAGE, st, S);
        } else {
            st.res_a = hbuf; st.res_raw = b.down_raw; st.raw_stride = E; st.raw_plane = ks_of(F) > 1 ? S * E : 0;
            rc = launch_stage_frag<WT, PRO_RESID>(KT_DEC_STAGE, st, S);
        }
        if (rc) return rc;
        const int QW = E + 2 * KV;
        // (measured: 160 x 2 workgroups of 4 row tiles run slower than 160; of 8 row tiles too: 11.8 against 9.0 us per launch)
        const int ks_qkv = S <= 32 ? ks_of(E) : 1;
        if ((rc = mmk(KT_DEC_GEMV_QKV, b.stg_q, b.stg_d, b.qkv_raw, QW, E, ks_qkv, L.wq, E, L.wk, KV, L.wv, KV))) return rc;
        AttnArgs t{};
        t.step = b.step; t.qkv_raw = b.qkv_raw; t.kv_pitch = kv_pitch; t.scores = b.scores; t.stats = b.stats;
        t.att_part = b.att_part; t.rope = dc->rope; t.rope_now = b.rope_now;
        t.adtype = d.adtype; t.n_heads = d.n_heads; t.n_kv = d.n_kv_heads; t.d_head = dh; t.max_ctx = d.max_ctx;
        t.n_chunks = dc->n_chunks; t.n_embd = E;
        t.kv_tab = (const void* const*)b.kv_tab; t.layer = l; t.n_layers = d.n_layers;
        t.qkv_stride = QW; t.scores_stride = d.n_heads * d.max_ctx; t.stats_stride = d.n_heads * dc->n_chunks * 2;
        t.part_stride = d.n_heads * dc->n_chunks * dh;
        t.qkv_plane = ks_qkv > 1 ? S * QW : 0;
        const dim3 agrid(d.n_heads, dc->n_chunks, S);
        const size_t smem1 = (size_t)(16 + 3 * dh + 16) * 4 + 32 + (size_t)3 * dh + 64;
        if (!grouped_known) { grouped = attention_grouped_ok(t, S); grouped_known = true; }
        if ((rc = grouped ? launch_attention_grouped(t, S) : launch_attention(t, agrid, smem1))) return rc;
        Gemv8Args sa = base;
        sa.d_in = E; sa.att_part = b.att_part; sa.d_head = dh; sa.d_head_shift = __builtin_ctz(dh); sa.n_chunks = dc->n_chunks;
        sa.att_stats = b.stats; sa.stats_stride = d.n_heads * dc->n_chunks * 2;
        sa.act_q = b.stg_q; sa.act_d = b.stg_d; sa.act_sum = b.stg_sum; sa.act_f = b.stg_f;
        rc = (grouped ? grouped_one_pass(t, S) : attention_one_pass(dh)) ? launch_stage_frag<WT, PRO_ATTW>(KT_DEC_STAGE, sa, S)
                                                                         : launch_stage_frag<WT, PRO_ATT>(KT_DEC_STAGE, sa, S);
        if (rc) return rc;
        if ((rc = mmk(KT_DEC_GEMV_O, b.stg_q, b.stg_d, b.proj_raw, E, E, ks_of(E), L.wo, E))) return rc;
        Gemv8Args sh = base;
        sh.d_in = E; sh.res_a = xbuf; sh.res_raw = b.proj_raw; sh.raw_stride = E; sh.x_out = hbuf;
        sh.raw_plane = ks_of(E) > 1 ? S * E : 0;
        sh.norm_w = (const uint16_t*)L.ffn_norm;
        sh.act_q = b.stg_q; sh.act_d = b.stg_d; sh.act_sum = b.stg_sum; sh.act_f = b.stg_f;
        if ((rc = launch_stage_frag<WT, PRO_RESID>(KT_DEC_STAGE, sh, S))) return rc;
        // gate|up has 176-352 workgroups already; the split measured +4 % at 16 sequences, +2 % at 64, -2 % at 32
        const int ks_gu = ((S + 15) / 16 != 2) ? ks_of(E) : 1;
        const bool fuse_ffn = folded && WT == GTEN_Q4 && E / 32 <= 32 * (MMV_MAXP / 4);      // (a tile's slab: <= 2 pieces per thread)
        if (fuse_ffn) {
            if ((rc = launch_mmvh_silu<WT>(KT_DEC_GEMV_GATEUP, (const uint16_t*)b.stg_q, L.wgate, L.wup, F, E, S, (uint16_t*)b.act_q))) return rc;
        } else {
        if ((rc = mmk(KT_DEC_GEMV_GATEUP, b.stg_q, b.stg_d, b.gu_raw, 2 * F, E, ks_gu, L.wgate, F, L.wup, F))) return rc;
        if (WT == GTEN_F16)
            DEC_LAUNCH(KT_DEC_GEMV_GATEUP, k_dec_silumul_rows_f16, dim3(F / 256, S), dim3(256), 0, (const float*)b.gu_raw, F,
                       ks_gu > 1 ? S * 2 * F : 0, (uint16_t*)b.act_q);
        else
            DEC_LAUNCH(KT_DEC_GEMV_GATEUP, k_dec_silumul_rows, dim3(F / 256, S), dim3(256), 0, (const float*)b.gu_raw, F, (S + 15) / 16,
                       ks_gu > 1 ? S * 2 * F : 0, b.act_q, b.act_d, b.act_sum, folded ? 1 : 0);
        }
        if ((rc = mmk(KT_DEC_GEMV_DOWN, b.act_q, b.act_d, b.down_raw, E, F, ks_of(F), L.wdown, E))) return rc;
    }
    Gemv8Args sf = base;
    sf.d_in = E; sf.res_a = hbuf; sf.res_raw = b.down_raw; sf.raw_stride = E; sf.norm_w = (const uint16_t*)d.final_norm;
    sf.raw_plane = ks_of(F) > 1 ? S * E : 0;
    sf.act_q = b.stg_q; sf.act_d = b.stg_d; sf.act_sum = b.stg_sum; sf.act_f = b.stg_f;
    if ((rc = launch_stage_frag<WT, PRO_RESID>(KT_DEC_STAGE, sf, S))) return rc;
    if ((rc = mm(KT_DEC_GEMV_HEAD, b.stg_q, b.stg_d, b.stg_sum, b.logits_m, V, E, d.lm_head, V))) return rc;
    DEC_LAUNCH(KT_DEC_ARGMAX, k_dec_argmax, dim3(S), dim3(1024), 0, (const float*)b.logits_m, (const int*)nullptr,
               V, b.step, b.result, V, d.max_ctx + 2, b.tokens, d.max_ctx + 1);
    return 0;
}

template <int WT>
static int enqueue_multi(gten_hip_decoder* dc, int lane)
{
    switch (dc->n_seq) {
    case 2: return enqueue_step_multi<WT, 2>(dc);
    case 4: return enqueue_step_multi<WT, 4>(dc);
    case 8: return enqueue_step_multi<WT, 8>(dc);
    }
    if (dc->n_seq >= 16) return enqueue_step_wide<WT>(dc, lane);
    return fail(-4, "decoder: n_seq %d not supported for this configuration", dc->n_seq);
}

// one step of one lane (a decoder of up to 64 sequences has the single lane 0)
#include "gten_decode_persist.h"

static int persist_prepare(gten_hip_decoder* dc);

static int enqueue_lane(gten_hip_decoder* dc, int lane)
{
    g_exact_now = dc->exact;
    // single sequence, Q8 activations: the whole step as one persistent launch (a family-restricted timing replay wants
    // the launch chain's kernels)
    if (dc->persist_on && dc->n_seq == 1 && !dc->exact && (g_only_family < 0 || g_only_family == KT_DEC_PERSIST)) {
        if (int rc = persist_prepare(dc)) return rc;
        if (dc->persist) return persist_launch<GTEN_Q4>(dc->persist);
    }
    if (dc->n_seq > 1) {
        switch (dc->d.wdtype) {
        case GTEN_F16: return enqueue_multi<GTEN_F16>(dc, lane);
        case GTEN_Q8: return enqueue_multi<GTEN_Q8>(dc, lane);
        case GTEN_Q4: return enqueue_multi<GTEN_Q4>(dc, lane);
        }
    }
    switch (dc->d.wdtype) {
    case GTEN_F16: return enqueue_step_q8act<GTEN_F16>(dc);
    case GTEN_Q8: return enqueue_step_q8act<GTEN_Q8>(dc);
    case GTEN_Q4: return enqueue_step_q8act<GTEN_Q4>(dc);
    }
    return fail(-4, "decoder: bad weight dtype %d", dc->d.wdtype);
}

// `count` consecutive steps.  Several lanes: lane 0's chain on the current stream, the others on the decoder's side
// streams, forked behind whatever the current stream holds and joined at the end -- under stream capture these become
// parallel branches of the graph, eagerly they are real events.  The lanes never touch each other's rows, and a lane's
// steps follow each other in its own stream: nothing else needs ordering.
static int enqueue(gten_hip_decoder* dc, int count = 1)
{
    if (dc->lanes <= 1) {
        for (int i = 0; i < count; i++)
            if (int rc = enqueue_lane(dc, 0)) return rc;
        return 0;
    }
    const unsigned all = (1u << dc->lanes) - 1u, mask = (dc->lane_mask & all) ? (dc->lane_mask & all) : all;
    int g0 = 0;
    while (!((mask >> g0) & 1u)) g0++;                       // the first lane taken runs on the current stream
    const bool side = (mask & ~(1u << g0)) != 0;
    hipStream_t main_s = stream();
    if (side) {
        GTR_CHECK(hipEventRecord(dc->lane_fork, main_s));
        for (int g = g0 + 1; g < dc->lanes; g++)
            if ((mask >> g) & 1u) GTR_CHECK(hipStreamWaitEvent(dc->lane_stream[g], dc->lane_fork, 0));
    }
    int rc = 0;
    for (int g = g0; g < dc->lanes && !rc; g++) {
        if (!((mask >> g) & 1u)) continue;
        if (g != g0) gtr::stream_override(dc->lane_stream[g]);
        for (int i = 0; i < count && !rc; i++) rc = enqueue_lane(dc, g);
        gtr::stream_override(nullptr);
    }
    // (join even after a failure: a capture must not be left with dangling branches)
    for (int g = g0 + 1; g < dc->lanes; g++) {
        if (!((mask >> g) & 1u)) continue;
        GTR_CHECK(hipEventRecord(dc->lane_join[g], dc->lane_stream[g]));
        GTR_CHECK(hipStreamWaitEvent(main_s, dc->lane_join[g], 0));
    }
    return rc;
}

static int decoder_build(gten_hip_decoder* dc, const gten_hip_decoder_desc& d, const gten_hip_layer_ptrs* layers,
                         const gten_hip_kv_ptrs* kv, int n_seq);

static int decoder_create_common(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers,
                                 const gten_hip_kv_ptrs* kv, int n_seq, gten_hip_decoder** out)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(desc && layers && out, "decoder_create: null argument");
    const gten_hip_decoder_desc& d = *desc;
    GTR_REQUIRE(d.n_layers > 0 && d.n_heads > 0 && d.n_kv_heads > 0 && d.n_heads % d.n_kv_heads == 0, "decoder_create: bad head/layer counts");
    GTR_REQUIRE(d.n_embd % d.n_heads == 0, "decoder_create: n_embd %% n_heads != 0");
    const int dh = d.n_embd / d.n_heads;
    GTR_REQUIRE(dh == 32 || dh == 64, "decoder_create: fast path supports d_head 32 or 64 (got %d)", dh);
    GTR_REQUIRE(d.n_embd % 32 == 0 && d.n_ffn % 32 == 0 && d.n_embd <= 2048 && d.n_ffn <= 6144, "decoder_create: unsupported widths (n_embd <= 2048, n_ffn <= 6144)");
    GTR_REQUIRE(d.n_vocab > 0 && d.n_vocab <= 65535, "decoder_create: n_vocab %d outside [1, 65535] (row counts travel as 16-bit fields of the preloaded arguments)", d.n_vocab);
    GTR_REQUIRE(d.max_ctx > 0 && d.max_ctx <= GTEN_ROPE_MAX_POS, "decoder_create: max_ctx %d beyond the RoPE table", d.max_ctx);
    const bool pair_ok = (d.wdtype == GTEN_F16 && d.adtype == GTEN_F16) || ((d.wdtype == GTEN_Q8 || d.wdtype == GTEN_Q4) && d.adtype == GTEN_Q8);
    GTR_REQUIRE(pair_ok, "decoder_create: unsupported dtype pair (%d,%d) (tinyllama.cpp:258-265)", d.wdtype, d.adtype);
    GTR_REQUIRE(d.embed && d.final_norm && d.lm_head, "decoder_create: null model pointer");
    const bool wide = n_seq >= 16;
    GTR_REQUIRE(n_seq == 1 || n_seq == 2 || n_seq == 4 || n_seq == 8 || (wide && n_seq <= 64 && n_seq % 16 == 0) ||
                (n_seq > 64 && n_seq <= 64 * DEC_MAX_LANES && n_seq % 64 == 0) || (n_seq > 256 && n_seq <= 128 * DEC_MAX_LANES && n_seq % 128 == 0),
                "decoder_create: n_seq %d not in {1, 2, 4, 8, 16, 32, 48, 64, 128, 192, 256, 384, 512}", n_seq);
    GTR_REQUIRE(!wide || (d.n_ffn % 256 == 0 && d.n_embd % 256 == 0 && (dh * d.n_kv_heads) % 16 == 0),
                "decoder_create: n_seq >= 16 runs the W.x on the matrix cores: n_embd and n_ffn %% 256 == 0");
    GTR_REQUIRE(n_seq == 1 || (kv && dh == 64), "decoder_create: multi-sequence decode needs the cache table and d_head 64");
    GTR_REQUIRE(n_seq > 1 || d.logits, "decoder_create: null logits pointer");
    if (n_seq > 1)
        for (size_t i = 0; i < (size_t)n_seq * d.n_layers; i++)
            GTR_REQUIRE(kv[i].kcache && kv[i].vcache, "decoder_create: null cache pointer (sequence %zu, layer %zu)", i / d.n_layers, i % d.n_layers);
    // every failure past this point goes through gten_hip_decoder_destroy: nothing allocated so far is leaked
    auto* dc = new gten_hip_decoder;
    if (const int rc = decoder_build(dc, d, layers, kv, n_seq)) {
        char msg[512];
        snprintf(msg, sizeof(msg), "%s", gten_hip_last_error());
        gten_hip_decoder_destroy(dc);
        return fail(rc, "%s", msg);
    }
    *out = dc;
    return 0;
}

static int decoder_build(gten_hip_decoder* dc, const gten_hip_decoder_desc& d, const gten_hip_layer_ptrs* layers,
                         const gten_hip_kv_ptrs* kv, int n_seq)
{
    const int dh = d.n_embd / d.n_heads;
    const bool wide = n_seq >= 16;
    dc->d = d;
    dc->n_seq = n_seq;
    dc->exact = g_decode_exact;
    dc->persist_on = g_persist_on;
    {
        // Rows per lane: 128 where the folded W.x form runs eight row tiles per workgroup (k_dec_mmvh<.., 8, ..>: every expanded
        // weight fragment feeds eight matrix instructions and the weights are read once per 128 sequences), else 64 (f16 weights,
        // the exact forms, 192 sequences).  Per sequence the same bits either way (tests/test_multiseq_gpu.py).  Measured (q4,
        // ctx -> 2048, tok/s): 128 sequences 50.8 k as two lanes of 64, 51.7 k as one of 128; 256 sequences 56.9 k as four
        // lanes of 64, 67.3 k as two of 128; serving 1024 prompts through 128 slots 29.9 k -> 31.5 k new ids/s.
        const bool can128 = !g_decode_exact && d.wdtype != GTEN_F16 && d.adtype == GTEN_Q8 && n_seq % 128 == 0;
        const int lane_rows = can128 ? 128 : 64;
        const int lanes = (n_seq + lane_rows - 1) / lane_rows;
        GTR_REQUIRE(lanes <= DEC_MAX_LANES, "decoder_create: %d sequences need %d lanes of %d (at most %d: f16 weights and the exact forms run lanes of 64)",
                    n_seq, lanes, lane_rows, DEC_MAX_LANES);
        dc->lanes = lanes;
    }
    for (int g = 1; g < dc->lanes; g++) {
        GTR_CHECK(hipStreamCreateWithFlags(&dc->lane_stream[g], hipStreamNonBlocking));
        GTR_CHECK(hipEventCreateWithFlags(&dc->lane_join[g], hipEventDisableTiming));
    }
    if (dc->lanes > 1) GTR_CHECK(hipEventCreateWithFlags(&dc->lane_fork, hipEventDisableTiming));
    dc->layers.assign(layers, layers + d.n_layers);
    dc->n_chunks = (d.max_ctx + DEC_CHUNK - 1) / DEC_CHUNK;
    static_assert(GTEN_ROPE_MAX_POS <= DEC_ATT_MAXCH * DEC_CHUNK, "max_ctx (checked above) bounds the attention chunks the consumers request up front");
    const int E = d.n_embd, F = d.n_ffn, KV = dh * d.n_kv_heads;
    const size_t S = (size_t)n_seq;
    GTR_CHECK(hipMalloc((void**)&dc->step, S * sizeof(DecStep)));
    GTR_CHECK(hipMemset(dc->step, 0, S * sizeof(DecStep)));
    GTR_CHECK(hipMalloc((void**)&dc->tokens, S * (size_t)(d.max_ctx + 1) * 4));
    GTR_CHECK(hipMemset(dc->tokens, 0, S * (size_t)(d.max_ctx + 1) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->result, S * (size_t)(d.max_ctx + 2) * 4));
    GTR_CHECK(hipMemset(dc->result, 0, S * (size_t)(d.max_ctx + 2) * 4));
    const size_t planes = wide ? 2 : 1;                       // k_dec_mmv may split K over two workgroups: one output plane each
    GTR_CHECK(hipMalloc((void**)&dc->qkv_raw, planes * S * (size_t)(E + 2 * KV) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->proj_raw, planes * S * (size_t)E * 4));
    GTR_CHECK(hipMalloc((void**)&dc->down_raw, planes * S * (size_t)E * 4));
    GTR_CHECK(hipMalloc((void**)&dc->scores, S * (size_t)d.n_heads * d.max_ctx * 4));
    GTR_CHECK(hipMalloc((void**)&dc->stats, (S * (size_t)d.n_heads * dc->n_chunks * 2 + 16) * 4));   // + 8 chunks of slack: the one-launch kernel reads 8 per head
    GTR_CHECK(hipMemset(dc->stats, 0, (S * (size_t)d.n_heads * dc->n_chunks * 2 + 16) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->att_part, S * (size_t)d.n_heads * dc->n_chunks * dh * 4));
    // residual rows between kernels: f32 rows of exact storage values
    GTR_CHECK(hipMalloc((void**)&dc->xbuf, S * (size_t)E * 4));
    GTR_CHECK(hipMalloc((void**)&dc->hbuf, S * (size_t)E * 4));
    // (wide f16 decode keeps its staged rows as f16 matrices of 16-row tiles: 2 bytes per element, rows padded to the tile)
    // (... and the quantized wide path stages f16(q * delta) fragments for k_dec_mmvh: 2 bytes per element as well)
    const size_t stage_rows = wide ? 2 * ((S + 15) / 16 * 16) : S;
    GTR_CHECK(hipMalloc((void**)&dc->act_q, stage_rows * (size_t)F));
    GTR_CHECK(hipMemset(dc->act_q, 0, stage_rows * (size_t)F));
    GTR_CHECK(hipMalloc((void**)&dc->act_d, S * (size_t)(F / 32) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->act_sum, S * (size_t)(F / 32) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->act_f, S * (size_t)F * 4));
    {
        // lm_head launch: 8 waves x R rows per workgroup (R: single 8 | 4 for f16; multi 4 | 2)
        const int r = (n_seq == 1) ? (d.wdtype == GTEN_F16 ? 4 : 8) : (d.wdtype == GTEN_F16 ? 2 : 4);
        dc->n_best = ((d.n_vocab + 8 * r - 1) / (8 * r)) * 8;
    }
    GTR_CHECK(hipMalloc((void**)&dc->best_val, S * (size_t)dc->n_best * 4));
    GTR_CHECK(hipMalloc((void**)&dc->best_idx, S * (size_t)dc->n_best * 4));
    if (n_seq > 1) {
        GTR_CHECK(hipMalloc((void**)&dc->stg_q, stage_rows * (size_t)E));
        GTR_CHECK(hipMemset(dc->stg_q, 0, stage_rows * (size_t)E));
        GTR_CHECK(hipMalloc((void**)&dc->stg_d, S * (size_t)(E / 32) * 4));
        GTR_CHECK(hipMalloc((void**)&dc->stg_sum, S * (size_t)(E / 32) * 4));
        GTR_CHECK(hipMalloc((void**)&dc->stg_f, S * (size_t)E * 4));
        GTR_CHECK(hipMalloc((void**)&dc->logits_m, S * (size_t)d.n_vocab * 4));
        if (wide) {
            GTR_CHECK(hipMalloc((void**)&dc->gu_raw, planes * S * (size_t)2 * F * 4));
            if (d.wdtype != GTEN_F16)
            {
                if (int rc = (d.wdtype == GTEN_Q4) ? mmv_prepare<GTEN_Q4>() : mmv_prepare<GTEN_Q8>()) return rc;
                if (int rc = (d.wdtype == GTEN_Q4) ? mmvh_prepare<GTEN_Q4>() : mmvh_prepare<GTEN_Q8>()) return rc;
            }
        }
        std::vector<const void*> tab(S * d.n_layers * 2);
        for (size_t q = 0; q < S; q++)
            for (int l = 0; l < d.n_layers; l++) {
                const gten_hip_kv_ptrs& p = kv[q * d.n_layers + l];
                tab[(q * d.n_layers + l) * 2] = p.kcache;
                tab[(q * d.n_layers + l) * 2 + 1] = p.vcache;
            }
        GTR_CHECK(hipMalloc((void**)&dc->kv_tab, tab.size() * sizeof(void*)));
        GTR_CHECK(hipMemcpy(dc->kv_tab, tab.data(), tab.size() * sizeof(void*), hipMemcpyHostToDevice));
        dc->kv_real = tab;
        dc->kv_parked.assign(S, 0);
        const size_t cache_bytes = (size_t)d.max_ctx * gten_hip_row_bytes(d.adtype, KV);
        GTR_CHECK(hipMalloc(&dc->dummy_kv, 2 * cache_bytes));
        GTR_CHECK(hipMemset(dc->dummy_kv, 0, 2 * cache_bytes));
    }
    if (int rc = rope_table(dh, &dc->rope)) return rc;
    GTR_CHECK(hipMalloc((void**)&dc->rope_now, S * (size_t)(dh / 2) * sizeof(float2)));
    GTR_CHECK(hipMemset(dc->rope_now, 0, S * (size_t)(dh / 2) * sizeof(float2)));
    // the persistent step's state is allocated now: the first step may already be a stream capture
    if (dc->persist_on && n_seq == 1 && !dc->exact)
        if (int rc = persist_prepare(dc)) return rc;
    return 0;
}

// ---- the persistent step's state (gten_decode_persist.h): per-layer pointer table, granule buffers, control words
static int g_cu_count = -1;
static int persist_prepare(gten_hip_decoder* dc)
{
    if (dc->persist) return 0;
    if (g_cu_count < 0) {
        int dev = 0;
        GTR_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        GTR_CHECK(hipGetDeviceProperties(&prop, dev));
        g_cu_count = prop.multiProcessorCount;
    }
    const gten_hip_decoder_desc& d = dc->d;
    if (!persist_supported(d, dc->n_seq, dc->n_chunks, g_cu_count)) { dc->persist_on = false; return 0; }
    const int G = g_cu_count, E = d.n_embd, F = d.n_ffn, dh = E / d.n_heads, KV = dh * d.n_kv_heads;
    auto* ps = new PersistState;
    dc->persist = ps;
    g_persist_all.push_back(ps);
    std::vector<PLayer> tab((size_t)d.n_layers);
    for (int l = 0; l < d.n_layers; l++) {
        const gten_hip_layer_ptrs& L = dc->layers[l];
        tab[l] = PLayer{(const uint8_t*)L.wq, (const uint8_t*)L.wk, (const uint8_t*)L.wv, (const uint8_t*)L.wo, (const uint8_t*)L.wgate,
                        (const uint8_t*)L.wup, (const uint8_t*)L.wdown, (const uint16_t*)L.attn_norm, (const uint16_t*)L.ffn_norm,
                        (uint8_t*)L.kcache, (uint8_t*)L.vcache};
    }
    GTR_CHECK(hipMalloc((void**)&ps->layers, tab.size() * sizeof(PLayer)));
    GTR_CHECK(hipMemcpy(ps->layers, tab.data(), tab.size() * sizeof(PLayer), hipMemcpyHostToDevice));
    GTR_CHECK(hipMalloc((void**)&ps->ctl, 64));
    const unsigned ctl0[16] = {1u, 0u};
    GTR_CHECK(hipMemcpy(ps->ctl, ctl0, 64, hipMemcpyHostToDevice));
    PArgs& a = ps->args;
    a.rpa = (E + 2 * KV + G - 1) / G; a.rpo = (E + G - 1) / G; a.rph = (d.n_vocab + G - 1) / G; a.rw = (a.rph + 7) / 8;
    // granule buffers (8-byte {value, tag}), zeroed once: tag 0 never occurs (the epoch starts at 1)
    const size_t n_q = (size_t)G * a.rpa, n_part = (size_t)d.n_heads * dc->n_chunks * 66, n_att = (size_t)E, n_proj = (size_t)G * a.rpo,
                 n_act = (size_t)10 * (F / 32) + 8, n_down = (size_t)G * a.rpo, n_best = (size_t)2 * G;
    const size_t total = n_q + n_part + n_att + n_proj + n_act + n_down + n_best + 64;
    GTR_CHECK(hipMalloc((void**)&ps->gran, total * 8));
    GTR_CHECK(hipMemset(ps->gran, 0, total * 8));
    pu64* p = ps->gran;
    a.gq = p; p += n_q; a.gpart = p; p += n_part; a.gatt = p; p += n_att; a.gproj = p; p += n_proj;
    a.gact = p; p += (n_act + 7) / 8 * 8; a.gdown = p; p += n_down; a.gbest = p;
    if (const char* e = getenv("GTEN_HIP_PERSIST_STAMPS"); e && e[0] == '1') {
        ps->n_stamps = d.n_layers * 20 + 8;
        GTR_CHECK(hipMalloc((void**)&ps->stamps, (size_t)(d.n_layers * 20 + 8) * 4));
        GTR_CHECK(hipMemset(ps->stamps, 0, (size_t)(d.n_layers * 20 + 8) * 4));
    }
    a.layers = ps->layers; a.step = dc->step; a.tokens = dc->tokens; a.result = dc->result;
    a.embed = (const uint8_t*)d.embed; a.final_norm = (const uint16_t*)d.final_norm; a.lm_head = (const uint8_t*)d.lm_head;
    a.logits = d.logits; a.rope = dc->rope; a.ctl = ps->ctl; a.stamps = ps->stamps;
    a.n_layers = d.n_layers; a.E = E; a.F = F; a.KV = KV; a.n_heads = d.n_heads; a.n_kv = d.n_kv_heads; a.n_vocab = d.n_vocab;
    a.max_ctx = d.max_ctx; a.n_chunks = dc->n_chunks; a.kv_pitch = (int)gten_hip_row_bytes(d.adtype, KV);
    ps->grid = G;
    ps->smem = persist_smem();
    GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_persist<GTEN_Q4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ps->smem));
    return 0;
}
static hipError_t persist_free(gten_hip_decoder* dc)
{
    if (!dc->persist) return hipSuccess;
    PersistState* ps = dc->persist;
    g_persist_all.erase(std::remove(g_persist_all.begin(), g_persist_all.end(), ps), g_persist_all.end());
    hipError_t first = hipSuccess;
    for (void* b : {(void*)ps->layers, (void*)ps->ctl, (void*)ps->gran, (void*)ps->stamps})
        if (b) { const hipError_t e = hipFree(b); if (e != hipSuccess && first == hipSuccess) first = e; }
    delete ps;
    dc->persist = nullptr;
    return first;
}

extern "C" {

int gten_hip_decoder_create(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers, gten_hip_decoder** out)
{
    return decoder_create_common(desc, layers, nullptr, 1, out);
}

int gten_hip_decoder_create_multi(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers,
                                  const gten_hip_kv_ptrs* kv, int n_seq, gten_hip_decoder** out)
{
    return decoder_create_common(desc, layers, kv, n_seq, out);
}

int gten_hip_decoder_destroy(gten_hip_decoder* dc)
{
    if (!dc) return 0;
    GTR_NEED_INIT();
    GTR_CHECK(hipStreamSynchronize(stream()));
    // (every release is attempted; the first failure is what the call reports)
    hipError_t first = hipSuccess;
    auto rel = [&](hipError_t e) { if (e != hipSuccess && first == hipSuccess) first = e; };
    if (dc->exec) rel(hipGraphExecDestroy(dc->exec));
    if (dc->graph) rel(hipGraphDestroy(dc->graph));
    if (dc->exec_k) rel(hipGraphExecDestroy(dc->exec_k));
    if (dc->graph_k) rel(hipGraphDestroy(dc->graph_k));
    for (int m = 0; m < (1 << DEC_MAX_LANES); m++) {
        if (dc->exec_m[m]) rel(hipGraphExecDestroy(dc->exec_m[m]));
        if (dc->graph_m[m]) rel(hipGraphDestroy(dc->graph_m[m]));
        if (dc->exec_km[m]) rel(hipGraphExecDestroy(dc->exec_km[m]));
        if (dc->graph_km[m]) rel(hipGraphDestroy(dc->graph_km[m]));
    }
    void* bufs[] = {dc->ids_stage, dc->step, dc->tokens, dc->result, dc->qkv_raw, dc->proj_raw, dc->down_raw,
                    dc->scores, dc->stats, dc->att_part, dc->xbuf, dc->hbuf, dc->best_val, dc->best_idx,
                    dc->act_q, dc->act_d, dc->act_sum, dc->act_f, dc->stg_q, dc->stg_d, dc->stg_sum, dc->stg_f,
                    dc->logits_m, (void*)dc->kv_tab, dc->gu_raw, dc->rope_now, dc->dummy_kv};
    for (void* b : bufs) if (b) rel(hipFree(b));
    rel(persist_free(dc));
    for (int g = 1; g < DEC_MAX_LANES; g++) {
        if (dc->lane_stream[g]) { rel(hipStreamSynchronize(dc->lane_stream[g])); rel(hipStreamDestroy(dc->lane_stream[g])); }
        if (dc->lane_join[g]) rel(hipEventDestroy(dc->lane_join[g]));
    }
    if (dc->lane_fork) rel(hipEventDestroy(dc->lane_fork));
    if (first != hipSuccess) {
        delete dc;
        return fail((int)first, "decoder_destroy: %s", hipGetErrorString(first));
    }
    delete dc;
    return 0;
}

int gten_hip_decoder_set_tokens(gten_hip_decoder* dc, const int32_t* tokens_host, int first, int count)
{
    return gten_hip_decoder_set_tokens_seq(dc, 0, tokens_host, first, count);
}

int gten_hip_decoder_set_tokens_seq(gten_hip_decoder* dc, int seq, const int32_t* tokens_host, int first, int count)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && tokens_host && first >= 0 && count > 0 && first + count <= dc->d.max_ctx + 1, "decoder_set_tokens: bad range");
    GTR_REQUIRE(seq >= 0 && seq < dc->n_seq, "decoder_set_tokens: sequence %d outside [0, %d)", seq, dc->n_seq);
    for (int i = 0; i < count; i++)        // the embedding kernels index the table with the raw id
        GTR_REQUIRE(tokens_host[i] >= 0 && tokens_host[i] < dc->d.n_vocab, "decoder_set_tokens: token id %d at position %d outside [0, %d)",
                    tokens_host[i], first + i, dc->d.n_vocab);
    GTR_CHECK(hipMemcpyAsync(dc->tokens + (size_t)seq * (dc->d.max_ctx + 1) + first, tokens_host, (size_t)count * 4, hipMemcpyHostToDevice, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    return 0;
}

// Consecutive free-running steps replay a graph that holds DEC_GRAPH_STEPS of them: between two graph replays the
// command processor idles for several microseconds (8.6 us between a step's last kernel and the next step's first one
// in the rocprofv3 kernel trace, profiles/r02_*), i.e. ~1.5 % of a batch-1 step; four steps per replay pay it once.
#define DEC_GRAPH_STEPS 4
static int slots_leave(gten_hip_decoder* dc);     // continuous batching (below): back to every sequence's own caches
static int run_step(gten_hip_decoder* dc, int use_graph);
// (the graphs of the lane subset dc->lane_mask selects: slot 0 = every lane)
static unsigned graph_slot(const gten_hip_decoder* dc)
{
    const unsigned all = (1u << dc->lanes) - 1u, m = dc->lane_mask & all;
    return (dc->lanes <= 1 || m == 0 || m == all) ? 0u : m;
}
static int run_steps_free(gten_hip_decoder* dc, int count)
{
    if (prof_on()) { for (int i = 0; i < count; i++) if (int rc = run_step(dc, 0)) return rc; return 0; }
    const unsigned gs = graph_slot(dc);
    hipGraph_t& graph_k = gs ? dc->graph_km[gs] : dc->graph_k;
    hipGraphExec_t& exec_k = gs ? dc->exec_km[gs] : dc->exec_k;
    while (count >= DEC_GRAPH_STEPS) {
        if (!exec_k) {
            GTR_CHECK(hipStreamBeginCapture(stream(), hipStreamCaptureModeThreadLocal));
            const int rc = enqueue(dc, DEC_GRAPH_STEPS);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(stream(), &g);
            if (rc) { if (g) hipGraphDestroy(g); return rc; }
            GTR_CHECK(e);
            graph_k = g;
            GTR_CHECK(hipGraphInstantiate(&exec_k, graph_k, nullptr, nullptr, 0));
        }
        GTR_CHECK(hipGraphLaunch(exec_k, stream()));
        count -= DEC_GRAPH_STEPS;
    }
    for (int i = 0; i < count; i++)
        if (int rc = run_step(dc, 1)) return rc;
    return 0;
}

static int run_step(gten_hip_decoder* dc, int use_graph)
{
    if (!use_graph || prof_on()) return enqueue(dc);    // event pairs cannot be recorded into a capture
    const unsigned gs = graph_slot(dc);
    hipGraph_t& graph = gs ? dc->graph_m[gs] : dc->graph;
    hipGraphExec_t& exec = gs ? dc->exec_m[gs] : dc->exec;
    if (!exec) {
        GTR_CHECK(hipStreamBeginCapture(stream(), hipStreamCaptureModeThreadLocal));
        const int rc = enqueue(dc);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(stream(), &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        GTR_CHECK(e);
        graph = g;
        GTR_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    }
    GTR_CHECK(hipGraphLaunch(exec, stream()));
    return 0;
}

int gten_hip_decoder_step(gten_hip_decoder* dc, int n, int use_graph)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && n >= 1 && n <= dc->d.max_ctx, "decoder_step: n=%d outside [1, %d]", n, dc ? dc->d.max_ctx : 0);
    // The step's position lives on the device and the argmax kernel advances it,
    // so consecutive steps need no host-side update at all.  (All sequences of a
    // multi-sequence decoder are at the same position.)
    if (dc->dev_n != n || !dc->dev_ns.empty()) {
        std::vector<DecStep> st((size_t)dc->n_seq, DecStep{n, 1});
        GTR_CHECK(hipMemcpyAsync(dc->step, st.data(), st.size() * sizeof(DecStep), hipMemcpyHostToDevice, stream()));
        GTR_CHECK(hipStreamSynchronize(stream()));    // `st` lives on this stack frame
    }
    dc->dev_n = n + 1;
    dc->dev_ns.clear();
    if (int rc_ = slots_leave(dc)) return rc_;
    return run_step(dc, use_graph);
}

// `count` consecutive steps n_first, n_first + 1, ... of every sequence (teacher-forced ids already on the device):
// asynchronous, free-running (each step's last kernel advances the position), DEC_GRAPH_STEPS steps per graph replay.
int gten_hip_decoder_steps(gten_hip_decoder* dc, int n_first, int count, int use_graph)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && count >= 0 && n_first >= 1 && n_first + count - 1 <= dc->d.max_ctx, "decoder_steps: steps [%d, %d) outside [1, %d]",
                n_first, n_first + count, dc ? dc->d.max_ctx : 0);
    if (count == 0) return 0;
    if (dc->dev_n != n_first || !dc->dev_ns.empty()) {
        std::vector<DecStep> st((size_t)dc->n_seq, DecStep{n_first, 1});
        GTR_CHECK(hipMemcpyAsync(dc->step, st.data(), st.size() * sizeof(DecStep), hipMemcpyHostToDevice, stream()));
        GTR_CHECK(hipStreamSynchronize(stream()));
    }
    dc->dev_n = n_first + count;
    dc->dev_ns.clear();
    if (int rc_ = slots_leave(dc)) return rc_;
    if (!use_graph) { for (int i = 0; i < count; i++) if (int rc = run_step(dc, 0)) return rc; return 0; }
    return run_steps_free(dc, count);
}

// Sequences at DIFFERENT positions (continuous batching): sequence q decodes row n[q] - 1.  Every kernel of the
// step already reads its position from the sequence's own step word, the attention grid covers the longest
// context and workgroups past a sequence's end return at once.
int gten_hip_decoder_step_ragged(gten_hip_decoder* dc, const int* n_per_seq, int use_graph)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && n_per_seq, "decoder_step_ragged: null argument");
    for (int q = 0; q < dc->n_seq; q++)
        GTR_REQUIRE(n_per_seq[q] >= 1 && n_per_seq[q] <= dc->d.max_ctx, "decoder_step_ragged: n[%d]=%d outside [1, %d]", q, n_per_seq[q], dc->d.max_ctx);
    bool same = (int)dc->dev_ns.size() == dc->n_seq;
    for (int q = 0; same && q < dc->n_seq; q++) same = dc->dev_ns[q] == n_per_seq[q];
    if (!same) {
        std::vector<DecStep> st((size_t)dc->n_seq);
        for (int q = 0; q < dc->n_seq; q++) st[q] = DecStep{n_per_seq[q], 1};
        GTR_CHECK(hipMemcpyAsync(dc->step, st.data(), st.size() * sizeof(DecStep), hipMemcpyHostToDevice, stream()));
        GTR_CHECK(hipStreamSynchronize(stream()));    // `st` lives on this stack frame
    }
    dc->dev_ns.assign(n_per_seq, n_per_seq + dc->n_seq);
    for (int& v : dc->dev_ns) v += 1;                 // the argmax kernel advances every sequence
    dc->dev_n = -1;
    if (int rc_ = slots_leave(dc)) return rc_;
    return run_step(dc, use_graph);
}

// Greedy generation without the host in the loop (single-sequence decoders): steps n_first, n_first + 1, ... replay the
// graph back to back, each one's argmax written on the device as the next step's input token; the host reads the ids in
// slices of `GEN_SLICE` steps and stops at `eos` (the steps queued past it only touched rows that no longer matter).
int gten_hip_decoder_generate(gten_hip_decoder* dc, int n_first, int max_new, int eos, int32_t* out_host, int* n_out)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && out_host && n_out && dc->n_seq == 1, "decoder_generate: bad arguments (single-sequence decoders)");
    GTR_REQUIRE(n_first >= 1 && n_first <= dc->d.max_ctx && max_new >= 0, "decoder_generate: n_first=%d outside [1, %d]", n_first, dc->d.max_ctx);
    GTR_REQUIRE(!prof_on(), "decoder_generate: switch the per-launch profiler off first");
    constexpr int GEN_SLICE = 32;
    const int last = std::min(dc->d.max_ctx, n_first + max_new - 1);           // last step that may run
    const DecStep st{n_first, 3};
    GTR_CHECK(hipMemcpyAsync(dc->step, &st, sizeof(DecStep), hipMemcpyHostToDevice, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    dc->dev_n = -1;
    dc->dev_ns.clear();
    if (int rc_ = slots_leave(dc)) return rc_;
    int got = 0;
    std::vector<int32_t> ids(GEN_SLICE);
    for (int n = n_first; n <= last;) {
        const int cnt = std::min(GEN_SLICE, last - n + 1);
        if (int rc = run_steps_free(dc, cnt)) return rc;
        GTR_CHECK(hipMemcpyAsync(ids.data(), dc->result + n, (size_t)cnt * 4, hipMemcpyDeviceToHost, stream()));
        GTR_CHECK(hipStreamSynchronize(stream()));
        bool stop = false;
        for (int i = 0; i < cnt; i++) {
            if (ids[i] == eos) { stop = true; break; }
            out_host[got++] = ids[i];
        }
        if (stop) break;
        n += cnt;
    }
    *n_out = got;
    return 0;
}

// The same for the sequences of a multi-sequence decoder, each from its own position (continuous batching): sequence q
// starts at step n_first[q]; one that produced `eos`, its max_new ids or a full context is PARKED (its step word stops
// advancing: it recomputes the same row, which changes nothing) while the others go on.  out_host is [n_seq][max_new].
int gten_hip_decoder_generate_multi(gten_hip_decoder* dc, const int* n_first, const int* max_new_seq, int max_new, int eos, int32_t* out_host, int* n_out)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && n_first && out_host && n_out && max_new >= 0, "decoder_generate_multi: bad arguments");
    GTR_REQUIRE(!prof_on(), "decoder_generate_multi: switch the per-launch profiler off first");
    const int S = dc->n_seq, ctx = dc->d.max_ctx;
    for (int q = 0; q < S; q++) GTR_REQUIRE(n_first[q] >= 1 && n_first[q] <= ctx, "decoder_generate_multi: n_first[%d]=%d outside [1, %d]", q, n_first[q], ctx);
    constexpr int GEN_SLICE = 32;
    std::vector<DecStep> st((size_t)S);
    std::vector<int> cur(n_first, n_first + S), last((size_t)S);
    std::vector<char> live((size_t)S, 1);
    int n_live = 0;
    for (int q = 0; q < S; q++) {
        const int room = max_new_seq ? std::min(std::max(max_new_seq[q], 0), max_new) : max_new;   // this sequence's own bound
        last[q] = std::min(ctx, n_first[q] + room - 1);
        live[q] = last[q] >= n_first[q];
        st[q] = DecStep{n_first[q], live[q] ? 3 : 0};
        n_out[q] = 0;
        n_live += live[q];
    }
    GTR_CHECK(hipMemcpyAsync(dc->step, st.data(), st.size() * sizeof(DecStep), hipMemcpyHostToDevice, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    dc->dev_n = -1;
    dc->dev_ns.clear();
    if (int rc_ = slots_leave(dc)) return rc_;
    std::vector<int32_t> ids((size_t)GEN_SLICE);
    while (n_live > 0) {
        int cnt = GEN_SLICE;
        for (int q = 0; q < S; q++) if (live[q]) cnt = std::min(cnt, last[q] - cur[q] + 1);   // nobody runs past its last step
        if (int rc = run_steps_free(dc, cnt)) return rc;
        GTR_CHECK(hipStreamSynchronize(stream()));
        for (int q = 0; q < S; q++) {
            if (!live[q]) continue;
            GTR_CHECK(hipMemcpy(ids.data(), dc->result + (size_t)q * (ctx + 2) + cur[q], (size_t)cnt * 4, hipMemcpyDeviceToHost));
            bool stop = false;
            for (int i = 0; i < cnt && !stop; i++) {
                if (ids[i] == eos) stop = true;
                else out_host[(size_t)q * max_new + n_out[q]++] = ids[i];
            }
            cur[q] += cnt;
            if (stop || cur[q] > last[q]) {
                live[q] = 0; n_live--;
                st[q] = DecStep{std::min(cur[q], ctx), 0};
                GTR_CHECK(hipMemcpy(dc->step + q, &st[q], sizeof(DecStep), hipMemcpyHostToDevice));
            }
        }
    }
    return 0;
}

// ---- continuous batching: slots started / parked independently, the batch replayed free-running
// A PARKED slot keeps taking part in the shared launches (the batch is one matrix of n_seq rows), at position 1 and on
// the decoder's dummy K / V caches: its own caches are then free to be filled by the prompt processing of the NEXT
// sequence -- on the library's second stream, beside the steps of the live slots.
static int slot_caches(gten_hip_decoder* dc, int seq, bool parked)
{
    if (dc->kv_parked.empty() || (bool)dc->kv_parked[(size_t)seq] == parked) return 0;
    const size_t L = (size_t)dc->d.n_layers, off = (size_t)seq * L * 2;
    std::vector<const void*> row(L * 2);
    const size_t cache_bytes = (size_t)dc->d.max_ctx * gten_hip_row_bytes(dc->d.adtype, (dc->d.n_embd / dc->d.n_heads) * dc->d.n_kv_heads);
    for (size_t l = 0; l < L; l++) {
        row[2 * l] = parked ? dc->dummy_kv : dc->kv_real[off + 2 * l];
        row[2 * l + 1] = parked ? (const void*)((const uint8_t*)dc->dummy_kv + cache_bytes) : dc->kv_real[off + 2 * l + 1];
    }
    GTR_CHECK(hipMemcpyAsync(dc->kv_tab + off, row.data(), row.size() * sizeof(void*), hipMemcpyHostToDevice, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    dc->kv_parked[(size_t)seq] = parked;
    return 0;
}

static int slots_view(gten_hip_decoder* dc)
{
    if (dc->slots.empty()) {
        // every slot parked at a valid position until it is started
        dc->slots.assign((size_t)dc->n_seq, DecStep{1, 0});
        GTR_CHECK(hipMemcpyAsync(dc->step, dc->slots.data(), dc->slots.size() * sizeof(DecStep), hipMemcpyHostToDevice, stream()));
        GTR_CHECK(hipStreamSynchronize(stream()));
        for (int q = 0; q < dc->n_seq; q++)
            if (int rc = slot_caches(dc, q, true)) return rc;
    }
    dc->dev_n = -1;
    dc->dev_ns.clear();
    return 0;
}

// the other entry points work on every sequence's own caches
static int slots_leave(gten_hip_decoder* dc)
{
    for (int q = 0; q < dc->n_seq && !dc->kv_parked.empty(); q++)
        if (int rc = slot_caches(dc, q, false)) return rc;
    dc->slots.clear();
    return 0;
}

int gten_hip_decoder_slot_start(gten_hip_decoder* dc, int seq, int n_first)
{
    return gten_hip_decoder_slot_start_until(dc, seq, n_first, 0);
}

int gten_hip_decoder_slot_start_until(gten_hip_decoder* dc, int seq, int n_first, int n_last)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && seq >= 0 && seq < dc->n_seq, "decoder_slot_start: sequence %d outside [0, %d)", seq, dc ? dc->n_seq : 0);
    GTR_REQUIRE(n_first >= 1 && n_first <= dc->d.max_ctx, "decoder_slot_start: n_first=%d outside [1, %d]", n_first, dc->d.max_ctx);
    GTR_REQUIRE(n_last == 0 || (n_last >= n_first && n_last <= dc->d.max_ctx), "decoder_slot_start: n_last=%d outside [%d, %d]", n_last, n_first, dc->d.max_ctx);
    if (int rc = slots_view(dc)) return rc;
    if (int rc = slot_caches(dc, seq, false)) return rc;
    dc->slots[(size_t)seq] = DecStep{n_first, 3, n_last};
    GTR_CHECK(hipMemcpyAsync(dc->step + seq, &dc->slots[(size_t)seq], sizeof(DecStep), hipMemcpyHostToDevice, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    return 0;
}

// Several slots at once (a harvest of a 128-slot queue parks ~8 slots and starts ~8: 40 small copies each followed by a wait):
// slot seqs[i] is started at n_first[i] with its last step n_last[i] (0: none) and, when tokens[i] is given, its ids
// [0, n_first[i]) set -- or parked when n_first[i] == 0.  The step words and the cache table go up once, one wait at the end.
int gten_hip_decoder_slots_apply(gten_hip_decoder* dc, int count, const int* seqs, const int* n_first, const int* n_last, const int32_t* const* tokens)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && count >= 0 && (count == 0 || (seqs && n_first && n_last)), "decoder_slots_apply: bad arguments");
    if (int rc = slots_view(dc)) return rc;
    if (count == 0) return 0;
    const size_t L = (size_t)dc->d.n_layers;
    const size_t cache_bytes = (size_t)dc->d.max_ctx * gten_hip_row_bytes(dc->d.adtype, (dc->d.n_embd / dc->d.n_heads) * dc->d.n_kv_heads);
    std::vector<char> seen((size_t)dc->n_seq, 0);
    for (int i = 0; i < count; i++) {
        const int q = seqs[i];
        GTR_REQUIRE(q >= 0 && q < dc->n_seq, "decoder_slots_apply: sequence %d outside [0, %d)", q, dc->n_seq);
        GTR_REQUIRE(!seen[(size_t)q], "decoder_slots_apply: sequence %d appears twice (two step words / cache rows for one slot)", q);
        seen[(size_t)q] = 1;
        GTR_REQUIRE(n_first[i] >= 0 && n_first[i] <= dc->d.max_ctx, "decoder_slots_apply: n_first=%d outside [0, %d]", n_first[i], dc->d.max_ctx);
        GTR_REQUIRE(n_first[i] == 0 || n_last[i] == 0 || (n_last[i] >= n_first[i] && n_last[i] <= dc->d.max_ctx), "decoder_slots_apply: n_last=%d outside [%d, %d]",
                    n_last[i], n_first[i], dc->d.max_ctx);
        if (tokens && tokens[i] && n_first[i] > 0)
            for (int k = 0; k < n_first[i]; k++)
                GTR_REQUIRE(tokens[i][k] >= 0 && tokens[i][k] < dc->d.n_vocab, "decoder_slots_apply: token id %d at position %d outside [0, %d)", tokens[i][k], k, dc->d.n_vocab);
    }
    std::vector<const void*> rows((size_t)count * L * 2);
    for (int i = 0; i < count; i++) {
        const int q = seqs[i];
        const bool park = n_first[i] == 0;
        dc->slots[(size_t)q] = park ? DecStep{1, 0, 0} : DecStep{n_first[i], 3, n_last[i]};
        if (!dc->kv_parked.empty() && (bool)dc->kv_parked[(size_t)q] != park) {
            const void** row = rows.data() + (size_t)i * L * 2;            // (alive until the wait below)
            for (size_t l = 0; l < L; l++) {
                const size_t o = (size_t)q * L * 2 + 2 * l;
                row[2 * l] = park ? dc->dummy_kv : dc->kv_real[o];
                row[2 * l + 1] = park ? (const void*)((const uint8_t*)dc->dummy_kv + cache_bytes) : dc->kv_real[o + 1];
            }
            dc->kv_parked[(size_t)q] = park;
            GTR_CHECK(hipMemcpyAsync(dc->kv_tab + (size_t)q * L * 2, row, L * 2 * sizeof(void*), hipMemcpyHostToDevice, stream()));
        }
        if (!park && tokens && tokens[i])
            GTR_CHECK(hipMemcpyAsync(dc->tokens + (size_t)q * (dc->d.max_ctx + 1), tokens[i], (size_t)n_first[i] * 4, hipMemcpyHostToDevice, stream()));
        GTR_CHECK(hipMemcpyAsync(dc->step + q, &dc->slots[(size_t)q], sizeof(DecStep), hipMemcpyHostToDevice, stream()));
    }
    GTR_CHECK(hipStreamSynchronize(stream()));
    return 0;
}

int gten_hip_decoder_slot_park(gten_hip_decoder* dc, int seq)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && seq >= 0 && seq < dc->n_seq, "decoder_slot_park: sequence %d outside [0, %d)", seq, dc ? dc->n_seq : 0);
    if (int rc = slots_view(dc)) return rc;
    if (int rc = slot_caches(dc, seq, true)) return rc;
    DecStep& s = dc->slots[(size_t)seq];
    s.n = 1;
    s.advance = 0;
    s.stop = 0;
    GTR_CHECK(hipMemcpyAsync(dc->step + seq, &s, sizeof(DecStep), hipMemcpyHostToDevice, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    return 0;
}

int gten_hip_decoder_run(gten_hip_decoder* dc, int steps)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && steps >= 0, "decoder_run: bad arguments");
    GTR_REQUIRE(!prof_on(), "decoder_run: switch the per-launch profiler off first");
    if (int rc = slots_view(dc)) return rc;
    for (const DecStep& s : dc->slots)
        GTR_REQUIRE(!(s.advance & 1) || s.stop > 0 || s.n + steps - 1 <= dc->d.max_ctx, "decoder_run: %d steps would take a slot at n=%d past max_ctx %d", steps, s.n,
                    dc->d.max_ctx);
    // lanes whose slots are ALL parked sit this run out (a parked slot only recomputes row 0 of the dummy caches; its ids are
    // never read).  Nobody live at all: every lane, as before.
    unsigned mask = 0;
    const int SL = dc->n_seq / dc->lanes;
    for (int q = 0; q < dc->n_seq; q++)
        if (dc->slots[(size_t)q].advance & 1) mask |= 1u << (q / SL);
    dc->lane_mask = (dc->lanes > 1 && g_lane_skip) ? mask : 0;
    dc->last_run_lanes = (dc->lanes > 1 && g_lane_skip && mask) ? __builtin_popcount(mask) : dc->lanes;
    const int rc_run = run_steps_free(dc, steps);
    dc->lane_mask = 0;
    if (rc_run) return rc_run;
    for (DecStep& s : dc->slots)
        if (s.advance & 1) s.n = s.stop > 0 ? std::min(s.n + steps, s.stop) : s.n + steps;   // (a slot that reaches max_ctx + 1 has to be parked or restarted before the next run)
    return 0;
}

/* rows (sequences) per lane, the number of lanes, and how many of them the last gten_hip_decoder_run took */
int gten_hip_decoder_lane_info(gten_hip_decoder* dc, int* lane_rows, int* lanes, int* last_run_lanes)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc, "decoder_lane_info: null decoder");
    if (lane_rows) *lane_rows = dc->n_seq / dc->lanes;
    if (lanes) *lanes = dc->lanes;
    if (last_run_lanes) *last_run_lanes = dc->last_run_lanes ? dc->last_run_lanes : dc->lanes;
    return 0;
}

/* test hook: 0 = gten_hip_decoder_run always takes every lane (the behaviour before round 4) */
int gten_hip_set_lane_skip(int on)
{
    g_lane_skip = on != 0;
    return 0;
}

int gten_hip_decoder_slot_ids(gten_hip_decoder* dc, int seq, int n_from, int count, int32_t* ids_host)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && ids_host && seq >= 0 && seq < dc->n_seq, "decoder_slot_ids: bad arguments");
    GTR_REQUIRE(n_from >= 1 && count >= 0 && n_from + count - 1 <= dc->d.max_ctx, "decoder_slot_ids: steps [%d, %d) outside [1, %d]", n_from, n_from + count, dc->d.max_ctx);
    GTR_CHECK(hipStreamSynchronize(stream()));
    if (count > 0)
        GTR_CHECK(hipMemcpy(ids_host, dc->result + (size_t)seq * (dc->d.max_ctx + 2) + n_from, (size_t)count * 4, hipMemcpyDeviceToHost));
    return 0;
}

// the ids of steps [n_from[q], n_from[q] + count) of EVERY sequence in one gather launch and one copy (a harvest of 128 slots was
// 128 synchronous 32-byte copies: 1.5 ms per slice of 8 shared steps); n_from by value in the kernel arguments
struct IdsFrom { int n[128 * DEC_MAX_LANES]; };
__global__ __launch_bounds__(64) void k_dec_gather_ids(const int32_t* __restrict__ result, int stride, int last, const IdsFrom from, int count, int32_t* __restrict__ out)
{
    const int q = blockIdx.x, t = threadIdx.x;
    if (t < count) out[(size_t)q * count + t] = result[(size_t)q * stride + min(max(from.n[q], 0) + t, last)];
}

int gten_hip_decoder_slot_ids_all(gten_hip_decoder* dc, const int* n_from, int count, int32_t* ids_host)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && n_from && ids_host && count >= 0 && count <= 64, "decoder_slot_ids_all: bad arguments (count %d, at most 64)", count);
    GTR_REQUIRE(dc->n_seq <= 128 * DEC_MAX_LANES, "decoder_slot_ids_all: %d sequences", dc->n_seq);
    if (count == 0) { GTR_CHECK(hipStreamSynchronize(stream())); return 0; }
    if (!dc->ids_stage) GTR_CHECK(hipMalloc((void**)&dc->ids_stage, (size_t)dc->n_seq * 64 * 4));
    IdsFrom from{};
    for (int q = 0; q < dc->n_seq; q++) {
        GTR_REQUIRE(n_from[q] >= 0 && n_from[q] <= dc->d.max_ctx + 1, "decoder_slot_ids_all: sequence %d from step %d", q, n_from[q]);
        from.n[q] = n_from[q];
    }
    // (steps past max_ctx + 1 read the row's last entry: the caller takes only the steps a slot really ran)
    GTR_LAUNCH(KT_DEC_ARGMAX, k_dec_gather_ids, dim3(dc->n_seq), dim3(64), 0, (const int32_t*)dc->result, dc->d.max_ctx + 2, dc->d.max_ctx + 1, from, count, dc->ids_stage);
    GTR_CHECK(hipStreamSynchronize(stream()));
    GTR_CHECK(hipMemcpy(ids_host, dc->ids_stage, (size_t)dc->n_seq * count * 4, hipMemcpyDeviceToHost));
    return 0;
}

int gten_hip_decoder_time_family(gten_hip_decoder* dc, int family, int n, int reps, double* avg_us, int* launches_per_replay)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && avg_us && reps > 0 && n >= 1 && n <= dc->d.max_ctx, "decoder_time_family: bad arguments");
    GTR_REQUIRE(!prof_on(), "decoder_time_family: switch the per-launch profiler off first");
    std::vector<DecStep> st((size_t)dc->n_seq, DecStep{n, 0});
    GTR_CHECK(hipMemcpyAsync(dc->step, st.data(), st.size() * sizeof(DecStep), hipMemcpyHostToDevice, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    dc->dev_n = -1;
    dc->dev_ns.clear();
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    g_only_family = family;
    GTR_CHECK(hipStreamBeginCapture(stream(), hipStreamCaptureModeThreadLocal));
    const int rc = enqueue(dc);
    const hipError_t e = hipStreamEndCapture(stream(), &g);
    g_only_family = -1;
    if (rc) { if (g) hipGraphDestroy(g); return rc; }
    GTR_CHECK(e);
    size_t n_nodes = 0;
    GTR_CHECK(hipGraphGetNodes(g, nullptr, &n_nodes));
    GTR_REQUIRE(n_nodes > 0, "decoder_time_family: family %d has no launch in a decode step", family);
    GTR_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b;
    GTR_CHECK(hipEventCreate(&a));
    GTR_CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) GTR_CHECK(hipGraphLaunch(ge, stream()));
    GTR_CHECK(hipEventRecord(a, stream()));
    for (int i = 0; i < reps; i++) GTR_CHECK(hipGraphLaunch(ge, stream()));
    GTR_CHECK(hipEventRecord(b, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    float ms = 0.f;
    GTR_CHECK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a); hipEventDestroy(b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    *avg_us = (double)ms * 1e3 / ((double)reps * (double)n_nodes);
    if (launches_per_replay) *launches_per_replay = (int)n_nodes;
    return 0;
}

/* the persistent step (gten_decode_persist.h), over every live decoder of the process: how many run it, how many of its
 * launches were enqueued so far, the abort code of a poll that gave up (0 = none; cleared by the call -- the step's results
 * are invalid), and workgroup 0's phase stamps of the newest such decoder (10 ns ticks; GTEN_HIP_PERSIST_STAMPS=1 at its
 * creation, else zeros).  Waits for the stream. */
int gten_hip_persist_status(int* n_decoders, unsigned long long* launches, unsigned* abort_code, unsigned* stamps_host, int n_stamps)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(n_decoders && launches && abort_code, "persist_status: null argument");
    GTR_CHECK(hipStreamSynchronize(stream()));
    *n_decoders = (int)g_persist_all.size();
    *launches = g_persist_launches;
    *abort_code = 0;
    for (PersistState* ps : g_persist_all) {
        unsigned ctl[2];
        GTR_CHECK(hipMemcpy(ctl, ps->ctl, 8, hipMemcpyDeviceToHost));
        if (ctl[1]) {
            *abort_code = ctl[1];
            const unsigned z = 0;
            GTR_CHECK(hipMemcpy(ps->ctl + 1, &z, 4, hipMemcpyHostToDevice));
        }
    }
    if (stamps_host && n_stamps > 0) {
        for (int i = 0; i < n_stamps; i++) stamps_host[i] = 0;
        for (auto it = g_persist_all.rbegin(); it != g_persist_all.rend(); ++it)
            if ((*it)->stamps) {
                GTR_CHECK(hipMemcpy(stamps_host, (*it)->stamps, (size_t)std::min((*it)->n_stamps, n_stamps) * 4, hipMemcpyDeviceToHost));
                break;
            }
    }
    return 0;
}

int gten_hip_decoder_result(gten_hip_decoder* dc, int n, int32_t* argmax_host)
{
    return gten_hip_decoder_result_seq(dc, 0, n, argmax_host);
}

int gten_hip_decoder_result_seq(gten_hip_decoder* dc, int seq, int n, int32_t* argmax_host)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && argmax_host && n >= 1 && n <= dc->d.max_ctx, "decoder_result: bad arguments");
    GTR_REQUIRE(seq >= 0 && seq < dc->n_seq, "decoder_result: sequence %d outside [0, %d)", seq, dc->n_seq);
    GTR_CHECK(hipMemcpyAsync(argmax_host, dc->result + (size_t)seq * (dc->d.max_ctx + 2) + n, 4, hipMemcpyDeviceToHost, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    return 0;
}

int gten_hip_decoder_logits_seq(gten_hip_decoder* dc, int seq, float* logits_host)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && logits_host && seq >= 0 && seq < dc->n_seq, "decoder_logits: bad arguments");
    const float* src = dc->n_seq > 1 ? dc->logits_m + (size_t)seq * dc->d.n_vocab : dc->d.logits;
    GTR_CHECK(hipMemcpyAsync(logits_host, src, (size_t)dc->d.n_vocab * 4, hipMemcpyDeviceToHost, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    return 0;
}

} // extern "C"
