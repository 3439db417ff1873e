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

#define WXP_PLANES 8             // K planes of k_dec_wxp_f16 (gten_decode_wxp.h), added by the staging launches
#include "gten_decode_wx.h"
#include "gten_decode_wide_wx.h"
#include "gten_decode_ffn.h"
#include "gten_decode_wxp.h"
#include "gten_decode_attn.h"
#include "gten_decode_attn_exact64.h"
#include "gten_decode_attn_wide.h"
#include "gten_decode_attn_hm.h"

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
    // ---- head-major shadows of the K / V caches (gten_decode_attn_hm.h): decoders of 16+ sequences, Q8 activations, fast forms.
    // [sequence][layer][K | V][kv head][chunk][HM_CHUNK_BYTES]; hm_dirty[q] != 0: sequence q's shadow must be re-imported from
    // its cache rows before the next step (set at creation, by every (re)start of the sequence, and -- through the watch
    // registry of gten_rt.h -- by every write of this library into one of its cache rows)
    uint8_t* hm = nullptr;
    size_t hm_seq_stride = 0, hm_cache_bytes = 0, hm_chunk_bytes = 0;
    std::vector<char> hm_dirty;
    unsigned long long hm_imports = 0, hm_import_launches = 0;
    // every decoder: do the caches it appends to overlap a watch of ANOTHER decoder (cached per registry epoch)
    unsigned long long watch_epoch = 0;
    bool watch_foreign = false;
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
    const size_t rplanes = dc->n_seq >= 16 ? (size_t)WXP_PLANES : 1;        // (k_dec_wxp_f16: eight K planes)
    b.qkv_raw = dc->qkv_raw + o * rplanes * (E + 2 * KV);
    b.proj_raw = dc->proj_raw + o * rplanes * E; b.down_raw = dc->down_raw + o * rplanes * E;
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
// decoders of 16+ sequences created AFTERWARDS keep head-major shadows of their K / V caches (gten_decode_attn_hm.h) or read
// the cache rows as they lie (0: k_dec_attn_mm_g, round 4's kernel -- kept for the A/B and as the reference the tests hold
// the new kernel to)
static bool g_kv_head_major = true;
extern "C" int gten_hip_set_kv_head_major(int on)
{
    g_kv_head_major = on != 0;
    return 0;
}
extern "C" int gten_hip_set_decode_exact(int on)
{
    g_decode_exact = on != 0;
    return 0;
}

// many sequences, Q8 activations, 64-wide heads, 8 (or 4, 2, 1) query heads per kv head: grouped kernels
static bool attention_grouped_ok(const AttnArgs& t, int n_seq)
{
    const int grp = t.n_heads / t.n_kv;
    const int min_seq = 8;                                       // measured (q4, ctx 2048): 8 sequences +6 %, 4 and 2 slower
    if (t.adtype == GTEN_F16 && n_seq < 16) return false;        // f16 below 16 sequences: the per-head kernels (one launch, k_dec_attn_one64)
    // (group sizes other than 8 / 4 / 2 / 1 and head widths other than 64 keep the per-head kernels at every sequence count:
    //  tests/test_multiseq_gpu.py decodes such a model -- 3 query heads per kv head -- against single-sequence decode)
    return n_seq >= min_seq && (t.adtype == GTEN_Q8 || t.adtype == GTEN_F16) && t.d_head == 64 && (grp == 8 || grp == 4 || grp == 2 || grp == 1);
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

// f16 activations, 16+ sequences, head-major shadows (round 5): one launch with chunk-local statistics (k_dec_attn_hm_f16)
static bool grouped_hm_f16(const AttnArgs& t, int n_seq)
{
    return n_seq >= 16 && t.adtype == GTEN_F16 && t.d_head == 64 && t.hm_k && !g_exact_now;
}

static bool grouped_one_pass(const AttnArgs& t, int n_seq)
{
    if (grouped_mm(t, n_seq) || grouped_hm_f16(t, n_seq)) return true;
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
    if constexpr (ADT == GTEN_Q8) if (grouped_mm(t, n_seq) && t.hm_k) {
        // head-major shadows: one wave per (sequence, kv head, chunk), HM_WAVES consecutive chunks per workgroup
        const int n_cq = (t.n_chunks + HM_WAVES - 1) / HM_WAVES;
        if (n_seq >= 128) DEC_LAUNCH(KT_DEC_ATTN_SCORE, (k_dec_attn_hm<GRP, true>), dim3(n_seq * n_cq * t.n_kv), dim3(64 * HM_WAVES), 2048, t, n_seq, n_cq);
        else DEC_LAUNCH(KT_DEC_ATTN_SCORE, (k_dec_attn_hm<GRP, false>), dim3(n_seq * n_cq * t.n_kv), dim3(64 * HM_WAVES), 2048, t, n_seq, n_cq);
        return 0;
    }
    if constexpr (ADT == GTEN_Q8) if (grouped_mm(t, n_seq)) {
        const size_t smem = (size_t)DEC_CHUNK * 68 + 2 * 8 * 264 * 2 + 16 * 64 + 32 * 4 + 8 * 4 + (size_t)4 * (DEC_MAXGRP + 2) * 2 + 128 + 128 * 4 + 80 + 64;
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
    if constexpr (ADT == GTEN_F16) if (grouped_hm_f16(t, n_seq)) {
        const int n_cq = (t.n_chunks + HM_WAVES - 1) / HM_WAVES;
        if (n_seq >= 128) DEC_LAUNCH(KT_DEC_ATTN_SCORE, (k_dec_attn_hm_f16<GRP, true>), dim3(n_seq * n_cq * t.n_kv), dim3(64 * HM_WAVES), 2048, t, n_seq, n_cq);
        else DEC_LAUNCH(KT_DEC_ATTN_SCORE, (k_dec_attn_hm_f16<GRP, false>), dim3(n_seq * n_cq * t.n_kv), dim3(64 * HM_WAVES), 2048, t, n_seq, n_cq);
        return 0;
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
// (A/B switch of the streamed gate | up and lm_head kernel, gten_decode_ffn.h: 0 keeps k_dec_mmvh<.., 8, 4, ..> for 128-row lanes too)
static bool g_ffn_streamed = true;
extern "C" int gten_hip_set_ffn_streamed(int on)
{
    g_ffn_streamed = on != 0;
    return 0;
}
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
    const MmvRest rest{a.w[1], a.w[2], a.d_out[1], a.d_out[2], a.plane};
    // lm_head at eight row tiles (round 5): FOUR feature tiles per workgroup.  Its 1001 two-tile workgroups each pulled the lane's
    // 512 KB of staged activations through L2 -- 512 MB per launch, which is what its 53 us were; 501 workgroups of 64 features
    // halve that (the same sums per row: a row's K slices and their order do not depend on the feature tiles beside it).
    if constexpr (RT == 8 && WT == GTEN_Q4) {
        if (a.n_mats == 1 && ks == 1 && cols >= 16384 && g_ffn_streamed && a.S == 128 && a.d_in == 2048) {
            // ... and as the streamed kernel of gten_decode_ffn.h (the same sums): no cross-wave sum, the weights slice by slice
            DEC_LAUNCH(tag, k_dec_ffn_q4<false>, dim3((cols + 63) / 64), dim3(512), (size_t)2 * 8 * 4 * 1024, a.ah, a.w[0], a.w[0], (uint16_t*)a.out, a.d_in,
                       a.d_out[0], a.S, a.out_cols, 8);
            return 0;
        }
        if (a.n_mats == 1 && ks == 1 && cols >= 16384 && ppr <= 32 * (MMV_MAXP / 4)) {
            const size_t smem4 = std::max((size_t)64 * nbs * 16, (size_t)8 * 16 * RT * 64) + 64 * nbs * 2;
            GTR_REQUIRE(smem4 <= 150 * 1024, "decoder: the weight slab of d_in %d does not fit LDS", a.d_in);
            DEC_LAUNCH(tag, (k_dec_mmvh<WT, 8, 4, false>), dim3((cols + 63) / 64, 1), dim3(512), smem4, a.ah, a.w[0], a.out, a.d_in, a.d_out[0], a.out_cols, a.S,
                       a.n_mats, rest);
            return 0;
        }
    }
    if constexpr (RT == 8 && WT == GTEN_Q8) {
        if (a.n_mats == 1 && ks == 1 && cols >= 16384 && g_ffn_streamed && a.S == 128 && a.d_in == 2048) {
            // q8 lm_head of a full lane: streamed, the sums of the two-tile launch below (one K plane of eight wave slices)
            DEC_LAUNCH(tag, k_dec_ffn_q8<false>, dim3((cols + 63) / 64), dim3(512), (size_t)2 * 8 * 4 * 1024, a.ah, a.w[0], a.w[0], (uint16_t*)a.out, a.d_in,
                       a.d_out[0], a.S, a.out_cols, 8);
            return 0;
        }
    }
    // ... and of a lane of four row tiles (49 .. 64 sequences): the same kernels, waves 4 .. 7 only expand weights (the sums of the
    // k_dec_mmvh launch below: one K plane of eight wave slices)
    if constexpr (RT == 4 && (WT == GTEN_Q4 || WT == GTEN_Q8)) {
        if (a.n_mats == 1 && ks == 1 && cols >= 16384 && g_ffn_streamed && a.d_in == 2048) {
            if constexpr (WT == GTEN_Q4)
                DEC_LAUNCH(tag, (k_dec_ffn_q4<false, true>), dim3((cols + 63) / 64), dim3(512), (size_t)2 * 8 * 4 * 1024, a.ah, a.w[0], a.w[0], (uint16_t*)a.out,
                           a.d_in, a.d_out[0], a.S, a.out_cols, 4);
            else
                DEC_LAUNCH(tag, (k_dec_ffn_q8<false, true>), dim3((cols + 63) / 64), dim3(512), (size_t)2 * 8 * 4 * 1024, a.ah, a.w[0], a.w[0], (uint16_t*)a.out,
                           a.d_in, a.d_out[0], a.S, a.out_cols, 4);
            return 0;
        }
    }
    const dim3 grid((cols + 16 * ft - 1) / (16 * ft), ks);
    if (ft > 1)
        DEC_LAUNCH(tag, (k_dec_mmvh<WT, RT, FTW, false>), grid, dim3(512), smem, a.ah, a.w[0], a.out, a.d_in, a.d_out[0], a.out_cols, a.S, a.n_mats, rest);
    else
        DEC_LAUNCH(tag, (k_dec_mmvh<WT, RT, 1, false>), grid, dim3(512), smem, a.ah, a.w[0], a.out, a.d_in, a.d_out[0], a.out_cols, a.S, a.n_mats, rest);
    return 0;
}
// o and down of every f16 decoder of 16+ sequences in eight K planes of 64-feature workgroups (gten_decode_wxp.h) -- or as k_dec_mmv_f16 in
// two (0: A/B; the sums differ in the association of the f32 additions, so the switch is read when a step is enqueued or captured and
// holds for every decoder alike)
static bool g_wx_planes = true;
extern "C" int gten_hip_set_wx_planes(int on)
{
    g_wx_planes = on != 0;
    return 0;
}
static bool wxp_shape(int wt, int d_in, int d_out, int planes = WXP_PLANES)
{
    const int nbk = d_in / 32 / planes;
    return g_wx_planes && !g_exact_now && wt == GTEN_F16 && d_in % (32 * planes) == 0 && d_out % 64 == 0 &&
           (nbk == 1 || nbk == 2 || nbk == 3 || nbk == 4 || nbk == 6 || nbk == 8 || nbk == 16 || nbk == 22);
}
#define WXP_QKV_PLANES 4         // q | k | v: four planes (every attention workgroup of a (sequence, kv head) re-reads them)
static int launch_wxp_f16(int tag, const uint16_t* ah, const void* w, float* out, int d_in, int d_out, int S, int out_cols,
                          const void* w1 = nullptr, int d1 = 0, const void* w2 = nullptr, int d2 = 0, int planes = WXP_PLANES)
{
    const int n_mats = w2 ? 3 : (w1 ? 2 : 1);
    const MmvRest rest{w1, w2, d1, d2, 0};
    const int rt_s = (S + 15) / 16, frt = rt_s <= 4 ? rt_s : 8, nbk = d_in / 32 / planes;
    GTR_REQUIRE(rt_s <= 4 || S == 128, "decoder: a lane of %d rows (one to four row tiles, or 128 rows)", S);
    const dim3 grid((d_out + d1 + d2) / 64, planes);
    const size_t smem = (size_t)nbk * 4096;
    static bool attr = false;
    if (!attr) {
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_wxp_f16<22>, hipFuncAttributeMaxDynamicSharedMemorySize, 22 * 4096));
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_wxp_f16<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 4096));
        attr = true;
    }
#define WXP_GO(N_) DEC_LAUNCH(tag, k_dec_wxp_f16<N_>, grid, dim3(512), smem, ah, (const uint16_t*)w, out, d_in, d_out, S, out_cols, S * out_cols, frt, n_mats, rest)
    switch (nbk) {
    case 1: WXP_GO(1); break;
    case 2: WXP_GO(2); break;
    case 3: WXP_GO(3); break;
    case 4: WXP_GO(4); break;
    case 6: WXP_GO(6); break;
    case 8: WXP_GO(8); break;
    case 16: WXP_GO(16); break;
    default: WXP_GO(22); break;
    }
#undef WXP_GO
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
    // full 128-row lanes at K = 2048, q4: the streamed form (gten_decode_ffn.h) -- the same sums bit for bit, 15.1 -> 12.2 us per launch
    if (WQ == GTEN_Q4 && g_ffn_streamed && rt == 8 && S == 128 && d_in == 2048) {
        DEC_LAUNCH(tag, k_dec_ffn_q4<true>, dim3(n_ffn / 32), dim3(512), (size_t)2 * 8 * 4 * 1024, ah, wgate, wup, out_frag, d_in, n_ffn, S, 0, 8);
        return 0;
    }
    if (WQ == GTEN_Q4 && g_ffn_streamed && rt == 4 && d_in == 2048) {          // four row tiles: waves 4 .. 7 only expand weights
        DEC_LAUNCH(tag, (k_dec_ffn_q4<true, true>), dim3(n_ffn / 32), dim3(512), (size_t)2 * 8 * 4 * 1024, ah, wgate, wup, out_frag, d_in, n_ffn, S, 0, 4);
        return 0;
    }
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
    if (WT == GTEN_Q4) MMVH_ATTR(8, 4);
#define MMVH_ATTR_S(RT_) GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_mmvh<WT, RT_, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
    MMVH_ATTR_S(1); MMVH_ATTR_S(2); MMVH_ATTR_S(3); MMVH_ATTR_S(4); MMVH_ATTR_S(8);
#undef MMVH_ATTR_S
#undef MMVH_ATTR
    if (WT == GTEN_Q4) {
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_ffn_q4<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4 * 1024));
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_ffn_q4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4 * 1024));
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_ffn_q4<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4 * 1024));
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_ffn_q4<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4 * 1024));
    }
    if (WT == GTEN_Q8) {
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_ffn_q8<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4 * 1024));
        GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_ffn_q8<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4 * 1024));
    }
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
    GTR_REQUIRE(a.d_in % 256 == 0 && a.S >= 1 && (a.S <= 64 || (WT == GTEN_F16 && a.S == 128)), "decoder: skinny W.x wants d_in %% 256 == 0 and <= 64 rows (f16: or 128)");
    if (WT == GTEN_F16) {
        for (int k = 0; k + 1 < a.n_mats; k++) GTR_REQUIRE(a.d_out[k] % 16 == 0, "decoder: concatenated matrices must be multiples of 16 wide");
        // the lm_head of a lane of four row tiles: streamed, 64 features per workgroup (gten_decode_ffn.h) -- the sums of k_dec_mmv_f16<4> at ks = 1
        const int frt16 = (a.S + 15) / 16;
        if (g_ffn_streamed && a.n_mats == 1 && a.ks <= 1 && a.d_out[0] >= 16384 && a.d_in == 2048 && (frt16 == 4 || a.S == 128)) {
            static bool attr = false;
            if (!attr) {
                GTR_CHECK(hipFuncSetAttribute((const void*)k_dec_ffn_f16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 4 * 1024));
                attr = true;
            }
            DEC_LAUNCH(tag, k_dec_ffn_f16<false>, dim3((a.d_out[0] + 63) / 64), dim3(512), (size_t)2 * 8 * 4 * 1024, (const uint16_t*)a.aq, (const uint16_t*)a.w[0],
                       (const uint16_t*)a.w[0], (uint16_t*)a.out, a.d_in, a.d_out[0], a.S, frt16, a.out_cols);
            return 0;
        }
        switch ((a.S + 15) / 16) {
        case 1: return launch_mmv_f16_rt<1>(tag, a);
        case 2: return launch_mmv_f16_rt<2>(tag, a);
        case 3: return launch_mmv_f16_rt<3>(tag, a);
        case 4: return launch_mmv_f16_rt<4>(tag, a);
        default: return launch_mmv_f16_rt<8>(tag, a);          // a full lane of 128 rows
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
    // f16: o and down as eight K planes of 64-feature workgroups (gten_decode_wxp.h)
    const bool down_planes = wxp_shape(WT, F, E), o_planes = wxp_shape(WT, E, E);
    for (int l = 0; l < d.n_layers; l++) {
        const gten_hip_layer_ptrs& L = dc->layers[l];
        Gemv8Args st = base;
        st.d_in = E; st.norm_w = (const uint16_t*)L.attn_norm; st.x_out = xbuf;
        st.act_q = b.stg_q; st.act_d = b.stg_d; st.act_sum = b.stg_sum; st.act_f = b.stg_f;
        if (l == 0) {
            st.table = d.embed; st.rope = dc->rope; st.rope_now = b.rope_now; st.rope_half = dh / 2; st.n_vocab = V; st.tokens = b.tokens;
            rc = launch_stage_frag<WT, PRO_EMBED>(KT_DEC_STAGE, st, S);
        } else {
            st.res_a = hbuf; st.res_raw = b.down_raw; st.raw_stride = E; st.raw_plane = (down_planes || ks_of(F) > 1) ? S * E : 0;
            st.raw_nplanes = down_planes ? WXP_PLANES : 0;
            rc = launch_stage_frag<WT, PRO_RESID>(KT_DEC_STAGE, st, S);
        }
        if (rc) return rc;
        const int QW = E + 2 * KV;
        // (measured: 160 x 2 workgroups of 4 row tiles run slower than 160; of 8 row tiles too: 11.8 against 9.0 us per launch)
        const int ks_qkv = S <= 32 ? ks_of(E) : 1;
        // f16 on head-major shadows: q | k | v in eight K planes too (k_dec_attn_hm_f16 adds them)
        const bool qkv_planes = wxp_shape(WT, E, E, WXP_QKV_PLANES) && dc->hm && dh == 64 && KV % 64 == 0 && !g_exact_now &&
                                (d.n_heads / d.n_kv_heads == 8 || d.n_heads / d.n_kv_heads == 4 || d.n_heads / d.n_kv_heads == 2 || d.n_heads / d.n_kv_heads == 1);
        if (qkv_planes) {
            if ((rc = launch_wxp_f16(KT_DEC_GEMV_QKV, (const uint16_t*)b.stg_q, L.wq, b.qkv_raw, E, E, S, QW, L.wk, KV, L.wv, KV, WXP_QKV_PLANES))) return rc;
        } else if ((rc = mmk(KT_DEC_GEMV_QKV, b.stg_q, b.stg_d, b.qkv_raw, QW, E, ks_qkv, L.wq, E, L.wk, KV, L.wv, KV))) return rc;
        AttnArgs t{};
        t.step = b.step; t.qkv_raw = b.qkv_raw; t.kv_pitch = kv_pitch; t.scores = b.scores; t.stats = b.stats;
        t.att_part = b.att_part; t.rope = dc->rope; t.rope_now = b.rope_now;
        t.adtype = d.adtype; t.n_heads = d.n_heads; t.n_kv = d.n_kv_heads; t.d_head = dh; t.max_ctx = d.max_ctx;
        t.n_chunks = dc->n_chunks; t.n_embd = E;
        t.kv_tab = (const void* const*)b.kv_tab; t.layer = l; t.n_layers = d.n_layers;
        t.qkv_stride = QW; t.scores_stride = d.n_heads * d.max_ctx; t.stats_stride = d.n_heads * dc->n_chunks * 2;
        t.part_stride = d.n_heads * dc->n_chunks * dh;
        t.qkv_plane = (qkv_planes || ks_qkv > 1) ? S * QW : 0;
        t.qkv_nplanes = qkv_planes ? WXP_QKV_PLANES : 0;
        if (dc->hm) {
            t.hm_k = dc->hm + (size_t)lane * S * dc->hm_seq_stride + (size_t)l * 2 * dc->hm_cache_bytes;
            t.hm_seq_stride = dc->hm_seq_stride; t.hm_cache_bytes = dc->hm_cache_bytes;
        }
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
        if (o_planes) {
            if ((rc = launch_wxp_f16(KT_DEC_GEMV_O, (const uint16_t*)b.stg_q, L.wo, b.proj_raw, E, E, S, E))) return rc;
        } else if ((rc = mmk(KT_DEC_GEMV_O, b.stg_q, b.stg_d, b.proj_raw, E, E, ks_of(E), L.wo, E))) return rc;
        Gemv8Args sh = base;
        sh.d_in = E; sh.res_a = xbuf; sh.res_raw = b.proj_raw; sh.raw_stride = E; sh.x_out = hbuf;
        sh.raw_plane = (o_planes || ks_of(E) > 1) ? S * E : 0;
        sh.raw_nplanes = o_planes ? WXP_PLANES : 0;
        sh.norm_w = (const uint16_t*)L.ffn_norm;
        sh.act_q = b.stg_q; sh.act_d = b.stg_d; sh.act_sum = b.stg_sum; sh.act_f = b.stg_f;
        if ((rc = launch_stage_frag<WT, PRO_RESID>(KT_DEC_STAGE, sh, S))) return rc;
        // gate|up has 176-352 workgroups already; the split measured +4 % at 16 sequences, +2 % at 64, -2 % at 32
        const int ks_gu = ((S + 15) / 16 != 2) ? ks_of(E) : 1;
        const bool fuse_ffn = folded && WT == GTEN_Q4 && E / 32 <= 32 * (MMV_MAXP / 4);      // (a tile's slab: <= 2 pieces per thread)
        if (fuse_ffn) {
            if ((rc = launch_mmvh_silu<WT>(KT_DEC_GEMV_GATEUP, (const uint16_t*)b.stg_q, L.wgate, L.wup, F, E, S, (uint16_t*)b.act_q))) return rc;
        } else if (folded && WT == GTEN_Q8 && g_ffn_streamed && S == 128 && E == 2048 && ks_gu == 2 && F % 32 == 0) {
            // q8 weights, a full 128-row lane: gate | up and the silu . mul chain as ONE streamed launch (gten_decode_ffn.h) -- the bits of the
            // k_dec_mmvh<Q8, 8, 2, false> + k_dec_silumul_rows pair below (two K planes of eight wave slices each)
            DEC_LAUNCH(KT_DEC_GEMV_GATEUP, k_dec_ffn_q8<true>, dim3(F / 32), dim3(512), (size_t)2 * 4 * 4 * 1024, (const uint16_t*)b.stg_q, L.wgate, L.wup,
                       (uint16_t*)b.act_q, E, F, S, 0, 8);
        } else if (folded && WT == GTEN_Q8 && g_ffn_streamed && (S + 15) / 16 == 4 && E == 2048 && ks_gu == 2 && F % 32 == 0) {
            DEC_LAUNCH(KT_DEC_GEMV_GATEUP, (k_dec_ffn_q8<true, true>), dim3(F / 32), dim3(512), (size_t)2 * 4 * 4 * 1024, (const uint16_t*)b.stg_q, L.wgate, L.wup,
                       (uint16_t*)b.act_q, E, F, S, 0, 4);
        } else if (WT == GTEN_F16 && g_ffn_streamed && ((S + 15) / 16 == 4 || S == 128) && E % 256 == 0 && (E / 128) % 2 == 0 && ks_gu == 2 && F % 32 == 0) {
            // f16: gate | up and the chain of a lane of four row tiles as one streamed launch (gten_decode_ffn.h) -- the bits of the
            // k_dec_mmv_f16 + k_dec_silumul_rows_f16 pair below
            DEC_LAUNCH(KT_DEC_GEMV_GATEUP, k_dec_ffn_f16<true>, dim3(F / 32), dim3(512), (size_t)2 * 4 * 4 * 1024, (const uint16_t*)b.stg_q, (const uint16_t*)L.wgate,
                       (const uint16_t*)L.wup, (uint16_t*)b.act_q, E, F, S, (S + 15) / 16, 0);
        } else {
        if ((rc = mmk(KT_DEC_GEMV_GATEUP, b.stg_q, b.stg_d, b.gu_raw, 2 * F, E, ks_gu, L.wgate, F, L.wup, F))) return rc;
        if (WT == GTEN_F16)
            DEC_LAUNCH(KT_DEC_GEMV_GATEUP, k_dec_silumul_rows_f16, dim3(F / 256, S), dim3(256), 0, (const float*)b.gu_raw, F,
                       ks_gu > 1 ? S * 2 * F : 0, (uint16_t*)b.act_q);
        else
            DEC_LAUNCH(KT_DEC_GEMV_GATEUP, k_dec_silumul_rows, dim3(F / 256, S), dim3(256), 0, (const float*)b.gu_raw, F, (S + 15) / 16,
                       ks_gu > 1 ? S * 2 * F : 0, b.act_q, b.act_d, b.act_sum, folded ? 1 : 0);
        }
        if (down_planes) {
            if ((rc = launch_wxp_f16(KT_DEC_GEMV_DOWN, (const uint16_t*)b.act_q, L.wdown, b.down_raw, F, E, S, E))) return rc;
        } else if ((rc = mmk(KT_DEC_GEMV_DOWN, b.act_q, b.act_d, b.down_raw, E, F, ks_of(F), L.wdown, E))) return rc;
    }
    Gemv8Args sf = base;
    sf.d_in = E; sf.res_a = hbuf; sf.res_raw = b.down_raw; sf.raw_stride = E; sf.norm_w = (const uint16_t*)d.final_norm;
    sf.raw_plane = (down_planes || ks_of(F) > 1) ? S * E : 0;
    sf.raw_nplanes = down_planes ? WXP_PLANES : 0;
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
// The batch-1 step as ONE persistent launch (round 4: built, bit-identical to the launch chain, 0.90 x its speed -- HISTORY.md):
// compiled only with -DGTEN_WITH_PERSIST=1 (GTEN_HIP_EXTRA_FLAGS for tinyllama.cpp_amd/build.py); the product library
// answers gten_hip_set_decode_persistent(1) with an error and runs the launch chain.
#ifndef GTEN_WITH_PERSIST
#define GTEN_WITH_PERSIST 0
#endif
#if GTEN_WITH_PERSIST
#include "gten_decode_persist.h"
#else
struct PersistState {};
static bool g_persist_on = false;
extern "C" int gten_hip_set_decode_persistent(int on)
{
    if (on) return fail(-4, "set_decode_persistent: this library was built without the persistent step (-DGTEN_WITH_PERSIST=1, HISTORY.md)");
    return 0;
}
#endif

static int persist_prepare(gten_hip_decoder* dc);

static int enqueue_lane(gten_hip_decoder* dc, int lane)
{
    g_exact_now = dc->exact;
    // single sequence, Q8 activations: the whole step as one persistent launch (a family-restricted timing replay wants
    // the launch chain's kernels)
#if GTEN_WITH_PERSIST
    if (dc->persist_on && dc->n_seq == 1 && !dc->exact && (g_only_family < 0 || g_only_family == KT_DEC_PERSIST)) {
        if (int rc = persist_prepare(dc)) return rc;
        if (dc->persist) return persist_launch<GTEN_Q4>(dc->persist);
    }
#endif
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
        // weight fragment feeds eight matrix instructions and the weights are read once per 128 sequences), else 64 (the exact forms,
        // 192 sequences, f16 at other widths).  Per sequence the same bits either way (tests/test_multiseq_gpu.py).  Measured (q4,
        // ctx -> 2048, tok/s): 128 sequences 50.8 k as two lanes of 64, 51.7 k as one of 128; 256 sequences 56.9 k as four
        // lanes of 64, 67.3 k as two of 128; serving 1024 prompts through 128 slots 29.9 k -> 31.5 k new ids/s.
        // (round 5: f16 weights too, at TinyLlama's width -- k_dec_mmv_f16<8> and the streamed k_dec_ffn_f16 with all eight waves on row tiles:
        //  256 sequences 5.71 -> 5.38 ms per step as two lanes of 128 instead of four of 64, 128 sequences 3.25 -> 3.17)
        const bool can128 = !g_decode_exact && n_seq % 128 == 0 &&
                            ((d.wdtype != GTEN_F16 && d.adtype == GTEN_Q8) || (d.wdtype == GTEN_F16 && d.adtype == GTEN_F16 && d.n_embd == 2048));
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
    GTR_CHECK(hipMalloc((void**)&dc->qkv_raw, (wide ? (size_t)WXP_PLANES : 1) * S * (size_t)(E + 2 * KV) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->proj_raw, (wide ? (size_t)WXP_PLANES : 1) * S * (size_t)E * 4));      // (k_dec_wxp_f16: eight K planes)
    GTR_CHECK(hipMalloc((void**)&dc->down_raw, (wide ? (size_t)WXP_PLANES : 1) * S * (size_t)E * 4));
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
        const int grp = d.n_heads / d.n_kv_heads;
        if (wide && g_kv_head_major && !dc->exact && (d.adtype == GTEN_Q8 || d.adtype == GTEN_F16) && dh == 64 && (grp == 8 || grp == 4 || grp == 2 || grp == 1)) {
            dc->hm_chunk_bytes = d.adtype == GTEN_Q8 ? HM_CHUNK_BYTES : HMF_CHUNK_BYTES;
            dc->hm_cache_bytes = (size_t)d.n_kv_heads * dc->n_chunks * dc->hm_chunk_bytes;
            dc->hm_seq_stride = (size_t)d.n_layers * 2 * dc->hm_cache_bytes;
            GTR_CHECK(hipMalloc((void**)&dc->hm, S * dc->hm_seq_stride));
            GTR_CHECK(hipMemset(dc->hm, 0, S * dc->hm_seq_stride));
            dc->hm_dirty.assign(S, 1);
            for (size_t q = 0; q < S; q++)
                for (size_t i = 0; i < (size_t)d.n_layers * 2; i++)
                    kv_watch_add(dc->kv_real[q * d.n_layers * 2 + i], cache_bytes, dc, &dc->hm_dirty[q]);
        }
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
#if GTEN_WITH_PERSIST
static int g_cu_count = -1;
static int persist_prepare(gten_hip_decoder* dc)
{
    if (dc->persist) return 0;
    // ONE live persistent decoder per device: the kernel holds 89 728 bytes of LDS and 512 threads, so only one of its workgroups
    // fits a CU and its grid takes every CU -- two such grids on two streams would each hold part of the chip and poll for
    // workgroups that can never be scheduled (every poll then runs into its spin limit and aborts the step).  A second decoder
    // keeps the launch chain.
    if (!g_persist_all.empty()) { dc->persist_on = false; return 0; }
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
#else
static int persist_prepare(gten_hip_decoder* dc) { dc->persist_on = false; return 0; }
static hipError_t persist_free(gten_hip_decoder*) { return hipSuccess; }
#endif

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
    kv_watch_remove(dc, nullptr);
    void* bufs[] = {dc->hm, dc->ids_stage, dc->step, dc->tokens, dc->result, dc->qkv_raw, dc->proj_raw, dc->down_raw,
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
// ---- what has to happen before a decoder's steps are launched (never inside a stream capture):
//   1. the cache rows this decoder is about to append to may be shadowed by ANOTHER decoder (a batch's wide decoder over the
//      same sequences): tell the registry, that decoder re-imports them;
//   2. this decoder's own head-major shadows: sequences whose rows were written since the last import (or that were
//      (re)started) are imported now, rows [0, n - 1) with n read from the sequence's step word on the device -- on the
//      stream the steps follow on, so whatever ordered the rows' writers ahead of the steps orders them ahead of this.
static void hm_mark_all(gten_hip_decoder* dc)
{
    std::fill(dc->hm_dirty.begin(), dc->hm_dirty.end(), (char)1);
}
static int pre_run(gten_hip_decoder* dc)
{
    const gten_hip_decoder_desc& d = dc->d;
    if (kv_watch_any()) {
        const size_t cache_bytes = (size_t)d.max_ctx * gten_hip_row_bytes(d.adtype, (d.n_embd / d.n_heads) * d.n_kv_heads);
        if (dc->watch_epoch != kv_watch_epoch()) {
            dc->watch_epoch = kv_watch_epoch();
            dc->watch_foreign = false;
            if (dc->n_seq == 1) {
                for (const gten_hip_layer_ptrs& L : dc->layers)
                    if (kv_watch_overlaps(L.kcache, cache_bytes, dc) || kv_watch_overlaps(L.vcache, cache_bytes, dc)) { dc->watch_foreign = true; break; }
            } else {
                for (const void* c : dc->kv_real)
                    if (kv_watch_overlaps(c, cache_bytes, dc)) { dc->watch_foreign = true; break; }
            }
        }
        if (dc->watch_foreign) {
            if (dc->n_seq == 1) for (const gten_hip_layer_ptrs& L : dc->layers) { kv_watch_touch(L.kcache, cache_bytes, dc); kv_watch_touch(L.vcache, cache_bytes, dc); }
            else for (const void* c : dc->kv_real) kv_watch_touch(c, cache_bytes, dc);
        }
    }
    if (!dc->hm) return 0;
    HmImportList items{};
    const size_t kv_pitch = gten_hip_row_bytes(d.adtype, (d.n_embd / d.n_heads) * d.n_kv_heads);
    auto flush = [&]() -> int {
        if (items.n == 0) return 0;
        if (d.adtype == GTEN_Q8)
            GTR_LAUNCH(KT_PACK, k_kv_import_hm, dim3(d.n_kv_heads * dc->n_chunks, d.n_layers * 2, items.n), dim3(256), (size_t)DEC_CHUNK * 68, items,
                       (const DecStep*)dc->step, (const void* const*)dc->kv_tab, dc->hm, dc->hm_seq_stride, dc->hm_cache_bytes, d.n_layers, d.n_kv_heads,
                       dc->n_chunks, d.max_ctx, kv_pitch);
        else
            GTR_LAUNCH(KT_PACK, k_kv_import_hm_f16, dim3(d.n_kv_heads * dc->n_chunks, d.n_layers * 2, items.n), dim3(256), (size_t)DEC_CHUNK * 33 * 4, items,
                       (const DecStep*)dc->step, (const void* const*)dc->kv_tab, dc->hm, dc->hm_seq_stride, dc->hm_cache_bytes, d.n_layers, d.n_kv_heads,
                       dc->n_chunks, d.max_ctx, kv_pitch);
        dc->hm_import_launches++;
        dc->hm_imports += (unsigned long long)items.n;
        items.n = 0;
        return 0;
    };
    for (int q = 0; q < dc->n_seq; q++) {
        if (!dc->hm_dirty[(size_t)q]) continue;
        dc->hm_dirty[(size_t)q] = 0;
        items.seq[items.n++] = q;
        if (items.n == (int)(sizeof(items.seq) / sizeof(items.seq[0])))
            if (int rc = flush()) return rc;
    }
    return flush();
}

// (the graphs of the lane subset dc->lane_mask selects: slot 0 = every lane)
static unsigned graph_slot(const gten_hip_decoder* dc)
{
    const unsigned all = (1u << dc->lanes) - 1u, m = dc->lane_mask & all;
    return (dc->lanes <= 1 || m == 0 || m == all) ? 0u : m;
}
static int run_steps_free(gten_hip_decoder* dc, int count)
{
    if (int rc = pre_run(dc)) return rc;
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
    if (int rc = pre_run(dc)) return rc;
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
        hm_mark_all(dc);                              // not the continuation of this decoder's own steps: the rows are the caller's
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
        hm_mark_all(dc);
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
        hm_mark_all(dc);
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
    hm_mark_all(dc);
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
        hm_mark_all(dc);
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
    if (!dc->slots.empty()) hm_mark_all(dc);           // (leaving the slot view: positions and cache bindings are the caller's again)
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
    if (dc->hm) dc->hm_dirty[(size_t)seq] = 1;         // a (re)started sequence: its rows [0, n_first - 1) are the caller's
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
        if (dc->hm && !park) dc->hm_dirty[(size_t)q] = 1;
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

// A PARKED slot gets another set of caches (continuous batching with prompts processed ahead of the slots that will take
// them: the host fills spare cache sets while every slot is busy and hands a ready set to the next slot that ends).  Only the
// host copy of the table changes here; the device rows follow when the slot is started (slots_apply / slot_start).
int gten_hip_decoder_slot_bind(gten_hip_decoder* dc, int seq, const gten_hip_kv_ptrs* kv)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && kv && seq >= 0 && seq < dc->n_seq, "decoder_slot_bind: sequence %d outside [0, %d)", seq, dc ? dc->n_seq : 0);
    GTR_REQUIRE(!dc->kv_parked.empty() && dc->kv_parked[(size_t)seq], "decoder_slot_bind: slot %d is not parked", seq);
    const size_t off = (size_t)seq * dc->d.n_layers * 2;
    for (int l = 0; l < dc->d.n_layers; l++)
        GTR_REQUIRE(kv[l].kcache && kv[l].vcache, "decoder_slot_bind: null cache pointer (layer %d)", l);
    for (int l = 0; l < dc->d.n_layers; l++) {
        dc->kv_real[off + 2 * (size_t)l] = kv[l].kcache;
        dc->kv_real[off + 2 * (size_t)l + 1] = kv[l].vcache;
    }
    if (dc->hm) {
        // the slot's shadow follows the set it is bound to: watch the new rows instead of the old ones
        const size_t cache_bytes = (size_t)dc->d.max_ctx * gten_hip_row_bytes(dc->d.adtype, (dc->d.n_embd / dc->d.n_heads) * dc->d.n_kv_heads);
        char* flag = &dc->hm_dirty[(size_t)seq];
        kv_watch_remove(dc, flag);
        for (size_t i = 0; i < (size_t)dc->d.n_layers * 2; i++) kv_watch_add(dc->kv_real[off + i], cache_bytes, dc, flag);
        *flag = 1;
    }
    return 0;
}

static int run_impl(gten_hip_decoder* dc, int steps, bool skip_lanes)
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
    dc->lane_mask = (dc->lanes > 1 && skip_lanes) ? mask : 0;
    dc->last_run_lanes = (dc->lanes > 1 && skip_lanes && mask) ? __builtin_popcount(mask) : dc->lanes;
    const int rc_run = run_steps_free(dc, steps);
    dc->lane_mask = 0;
    if (rc_run) return rc_run;
    for (DecStep& s : dc->slots)
        if (s.advance & 1) s.n = s.stop > 0 ? std::min(s.n + steps, s.stop) : s.n + steps;   // (a slot that reaches max_ctx + 1 has to be parked or restarted before the next run)
    return 0;
}

int gten_hip_decoder_run(gten_hip_decoder* dc, int steps) { return run_impl(dc, steps, g_lane_skip); }
/* ... with the lanes whose slots are all parked left out of THIS run whatever gten_hip_set_lane_skip says (the tail of a queue,
 * whose last sequences the caller has moved into as few lanes as they fit: TinyLlamaBatch::serve) */
int gten_hip_decoder_run_lanes(gten_hip_decoder* dc, int steps, int skip_empty_lanes) { return run_impl(dc, steps, skip_empty_lanes != 0 || g_lane_skip); }

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
    hm_mark_all(dc);
    if (int rc_ = pre_run(dc)) return rc_;
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

// a persistent step that gave up (a poll ran into its spin limit) leaves stale ids and logits behind: every read of a result says so
static int persist_aborted(gten_hip_decoder* dc)
{
#if GTEN_WITH_PERSIST
    if (dc->persist) {
        unsigned code = 0;
        GTR_CHECK(hipMemcpy(&code, dc->persist->ctl + 1, 4, hipMemcpyDeviceToHost));
        if (code) return fail(-5, "decoder: the persistent step aborted (code %u): results are invalid; gten_hip_persist_status clears the state", code);
    }
#else
    (void)dc;
#endif
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
    *n_decoders = 0;
    *launches = 0;
    *abort_code = 0;
    if (stamps_host && n_stamps > 0)
        for (int i = 0; i < n_stamps; i++) stamps_host[i] = 0;
#if GTEN_WITH_PERSIST
    *n_decoders = (int)g_persist_all.size();
    *launches = g_persist_launches;
    for (PersistState* ps : g_persist_all) {
        unsigned ctl[2];
        GTR_CHECK(hipMemcpy(ctl, ps->ctl, 8, hipMemcpyDeviceToHost));
        if (ctl[1]) {
            *abort_code = ctl[1];
            // the aborted step never bumped the epoch, and the granules it did publish carry exactly the tags the next launch
            // would wait for: clear the abort word AND move to the next epoch, so nothing of the partial step can be accepted
            const unsigned next[2] = {ctl[0] + 1u, 0u};
            GTR_CHECK(hipMemcpy(ps->ctl, next, 8, hipMemcpyHostToDevice));
        }
    }
    if (stamps_host && n_stamps > 0) {
        for (auto it = g_persist_all.rbegin(); it != g_persist_all.rend(); ++it)
            if ((*it)->stamps) {
                GTR_CHECK(hipMemcpy(stamps_host, (*it)->stamps, (size_t)std::min((*it)->n_stamps, n_stamps) * 4, hipMemcpyDeviceToHost));
                break;
            }
    }
#endif
    return 0;
}

/* head-major K / V shadows of a decoder (gten_decode_attn_hm.h): whether it keeps them, how many sequence imports (rows ->
 * shadow) it has launched so far and in how many launches -- tests watch these to see that a write into a cache row is
 * followed by a re-import */
int gten_hip_decoder_kv_info(gten_hip_decoder* dc, int* head_major, unsigned long long* seq_imports, unsigned long long* import_launches)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc, "decoder_kv_info: null decoder");
    if (head_major) *head_major = dc->hm ? 1 : 0;
    if (seq_imports) *seq_imports = dc->hm_imports;
    if (import_launches) *import_launches = dc->hm_import_launches;
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
    return persist_aborted(dc);
}

int gten_hip_decoder_logits_seq(gten_hip_decoder* dc, int seq, float* logits_host)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && logits_host && seq >= 0 && seq < dc->n_seq, "decoder_logits: bad arguments");
    const float* src = dc->n_seq > 1 ? dc->logits_m + (size_t)seq * dc->d.n_vocab : dc->d.logits;
    GTR_CHECK(hipMemcpyAsync(logits_host, src, (size_t)dc->d.n_vocab * 4, hipMemcpyDeviceToHost, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    return persist_aborted(dc);
}

} // extern "C"
