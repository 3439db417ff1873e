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
//     kernel in its prologue, entirely on chip (it is 2048..5632 elements), so
//     no kernel waits on a 1-workgroup element-wise launch;
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

using namespace gtd;

extern __shared__ __attribute__((aligned(16))) uint8_t g_smem[];

#define DEC_CHUNK 256            // attention positions per workgroup

struct DecStep {
    int n;                        // context length of this step; the new row is n-1
    int advance;                  // argmax kernel bumps n afterwards (free-running replay)
};

// ---------------------------------------------------------------- prologues
//
// All prologues leave the W.x input vector staged in LDS: ActQ8 form for Q8
// activations, an f32 row (of exact fp16 values) for f16 activations.
// `row` is an f32 LDS scratch row of length d.  Thread t owns elements
// t, t+256, ... in every element-wise step, so only reductions need barriers.

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

// final step of every prologue: the (unrounded) f32 row becomes the staged input
__device__ __forceinline__ void finish_stage(ActStage s, int adtype, int d)
{
    if (adtype == GTEN_Q8) quantize_to_actq8(s.row, d >> 5, s.q8);
    else round_row_inplace(s.row, GTEN_F16, d);
    __syncthreads();
}

// RMSNorm of s.row in place (gten/ops.h:762-778)
__device__ __forceinline__ void rms_norm_row(ActStage s, const uint16_t* __restrict__ w, int d)
{
    float ss = 0.f;
    for (int i = threadIdx.x; i < d; i += blockDim.x) ss += s.row[i] * s.row[i];
    ss = block_sum(ss, s.red);
    const float rms = sqrtf(ss / (float)d);
    for (int i = threadIdx.x; i < d; i += blockDim.x) s.row[i] = s.row[i] / (rms + 1e-6f) * h2f(w[i]);
}

// x = embedding row of the step's token, in activation dtype (gten/ops.h:514-533)
template <int WT>
__device__ __forceinline__ void pro_embed(ActStage s, const void* table, int n_vocab, int tok, int adtype, int d)
{
    const int nb = d >> 5;
    if (WT == GTEN_F16) {
        const uint16_t* src = (const uint16_t*)table + (size_t)tok * d;
        for (int i = threadIdx.x; i < d; i += blockDim.x) s.row[i] = h2f(src[i]);
    } else if (WT == GTEN_Q8) {
        const PackedW p = packed_view(table, GTEN_Q8, n_vocab, d);
        const uint8_t* q0 = p.qs + (size_t)tok * nb * 32;
        for (int i = threadIdx.x; i < d; i += blockDim.x) {
            const int b = i >> 5, e = i & 31;
            const int qv = (int)(int8_t)q0[(size_t)(e >> 4) * nb * 16 + (size_t)b * 16 + (e & 15)];
            s.row[i] = (float)qv * h2f(p.ds[(size_t)tok * nb + b]);      // copied blocks dequantize to this
        }
    } else {
        const PackedW p = packed_view(table, GTEN_Q4, n_vocab, d);
        const uint8_t* q = p.qs + (size_t)tok * nb * 16;
        for (int i = threadIdx.x; i < d; i += blockDim.x) {
            const int b = i >> 5, e = i & 31;
            const uint8_t byte = q[(size_t)b * 16 + (e & 15)];
            const int nib = (e < 16) ? (byte >> 4) : (byte & 0x0f);
            s.row[i] = (float)(nib - 7) * h2f(p.ds[(size_t)tok * nb + b]);
        }
        round_row_inplace(s.row, adtype, d);     // Q4 row is re-quantized to Q8 (gten/ops.h:522-528)
    }
}

// x = write(add(a, write(raw)))  i.e. Linear output in activation dtype, then
// the residual add, then stored (gten/modules.cpp:33-43, 52-63)
__device__ __forceinline__ void pro_residual(ActStage s, const uint8_t* __restrict__ a, const float* __restrict__ raw, int adtype, int d)
{
    for (int i = threadIdx.x; i < d; i += blockDim.x) s.row[i] = raw[i];
    round_row_inplace(s.row, adtype, d);
    for (int i = threadIdx.x; i < d; i += blockDim.x) s.row[i] = load_elem(a, adtype, i) + s.row[i];
}

// ------------------------------------------------------------ GEMV kernels

struct GemvMat { const void* w; int rows; };

template <int WT>
__device__ __forceinline__ float dec_dot(const void* w, int rows, int d_in, int r, const ActStage& s)
{
    if (WT == GTEN_F16) return wave_dot_f16((const uint16_t*)w + (size_t)r * d_in, s.row, d_in);
    const PackedW pw = packed_view(w, WT, rows, d_in);
    if (WT == GTEN_Q8) return wave_dot_q8(pw, (size_t)r, s.q8);
    return wave_dot_q4(pw, (size_t)r, s.q8);
}

enum { PRO_EMBED = 0, PRO_RESID = 1, PRO_ATT = 2, PRO_SILUMUL = 3 };

struct GemvArgs {
    const DecStep* step;
    // matrices computed by this launch, outputs are concatenated in `out`
    GemvMat m[3];
    int n_mats;
    int d_in;
    float* out;                   // raw f32 dot products
    int adtype;
    // prologue inputs
    const void* table; int n_vocab; const int32_t* tokens;     // PRO_EMBED
    const uint8_t* res_a; const float* res_raw;                 // PRO_RESID
    uint8_t* x_out;                                             // PRO_EMBED/PRO_RESID: the new residual row (storage dtype)
    const uint16_t* norm_w;                                     // PRO_EMBED/PRO_RESID
    const float* att_part; int n_heads, d_head, n_chunks;       // PRO_ATT
    const float* gate_raw; const float* up_raw;                 // PRO_SILUMUL
    int rows_per_wave;
};

template <int WT, int PRO>
__global__ __launch_bounds__(256) void k_dec_gemv(const GemvArgs a)
{
    const int d = a.d_in;
    ActStage s = carve_stage(d);
    const int n = a.step->n;

    if (PRO == PRO_EMBED || PRO == PRO_RESID) {
        if (PRO == PRO_EMBED) pro_embed<WT>(s, a.table, a.n_vocab, a.tokens[n - 1], a.adtype, d);
        else pro_residual(s, a.res_a, a.res_raw, a.adtype, d);
        // the residual stream row is stored once, by workgroup 0, for the later add
        if (a.x_out && blockIdx.x == 0) store_row(s.row, a.adtype, d, a.x_out);
        round_row_inplace(s.row, a.adtype, d);
        if (a.norm_w) {
            rms_norm_row(s, a.norm_w, d);
        }
    } else if (PRO == PRO_ATT) {
        // attention output row = sum of the per-chunk partials (fixed order)
        const int nch = (n + DEC_CHUNK - 1) / DEC_CHUNK;
        for (int i = threadIdx.x; i < d; i += blockDim.x) {
            const int h = i / a.d_head, e = i % a.d_head;
            float v = 0.f;
            for (int j = 0; j < nch; j++) v += a.att_part[((size_t)h * a.n_chunks + j) * a.d_head + e];
            s.row[i] = v;
        }
    } else {
        // silu(write(gate)) then * write(up), each written in the activation dtype
        // (gten/modules.cpp:238-247: silu and mul are in place on the gate buffer)
        for (int i = threadIdx.x; i < d; i += blockDim.x) s.row[i] = a.gate_raw[i];
        round_row_inplace(s.row, a.adtype, d);
        for (int i = threadIdx.x; i < d; i += blockDim.x) { const float x = s.row[i]; s.row[i] = x / (1.0f + expf(-x)); }
        round_row_inplace(s.row, a.adtype, d);
        // up: round in registers with the same 32-lane grouping
        const int padded = d;   // d % 32 == 0
        for (int i = threadIdx.x; i < padded; i += blockDim.x) {
            float u = a.up_raw[i];
            if (a.adtype == GTEN_Q8) {
                const float amax = group_max<32>(fabsf(u));
                const Q8Scale sc = q8_scale_from_absmax(amax);
                u = (float)q8_round(u, sc.scale) * sc.ddeq;
            } else {
                u = h2f(f2h(u));
            }
            s.row[i] = s.row[i] * u;
        }
    }
    finish_stage(s, a.adtype, d);

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int rpw = a.rows_per_wave;
    const int r0 = (blockIdx.x * 4 + wid) * rpw;
    int total = 0;
    for (int k = 0; k < a.n_mats; k++) total += a.m[k].rows;
    for (int j = 0; j < rpw; j++) {
        const int r = r0 + j;
        if (r >= total) break;
        int lr = r, k = 0;
        while (lr >= a.m[k].rows) { lr -= a.m[k].rows; k++; }
        const float v = dec_dot<WT>(a.m[k].w, a.m[k].rows, d, lr, s);
        if (lane == 0) a.out[r] = v;
    }
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
    int adtype, n_heads, n_kv, d_head, max_ctx, n_chunks, n_embd;
};

// write(raw) -> rope -> write, for one head vector of d_head (32 or 64) elements
// held by lanes [0, d_head) of wave 0; returns the final f32 value (exact storage
// value) and, for Q8, leaves quants/deltas in qi8/qd/qd16.  The rotate-half
// partner (j, j + d_head/2) lives d_head/2 lanes away: one xor-shuffle.
// gten/modules.cpp:196-201 + gten/ops.h:714-755
__device__ __forceinline__ float head_prep(float raw, bool act, bool do_rope, int pos, int d_head, int adtype,
                                           const float2* __restrict__ rope, int8_t* qi8, float* qd, uint16_t* qd16)
{
    const int t = threadIdx.x;
    float v = act ? raw : 0.f;
    // Linear output written in the activation dtype
    if (adtype == GTEN_Q8) {
        const Q8Scale sc = q8_scale_from_absmax(group_max<32>(fabsf(v)));
        v = (float)q8_round(v, sc.scale) * sc.ddeq;
    } else {
        v = h2f(f2h(v));
    }
    if (do_rope) {
        const int half = d_head >> 1;
        const float other = __shfl_xor(v, half, 64);
        const bool lo = (t & half) == 0;
        const int j = t & (half - 1);
        const float2 cs = rope[(size_t)pos * half + j];
        const float x0 = lo ? v : other, x1 = lo ? other : v;
        v = lo ? (x0 * cs.x - x1 * cs.y) : (x0 * cs.y + x1 * cs.x);
        if (!act) v = 0.f;
    }
    if (adtype == GTEN_Q8) {
        const Q8Scale sc = q8_scale_from_absmax(group_max<32>(fabsf(v)));
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

// greedy argmax, strict '>' so the first maximum wins (tinyllama.cpp:416-424)
__global__ __launch_bounds__(1024) void k_dec_argmax(const float* __restrict__ logits, int n_vocab, DecStep* step,
                                                     int32_t* __restrict__ result)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n_vocab; i += blockDim.x) {
        const float v = logits[i];
        if (v > best) { best = v; idx = i; }
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
        if (step->advance) step->n = n + 1;
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
    float *qkv_raw = nullptr, *proj_raw = nullptr, *gu_raw = nullptr, *down_raw = nullptr;
    float *scores = nullptr, *stats = nullptr, *att_part = nullptr;
    uint8_t *xbuf = nullptr, *hbuf = nullptr;
    int n_chunks = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const float2* rope = nullptr;
};

template <int WT, int PRO>
static int launch_gemv(int tag, const GemvArgs& a, int total_rows)
{
    const int rows_per_wg = 4 * a.rows_per_wave;
    const dim3 grid((total_rows + rows_per_wg - 1) / rows_per_wg), block(256);
    GTR_LAUNCH(tag, (k_dec_gemv<WT, PRO>), grid, block, stage_bytes(a.d_in), a);
    return 0;
}

template <int WT>
static int enqueue_step(gten_hip_decoder* dc)
{
    const gten_hip_decoder_desc& d = dc->d;
    const int E = d.n_embd, F = d.n_ffn, dh = E / d.n_heads, KV = dh * d.n_kv_heads;
    const size_t kv_pitch = gten_hip_row_bytes(d.adtype, KV);
    for (int l = 0; l < d.n_layers; l++) {
        const gten_hip_layer_ptrs& L = dc->layers[l];
        // ---- q,k,v projections of norm(x)
        GemvArgs a{};
        a.step = dc->step; a.adtype = d.adtype; a.d_in = E;
        a.m[0] = {L.wq, E}; a.m[1] = {L.wk, KV}; a.m[2] = {L.wv, KV}; a.n_mats = 3;
        a.out = dc->qkv_raw; a.norm_w = (const uint16_t*)L.attn_norm; a.x_out = dc->xbuf;
        a.rows_per_wave = 2;
        int rc;
        if (l == 0) {
            a.table = d.embed; a.n_vocab = d.n_vocab; a.tokens = dc->tokens;
            rc = launch_gemv<WT, PRO_EMBED>(KT_DEC_GEMV_QKV, a, E + 2 * KV);
        } else {
            a.res_a = dc->hbuf; a.res_raw = dc->down_raw;
            rc = launch_gemv<WT, PRO_RESID>(KT_DEC_GEMV_QKV, a, E + 2 * KV);
        }
        if (rc) return rc;
        // ---- attention over the caches
        AttnArgs t{};
        t.step = dc->step; t.qkv_raw = dc->qkv_raw; t.kcache = (uint8_t*)L.kcache; t.vcache = (uint8_t*)L.vcache;
        t.kv_pitch = kv_pitch; t.scores = dc->scores; t.stats = dc->stats; t.att_part = dc->att_part; t.rope = dc->rope;
        t.adtype = d.adtype; t.n_heads = d.n_heads; t.n_kv = d.n_kv_heads; t.d_head = dh; t.max_ctx = d.max_ctx;
        t.n_chunks = dc->n_chunks; t.n_embd = E;
        const dim3 agrid(d.n_heads, dc->n_chunks);
        const size_t smem1 = (size_t)(16 + 3 * dh + 16) * 4 + 32 + (size_t)3 * dh + 64;
        GTR_LAUNCH(KT_DEC_ATTN_SCORE, k_dec_attn_score, agrid, dim3(256), smem1, t);
        GTR_LAUNCH(KT_DEC_ATTN_PV, k_dec_attn_pv, agrid, dim3(256), (size_t)2 * DEC_CHUNK * 4, t);
        // ---- output projection of the attention row
        GemvArgs o{};
        o.step = dc->step; o.adtype = d.adtype; o.d_in = E; o.m[0] = {L.wo, E}; o.n_mats = 1; o.out = dc->proj_raw;
        o.att_part = dc->att_part; o.n_heads = d.n_heads; o.d_head = dh; o.n_chunks = dc->n_chunks; o.rows_per_wave = 2;
        if ((rc = launch_gemv<WT, PRO_ATT>(KT_DEC_GEMV_O, o, E))) return rc;
        // ---- h = x + proj ; gate, up of norm(h)
        GemvArgs gu{};
        gu.step = dc->step; gu.adtype = d.adtype; gu.d_in = E;
        gu.m[0] = {L.wgate, F}; gu.m[1] = {L.wup, F}; gu.n_mats = 2; gu.out = dc->gu_raw;
        gu.res_a = dc->xbuf; gu.res_raw = dc->proj_raw; gu.x_out = dc->hbuf; gu.norm_w = (const uint16_t*)L.ffn_norm;
        gu.rows_per_wave = 4;
        if ((rc = launch_gemv<WT, PRO_RESID>(KT_DEC_GEMV_GATEUP, gu, 2 * F))) return rc;
        // ---- down( silu(gate) * up )
        GemvArgs dn{};
        dn.step = dc->step; dn.adtype = d.adtype; dn.d_in = F; dn.m[0] = {L.wdown, E}; dn.n_mats = 1; dn.out = dc->down_raw;
        dn.gate_raw = dc->gu_raw; dn.up_raw = dc->gu_raw + F; dn.rows_per_wave = 2;
        if ((rc = launch_gemv<WT, PRO_SILUMUL>(KT_DEC_GEMV_DOWN, dn, E))) return rc;
    }
    // ---- x = h + down ; logits = lm_head(norm(x))
    GemvArgs hd{};
    hd.step = dc->step; hd.adtype = d.adtype; hd.d_in = E; hd.m[0] = {d.lm_head, d.n_vocab}; hd.n_mats = 1; hd.out = d.logits;
    hd.res_a = dc->hbuf; hd.res_raw = dc->down_raw; hd.x_out = nullptr; hd.norm_w = (const uint16_t*)d.final_norm;
    hd.rows_per_wave = 8;
    if (int rc = launch_gemv<WT, PRO_RESID>(KT_DEC_GEMV_HEAD, hd, d.n_vocab)) return rc;
    GTR_LAUNCH(KT_DEC_ARGMAX, k_dec_argmax, dim3(1), dim3(1024), 0, (const float*)d.logits, d.n_vocab, dc->step, dc->result);
    return 0;
}

static int enqueue(gten_hip_decoder* dc)
{
    switch (dc->d.wdtype) {
    case GTEN_F16: return enqueue_step<GTEN_F16>(dc);
    case GTEN_Q8: return enqueue_step<GTEN_Q8>(dc);
    case GTEN_Q4: return enqueue_step<GTEN_Q4>(dc);
    }
    return fail(-4, "decoder: bad weight dtype %d", dc->d.wdtype);
}

extern "C" {

int gten_hip_decoder_create(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers, gten_hip_decoder** out)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(desc && layers && out, "decoder_create: null argument");
    const gten_hip_decoder_desc& d = *desc;
    GTR_REQUIRE(d.n_layers > 0 && d.n_heads > 0 && d.n_kv_heads > 0 && d.n_heads % d.n_kv_heads == 0, "decoder_create: bad head/layer counts");
    GTR_REQUIRE(d.n_embd % d.n_heads == 0, "decoder_create: n_embd %% n_heads != 0");
    const int dh = d.n_embd / d.n_heads;
    GTR_REQUIRE(dh == 32 || dh == 64, "decoder_create: fast path supports d_head 32 or 64 (got %d)", dh);
    GTR_REQUIRE(d.n_embd % 32 == 0 && d.n_ffn % 32 == 0 && d.n_embd <= 8192 && d.n_ffn <= 12288, "decoder_create: unsupported widths");
    GTR_REQUIRE(d.max_ctx > 0 && d.max_ctx <= GTEN_ROPE_MAX_POS, "decoder_create: max_ctx %d beyond the RoPE table", d.max_ctx);
    const bool pair_ok = (d.wdtype == GTEN_F16 && d.adtype == GTEN_F16) || ((d.wdtype == GTEN_Q8 || d.wdtype == GTEN_Q4) && d.adtype == GTEN_Q8);
    GTR_REQUIRE(pair_ok, "decoder_create: unsupported dtype pair (%d,%d) (tinyllama.cpp:258-265)", d.wdtype, d.adtype);
    GTR_REQUIRE(d.embed && d.final_norm && d.lm_head && d.logits, "decoder_create: null model pointer");
    auto* dc = new gten_hip_decoder;
    dc->d = d;
    dc->layers.assign(layers, layers + d.n_layers);
    dc->n_chunks = (d.max_ctx + DEC_CHUNK - 1) / DEC_CHUNK;
    const int E = d.n_embd, F = d.n_ffn, KV = dh * d.n_kv_heads;
    GTR_CHECK(hipMalloc((void**)&dc->step, sizeof(DecStep)));
    GTR_CHECK(hipMemset(dc->step, 0, sizeof(DecStep)));
    GTR_CHECK(hipMalloc((void**)&dc->tokens, (size_t)(d.max_ctx + 1) * 4));
    GTR_CHECK(hipMemset(dc->tokens, 0, (size_t)(d.max_ctx + 1) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->result, (size_t)(d.max_ctx + 2) * 4));
    GTR_CHECK(hipMemset(dc->result, 0, (size_t)(d.max_ctx + 2) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->qkv_raw, (size_t)(E + 2 * KV) * 4));
    GTR_CHECK(hipMalloc((void**)&dc->proj_raw, (size_t)E * 4));
    GTR_CHECK(hipMalloc((void**)&dc->gu_raw, (size_t)2 * F * 4));
    GTR_CHECK(hipMalloc((void**)&dc->down_raw, (size_t)E * 4));
    GTR_CHECK(hipMalloc((void**)&dc->scores, (size_t)d.n_heads * d.max_ctx * 4));
    GTR_CHECK(hipMalloc((void**)&dc->stats, (size_t)d.n_heads * dc->n_chunks * 2 * 4));
    GTR_CHECK(hipMalloc((void**)&dc->att_part, (size_t)d.n_heads * dc->n_chunks * dh * 4));
    GTR_CHECK(hipMalloc((void**)&dc->xbuf, gten_hip_row_bytes(d.adtype, E)));
    GTR_CHECK(hipMalloc((void**)&dc->hbuf, gten_hip_row_bytes(d.adtype, E)));
    if (int rc = rope_table(dh, &dc->rope)) { delete dc; return rc; }
    *out = dc;
    return 0;
}

int gten_hip_decoder_destroy(gten_hip_decoder* dc)
{
    if (!dc) return 0;
    GTR_NEED_INIT();
    GTR_CHECK(hipStreamSynchronize(stream()));
    if (dc->exec) hipGraphExecDestroy(dc->exec);
    if (dc->graph) hipGraphDestroy(dc->graph);
    void* bufs[] = {dc->step, dc->tokens, dc->result, dc->qkv_raw, dc->proj_raw, dc->gu_raw, dc->down_raw,
                    dc->scores, dc->stats, dc->att_part, dc->xbuf, dc->hbuf};
    for (void* b : bufs) if (b) hipFree(b);
    delete dc;
    return 0;
}

int gten_hip_decoder_set_tokens(gten_hip_decoder* dc, const int32_t* tokens_host, int first, int count)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && tokens_host && first >= 0 && count > 0 && first + count <= dc->d.max_ctx + 1, "decoder_set_tokens: bad range");
    GTR_CHECK(hipMemcpyAsync(dc->tokens + first, tokens_host, (size_t)count * 4, hipMemcpyHostToDevice, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    return 0;
}

int gten_hip_decoder_step(gten_hip_decoder* dc, int n, int use_graph)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && n >= 1 && n <= dc->d.max_ctx, "decoder_step: n=%d outside [1, %d]", n, dc ? dc->d.max_ctx : 0);
    GTR_CHECK(hipMemsetD32Async((hipDeviceptr_t)&dc->step->n, n, 1, stream()));
    if (!use_graph || prof_on()) return enqueue(dc);    // event pairs cannot be recorded into a capture
    if (!dc->exec) {
        GTR_CHECK(hipStreamBeginCapture(stream(), hipStreamCaptureModeThreadLocal));
        const int rc = enqueue(dc);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(stream(), &g);
        if (rc) { if (g) hipGraphDestroy(g); return rc; }
        GTR_CHECK(e);
        dc->graph = g;
        GTR_CHECK(hipGraphInstantiate(&dc->exec, dc->graph, nullptr, nullptr, 0));
    }
    GTR_CHECK(hipGraphLaunch(dc->exec, stream()));
    return 0;
}

int gten_hip_decoder_result(gten_hip_decoder* dc, int n, int32_t* argmax_host)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(dc && argmax_host && n >= 1 && n <= dc->d.max_ctx, "decoder_result: bad arguments");
    GTR_CHECK(hipMemcpyAsync(argmax_host, dc->result + n, 4, hipMemcpyDeviceToHost, stream()));
    GTR_CHECK(hipStreamSynchronize(stream()));
    return 0;
}

} // extern "C"
