// gten_decode_attn.h: attention of the single-sequence step and of 2 .. 8 sequences (generic d_head kernels, the one-launch
// d_head 64 kernel k_dec_attn_one64 that ships) and the argmax kernel -- part of the single-token decode translation unit: included by gten_decode.hip (which owns the includes, the LDS
// symbol, the launch macros and the host side).  Split out in round 4; the code is unchanged.
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
    int qkv_nplanes;              // k_dec_attn_hm_f16 only: 4 planes qkv_plane floats apart (k_dec_wxp_f16 at WXP_QKV_PLANES), else 0: one or two
    // head-major shadows of the caches (gten_decode_attn_hm.h; decoders of 16+ sequences): this layer's K shadow of the lane's
    // first sequence, bytes from one sequence's shadows to the next, bytes of one shadow (the V shadow follows the K shadow)
    uint8_t* hm_k; size_t hm_seq_stride, hm_cache_bytes;
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
    t.hm_k = a.hm_k + (size_t)seq * a.hm_seq_stride;
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
