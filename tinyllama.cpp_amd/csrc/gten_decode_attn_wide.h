// gten_decode_attn_wide.h: attention of MANY sequences (GQA-grouped kernels: the VALU pair, the f16 matrix-core scores, the grouped
// one-launch form of 8 sequences, and k_dec_attn_mm_g -- the whole attention of a (sequence, chunk, kv head) on the matrix cores) -- part of the single-token decode translation unit: included by gten_decode.hip (which owns the includes, the LDS
// symbol, the launch macros and the host side).  Split out in round 4; the code is unchanged.
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
    // (16-byte alignment by OFFSET arithmetic on g_smem: through an integer the pointer would come back as a generic one and every access
    //  behind it would be a FLAT instruction)
    float* qfa = (float*)(g_smem + (((size_t)((uint8_t*)(vi8 + dh) - g_smem) + 15) & ~(size_t)15));   // f16 activations: [GRP][64] q values, then the new k row [64]
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
    uint16_t* qh = (uint16_t*)(g_smem + (((size_t)((uint8_t*)(qi8 + 64) - g_smem) + 15) & ~(size_t)15));   // [16][64] f16: the group's q vectors, zero rows beyond (offset arithmetic: see k_dec_attn_score_g)
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
    // (16-byte alignment by OFFSET arithmetic on g_smem: through an integer the pointer would come back as a generic one and every access
    //  behind it would be a FLAT instruction)
    float* qfa = (float*)(g_smem + (((size_t)((uint8_t*)(vi8 + dh) - g_smem) + 15) & ~(size_t)15));   // f16 activations: [GRP][64] q values, then the new k row [64]
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
//   * the K and the V chunk are requested at entry exactly as they lie in the cache (256 positions x 17 dwords each; five
//     16-byte requests per thread and matrix, five neighbouring lanes per row: see the requests below); K is parked in LDS at
//     once, V stays in registers until the scores are done (its arrival hides behind them) and then takes K's place in LDS;
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
// compute phase into scratch, and a wait for a scratch reload is a wait for every older request -- the prefetch itself.
// Round 4, without scratch: a workgroup walking 2 / 4 / 8 consecutive chunks of its (sequence, kv head) with the next chunk's
// requests in flight during the current one's arithmetic, and the other experiments of DESIGN.md 3.4 (5) -- none faster.)
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

    // (28 KB per workgroup, five workgroups per CU: the V chunk waits in registers until the scores are done and then takes
    //  K's place; with a region of its own -- 36 KB, four per CU -- 256 sequences decoded at 68.9 k instead of 71.0 k tok/s,
    //  64 at 37.0 k instead of 37.3 k.  SIX per CU -- 27 024 bytes with the new row's scratch over the reduction buffer and the
    //  zero rows of the A operand supplied by the lanes, 78 VGPRs with the pieces' coordinates recomputed where used, no spill,
    //  bit-identical -- measured 3.70 against 3.61 ms per step at 256 sequences, 1.762 against 1.717 at 64: not kept)
    unsigned* kl = (unsigned*)g_smem;                             // [256][17]: the chunk's K slices as they lie in the cache; dead after
    unsigned* vl = kl;                                            // the scores: the V chunk [256][17] takes their place
    _Float16* pl = (_Float16*)(kl + DEC_CHUNK * NW);              // the probability rows [2 halves][8 heads][PP]
    int8_t* qi8 = (int8_t*)(pl + 2 * 8 * PP);                     // [16][64], rows GRP..15 zero
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
    // (round 4: 16-byte requests, FIVE per thread and matrix instead of seventeen dword ones -- the texture path spends ~25 cycles
    //  per wave instruction whatever its width.  Piece q = tid + 256 k, k < 5: row q / 5, part q % 5 = bytes 16 part .. + 15 of the
    //  row's 68-byte slice; the fifth part holds the slice's last dword and 12 bytes of whatever follows (the next head's slice or
    //  the next row; for the cache's very last row the piece starts 12 bytes early instead).  Five neighbouring lanes fetch one
    //  row's slice in ONE instruction: with four pieces per row and the last dwords as a fifth request of their own -- 256 rows x
    //  4 bytes out of 256 different lines, long after those lines were first touched -- the requests alone took 21.1 / 39.9 us
    //  per launch at 64 / 128 sequences against 18.2 / 33.5 us this way.  A slice is 4-byte aligned only, which gfx9 global loads take.)
    typedef unsigned u4u __attribute__((ext_vector_type(4), aligned(4)));
    u4u kw[5], vw[5];
    int prow[5], ppart[5];
    bool pback[5];
    {
        const gmem_u32 kbase = as_global(a.kcache + (size_t)g * head_bytes), vbase = as_global(a.vcache + (size_t)g * head_bytes);
        const unsigned pitch_w = (unsigned)(a.kv_pitch >> 2);
        const int last = a.max_ctx - 1 - c0;
        unsigned off[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int q = (int)threadIdx.x + 256 * k;
            prow[k] = q / 5; ppart[k] = q - 5 * prow[k];
            pback[k] = ppart[k] == 4 && prow[k] >= last;
            off[k] = (unsigned)(c0 + min(prow[k], last)) * pitch_w + (unsigned)(4 * ppart[k]) - (pback[k] ? 3u : 0u);
        }
        // (all of K first, then all of V: requests return in order, and the scores must not wait for the V chunk)
#pragma unroll
        for (int k = 0; k < 5; k++) kw[k] = *(const __attribute__((address_space(1))) u4u*)(kbase + off[k]);
#pragma unroll
        for (int k = 0; k < 5; k++) vw[k] = *(const __attribute__((address_space(1))) u4u*)(vbase + off[k]);
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
    // the K chunk goes to LDS now ([row][17 dwords], as the slices lie in the cache); the V chunk stays in its registers, in
    // flight, until the scores are done
#pragma unroll
    for (int k = 0; k < 5; k++) {
        unsigned* dst = kl + prow[k] * NW + 4 * ppart[k];
        dst[0] = pback[k] ? kw[k].w : kw[k].x;
        if (ppart[k] < 4) { dst[1] = kw[k].y; dst[2] = kw[k].z; dst[3] = kw[k].w; }
    }
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
        const int newrow = has_new ? pos - c0 : -1;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            unsigned* dst = vl + prow[k] * NW + 4 * ppart[k];
            const unsigned* nw = vnew + 4 * ppart[k];
            const bool isnew = prow[k] == newrow;
            dst[0] = isnew ? nw[0] : (pback[k] ? vw[k].w : vw[k].x);
            if (ppart[k] < 4) { dst[1] = isnew ? nw[1] : vw[k].y; dst[2] = isnew ? nw[2] : vw[k].z; dst[3] = isnew ? nw[3] : vw[k].w; }
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
