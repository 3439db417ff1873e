// gten_decode_attn_exact64.h: the TWO-launch d_head 64 attention (k_dec_attn_score64 / k_dec_attn_pv64): row-global rounding points of the
// probabilities, only behind gten_hip_set_decode_exact(1) -- part of the single-token decode translation unit: included by gten_decode.hip (which owns the includes, the LDS
// symbol, the launch macros and the host side).  Split out in round 4; the code is unchanged.
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
