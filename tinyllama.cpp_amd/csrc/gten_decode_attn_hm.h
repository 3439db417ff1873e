// gten_decode_attn_hm.h: decode attention of 16+ sequences on HEAD-MAJOR K / V shadows (round 5) -- part of the single-token
// decode translation unit: included by gten_decode.hip (which owns the includes, the LDS symbol, the launch macros and the
// host side).
//
// The reference keeps a K / V cache as rows [max_ctx][n_kv x 68 bytes] (gten/modules.cpp:188-201; a kv head's slice of a
// Q8 row = two 34-byte blocks, gten/quants.h:17-23), and so do the module tensors and every operator of this library.  A
// decode step of many sequences reads each cache once per step and little else: 92 % of the bytes of a 256-sequence step at
// ctx 2048.  A (kv head, 256-position chunk) workgroup reading 68-byte slices of 272-byte rows uses half of every
// 128-byte line it touches (round 4: 3.4-3.7 TB/s for the requests alone, against 5.1-5.5 TB/s for one contiguous run), and
// the slices then have to be re-laid in LDS before the matrix cores can take them.  A decoder of 16+ sequences (Q8
// activations, fast forms) therefore keeps a SHADOW of every sequence's caches, laid out for exactly this kernel:
//
//   shadow of one (sequence, layer, K | V):  [kv head][chunk of 256 positions][HM_CHUNK_BYTES = 17 408 = 256 x 68]
//   a K chunk   bytes [0, 16384)  the quants in MATRIX-OPERAND order: 16 tiles of 16 positions x 1 KiB; inside a tile lane
//               l = 16 lq + lc of a wave owns bytes [16 l, 16 l + 16) = elements 16 lq .. 16 lq + 15 of position 16 T + lc --
//               a tile is ONE coalesced 16-byte-per-lane load straight into the A operand of v_mfma_i32_16x16x64_i8
//               bytes [16384, 17408)  the block deltas (f16) as [block][lq][tile][4 positions]: a lane's 64 deltas are one
//               contiguous 128-byte run
//   a V chunk   bytes [0, 16384)  the quants TRANSPOSED and biased (q ^ 0x80), 8 steps of 32 positions x 2 KiB; lane l owns
//               16 bytes = two element tiles x 8 positions in the order the score tiles leave their probabilities in the
//               registers (slot j < 4: position 32 s + 4 lq + j, j >= 4: 32 s + 16 + 4 lq + j - 4): the A operand of
//               v_mfma_f32_16x16x32_f16 after a byte -> f16 expansion, no transpose through LDS
//               bytes [16384, 17408)  the block deltas as [lq][position pair][step][tile][position][half]: again one
//               128-byte run per lane
// The same 17 408 bytes per chunk as the cache rows hold, as ONE contiguous run per matrix.
//
// The row caches stay the truth: the shadows are filled from them (k_kv_import_hm) whenever a sequence (re)starts or
// anything wrote into its rows (gten_rt.h, kv_watch_*), the decode appends go to both, and every other path -- operators,
// the prompt kernels, decoders of up to 8 sequences, the exact forms, f16 activations -- reads the rows as before.
//
// k_dec_attn_hm: ONE WAVE per (sequence, kv head, chunk), no barrier against another wave, LDS only for the 8 head
// vectors of the group.  Scores as K . Q^T (positions on the rows): one v_mfma_i32_16x16x64_i8 per 16 positions gives, for
// the group's 8 heads, the exact integer dots of BOTH quant blocks at once -- columns 0..7 hold head j's block 0 (the B
// operand's lanes lq >= 2 are zero there), columns 8..15 head j's block 1 -- scaled (isum dq dk) and added across the
// column pair by one DPP rotate, exactly the two terms k_dec_attn_mm_g adds.  A lane then owns 2 of its tile's 4
// positions under one head: 32 scores per lane, every lane busy; maxima, sums of exponentials and the Q8 block maxima of
// the probabilities (32 positions along the context, gten/ops.h:996-997) are in-lane loops plus three cross-lane steps.
// The probabilities, rounded to Q8 and with the V row's block delta folded in (one fp16 rounding: k_dec_attn_mm_g's A
// operand), ARE the B operand of p.V as they stand (V^T x P^T, v_mfma_f32_16x16x32_f16): no LDS, no transpose.
// The new position never touches the chunk's registers: its score, probability and p.V term are formed from the chip's
// own K / V row beside the matrix instructions (its probability joins its Q8 block's maximum).
// Same chunk-local statistics and partials as k_dec_attn_mm_g (the consumer joins the chunks, PRO_ATTW); per (head,
// position) the same operations -- what differs is the order of the f32 additions in a chunk's sum of exponentials and in
// p.V (the matrix core's order over another assignment of positions to its steps).
// three waves per SIMD: the kernel fits 160 registers without a spill (hipcc takes 176 when left alone)
#ifndef HM_OCC
#define HM_OCC __attribute__((amdgpu_waves_per_eu(3)))
#endif
#define HM_CHUNK_BYTES 17408
#define HM_Q_BYTES 16384

// byte offsets inside a chunk: position p in [0, 256), element e in [0, 64), quant block / half in {0, 1}
__host__ __device__ __forceinline__ unsigned hm_k_q_off(unsigned p, unsigned e) { return (p >> 4) * 1024u + (((e >> 4) * 16u + (p & 15u)) * 16u) + (e & 15u); }
__host__ __device__ __forceinline__ unsigned hm_k_d_off(unsigned p, unsigned blk) { return HM_Q_BYTES + blk * 512u + ((p & 15u) >> 2) * 128u + (p >> 4) * 8u + (p & 3u) * 2u; }
__host__ __device__ __forceinline__ unsigned hm_v_q_off(unsigned p, unsigned e)
{
    const unsigned s = p >> 5, tp = (p >> 4) & 1u, r = p & 15u, et = e >> 4;
    return s * 2048u + (et >> 1) * 1024u + (((r >> 2) * 16u + (e & 15u)) * 16u) + (et & 1u) * 8u + 4u * tp + (r & 3u);
}
__host__ __device__ __forceinline__ unsigned hm_v_d_off(unsigned p, unsigned half)
{
    const unsigned s = p >> 5, tp = (p >> 4) & 1u, r = p & 15u;
    return HM_Q_BYTES + ((r >> 2) * 2u + ((r >> 1) & 1u)) * 128u + s * 16u + (tp * 4u + (r & 1u) * 2u + half) * 2u;
}

// ---- row caches -> shadows: one workgroup per (kv head, chunk) x (layer, K | V) x listed sequence.  Rows [0, n - 1) of a
// sequence at step n are its context (the step itself appends row n - 1); chunks without such a row return at once.
struct HmImportList { int n; int seq[512]; };

__global__ __launch_bounds__(256) void k_kv_import_hm(const HmImportList items, const DecStep* __restrict__ step, const void* const* __restrict__ kv_tab,
                                                      uint8_t* __restrict__ hm_base, size_t hm_seq_stride, size_t hm_cache_bytes, int n_layers, int n_kv,
                                                      int n_chunks, int max_ctx, size_t kv_pitch)
{
    const int seq = items.seq[blockIdx.z], layer = blockIdx.y >> 1, kv = blockIdx.y & 1;
    const int g = blockIdx.x % n_kv, chunk = blockIdx.x / n_kv, c0 = chunk * DEC_CHUNK;
    const int rows = step[seq].n - 1;                             // cached positions of the sequence
    if (c0 >= rows) return;
    const uint8_t* src = (const uint8_t*)kv_tab[((size_t)seq * n_layers + layer) * 2 + kv] + (size_t)g * 68;
    uint8_t* dst = hm_base + (size_t)seq * hm_seq_stride + (size_t)(layer * 2 + kv) * hm_cache_bytes + (size_t)(g * n_chunks + chunk) * HM_CHUNK_BYTES;
    unsigned* raw = (unsigned*)g_smem;                            // [256][17]: the slices as they lie in the cache
    {
        const int p = threadIdx.x, row = min(c0 + p, max_ctx - 1);
        const gmem_u32 s = as_global(src + (size_t)row * kv_pitch);
#pragma unroll
        for (int j = 0; j < 17; j++) raw[p * 17 + j] = s[j];
    }
    __syncthreads();
    const uint8_t* rb = (const uint8_t*)raw;
    // element e of a slice sits at byte 2 + e (block 0) or 4 + e (block 1: behind the second delta)
    auto elem = [&](unsigned p, unsigned e) -> unsigned { return rb[p * 68u + e + (e < 32u ? 2u : 4u)]; };
    for (unsigned q = threadIdx.x; q < 1024u; q += 256u) {
        unsigned w[4] = {0, 0, 0, 0};
        if (kv == 0) {
            // K piece q = 64 T + 16 lq + lc: elements 16 lq .. 16 lq + 15 of position 16 T + lc
            const unsigned T = q >> 6, lq = (q >> 4) & 3u, lc = q & 15u, p = 16u * T + lc;
#pragma unroll
            for (unsigned j = 0; j < 16; j++) w[j >> 2] |= elem(p, 16u * lq + j) << (8u * (j & 3u));
        } else {
            // V piece q = 128 s + 64 ep + 16 lq + lc: element tiles 2 ep, 2 ep + 1 at column lc, 8 positions each
            const unsigned s = q >> 7, ep = (q >> 6) & 1u, lq = (q >> 4) & 3u, lc = q & 15u;
#pragma unroll
            for (unsigned j = 0; j < 16; j++) {
                const unsigned et = 2u * ep + (j >> 3), slot = j & 7u, p = 32u * s + 16u * (slot >> 2) + 4u * lq + (slot & 3u);
                w[j >> 2] |= (elem(p, 16u * et + lc) ^ 0x80u) << (8u * (j & 3u));
            }
        }
        *(uint4*)(dst + (size_t)q * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    {
        const unsigned p = threadIdx.x;
        // (rows the sequence does not have yet hold whatever the cache holds: their deltas become 0, so that the attention kernel
        //  may multiply a zero probability by them without a select -- the appends of later steps write real, finite deltas)
        const bool known = c0 + (int)p < rows;
        const uint16_t d0 = known ? *(const uint16_t*)(rb + p * 68u) : (uint16_t)0, d1 = known ? *(const uint16_t*)(rb + p * 68u + 34u) : (uint16_t)0;
        if (kv == 0) { *(uint16_t*)(dst + hm_k_d_off(p, 0)) = d0; *(uint16_t*)(dst + hm_k_d_off(p, 1)) = d1; }
        else { *(uint16_t*)(dst + hm_v_d_off(p, 0)) = d0; *(uint16_t*)(dst + hm_v_d_off(p, 1)) = d1; }
    }
}

// ---- cross-lane steps of the one-wave kernel
__device__ __forceinline__ float hm_ror8(float v)               // the lane 8 columns away in its row of 16 (row_ror:8)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
}
__device__ __forceinline__ unsigned hm_ror8_u(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);
}
__device__ __forceinline__ float hm_rows_max(float v)           // over the four rows of 16 lanes (same column); every lane gets it
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float hm_rows_sum(float v)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// four biased bytes (q ^ 0x80) -> four exact f16 integers: 0x6400 | b is 1024 + (q + 128); minus 1152
typedef _Float16 hm_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void hm_bytes_to_h4(unsigned b4, unsigned& lo, unsigned& hi)
{
    const unsigned x0 = __builtin_amdgcn_perm(0x64646464u, b4, 0x04010400u);      // [b0, 0x64, b1, 0x64]
    const unsigned x1 = __builtin_amdgcn_perm(0x64646464u, b4, 0x04030402u);      // [b2, 0x64, b3, 0x64]
    const hm_h2 k = {(_Float16)1152.0f, (_Float16)1152.0f};
    lo = __builtin_bit_cast(unsigned, __builtin_bit_cast(hm_h2, x0) - k);
    hi = __builtin_bit_cast(unsigned, __builtin_bit_cast(hm_h2, x1) - k);
}

// Measured on the 256- / 64-sequence step (ms per step, two runs each, one box): waves per workgroup 4 | 2 and the
// K / V requests nontemporal or not -- 4: 3.145 3.155 / 1.506 1.514; 2: 3.201 3.192 / 1.501 1.497; 4 + nt: 3.136 3.106 / 1.539 1.538;
// 2 + nt: 3.105 3.110 / 1.516 1.511; one wave per workgroup (every wave prepares all eight head vectors) 3.153 / 1.515, two waves
// per SIMD instead of three 3.161 / 1.517.  So: pairs of chunks per workgroup, and the K / V stream -- read once per step, 6 GB
// at 256 sequences, far beyond the memory-side cache -- nontemporal from 128 rows per lane up (it then stops displacing the
// weights the other lane is about to read), default policy below.
// (Built and measured, round 5, not kept: ALL eight chunks of a (sequence, kv head) in one workgroup of eight waves, the partials and
//  statistics parked in LDS and joined on the chip with the consumer's arithmetic -- bit-identical logits, 1 MB of joined rows per
//  lane launch instead of 8.4 MB of partials written and read back -- 256 sequences 3.183 / 3.184 ms against 3.042 / 3.064 on the
//  same box, the launch 30.0 against 27.9 us; 64 sequences 1.500 against 1.488 ms.  One 512-thread workgroup per CU at 159 registers
//  runs its eight waves through request / softmax / p.V in lockstep and waits at the barrier for its slowest chunk, where six
//  independent pairs of waves per CU interleave those phases: the 17 MB per launch are cheaper than that.  HISTORY.md, round 5.)
#ifndef HM_WAVES
#define HM_WAVES 2            // waves per workgroup = consecutive chunks of one (sequence, kv head) that share the head vectors
#endif
#define HM_LD(p) (NT ? __builtin_nontemporal_load(p) : *(p))

template <int GRP, bool NT>
__global__ __launch_bounds__(64 * HM_WAVES) HM_OCC void k_dec_attn_hm(const AttnArgs a0, const int n_seq, const int n_cq)
{
    constexpr int dh = 64, NWV = HM_WAVES;
    // id -> (kv head, sequence, chunk quad): the live workgroups of short contexts (quad 0) are the first ids, spread over all XCDs.
    // Wave w of the workgroup owns chunk NWV cq + w; the group's head vectors are prepared ONCE per workgroup (wave w: heads w,
    // w + NWV, ...) and shared through LDS -- one barrier at the start, none afterwards.
    const int g = blockIdx.x % a0.n_kv, sci = blockIdx.x / a0.n_kv, seq = sci % n_seq, cq = sci / n_seq;
    const int w = threadIdx.x >> 6, t = threadIdx.x & 63, lc = t & 15, lq = t >> 4, hd = lc & 7, hi = lc >> 3;
    const int chunk = NWV * cq + w, c0 = chunk * DEC_CHUNK;
    const AttnArgs a = attn_for_seq(a0, seq);
    const int n = a.step->n, pos = n - 1;
    if (NWV * cq * DEC_CHUNK >= n) return;                        // (the whole workgroup)
    const bool alive = c0 < n;                                    // (uniform per wave)
    // cached positions of this chunk: [c0, c0 + len); position `pos` itself comes from the chip when it lies in this chunk
    const int len = min(DEC_CHUNK, pos - c0);
    const bool has_new = alive && pos < c0 + DEC_CHUNK;
    const int kv_dim = a.n_kv * dh;

    int8_t* qi8 = (int8_t*)g_smem;                                // [8][64] head vectors (shared)
    float* qd = (float*)(qi8 + 8 * dh);                           // [8][2]
    uint16_t* d16 = (uint16_t*)(qd + 16);                         // [8][4] halves
    int8_t* ki8 = (int8_t*)(d16 + 32) + w * 192;                  // per wave: the new K row [64], the new V row [64],
    int8_t* vi8 = ki8 + dh;                                       //           their deltas as floats [8] and halves [2][4]
    float* kd = (float*)(vi8 + dh);
    uint16_t* kvd16 = (uint16_t*)(kd + 8);

    // ---- requests: this wave's share of the raw projections, the rotation, then the chunk's K tiles and deltas (tiles
    //      past the context re-read the last live tile: no traffic, no branch around a request)
    constexpr int NJ = (GRP + NWV - 1) / NWV;
    float qraw[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) qraw[jj] = a.qkv_raw[(g * GRP + min(w + NWV * jj, GRP - 1)) * dh + t];
    float kraw = a.qkv_raw[a.n_embd + g * dh + t], vraw = a.qkv_raw[a.n_embd + kv_dim + g * dh + t];
    if (a.qkv_plane) {                                            // second K-split plane of the projections (uniform)
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw[jj] += a.qkv_raw[a.qkv_plane + (g * GRP + min(w + NWV * jj, GRP - 1)) * dh + t];
        kraw += a.qkv_raw[a.qkv_plane + a.n_embd + g * dh + t];
        vraw += a.qkv_raw[a.qkv_plane + a.n_embd + kv_dim + g * dh + t];
    }
    const float2 rot = a.rope_now[t & 31];
    const uint8_t* kc = a.hm_k + (size_t)(g * a.n_chunks + (alive ? chunk : 0)) * HM_CHUNK_BYTES;
    const uint8_t* vc = kc + a.hm_cache_bytes;
    const int Tl = max(len - 1, 0) >> 4, Sl = Tl >> 1;
    typedef int hm_v4i __attribute__((ext_vector_type(4)));
    typedef const hm_v4i __attribute__((address_space(1)))* gmem_v4i;
    hm_v4i ka[16], kdw[8];
    if (alive) {
        const gmem_v4i kq = (gmem_v4i)(uintptr_t)(kc + t * 16);
#pragma unroll
        for (int T = 0; T < 16; T++) ka[T] = HM_LD(kq + min(T, Tl) * 64);
        const gmem_v4i kdp = (gmem_v4i)(uintptr_t)(kc + HM_Q_BYTES + hi * 512 + lq * 128);
#pragma unroll
        for (int j = 0; j < 8; j++) kdw[j] = HM_LD(kdp + j);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- head vectors of the group (write -> rope -> write, gten/modules.cpp:196-201): this wave's share
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
        const int j = w + NWV * jj;
        if (j < GRP) head_prep_cs(qraw[jj], true, true, rot, dh, GTEN_Q8, qi8 + j * dh, qd + 2 * j, d16 + 4 * j);
    }
    if (has_new) {
        head_prep_cs(kraw, true, true, rot, dh, GTEN_Q8, ki8, kd, kvd16);
        head_prep_cs(vraw, true, false, rot, dh, GTEN_Q8, vi8, kd + 4, kvd16 + 4);
    }
    __syncthreads();
    if (!alive) return;
    // B operand: column lc = head hd's block hi -- its 16 bytes where the lane's K bytes belong to that block, else zero
    const bool bsel = ((hi == 0) == (lq < 2)) && hd < GRP;
    hm_v4i qb = *(const hm_v4i*)(qi8 + hd * dh + 16 * lq);
    {
        const hm_v4i z = {0, 0, 0, 0};
        qb = bsel ? qb : z;
    }
    const float dq = qd[2 * hd + hi];
    const unsigned pn = (unsigned)(pos - c0);                     // the new position inside the chunk (has_new)
    float scn = -INFINITY;                                        // its score under head hd
    if (has_new) {
        // appends: the cache rows (as every decode path leaves them) and the shadows
        {
            uint8_t* krow = a.kcache + (size_t)pos * a.kv_pitch + (size_t)g * 68, *vrow = a.vcache + (size_t)pos * a.kv_pitch + (size_t)g * 68;
            const unsigned kb = (uint8_t)ki8[t], vb = (uint8_t)vi8[t];
            store_global<uint8_t>(krow + (t >> 5) * GTEN_Q8_BYTES + 2 + (t & 31), (uint8_t)kb);
            store_global<uint8_t>(vrow + (t >> 5) * GTEN_Q8_BYTES + 2 + (t & 31), (uint8_t)vb);
            uint8_t* kcw = a.hm_k + (size_t)(g * a.n_chunks + chunk) * HM_CHUNK_BYTES;
            uint8_t* vcw = kcw + a.hm_cache_bytes;
            store_global<uint8_t>(kcw + hm_k_q_off(pn, (unsigned)t), (uint8_t)kb);
            store_global<uint8_t>(vcw + hm_v_q_off(pn, (unsigned)t), (uint8_t)(vb ^ 0x80u));
            if ((t & 31) == 0) {
                const unsigned b = (unsigned)t >> 5;
                const uint16_t kdl = kvd16[b], vdl = kvd16[4 + b];
                store_global<uint16_t>(krow + b * GTEN_Q8_BYTES, kdl);
                store_global<uint16_t>(vrow + b * GTEN_Q8_BYTES, vdl);
                store_global<uint16_t>(kcw + hm_k_d_off(pn, b), kdl);
                store_global<uint16_t>(vcw + hm_v_d_off(pn, b), vdl);
            }
        }
        // its score: every row of the A operand is the new K row, so every lane gets its column's block dot in place
        const hm_v4i kn = *(const hm_v4i*)(ki8 + 16 * lq);
        const hm_v4i z = {0, 0, 0, 0};
        const hm_v4i cn = __builtin_amdgcn_mfma_i32_16x16x64_i8(kn, qb, z, 0, 0, 0);
        const float tn = (float)cn[0] * (dq * kd[hi]);
        scn = (tn + hm_ror8(tn)) * 0.125f;                        // 1 / sqrt(64)
    }

    // ---- scores: lane (lc, lq) keeps positions 16 T + 4 lq + 2 hi + u (u = 0, 1) under head hd
    float sc[16][2];
#pragma unroll
    for (int T = 0; T < 16; T++) {
        const hm_v4i z = {0, 0, 0, 0};
        const hm_v4i c = __builtin_amdgcn_mfma_i32_16x16x64_i8(ka[T], qb, z, 0, 0, 0);
        // this lane's block deltas of positions 16 T + 4 lq + i
        const unsigned w0 = (unsigned)kdw[T >> 1][(T & 1) * 2], w1 = (unsigned)kdw[T >> 1][(T & 1) * 2 + 1];
        float term[4];
        term[0] = (float)c[0] * (dq * h2f((uint16_t)(w0 & 0xffffu)));
        term[1] = (float)c[1] * (dq * h2f((uint16_t)(w0 >> 16)));
        term[2] = (float)c[2] * (dq * h2f((uint16_t)(w1 & 0xffffu)));
        term[3] = (float)c[3] * (dq * h2f((uint16_t)(w1 >> 16)));
#pragma unroll
        for (int u = 0; u < 2; u++) {
            // the column 8 away holds the other block's term of the same (head, position): it needs mine of ITS positions
            const float mine = hi ? term[2 + u] : term[u], send = hi ? term[u] : term[2 + u];
            const float both = mine + hm_ror8(send);
            const bool live = 16 * T + 4 * lq + 2 * hi + u < len;
            sc[T][u] = live ? both * 0.125f : -INFINITY;
        }
    }
    // ---- the V chunk and its deltas are requested now (the K registers are free; their flight hides behind the softmax)
    hm_v4i va[8][2], vdw[8];
    {
        const gmem_v4i vq = (gmem_v4i)(uintptr_t)(vc + t * 16);
#pragma unroll
        for (int s = 0; s < 8; s++) { va[s][0] = HM_LD(vq + min(s, Sl) * 128); va[s][1] = HM_LD(vq + min(s, Sl) * 128 + 64); }
        const gmem_v4i vdp = (gmem_v4i)(uintptr_t)(vc + HM_Q_BYTES + (lq * 2 + hi) * 128);
#pragma unroll
        for (int s = 0; s < 8; s++) vdw[s] = HM_LD(vdp + s);
    }
    // ---- chunk maximum and sum of exponentials per head (hardware exponential, as k_dec_attn_mm_g)
    float M = scn;
#pragma unroll
    for (int T = 0; T < 16; T++) M = fmaxf(M, fmaxf(sc[T][0], sc[T][1]));
    M = fmaxf(M, hm_ror8(M));
    M = hm_rows_max(M);
    float L = 0.f;
#pragma unroll
    for (int T = 0; T < 16; T++) {
        sc[T][0] = __expf(sc[T][0] - M);                          // exp(-inf) = 0 for masked positions
        sc[T][1] = __expf(sc[T][1] - M);
        L += sc[T][0] + sc[T][1];
    }
    L += hm_ror8(L);
    L = hm_rows_sum(L);
    float en = 0.f;
    if (has_new) { en = __expf(scn - M); L += en; }
    if (lq == 0 && hi == 0 && hd < GRP)
        *(float2*)(a.stats + ((size_t)(g * GRP + hd) * a.n_chunks + chunk) * 2) = make_float2(M, L);
    const float rL = recip_rn(L);

    // ---- the probabilities and the Q8 scale of every block of 32 positions (tiles 2 s, 2 s + 1): the cross-lane steps of
    //      the eight blocks side by side (each is a chain of dependent moves with wait states between them)
    const int sn = (int)(pn >> 5);
    const float pnew = en * rL;
    float am[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        sc[2 * s][0] *= rL; sc[2 * s][1] *= rL; sc[2 * s + 1][0] *= rL; sc[2 * s + 1][1] *= rL;
        am[s] = fmaxf(fmaxf(sc[2 * s][0], sc[2 * s][1]), fmaxf(sc[2 * s + 1][0], sc[2 * s + 1][1]));
    }
#pragma unroll
    for (int s = 0; s < 8; s++) am[s] = fmaxf(am[s], hm_ror8(am[s]));
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(am[s]), __float_as_uint(am[s]), false, false);
        am[s] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(am[s]), __float_as_uint(am[s]), false, false);
        am[s] = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        if (has_new && s == sn) am[s] = fmaxf(am[s], pnew);      // (uniform) the new position's probability joins its block
    }
    float pnq = 0.f;                                              // the new position's probability as stored (Q8)
    att_f4 acc[4];
#pragma unroll
    for (int et = 0; et < 4; et++) acc[et] = att_f4{0.f, 0.f, 0.f, 0.f};
    // ---- per block: Q8 rounding, the V deltas folded in (one fp16 rounding), p.V.  A position past the context has
    //      probability exactly 0 and a FINITE delta (the import zeroes the deltas of rows it does not know, appends write real
    //      ones), so its operand is 0 without a select.
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const Q8Scale qs = q8_scale_from_absmax(am[s]);
        if (has_new && s == sn) pnq = (float)q8_round(pnew, qs.scale) * qs.ddeq;
        unsigned own[2][2];                                       // [tile][half]: f16 pairs (u = 0, 1)
#pragma unroll
        for (int tp = 0; tp < 2; tp++) {
            const unsigned d0 = (unsigned)vdw[s][2 * tp], d1 = (unsigned)vdw[s][2 * tp + 1];   // u = 0 | u = 1: (half 0, half 1)
            const float q0 = (float)q8_round(sc[2 * s + tp][0], qs.scale) * qs.ddeq, q1 = (float)q8_round(sc[2 * s + tp][1], qs.scale) * qs.ddeq;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float dv0 = h2f((uint16_t)(h ? d0 >> 16 : d0 & 0xffffu)), dv1 = h2f((uint16_t)(h ? d1 >> 16 : d1 & 0xffffu));
                const hm_h2 pr = {f2hv(q0 * dv0), f2hv(q1 * dv1)};
                own[tp][h] = __builtin_bit_cast(unsigned, pr);
            }
        }
        // B operand of half h for the head columns (hi = 0): slots {own, the column 8 away} per tile
        att_h8 bp[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned wv[4] = {own[0][h], hm_ror8_u(own[0][h]), own[1][h], hm_ror8_u(own[1][h])};
            __builtin_memcpy(&bp[h], wv, 16);
        }
#pragma unroll
        for (int et = 0; et < 4; et++) {
            const unsigned b0 = (unsigned)va[s][et >> 1][(et & 1) * 2], b1 = (unsigned)va[s][et >> 1][(et & 1) * 2 + 1];
            unsigned wv[4];
            hm_bytes_to_h4(b0, wv[0], wv[1]);
            hm_bytes_to_h4(b1, wv[2], wv[3]);
            att_h8 av;
            __builtin_memcpy(&av, wv, 16);
            acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bp[et >> 1], acc[et], 0, 0, 0);
        }
    }
    // ---- the new position's term from the chip: out[head][e] += f16(p dv[half]) v[e]
    if (has_new) {
#pragma unroll
        for (int et = 0; et < 4; et++) {
            const float pv = (float)f2hv(pnq * h2f(kvd16[4 + (et >> 1)]));
            const unsigned vb = *(const unsigned*)(vi8 + 16 * et + 4 * lq);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[et][i] += pv * (float)(int8_t)((vb >> (8 * i)) & 0xffu);
        }
    }
    // C layout: column lc = head, rows e = 16 et + 4 lq + i
    if (hi == 0 && hd < GRP) {
        float* o = a.att_part + ((size_t)(g * GRP + hd) * a.n_chunks + chunk) * dh + 4 * lq;
#pragma unroll
        for (int et = 0; et < 4; et++) *(float4*)(o + 16 * et) = make_float4(acc[et][0], acc[et][1], acc[et][2], acc[et][3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// f16 activations (the f16 configuration's wide decoders; round 5).  The same idea with nothing to dequantize: a shadow chunk
// holds 256 positions x 64 halfs = 32 768 bytes per matrix, K as the A operands of v_mfma_f32_16x16x32_f16 (16 tiles x 2 steps
// of 32 elements: lane l owns the 8 halfs [16 l, 16 l + 16) of a (tile, step)), V TRANSPOSED as the A operands of the p.V
// products (8 steps of 32 positions x 4 element tiles; slot order as in the Q8 shadow).  One wave per (sequence, kv head, chunk):
// scores = K . Q^T (f16 products are exact in f32; the matrix core adds them), columns 0 .. GRP - 1 = the group's heads; a lane of
// a head column owns 4 positions per tile, so chunk maximum, sum of exponentials and the fp16 rounding of the probabilities
// (gten/ops.h:996-997 stores them as f16) are in-lane loops + two cross-lane steps, and the rounded probabilities are the B
// operand of p.V as they stand.  Chunk-local statistics, joined by the consumer (PRO_ATTW) -- the single-sequence f16 step's
// scheme instead of the row-global rounding of the two-launch pair this replaces.
#define HMF_CHUNK_BYTES 32768
__host__ __device__ __forceinline__ unsigned hmf_k_off(unsigned p, unsigned e)
{
    return (p >> 4) * 2048u + (e >> 5) * 1024u + (((((e & 31u) >> 3) * 16u) + (p & 15u)) * 16u) + (e & 7u) * 2u;
}
__host__ __device__ __forceinline__ unsigned hmf_v_off(unsigned p, unsigned e)
{
    return (p >> 5) * 4096u + (e >> 4) * 1024u + (((((p & 15u) >> 2) * 16u) + (e & 15u)) * 16u) + (4u * ((p >> 4) & 1u) + (p & 3u)) * 2u;
}

__global__ __launch_bounds__(256) void k_kv_import_hm_f16(const HmImportList items, const DecStep* __restrict__ step, const void* const* __restrict__ kv_tab,
                                                          uint8_t* __restrict__ hm_base, size_t hm_seq_stride, size_t hm_cache_bytes, int n_layers,
                                                          int n_kv, int n_chunks, int max_ctx, size_t kv_pitch)
{
    const int seq = items.seq[blockIdx.z], layer = blockIdx.y >> 1, kv = blockIdx.y & 1;
    const int g = blockIdx.x % n_kv, chunk = blockIdx.x / n_kv, c0 = chunk * DEC_CHUNK;
    const int rows = step[seq].n - 1;
    if (c0 >= rows) return;
    const uint8_t* src = (const uint8_t*)kv_tab[((size_t)seq * n_layers + layer) * 2 + kv] + (size_t)g * 128;
    uint8_t* dst = hm_base + (size_t)seq * hm_seq_stride + (size_t)(layer * 2 + kv) * hm_cache_bytes + (size_t)(g * n_chunks + chunk) * HMF_CHUNK_BYTES;
    unsigned* raw = (unsigned*)g_smem;                            // [256][33]: the slices (32 dwords) as they lie in the cache, padded
    {
        const int p = threadIdx.x, row = min(c0 + p, max_ctx - 1);
        const gmem_u32 s = as_global(src + (size_t)row * kv_pitch);
        const bool known = c0 + p < rows;                         // (rows the sequence does not have yet become zeros: V must stay finite)
#pragma unroll
        for (int j = 0; j < 32; j++) { const unsigned v = s[j]; raw[p * 33 + j] = known ? v : 0u; }
    }
    __syncthreads();
    const uint16_t* rh = (const uint16_t*)raw;
    auto elem = [&](unsigned p, unsigned e) -> unsigned { return rh[p * 66u + e]; };
    for (unsigned q = threadIdx.x; q < 2048u; q += 256u) {
        unsigned w[4] = {0, 0, 0, 0};
        if (kv == 0) {
            // K piece q = 128 T + 64 ks + 16 lq + lc: elements 32 ks + 8 lq .. + 7 of position 16 T + lc
            const unsigned T = q >> 7, ks = (q >> 6) & 1u, lq = (q >> 4) & 3u, lc = q & 15u, p = 16u * T + lc;
#pragma unroll
            for (unsigned j = 0; j < 8; j++) w[j >> 1] |= elem(p, 32u * ks + 8u * lq + j) << (16u * (j & 1u));
        } else {
            // V piece q = 256 s + 64 et + 16 lq + lc: element 16 et + lc at 8 positions
            const unsigned s = q >> 8, et = (q >> 6) & 3u, lq = (q >> 4) & 3u, lc = q & 15u;
#pragma unroll
            for (unsigned j = 0; j < 8; j++) {
                const unsigned p = 32u * s + 16u * (j >> 2) + 4u * lq + (j & 3u);
                w[j >> 1] |= elem(p, 16u * et + lc) << (16u * (j & 1u));
            }
        }
        *(uint4*)(dst + (size_t)q * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

template <int GRP, bool NT>
__global__ __launch_bounds__(64 * HM_WAVES) void k_dec_attn_hm_f16(const AttnArgs a0, const int n_seq, const int n_cq)
{
    constexpr int dh = 64, NWV = HM_WAVES;
    const int g = blockIdx.x % a0.n_kv, sci = blockIdx.x / a0.n_kv, seq = sci % n_seq, cq = sci / n_seq;
    const int w = threadIdx.x >> 6, t = threadIdx.x & 63, lc = t & 15, lq = t >> 4;
    const int chunk = NWV * cq + w, c0 = chunk * DEC_CHUNK;
    const AttnArgs a = attn_for_seq(a0, seq);
    const int n = a.step->n, pos = n - 1;
    if (NWV * cq * DEC_CHUNK >= n) return;
    const bool alive = c0 < n;
    const int len = min(DEC_CHUNK, pos - c0);
    const bool has_new = alive && pos < c0 + DEC_CHUNK;
    const int kv_dim = a.n_kv * dh;

    uint16_t* qh = (uint16_t*)g_smem;                             // [8][64] head vectors as f16 (shared)
    uint16_t* kh = qh + 8 * dh + w * 128;                         // per wave: the new K row [64], the new V row [64]
    uint16_t* vh = kh + dh;
    // head_prep_cs's Q8 outputs are unused for f16: a scratch corner they may write nothing into (act gates the stores on Q8 only)
    int8_t* nul8 = (int8_t*)(qh + 8 * dh + NWV * 128);
    float* nulf = (float*)(nul8 + 64);
    uint16_t* nulh = (uint16_t*)(nulf + 2);

    constexpr int NJ = (GRP + NWV - 1) / NWV;
    float qraw[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) qraw[jj] = a.qkv_raw[(g * GRP + min(w + NWV * jj, GRP - 1)) * dh + t];
    float kraw = a.qkv_raw[a.n_embd + g * dh + t], vraw = a.qkv_raw[a.n_embd + kv_dim + g * dh + t];
    if (a.qkv_plane) {
#pragma unroll
        for (int jj = 0; jj < NJ; jj++) qraw[jj] += a.qkv_raw[a.qkv_plane + (g * GRP + min(w + NWV * jj, GRP - 1)) * dh + t];
        kraw += a.qkv_raw[a.qkv_plane + a.n_embd + g * dh + t];
        vraw += a.qkv_raw[a.qkv_plane + a.n_embd + kv_dim + g * dh + t];
        if (a.qkv_nplanes == 4) {                                 // (uniform) four planes of k_dec_wxp_f16: 2 and 3 behind the first two, in order
            float qx[2][NJ], kx[2], vx[2];
#pragma unroll
            for (int q = 2; q < 4; q++) {
                const float* pl = a.qkv_raw + (size_t)q * a.qkv_plane;
#pragma unroll
                for (int jj = 0; jj < NJ; jj++) qx[q - 2][jj] = pl[(g * GRP + min(w + NWV * jj, GRP - 1)) * dh + t];
                kx[q - 2] = pl[a.n_embd + g * dh + t];
                vx[q - 2] = pl[a.n_embd + kv_dim + g * dh + t];
            }
#pragma unroll
            for (int q = 2; q < 4; q++) {
#pragma unroll
                for (int jj = 0; jj < NJ; jj++) qraw[jj] += qx[q - 2][jj];
                kraw += kx[q - 2];
                vraw += vx[q - 2];
            }
        }
    }
    const float2 rot = a.rope_now[t & 31];
    const uint8_t* kc = a.hm_k + (size_t)(g * a.n_chunks + (alive ? chunk : 0)) * HMF_CHUNK_BYTES;
    const uint8_t* vc = kc + a.hm_cache_bytes;
    const int Tl = max(len - 1, 0) >> 4, Sl = Tl >> 1;
    typedef int hm_v4i __attribute__((ext_vector_type(4)));
    typedef const hm_v4i __attribute__((address_space(1)))* gmem_v4i;
    hm_v4i ka[16][2];
    const gmem_v4i kq = (gmem_v4i)(uintptr_t)(kc + t * 16);
    if (alive) {
#pragma unroll
        for (int T = 0; T < 8; T++) { ka[T][0] = HM_LD(kq + min(T, Tl) * 128); ka[T][1] = HM_LD(kq + min(T, Tl) * 128 + 64); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < NJ; jj++) {
        const int j = w + NWV * jj;
        if (j < GRP) {
            const float v = head_prep_cs(qraw[jj], true, true, rot, dh, GTEN_F16, nul8, nulf, nulh);
            qh[j * dh + t] = f2h(v);
        }
    }
    float vnew = 0.f;
    if (has_new) {
        const float kf = head_prep_cs(kraw, true, true, rot, dh, GTEN_F16, nul8, nulf, nulh);
        vnew = head_prep_cs(vraw, true, false, rot, dh, GTEN_F16, nul8, nulf, nulh);
        kh[t] = f2h(kf);
        vh[t] = f2h(vnew);
    }
    __syncthreads();
    if (!alive) return;
    // B operand: column lc = head lc (lc < GRP), two steps of 32 elements
    att_h8 qb[2];
    {
        const hm_v4i z = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const hm_v4i v = *(const hm_v4i*)(qh + min(lc, GRP - 1) * dh + 32 * ks + 8 * lq);
            qb[ks] = __builtin_bit_cast(att_h8, lc < GRP ? v : z);
        }
    }
    const unsigned pn = (unsigned)(pos - c0);
    float scn = -INFINITY;
    if (has_new) {
        uint8_t* krow = a.kcache + (size_t)pos * a.kv_pitch + (size_t)g * 128, *vrow = a.vcache + (size_t)pos * a.kv_pitch + (size_t)g * 128;
        const uint16_t kb = kh[t], vb = vh[t];
        store_global<uint16_t>(krow + 2 * t, kb);
        store_global<uint16_t>(vrow + 2 * t, vb);
        uint8_t* kcw = a.hm_k + (size_t)(g * a.n_chunks + chunk) * HMF_CHUNK_BYTES;
        store_global<uint16_t>(kcw + hmf_k_off(pn, (unsigned)t), kb);
        store_global<uint16_t>(kcw + a.hm_cache_bytes + hmf_v_off(pn, (unsigned)t), vb);
        att_f4 cn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const att_h8 kn = __builtin_bit_cast(att_h8, *(const hm_v4i*)(kh + 32 * ks + 8 * lq));     // every row of A is the new K row
            cn = __builtin_amdgcn_mfma_f32_16x16x32_f16(kn, qb[ks], cn, 0, 0, 0);
        }
        scn = cn[0] * 0.125f;
    }
    // ---- scores: lane (lc, lq) holds positions 16 T + 4 lq + i under head lc
    float sc[16][4];
#pragma unroll
    for (int T = 0; T < 16; T++) {
        if (T == 8) {
#pragma unroll
            for (int U = 8; U < 16; U++) { ka[U][0] = HM_LD(kq + min(U, Tl) * 128); ka[U][1] = HM_LD(kq + min(U, Tl) * 128 + 64); }
        }
        att_f4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(att_h8, ka[T][0]), qb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(att_h8, ka[T][1]), qb[1], c, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) sc[T][i] = (16 * T + 4 * lq + i < len) ? c[i] * 0.125f : -INFINITY;
    }
    // ---- the V chunk (the K registers are free)
    hm_v4i va[8][4];
    {
        const gmem_v4i vq = (gmem_v4i)(uintptr_t)(vc + t * 16);
#pragma unroll
        for (int s = 0; s < 8; s++)
#pragma unroll
            for (int et = 0; et < 4; et++) va[s][et] = HM_LD(vq + min(s, Sl) * 256 + et * 64);
    }
    float M = scn;
#pragma unroll
    for (int T = 0; T < 16; T++) M = fmaxf(M, fmaxf(fmaxf(sc[T][0], sc[T][1]), fmaxf(sc[T][2], sc[T][3])));
    M = hm_rows_max(M);
    float L = 0.f;
#pragma unroll
    for (int T = 0; T < 16; T++)
#pragma unroll
        for (int i = 0; i < 4; i++) { sc[T][i] = __expf(sc[T][i] - M); L += sc[T][i]; }
    L = hm_rows_sum(L);
    float en = 0.f;
    if (has_new) { en = __expf(scn - M); L += en; }
    if (lq == 0 && lc < GRP)
        *(float2*)(a.stats + ((size_t)(g * GRP + lc) * a.n_chunks + chunk) * 2) = make_float2(M, L);
    const float rL = recip_rn(L);
    // ---- p.V: the probabilities rounded to f16 (as the reference stores them) are the B operand as they stand
    att_f4 acc[4];
#pragma unroll
    for (int et = 0; et < 4; et++) acc[et] = att_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; s++) {
        att_h8 bp;
#pragma unroll
        for (int j = 0; j < 8; j++) bp[j] = f2hv(sc[2 * s + (j >> 2)][j & 3] * rL);
#pragma unroll
        for (int et = 0; et < 4; et++) acc[et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(att_h8, va[s][et]), bp, acc[et], 0, 0, 0);
    }
    if (has_new) {
        const float pv = (float)f2hv(en * rL);
#pragma unroll
        for (int et = 0; et < 4; et++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[et][i] += pv * h2f(vh[16 * et + 4 * lq + i]);
    }
    if (lc < GRP) {
        float* o = a.att_part + ((size_t)(g * GRP + lc) * a.n_chunks + chunk) * dh + 4 * lq;
#pragma unroll
        for (int et = 0; et < 4; et++) *(float4*)(o + 16 * et) = make_float4(acc[et][0], acc[et][1], acc[et][2], acc[et][3]);
    }
}
