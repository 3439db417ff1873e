// gten_decode_persist.h -- the batch-1 decode step as ONE persistent launch (round 4).  Included by gten_decode.hip.
//
// The fused step of gten_decode.hip is 113 dependent launches of 3-6 us, each of which starts its weight requests only
// after the launch before it has drained: the chip idles between them.  Here the whole step --
// TinyLlama::logits for one new row (tinyllama.cpp:45-61; op order gten/modules.cpp:193-254) plus the greedy argmax
// (tinyllama.cpp:416-424) -- is one launch of one 512-thread workgroup per CU:
//
//   * every workgroup owns a fixed share of every W.x (rows of q|k|v, o, down, lm_head; one 32-wide FFN slice of
//     gate|up) and one (head, 256-position chunk) of the attention; its weight rows travel HBM -> registers and are
//     requested a whole transformer block AHEAD of the dependency chain (the registers of a matrix are refilled with
//     the next block's rows as soon as its dot products are done), so no dependency edge waits for the weight stream;
//   * an activation vector crosses the chip as 8-byte {value, tag} GRANULES: the producer workgroup gathers its few
//     results in LDS and ONE wave instruction stores them write-through (sc1) -- 64-80 contiguous bytes, the data is
//     the flag -- and every consumer thread polls exactly the granules its prologue arithmetic starts from with sc1
//     loads (thread t: elements 4t .. 4t+3, the register layout of k_dec_gemv8's prologue).  tools/microbench_edge.hip
//     measured this edge at 1.35-1.45 us against 3.0-3.9 us for the same phase as its own launch; eight separate
//     8-byte stores per workgroup instead of one 64-byte store cost 3.1 us, polling all partials of the attention
//     join from every consumer 12 us (hence the small join phase B2 below);
//   * tag = epoch * 256 + layer * 8 + phase + 1, the epoch a device word that the step's last workgroup bumps: tags
//     never repeat, nothing is zeroed between steps, and a graph replays the launch unchanged;
//   * every spin is bounded: a stalled poll sets the abort word, every other poll sees it and the grid drains.
//
// Phases of a block (all 256 workgroups take part in every all-to-all edge):
//   A   poll down_raw(l-1) [l = 0: embedding row]  -> x = A(h + A(down)) -> RMSNorm -> Q8 stage -> q|k|v rows -> publish
//   B   (head, chunk) workgroup: poll its head's q (and the new k, v)   -> k_dec_attn_one64's arithmetic  -> publish
//       chunk-local partial o_c, (m_c, l_c)
//   B2  the 8 workgroups of a head join 8 elements each (PRO_ATTW's arithmetic)                           -> publish
//   C   poll the joined attention row -> Q8 stage -> o rows                                               -> publish
//   D   poll proj_raw -> h = A(x + A(proj)) -> RMSNorm -> stage -> slice workgroups: gate|up rows, silu.mul -> publish Q8
//   E   poll the FFN activation -> down rows                                                              -> publish
// then  poll down_raw -> final RMSNorm -> lm_head rows (logits, plain stores) -> per-workgroup best -> publish;
//       workgroup 0 polls the 256 candidates, takes the first maximum, stores the id and advances the step word.
//
// Arithmetic, rounding points and reduction trees are those of k_dec_gemv8 / k_dec_attn_one64 / k_dec_argmax, so the
// step's bytes (K / V rows, logits, ids) are those of the launch chain (tests/test_decode_gpu.py).
//
// Residency: the grid (one workgroup per CU) must be co-resident.  The kernel holds 89 728 bytes of LDS (persist_smem()) and
// 512 threads, so ONE of its workgroups fits a CU and its grid takes every CU: a second persistent decoder on another stream
// could never be scheduled beside it, so the library keeps one per device (persist_prepare: the next decoder runs the launch
// chain) and other kernels only delay it.  gten_hip_set_decode_persistent(0) keeps the launch chain.

typedef unsigned long long pu64;
typedef __attribute__((address_space(1))) pu64 pgu64;
typedef __attribute__((address_space(1))) unsigned pgu32;

struct PLayer {
    const uint8_t *wq, *wk, *wv, *wo, *wgate, *wup, *wdown;
    const uint16_t *attn_norm, *ffn_norm;
    uint8_t *kcache, *vcache;
};


enum { PL_WQ = 0, PL_WK, PL_WV, PL_WO, PL_WGATE, PL_WUP, PL_WDOWN, PL_ANORM, PL_FNORM, PL_KC, PL_VC, PL_N };
static_assert(sizeof(PLayer) == PL_N * 8, "PLayer is a row of eleven pointers");

struct PArgs {
    const PLayer* layers;
    DecStep* step;
    int32_t* tokens;
    int32_t* result;
    const uint8_t* embed;
    const uint16_t* final_norm;
    const uint8_t* lm_head;
    float* logits;
    const float2* rope;
    unsigned* ctl;                 // [0] epoch, [1] abort code (sticky until the host clears it)
    pu64 *gq, *gpart, *gatt, *gproj, *gact, *gdown, *gbest;
    unsigned* stamps;              // optional [n_layers * 8 + 8]: workgroup 0's phase stamps (s_memrealtime), else null
    int n_layers, E, F, KV, n_heads, n_kv, n_vocab, max_ctx, n_chunks, kv_pitch;
    int rpa, rpo, rph, rw;         // rows per workgroup: q|k|v, o / down, lm_head; lm_head rows per wave
};

#define PERSIST_NT 512
#define PERSIST_MAX_LAYERS 64
#define PERSIST_SPIN_LIMIT (1u << 21)

// a value the optimiser cannot see through: address arithmetic derived from it stays where it is used instead of being
// hoisted out of the layer loop and held (or spilled) across it
__device__ __forceinline__ int p_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// ... and one that also waits for `dep`: a request formed from it cannot be scheduled ahead of the arithmetic that
// consumed the registers it refills
__device__ __forceinline__ int p_opaque_dep(int v, float dep) { asm volatile("" : "+v"(v) : "v"(dep)); return v; }

// ---- bounded polling
__device__ __forceinline__ bool p_give_up(unsigned& spins, unsigned* ctl, unsigned code, int lazy = 0)
{
    __builtin_amdgcn_s_sleep(1);
    for (int z = 0; z < lazy; z++) __builtin_amdgcn_s_sleep(8);       // (a workgroup with nothing to do in the producing phase: ~0.2 us per unit)
    ++spins;
    if ((spins & 63u) == 0 && __hip_atomic_load((pgu32*)(ctl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    if (spins > PERSIST_SPIN_LIMIT) {
        __hip_atomic_store((pgu32*)(ctl + 1), code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

// N consecutive granules per thread (threads with !act take no part); false = the step was aborted.
// Four granules = 32 aligned bytes = TWO 16-byte sc1 loads: the texture path spends ~25 cycles per wave INSTRUCTION whatever
// its width, and while a phase waits every wave of the chip is polling -- half the instructions, half the queue in front
// of the loads and stores that matter (hipcc has no 16-byte atomic load: inline assembly, the wait inside the statement).
typedef unsigned p_u4v __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ bool p_poll(const pu64* g, unsigned idx, unsigned tag, bool act, unsigned (&val)[N], unsigned* ctl, unsigned code, int lazy = 0)
{
    const pgu64* p = (const pgu64*)(uintptr_t)g + idx;              // (uniform base + 32-bit lane index)
    for (unsigned spins = 0;;) {
        bool ok = true;
        if constexpr (N == 4) {
            p_u4v x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
            if (act) {
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(x0), "=&v"(x1) : "v"(p) : "memory");
                ok = (x0[1] == tag) & (x0[3] == tag) & (x1[1] == tag) & (x1[3] == tag);
                val[0] = x0[0]; val[1] = x0[2]; val[2] = x1[0]; val[3] = x1[2];
            }
        } else {
            pu64 x[N];
            if (act) {
#pragma unroll
                for (int i = 0; i < N; i++) x[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < N; i++) { ok = ok & ((unsigned)(x[i] >> 32) == tag); val[i] = (unsigned)x[i]; }
            }
        }
        if (__all(ok)) return true;
        if (p_give_up(spins, ctl, code, lazy)) return false;
    }
}

__device__ __forceinline__ void p_store(pu64* g, unsigned idx, unsigned tag, unsigned val)
{
    __hip_atomic_store((pgu64*)(uintptr_t)g + idx, ((pu64)tag << 32) | (pu64)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// head_prep_cs (gten_decode.hip) for Q8 activations with the lane passed in: write(raw) -> rope -> write for one head
// vector of 64 elements held by the 64 lanes of a wave (gten/modules.cpp:196-201 + gten/ops.h:714-755)
__device__ __forceinline__ float p_head_prep(float raw, bool do_rope, const float2 cs, int t, int8_t* qi8, float* qd, uint16_t* qd16)
{
    float v = raw;
    {
        const Q8Scale sc = q8_scale_from_absmax(nn_max32(fabsf(v)));
        v = (float)q8_round(v, sc.scale) * sc.ddeq;
    }
    if (do_rope) {
        const float other = __shfl_xor(v, 32, 64);
        const bool lo = (t & 32) == 0;
        const float x0 = lo ? v : other, x1 = lo ? other : v;
        v = lo ? (x0 * cs.x - x1 * cs.y) : (x0 * cs.y + x1 * cs.x);
    }
    const Q8Scale sc = q8_scale_from_absmax(nn_max32(fabsf(v)));
    const int qv = q8_round(v, sc.scale);
    qi8[t] = (int8_t)qv;
    if ((t & 31) == 0) { qd[t >> 5] = sc.ddeq; qd16[t >> 5] = sc.d16; }
    return (float)qv * sc.ddeq;
}

// one lane's share of a weight row in flight / in registers (lane = K block of a pass): quants + delta
template <int WT> struct PRow { uint4 w0; uint16_t wd; };
template <> struct PRow<GTEN_Q8> { uint4 w0, w1; uint16_t wd; };
// (the table's pointers are only known as generic ones: say "global", or every request becomes a FLAT load -- which also
// counts on the LDS counter and returns out of order; the row base is wave-uniform, the lane adds a 32-bit byte offset)
typedef const __attribute__((address_space(1))) uint8_t* pg_u8;
__device__ __forceinline__ uint4 p_ld16(pg_u8 row, unsigned off)
{
    const gt_u4v v = __builtin_nontemporal_load((const __attribute__((address_space(1))) gt_u4v*)(row + off));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint16_t p_ld2(pg_u8 row, unsigned off)
{
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) uint16_t*)(row + off));
}
template <int WT>
__device__ __forceinline__ void p_ld_row(const uint8_t* qbase, int rows_m, int nb, int lr, int c, int lane, PRow<WT>& r)
{
    constexpr int WB = (WT == GTEN_Q4) ? 16 : 32;
    const unsigned b = (c * 64 + lane < nb) ? (unsigned)(c * 64 + lane) : 0u;
    const pg_u8 gq = (pg_u8)(uintptr_t)qbase;
    const pg_u8 qrow = gq + (size_t)lr * nb * WB;                              // scalar arithmetic: lr is wave-uniform
    const pg_u8 drow = gq + (size_t)rows_m * nb * WB + (size_t)lr * nb * 2;
    if constexpr (WT == GTEN_Q4) {
        r.w0 = p_ld16(qrow, b * 16u);
    } else {
        r.w0 = p_ld16(qrow, b * 16u);
        r.w1 = p_ld16(qrow, ((unsigned)nb + b) * 16u);
    }
    r.wd = p_ld2(drow, b * 2u);
}

// this lane's activation block of pass c from an LDS stage
struct PAct { int av[8]; int asum; float ad; };
__device__ __forceinline__ PAct p_act(const ActQ8& s, int nb, int c, int lane)
{
    PAct r;
    const int b = c * 64 + lane;
    const bool in = b < nb;
    const int bs = in ? b : 0;
    const int4* ap = (const int4*)(s.q + (size_t)bs * 32);
    const int4 a0 = ap[0], a1 = ap[1];
    r.av[0] = a0.x; r.av[1] = a0.y; r.av[2] = a0.z; r.av[3] = a0.w;
    r.av[4] = a1.x; r.av[5] = a1.y; r.av[6] = a1.z; r.av[7] = a1.w;
    r.ad = in ? s.d[bs] : 0.f;
    r.asum = in ? s.sum[bs] : 0;
    return r;
}
template <int WT>
__device__ __forceinline__ float p_term(const PAct& x, const PRow<WT>& r)
{
    int isum;
    if constexpr (WT == GTEN_Q4) isum = dot_q8_q4_block(x.av, x.asum, r.w0);
    else isum = dot_q8_q8_block(x.av, r.w0, r.w1);
    return (float)isum * (x.ad * h2f(r.wd));
}

template <int WT>
__global__ __launch_bounds__(PERSIST_NT) void k_dec_persist(const PArgs a)
{
    constexpr int EPT = 4, LPB = 8, NW = 8;
    constexpr int dh = 64, NWD = 17;                     // d_head; dwords per Q8 kv-head slice
    // (the wave index as a scalar: every row base below is then an SGPR pair and a request is base + 32-bit lane offset)
    const int lane0 = threadIdx.x & 63, lane = lane0, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), b = blockIdx.x;
    const int E = a.E, nb = E >> 5, F = a.F, nbf = F >> 5, KV = a.KV;
    const int NC = a.n_chunks;
    unsigned* ctl = a.ctl;
    if (__hip_atomic_load((pgu32*)(ctl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // an earlier step aborted
    const unsigned ep = __hip_atomic_load((pgu32*)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n = a.step->n, pos = n - 1;
#define PTAG(l, ph) (ep * 256u + (unsigned)((l) * 8 + (ph) + 1))
#define PSTAMP(i) do { if (a.stamps && b == 0 && threadIdx.x == 0) a.stamps[i] = (unsigned)__builtin_amdgcn_s_memrealtime(); } while (0)

    // ---- LDS
    float* red = (float*)g_smem;                         // 16 floats
    float* pubv = red + 16;                              // 64 words: a workgroup's results meet here before ONE store
    float* bestv = pubv + 64;                            // 16 + 16 words (lm_head candidates)
    int* besti = (int*)(bestv + 16);
    ActQ8 sE = actq8_carve(g_smem + 1024, nb);           // <= 64 blocks: 2560 bytes
    ActQ8 sF = actq8_carve(g_smem + 1024 + 2560, nbf);   // <= 192 blocks: 7680 bytes
    uint8_t* att = g_smem + 1024 + 2560 + 7680;          // 1152 + 2048 + 17408 bytes (k_dec_attn_one64's layout)
    float* a_red = (float*)att;                          // 16
    float* qf = a_red + 16 + dh;
    float* kf = qf + dh;
    float* qd = kf + dh;
    float* kd = qd + 8;
    uint16_t* d16 = (uint16_t*)(kd + 8);
    int8_t* qi8 = (int8_t*)(d16 + 16);
    int8_t* ki8 = qi8 + dh;
    int8_t* vi8 = ki8 + dh;
    float* pr_l = (float*)(att + 1152);                  // 256 probabilities
    float* part = pr_l + DEC_CHUNK;                      // 256
    // the chunk's V and K slices, row-major as they lie in the caches, TWO of each: while waves 0-3 work on this block's (buffer
    // l & 1), waves 4-7 -- idle through the attention phase -- fetch the next block's into the other one
    unsigned* vl0 = (unsigned*)(part + DEC_CHUNK);       // [2][256 x 17 dwords]
    unsigned* kl0 = vl0 + 2 * DEC_CHUNK * NWD;
    pu64* tabl = (pu64*)(kl0 + 2 * DEC_CHUNK * NWD);          // the per-layer pointer table (PLayer x n_layers), copied once: a pointer is then an
                                                         // LDS read (~0.1 us) instead of a scalar-cache miss per block and matrix (~1 us, measured)

    for (int i = threadIdx.x; i < a.n_layers * PL_N; i += PERSIST_NT) tabl[i] = ((const pu64*)a.layers)[i];
    __syncthreads();
    // field k of layer l as a wave-uniform pointer
    auto tab = [&](int l, int k) -> const uint8_t* {
        const pu64 v = tabl[l * PL_N + k];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const uint8_t*)(uintptr_t)(((pu64)hi << 32) | lo);
    };
    const int gi = threadIdx.x, base = gi * EPT, blk = gi / LPB, sub = gi % LPB;
    const bool on = base < E;
    const int sbase = on ? base : 0;

    // ---- the attention item of this workgroup
    const bool att_item = b < a.n_heads * NC;
    const int chunk = b % NC, h = att_item ? b / NC : 0, c0 = chunk * DEC_CHUNK;
    const int grp = a.n_heads / a.n_kv, g = h / grp;
    const bool att_on = att_item && c0 < n;
    const bool has_new = att_on && pos >= c0 && pos < c0 + DEC_CHUNK;
    const bool writer = has_new && (h == g * grp);
    const int nch = (n + DEC_CHUNK - 1) / DEC_CHUNK;
    const int JE = dh / NC;                              // elements of its head a workgroup joins (phase B2)
    const float2 rot = a.rope[(size_t)pos * (dh / 2) + (lane & 31)];

    // ---- row shares
    const int RQ = E + 2 * KV;
    const bool slice = b < nbf;
    const uint8_t* lmq = a.lm_head;

    // ---- register sets of the weight stream
    PRow<WT> wA[2];                                      // q|k|v: rows b*rpa + wid + 8j
    PRow<WT> wO;                                         // o: row b*rpo + wid
    PRow<WT> wG[8];                                      // gate|up slice rows (later: lm_head rows)
    PRow<WT> wD[3];                                      // down: row b*rpo + wid, three passes over K
    unsigned kw[NWD], vw[NWD];                           // the attention item's K rows / V chunk

    auto issue_qkv = [&](int Ll, bool real, float dep) {
        const int lane = p_opaque_dep(lane0, dep);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int r = b * a.rpa + wid + 8 * j;
            r = (real && wid + 8 * j < a.rpa && r < RQ) ? r : 0;
            const uint8_t* qb = tab(Ll, PL_WQ); int rows_m = E, lr = r;
            if (r >= E + KV) { qb = tab(Ll, PL_WV); rows_m = KV; lr = r - E - KV; }
            else if (r >= E) { qb = tab(Ll, PL_WK); rows_m = KV; lr = r - E; }
            p_ld_row<WT>(qb, rows_m, nb, lr, 0, lane, wA[j]);
        }
    };
    auto issue_o = [&](int Ll, bool real, float dep) {
        const int lane = p_opaque_dep(lane0, dep);
        const int r = real ? min(b * a.rpo + wid, E - 1) : 0;
        p_ld_row<WT>(tab(Ll, PL_WO), E, nb, r, 0, lane, wO);
    };
    auto issue_gu = [&](int Ll, bool real, float dep) {
        const int lane = p_opaque_dep(lane0, dep);
        const uint8_t* qb = (wid < 4) ? tab(Ll, PL_WGATE) : tab(Ll, PL_WUP);
        const int r0 = (real ? b * 32 + (wid & 3) * 8 : 0), rs = real ? 1 : 0;      // (not a slice workgroup: one dummy row)
#pragma unroll
        for (int j = 0; j < 8; j++) p_ld_row<WT>(qb, F, nb, r0 + j * rs, 0, lane, wG[j]);
    };
    auto issue_down = [&](int Ll, bool real, float dep) {
        const int lane = p_opaque_dep(lane0, dep);
        const int r = real ? min(b * a.rpo + wid, E - 1) : 0;
#pragma unroll
        for (int c = 0; c < 3; c++) p_ld_row<WT>(tab(Ll, PL_WDOWN), E, nbf, r, real ? c : 0, lane, wD[c]);
    };
    // requests [k0, k1) of the 17 + 17 that fetch a block's K and V chunks (waves 4-7; both chunks as coalesced dwords, exactly
    // as they lie in the cache: a thread's OWN K row -- 17 dwords at a pitch of 272 bytes -- would be 64 different lines per wave
    // instruction).  The texture path takes ~25 cycles per wave instruction, i.e. ~1.5 us for all of them: they are issued in
    // four pieces BETWEEN the barriers of the attention phase, so that no barrier waits for the issuing waves.
    auto issue_kv = [&](int Ll, int k0, int k1) {
        const int tid = p_opaque((int)threadIdx.x) - 256;
        const unsigned pitch_w = (unsigned)a.kv_pitch >> 2;
        const gmem_u32 kbase = as_global(tab(Ll, PL_KC) + (size_t)g * (2 * GTEN_Q8_BYTES));
        const gmem_u32 vbase = as_global(tab(Ll, PL_VC) + (size_t)g * (2 * GTEN_Q8_BYTES));
        const int last = a.max_ctx - 1 - c0;
#pragma unroll
        for (int k = 0; k < NWD; k++) {
            if (k < k0 || k >= k1) continue;
            const int d = tid + 256 * k, row = d / NWD, w = d - row * NWD;     // dword d of the chunk
            const unsigned off = (unsigned)(c0 + min(row, last)) * pitch_w + (unsigned)w;
            kw[k] = kbase[off];
            vw[k] = vbase[off];
        }
    };
    auto park_kv = [&](int buf) {                         // waves 4-7: the requested chunks into LDS buffer `buf`
        const int t2 = (int)threadIdx.x - 256;
#pragma unroll
        for (int k = 0; k < NWD; k++) { vl0[buf * DEC_CHUNK * NWD + t2 + k * 256] = vw[k]; kl0[buf * DEC_CHUNK * NWD + t2 + k * 256] = kw[k]; }
    };
    // lm_head rows of this wave, batch jb (8 rows): row index clamped for the request, discarded later
    auto lm_row = [&](int jb, int j) { return b * a.rph + wid * a.rw + jb * 8 + j; };
    auto issue_lm = [&](int jb, bool real, float dep) {
        const int lane = p_opaque_dep(lane0, dep);
#pragma unroll
        for (int j = 0; j < 8; j++) p_ld_row<WT>(lmq, a.n_vocab, nb, real ? min(lm_row(jb, j), a.n_vocab - 1) : 0, 0, lane, wG[j]);
    };

    // ---- first requests: the embedding row of layer 0's prologue, then block 0's weights
    unsigned emb[2] = {0, 0};
    float emb_delta = 0.f;
    unsigned nw0[2];
    {
        const int tok = a.tokens[pos];
        const int byte0 = (sub * EPT) & 15;
        const uint8_t* src;
        const uint16_t* dsp;
        if (WT == GTEN_Q4) {
            src = a.embed + ((size_t)tok * nb + (on ? blk : 0)) * 16 + byte0;
            dsp = (const uint16_t*)(a.embed + (size_t)a.n_vocab * nb * 16);
        } else {
            src = a.embed + (size_t)tok * nb * 32 + (size_t)((sub * EPT) >> 4) * nb * 16 + (size_t)(on ? blk : 0) * 16 + byte0;
            dsp = (const uint16_t*)(a.embed + (size_t)a.n_vocab * nb * 32);
        }
        emb[0] = *(const unsigned*)src;
        emb_delta = h2f(dsp[(size_t)tok * nb + (on ? blk : 0)]);
        const uint2 t = *(const uint2*)(a.layers[0].attn_norm + sbase);   // (before the table is used: a plain load)
        nw0[0] = t.x; nw0[1] = t.y;
    }
    {
        issue_qkv(0, true, 0.f);
        if (att_on && wid >= 4) { issue_kv(0, 0, NWD); park_kv(0); }      // block 0's K / V chunks (the barriers of phase A order them)
    }
    __builtin_amdgcn_sched_barrier(0);
    PSTAMP(0);

    float x[EPT], hres[EPT];                              // residual rows (exact storage values), this thread's elements
#pragma unroll
    for (int i = 0; i < EPT; i++) { x[i] = 0.f; hres[i] = 0.f; }

    // RMSNorm of v with weights nwv, then the row staged as Q8 in sE (k_dec_gemv8 prologue, PRO_EMBED / PRO_RESID)
    auto norm_stage = [&](float (&v)[EPT], const unsigned (&nwv)[2]) {
        float ss = on ? sumsq_treeN<EPT>(v) : 0.f;
        ss = block_sum_tree_n<NW, false>(ss, red);
        const float ms = ((E & (E - 1)) == 0) ? __builtin_ldexpf(ss, -__builtin_ctz(E)) : ss / (float)E;
        const float inv = recip_rn(sqrtf(ms) + 1e-6f);
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            const uint16_t hw = (uint16_t)((i & 1) ? (nwv[i >> 1] >> 16) : (nwv[i >> 1] & 0xffffu));
            v[i] = v[i] * inv * h2f(hw);
        }
        if (on) q8_stageN<EPT>(v, blk, sub, sE);
        __syncthreads();
    };
    // a workgroup's `count` results (pubv[0 .. count)) leave as ONE store instruction
    auto publish = [&](pu64* dst, int count, unsigned tag) {
        __syncthreads();
        if (wid == 0) {
            const int ln = p_opaque(lane0);
            if (ln < count) p_store(dst, (unsigned)ln, tag, __float_as_uint(pubv[ln]));
        }
    };

    for (int l = 0; l < a.n_layers; l++) {
        const int Lc = l;
        // (thread-derived indices are re-derived from an opaque copy in every block: see p_opaque)
        const int gi = p_opaque((int)threadIdx.x), base = gi * EPT, blk = gi / LPB, sub = gi % LPB;
        const bool on = base < E;
        const int sbase = on ? base : 0;
        const int lane = gi & 63;
        const bool more = l + 1 < a.n_layers;
        const int Ln = l + (more ? 1 : 0);

        // ================= A: residual + RMSNorm + q|k|v rows
        {
            float v[EPT];
            unsigned nwv[2];
            if (l == 0) {
#pragma unroll
                for (int i = 0; i < EPT; i++) {
                    const unsigned byte = (emb[i >> 2] >> ((i & 3) * 8)) & 0xffu;
                    if (WT == GTEN_Q4) v[i] = (float)((((sub * EPT) < 16) ? (int)(byte >> 4) : (int)(byte & 0x0fu)) - 7) * emb_delta;
                    else v[i] = (float)(int)(int8_t)byte * emb_delta;
                }
                if (WT == GTEN_Q4) q8_roundN<EPT>(v);
                nwv[0] = nw0[0]; nwv[1] = nw0[1];
            } else {
                const uint2 t = *(const uint2*)((const uint16_t*)tab(Lc, PL_ANORM) + sbase);
                nwv[0] = t.x; nwv[1] = t.y;
                unsigned raw[EPT];
                if (!p_poll<EPT>(a.gdown, (unsigned)sbase, PTAG(l - 1, 5), on, raw, ctl, 0x100u + l)) return;
#pragma unroll
                for (int i = 0; i < EPT; i++) v[i] = __uint_as_float(raw[i]);
                q8_roundN<EPT>(v);                                    // Linear output written in the activation dtype
#pragma unroll
                for (int i = 0; i < EPT; i++) v[i] = hres[i] + v[i];
                q8_roundN<EPT>(v);                                    // Residual output written
            }
            PSTAMP(l * 20 + 1);
            // this block's K rows / V chunk, o and down rows: requested now, consumed one to four phases on -- everything but
            // the q|k|v rows is requested and consumed inside ONE iteration of the block loop, so no in-flight register
            // crosses the loop's back edge (the compiler would copy it there, behind a wait for the request)
            issue_o(Lc, true, v[0]);
            issue_down(Lc, true, v[0]);
            PSTAMP(l * 20 + 2);
#pragma unroll
            for (int i = 0; i < EPT; i++) x[i] = v[i];
            norm_stage(v, nwv);
            PSTAMP(l * 20 + 3);
            const PAct xa = p_act(sE, nb, 0, p_opaque(lane0));
            float dep = 0.f;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const float acc = wave_sum(p_term<WT>(xa, wA[j]));
                if (lane == 0) pubv[wid + 8 * j] = acc;
                dep += acc;
            }
            PSTAMP(l * 20 + 4);
            if (more) issue_qkv(Ln, true, dep);
            publish(a.gq + (size_t)b * a.rpa, min(a.rpa, RQ - b * a.rpa), PTAG(l, 0));
        }
        PSTAMP(l * 20 + 5);

        // ================= B: attention of (head h, chunk) -- k_dec_attn_one64<GTEN_Q8, false>
#ifndef PERSIST_NO_ATT
        float sc = -INFINITY;
        if (att_on) {
            const int tid = p_opaque((int)threadIdx.x);
            const int t = tid & 63, pw = wid;
            float vnew = 0.f;
            if (pw == 0 || (has_new && pw < 3)) {
                const int roff = (pw == 1) ? E + g * dh : (pw == 2) ? E + KV + g * dh : h * dh;
                unsigned rawu[1];
                if (!p_poll<1>(a.gq, (unsigned)(roff + t), PTAG(l, 0), true, rawu, ctl, 0x200u + l)) return;
                int8_t* dq = (pw == 0) ? qi8 : (pw == 1) ? ki8 : vi8;
                float* dd = (pw == 0) ? qd : (pw == 1) ? kd : kd + 4;
                const float v = p_head_prep(__uint_as_float(rawu[0]), pw != 2, rot, t, dq, dd, d16 + 4 * pw);
                if (pw == 0) qf[t] = v;
                if (pw == 1) kf[t] = v;
                if (pw == 2) vnew = v;
                if (pw >= 1 && writer) {
                    uint8_t* row = (uint8_t*)tab(Lc, (pw == 1) ? PL_KC : PL_VC) + (size_t)pos * a.kv_pitch + (size_t)g * (2 * GTEN_Q8_BYTES);
                    uint8_t* bk = row + (size_t)(t >> 5) * GTEN_Q8_BYTES;
                    store_global<uint8_t>(bk + 2 + (t & 31), (uint8_t)dq[t]);
                    if ((t & 31) == 0) store_global<uint16_t>(bk, d16[4 * pw + (t >> 5)]);
                }
            }
            PSTAMP(l * 20 + 6);
            const unsigned* kl = kl0 + (l & 1) * DEC_CHUNK * NWD;
            unsigned* vl = vl0 + (l & 1) * DEC_CHUNK * NWD;
            __syncthreads();
            // waves 4-7: the NEXT block's chunks are requested now and parked after the phase's last barrier (below)
            if (more && tid >= 256) issue_kv(Ln, 0, 4);
            const int c = c0 + (int)tid;
            if (tid < 256) {
#pragma unroll
                for (int j = 0; j < NWD; j++) kw[j] = kl[tid * NWD + j];          // this position's K slice (17 is odd: no bank conflict)
                const float scale = 1.0f / sqrtf((float)dh);
                float acc = 0.f;
                const int* qi = (const int*)qi8;
                int isum = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) isum = dot4(qi[j], (int)__builtin_amdgcn_alignbit(kw[j + 1], kw[j], 16), isum);
                acc += (float)isum * (qd[0] * h2f((uint16_t)(kw[0] & 0xffffu)));
                isum = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) isum = dot4(qi[8 + j], (int)kw[9 + j], isum);
                acc += (float)isum * (qd[1] * h2f((uint16_t)(kw[8] >> 16)));
                if (has_new) {
                    float accn = 0.f;
                    const int* ki = (const int*)ki8;
#pragma unroll
                    for (int bb = 0; bb < 2; bb++) {
                        int is2 = 0;
#pragma unroll
                        for (int j = 0; j < 8; j++) is2 = dot4(qi[bb * 8 + j], ki[bb * 8 + j], is2);
                        accn += (float)is2 * (qd[bb] * kd[bb]);
                    }
                    if (c == pos) acc = accn;
                    if (pw == 2) {
                        uint8_t* vrow = (uint8_t*)vl + (size_t)(pos - c0) * (NWD * 4);
                        vrow[(t >> 5) * GTEN_Q8_BYTES + 2 + (t & 31)] = (uint8_t)vi8[t];
                        if ((t & 31) == 0) *(uint16_t*)(vrow + (t >> 5) * GTEN_Q8_BYTES) = d16[8 + (t >> 5)];
                    }
                }
                sc = (c < n) ? acc * scale : -INFINITY;
            }
        }
        // the K / V registers are free: THIS block's gate|up slice rows take their place (needed three phases on); one
        // unconditional request site (workgroups without a slice: a dummy row)
        PSTAMP(l * 20 + 7);
        if (slice) issue_gu(Lc, true, sc);
        if (att_on) {
            const int tid = p_opaque((int)threadIdx.x);
            const int t = tid & 63;
            const int c = c0 + (int)tid;
            const unsigned* vl = vl0 + (l & 1) * DEC_CHUNK * NWD;
            // chunk maximum and sum of exponentials (block_max_n<4> / block_sum_n<4> of waves 0-3)
            {
                const float wm = wave_max_dpp(sc);
                if (lane == 0 && wid < 4) a_red[wid] = wm;
            }
            __syncthreads();
            if (more && tid >= 256) issue_kv(Ln, 4, 8);
            float mx, sm;
            {
                const float4 m4 = *(const float4*)a_red;
                mx = fmaxf(fmaxf(fmaxf(m4.x, m4.y), m4.z), m4.w);
            }
            const float ex = (tid < 256 && c < n) ? expf(sc - mx) : 0.f;
            {
                const float ws = wave_sum(ex);
                if (lane == 0 && wid < 4) a_red[4 + wid] = ws;
            }
            __syncthreads();
            if (more && tid >= 256) issue_kv(Ln, 8, 12);
            PSTAMP(l * 20 + 18);
            {
                const float4 s4 = *(const float4*)(a_red + 4);
                sm = 0.f;
                sm += s4.x; sm += s4.y; sm += s4.z; sm += s4.w;
            }
            if (tid < 256) {
                float pr = (c < n) ? ex / sm : 0.f;
                const Q8Scale s8 = q8_scale_from_absmax(max32(fabsf(pr)));
                if (c < n) pr = (float)q8_round(pr, s8.scale) * s8.ddeq;
                pr_l[tid] = pr;
            }
            __syncthreads();
            if (more && tid >= 256) issue_kv(Ln, 12, NWD);
            if (tid < 256) {
                const int len = min(DEC_CHUNK, n - c0);
                const int e = tid & 63, cg = tid >> 6;
                const uint8_t* vb = (const uint8_t*)vl;
                float o = 0.f;
                const int qoff = (e < 32) ? 2 + e : 36 + (e - 32), doff = (e < 32) ? 0 : 34;
                if (len == DEC_CHUNK) {
                    // (k_dec_attn_one64's sum, term by term in the same order; the operands are fetched four terms ahead)
                    float pp[2][4];
                    int qv[2][4];
                    unsigned dv[2][4];
                    auto fetch = [&](int r, int slot) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint8_t* row = vb + (size_t)(cg + 4 * (4 * r + u)) * 68;
                            pp[slot][u] = pr_l[cg + 4 * (4 * r + u)];
                            qv[slot][u] = (int)(int8_t)row[qoff];
                            dv[slot][u] = *(const uint16_t*)(row + doff);
                        }
                    };
                    fetch(0, 0);
#pragma unroll
                    for (int r = 0; r < DEC_CHUNK / 16; r++) {
                        if (r + 1 < DEC_CHUNK / 16) fetch(r + 1, (r + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 4; u++) o += pp[r & 1][u] * ((float)qv[r & 1][u] * h2f((uint16_t)dv[r & 1][u]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll 8
                    for (int cl = cg; cl < len; cl += 4) {
                        const uint8_t* row = vb + (size_t)cl * 68;
                        o += pr_l[cl] * ((float)(int8_t)row[qoff] * h2f(*(const uint16_t*)(row + doff)));
                    }
                }
                part[tid] = o;
            }
            PSTAMP(l * 20 + 19);
            __syncthreads();
            if (more && tid >= 256) park_kv((l + 1) & 1);           // (the data has had the whole phase to arrive; nobody waits for these waves)
            // publish o_c[64] and (m_c, l_c): wave 0, two store instructions (512 + 16 contiguous bytes)
            if (wid == 0) {
                float r = 0.f;
                for (int q = 0; q < 4; q++) r += part[q * dh + t];
                pu64* dst = a.gpart + (size_t)(h * NC + chunk) * 66;
                p_store(dst, (unsigned)t, PTAG(l, 1), __float_as_uint(r));
                if (t < 2) p_store(dst, 64u + (unsigned)t, PTAG(l, 1), __float_as_uint(t == 0 ? mx : sm));
            }
        }
#else
        if (slice) issue_gu(Lc, true, 0.f);
#endif
        PSTAMP(l * 20 + 8);

        // ================= B2: join the chunks of JE elements of head h (PRO_ATTW's arithmetic), wave 0
        if (att_item && wid == 0) {
            const int j = lane / JE, i = lane % JE;
            unsigned pv[1], sv[1];
            if (!p_poll<1>(a.gpart + (size_t)h * NC * 66, (unsigned)(j * 66 + chunk * JE + i), PTAG(l, 1), j < nch, pv, ctl, 0x300u + l)) return;
            const int sj = min(lane >> 1, NC - 1);
            if (!p_poll<1>(a.gpart + (size_t)h * NC * 66, (unsigned)(sj * 66 + 64 + (lane & 1)), PTAG(l, 1), lane < 2 * NC && sj < nch, sv, ctl, 0x380u + l)) return;
            PSTAMP(l * 20 + 9);
            float cm[DEC_ATT_MAXCH], cl_[DEC_ATT_MAXCH], pj[DEC_ATT_MAXCH];
#pragma unroll
            for (int q = 0; q < DEC_ATT_MAXCH; q++) {
                const int qq = min(q, NC - 1);
                cm[q] = __uint_as_float((unsigned)__shfl((int)sv[0], 2 * qq, 64));
                cl_[q] = __uint_as_float((unsigned)__shfl((int)sv[0], 2 * qq + 1, 64));
                pj[q] = __uint_as_float((unsigned)__shfl((int)pv[0], qq * JE + i, 64));
            }
            float M = -INFINITY;
#pragma unroll
            for (int q = 0; q < DEC_ATT_MAXCH; q++) M = fmaxf(M, (q < nch) ? cm[q] : -INFINITY);
            float w[DEC_ATT_MAXCH], Lsum = 0.f;
#pragma unroll
            for (int q = 0; q < DEC_ATT_MAXCH; q++) {
                w[q] = (q < nch) ? cl_[q] * __expf(cm[q] - M) : 0.f;
                Lsum += w[q];
            }
            const float rL = recip_rn(Lsum);
#pragma unroll
            for (int q = 0; q < DEC_ATT_MAXCH; q++) w[q] = (nch == 1) ? 1.0f : w[q] * rL;
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < DEC_ATT_MAXCH; q++) v += (q < nch) ? w[q] * pj[q] : 0.f;
            if (lane < JE) p_store(a.gatt + h * dh + chunk * JE, (unsigned)lane, PTAG(l, 2), __float_as_uint(v));
        }
        // (the other seven waves wait HERE, not in the poll below: 7/8 of the chip's threads polling through the join measured
        //  ~1 us more on every edge of the block)
        __syncthreads();
        PSTAMP(l * 20 + 10);

        // ================= C: o projection
        {
            unsigned raw[EPT];
            if (!p_poll<EPT>(a.gatt, (unsigned)sbase, PTAG(l, 2), on, raw, ctl, 0x400u + l)) return;
            PSTAMP(l * 20 + 11);
            float v[EPT];
#pragma unroll
            for (int i = 0; i < EPT; i++) v[i] = __uint_as_float(raw[i]);
            __syncthreads();                                           // sE: phase A's dots are done in every wave
            if (on) q8_stageN<EPT>(v, blk, sub, sE);
            __syncthreads();
            const PAct xa = p_act(sE, nb, 0, p_opaque(lane0));
            const float acc = wave_sum(p_term<WT>(xa, wO));
            if (lane == 0) pubv[wid] = acc;
            publish(a.gproj + (size_t)b * a.rpo, min(a.rpo, E - b * a.rpo), PTAG(l, 3));
            PSTAMP(l * 20 + 12);
        }

        // ================= D: residual + RMSNorm + gate|up slice + silu.mul
        {
            const uint2 tn = *(const uint2*)((const uint16_t*)tab(Lc, PL_FNORM) + sbase);
            const unsigned nwv[2] = {tn.x, tn.y};
            unsigned raw[EPT];
            if (!p_poll<EPT>(a.gproj, (unsigned)sbase, PTAG(l, 3), on, raw, ctl, 0x500u + l)) return;
            PSTAMP(l * 20 + 13);
            float v[EPT];
#pragma unroll
            for (int i = 0; i < EPT; i++) v[i] = __uint_as_float(raw[i]);
            q8_roundN<EPT>(v);
#pragma unroll
            for (int i = 0; i < EPT; i++) v[i] = x[i] + v[i];
            q8_roundN<EPT>(v);
#pragma unroll
            for (int i = 0; i < EPT; i++) hres[i] = v[i];
            if (slice) {
                norm_stage(v, nwv);
                const PAct xa = p_act(sE, nb, 0, p_opaque(lane0));
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float acc = wave_sum(p_term<WT>(xa, wG[j]));
                    if (lane == 0) pubv[(wid >> 2) * 32 + (wid & 3) * 8 + j] = acc;
                }
                PSTAMP(l * 20 + 14);
                __syncthreads();
                if (wid == 0) {
                    const int e = lane & 31;
                    float gt = q8_round32(pubv[e]);
                    gt = q8_round32(gt / (1.0f + expf(-gt)));
                    const float u = q8_round32(pubv[32 + e]);
                    const float pv = gt * u;
                    const Q8Scale sc8 = q8_scale_from_absmax(max32(fabsf(pv)));
                    const int q = q8_round(pv, sc8.scale);
                    const int qs = sum32_i(q);
                    // 32 quants -> 8 granules of 4 bytes, then the delta and the sum: ten contiguous granules, one store
                    int pk = (q & 0xff) << ((e & 3) * 8);
                    pk |= dpp_mov_i<0xB1>(pk);
                    pk |= dpp_mov_i<0x4E>(pk);
                    pu64* dst = a.gact + (size_t)b * 10;
                    const bool st_q = lane < 32 && (lane & 3) == 0;
                    const int idx = st_q ? (lane >> 2) : (lane == 1) ? 8 : 9;
                    const unsigned val = st_q ? (unsigned)pk : (lane == 1) ? __float_as_uint(sc8.ddeq) : (unsigned)qs;
                    if (st_q || lane == 1 || lane == 2) p_store(dst, (unsigned)idx, PTAG(l, 4), val);
                }
                PSTAMP(l * 20 + 15);
            }
        }

        // ================= E: down projection
        {
            const int ng = 10 * nbf;
            unsigned raw[EPT];
            const bool pa = base < ng;
            if (!p_poll<EPT>(a.gact, (unsigned)(pa ? base : 0), PTAG(l, 4), pa, raw, ctl, 0x600u + l, slice ? 0 : 1)) return;
            PSTAMP(l * 20 + 16);
            if (pa) {
#pragma unroll
                for (int i = 0; i < EPT; i++) {
                    const int gidx = base + i, s = gidx / 10, r = gidx - s * 10;
                    if (r < 8) ((unsigned*)sF.q)[s * 8 + r] = raw[i];
                    else if (r == 8) sF.d[s] = __uint_as_float(raw[i]);
                    else sF.sum[s] = (int)raw[i];
                }
            }
            __syncthreads();
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const PAct xa = p_act(sF, nbf, c, p_opaque(lane0));
                acc += p_term<WT>(xa, wD[c]);
            }
            acc = wave_sum(acc);
            PSTAMP(l * 20 + 17);
            if (lane == 0) pubv[wid] = acc;
            if (!more) issue_lm(0, true, acc);                                  // (the gate|up registers are free; before the last block: a dummy row)
            publish(a.gdown + (size_t)b * a.rpo, min(a.rpo, E - b * a.rpo), PTAG(l, 5));
        }
        PSTAMP(l * 20 + 20);
    }

    // ================= final norm + lm_head + argmax
    {
        const int LL = a.n_layers;
        const uint2 tn = *(const uint2*)(a.final_norm + sbase);
        const unsigned nwv[2] = {tn.x, tn.y};
        unsigned raw[EPT];
        if (!p_poll<EPT>(a.gdown, (unsigned)sbase, PTAG(LL - 1, 5), on, raw, ctl, 0x700u)) return;
        float v[EPT];
#pragma unroll
        for (int i = 0; i < EPT; i++) v[i] = __uint_as_float(raw[i]);
        q8_roundN<EPT>(v);
#pragma unroll
        for (int i = 0; i < EPT; i++) v[i] = hres[i] + v[i];
        q8_roundN<EPT>(v);
        norm_stage(v, nwv);
        const PAct xa = p_act(sE, nb, 0, p_opaque(lane0));
        float best = -INFINITY;
        int best_i = 0x7fffffff;
#ifndef PERSIST_NO_LM
        for (int jb = 0; jb < 2; jb++) {
            float accs[8];
#pragma unroll
            for (int j = 0; j < 8; j++) accs[j] = wave_sum(p_term<WT>(xa, wG[j]));
            if (jb == 0 && a.rw > 8) issue_lm(1, true, accs[7]);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int r = lm_row(jb, j);
                const bool valid = (jb * 8 + j < a.rw) && (wid * a.rw + jb * 8 + j < a.rph) && r < a.n_vocab;
                if (valid) {
                    if (lane == 0) a.logits[r] = accs[j];
                    if (accs[j] > best) { best = accs[j]; best_i = r; }
                }
            }
            if (a.rw <= 8) break;
        }
#endif
        if (lane == 0) { bestv[wid] = best; besti[wid] = best_i; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < NW; w++)
                if (bestv[w] > best || (bestv[w] == best && besti[w] < best_i)) { best = bestv[w]; best_i = besti[w]; }
        }
        if (wid == 0 && lane < 2) {
            const float bb = __shfl(best, 0, 64);
            const int bi = __shfl(best_i, 0, 64);
            p_store(a.gbest + 2 * (size_t)b, (unsigned)lane, PTAG(LL, 0), lane == 0 ? __float_as_uint(bb) : (unsigned)bi);
        }
        PSTAMP(LL * 20 + 1);
        if (b != 0) return;
        // workgroup 0: first maximum over the workgroups' candidates (k_dec_argmax)
        const int G = gridDim.x;
        unsigned cand[2];
        const bool ca = (int)threadIdx.x < G;
        if (!p_poll<2>(a.gbest, 2u * (ca ? threadIdx.x : 0u), PTAG(LL, 0), ca, cand, ctl, 0x800u)) return;
        float bv = ca ? __uint_as_float(cand[0]) : -INFINITY;
        int bi = ca ? (int)cand[1] : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if (lane == 0) { bestv[wid] = bv; besti[wid] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < NW; w++)
                if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
            if (bi == 0x7fffffff) bi = 0;
            a.result[n] = bi;
            const int adv = a.step->advance;
            if (adv & 2) a.tokens[n] = bi;
            const int stop = a.step->stop;
            if ((adv & 1) && (stop <= 0 || n < stop)) a.step->n = n + 1;
            __hip_atomic_store((pgu32*)ctl, ep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PSTAMP(LL * 20 + 2);
        }
    }
#undef PTAG
#undef PSTAMP
}

// ---------------------------------------------------------------- host side

struct PersistState {
    PLayer* layers = nullptr;
    unsigned* ctl = nullptr;
    pu64* gran = nullptr;          // all granule buffers, one allocation
    unsigned* stamps = nullptr;
    int n_stamps = 0;
    PArgs args{};
    int grid = 0;
    size_t smem = 0;
};

static std::vector<PersistState*> g_persist_all;      // every live decoder's state (gten_hip_persist_status)
static unsigned long long g_persist_launches = 0;
// off by default: on MI355X the persistent step measured 0.54 ms against the launch chain's 0.49 ms
static bool g_persist_on = false;
extern "C" int gten_hip_set_decode_persistent(int on)
{
    g_persist_on = on != 0;
    return 0;
}

static bool persist_supported(const gten_hip_decoder_desc& d, int n_seq, int n_chunks, int cus)
{
    // (Q8 weights: twice the registers per row in flight -- the kernel as it stands would spill; they keep the launch chain)
    if (n_seq != 1 || d.adtype != GTEN_Q8 || d.wdtype != GTEN_Q4 || d.n_layers > PERSIST_MAX_LAYERS) return false;
    const int E = d.n_embd, F = d.n_ffn, dh = E / d.n_heads, KV = dh * d.n_kv_heads, G = cus;
    if (dh != 64 || E > 2048 || E % 128 != 0 || F % 64 != 0 || F > 6144 || G > PERSIST_NT) return false;
    if (n_chunks < 1 || n_chunks > 8 || (n_chunks & (n_chunks - 1)) != 0) return false;
    if (d.n_heads * n_chunks > G || F / 32 > G) return false;
    if ((E + 2 * KV + G - 1) / G > 16 || (E + G - 1) / G > 8) return false;
    if (((d.n_vocab + G - 1) / G + 7) / 8 > 16) return false;
    return true;
}

static size_t persist_smem() { return 1024 + 2560 + 7680 + 1152 + 2048 + 4 * 17408 + PERSIST_MAX_LAYERS * sizeof(PLayer); }

template <int WT>
static int persist_launch(const PersistState* ps)
{
    DEC_LAUNCH(KT_DEC_PERSIST, (k_dec_persist<WT>), dim3(ps->grid), dim3(PERSIST_NT), ps->smem, ps->args);
    g_persist_launches++;
    return 0;
}
