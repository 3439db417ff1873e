/*
 * gten_hip.h -- C-ABI of libgten_hip.so: the MI355X (gfx950) replacement for the
 * arithmetic underneath tinyllama.cpp's gten API.
 *
 * What it replaces (upstream file:line):
 *   - the ten gten::ops:: entry points gten/modules.cpp calls
 *       token_embed   gten/ops.h:554   (modules.cpp:23)
 *       matmul_2d     gten/ops.h:651   (modules.cpp:60,77)
 *       rms_norm      gten/ops.h:806   (modules.cpp:97)
 *       rotary_emb    gten/ops.h:757   (modules.cpp:171)
 *       silu(_inplace)gten/ops.h:700,708 (modules.cpp:144,152)
 *       mul(_inplace) gten/ops.h:853,861 (modules.cpp:116,125)
 *       add           gten/ops.h:900   (modules.cpp:40)
 *       qkv_attn      gten/ops.h:1118  (modules.cpp:219)
 *   - Tensor storage allocation, gten/tensor.cpp:61 (malloc -> HBM)
 *   - the quantization block formats of gten/quants.h:17-31 as the activation
 *     layout in HBM, and a load-time repack of weight blocks (below).
 *
 * Conventions (same as the reference operators, gten/ops.h):
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - outputs are caller-allocated; nothing passed in is owned by the library;
 *   - shapes are [n][d] row-major, pitches are BYTES per row, rows
 *     [start_pos, n) are computed, rows < start_pos are left untouched;
 *   - dtype codes follow `enum class Dtype` (gten/gten_types.h:20-26);
 *   - calls are asynchronous on the library's stream (gten_hip_stream()); host
 *     visibility needs gten_hip_memcpy_d2h (synchronous) or gten_hip_sync();
 *   - one calling thread per process/device (the reference ops are not
 *     re-entrant either, gten/ops.h:37);
 *   - return 0 on success, a non-zero code otherwise with a message available
 *     from gten_hip_last_error().  The C++ wrappers in
 *     tinyllama.cpp_amd/gten/ops.h turn a non-zero return into the reference's
 *     GTEN_ASSERT behaviour (gten/log.h:6-10: red message, exit(EXIT_FAILURE)).
 *
 * Activation layouts in HBM are byte-identical to the reference's host layouts:
 *   f16: IEEE half;  Q8: 34-byte blocks {f16 delta, int8 q[32]} (quants.h:17-23).
 * Weight layouts: f16 as in the file; Q8/Q4 are REPACKED once at load by
 * gten_hip_pack_weight() from the .gten block stream (quants.h:17-31) into
 *   Q4: [rows][nb][16 B nibbles]            then [rows][nb] f16 deltas
 *   Q8: [rows][2][nb][16 B int8 half-block] then [rows][nb] f16 deltas
 * (nb = cols/32), same total bytes, so that one wavefront instruction reads
 * 1 KiB of contiguous quants.  All ops below take Q8/Q4 WEIGHTS in this packed
 * form and f16 weights as is.
 */
#ifndef GTEN_HIP_H
#define GTEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { GTEN_I32 = 0, GTEN_F16 = 1, GTEN_F32 = 2, GTEN_Q8 = 3, GTEN_Q4 = 4 };

/* ---- runtime ---------------------------------------------------------- */
int         gten_hip_device_count(void);             /* does not initialise the GPU */
int         gten_hip_init(int device);               /* idempotent per process */
const char* gten_hip_last_error(void);
void*       gten_hip_stream(void);                   /* hipStream_t work is being queued on (the selected one) */
int         gten_hip_sync(void);                     /* waits for the selected stream */
/* The library owns two streams (the second is created when it is first selected); every call queues on the selected one (0 at start).  Stream 1 exists so that the prompt of a
 * NEW sequence can be processed beside the decode steps of the others (continuous batching, TinyLlamaBatch::serve): work on
 * different streams is unordered unless gten_hip_stream_wait(waiter, on) makes `waiter` wait for what `on` holds so far.
 * One calling thread; buffers touched on both streams are the caller's to order. */
int         gten_hip_select_stream(int idx);
int         gten_hip_stream_wait(int waiter, int on);
int         gten_hip_stream_idle(int idx, int* idle);   /* *idle = 1 when everything queued on stream idx has finished (no wait) */

/* replaces std::malloc / std::free of Tensor storage, gten/tensor.cpp:23-25,61 */
int gten_hip_malloc(void** dptr, size_t nbytes);
int gten_hip_free(void* dptr);
int gten_hip_memset(void* dptr, int byte, size_t nbytes);
/* loader staging, tinyllama.cpp:320 (host file bytes -> HBM) */
int gten_hip_memcpy_h2d(void* dst, const void* src_host, size_t nbytes);
/* sampler / debug reads, tinyllama.cpp:414,464 (synchronous) */
int gten_hip_memcpy_d2h(void* dst_host, const void* src, size_t nbytes);
int gten_hip_memcpy_d2d(void* dst, const void* src, size_t nbytes);

/* HIP-event profiler: while enabled, every kernel launch is bracketed by two
 * events on the library's stream.  `family` is a kernel-family index
 * (gten_hip_prof_family_name lists them).  Not usable during graph capture. */
int         gten_hip_prof_enable(int on);
int         gten_hip_prof_read(int family, int* launches, double* total_ms);
const char* gten_hip_prof_family_name(int family);   /* NULL past the last family */

/* Self-test of the Q8 quantizer's scale arithmetic (gten/quants.h:52-58: delta = absmax / 127.0f, scale = 1.0f /
 * delta).  The kernels compute both with short exact sequences (csrc/gten_dev.h: div127, recip_rn) instead of the
 * compiler's IEEE division expansion; this runs them against that expansion on the device over every binary32
 * significand of several binades and returns the number of differing results (must be 0). */
int gten_hip_selftest_q8scale(unsigned long long* mismatches_div127, unsigned long long* mismatches_recip);

/* bytes of one storage row (gten/tensor.h:97-117, gten/tensor.cpp:37-57) */
size_t gten_hip_row_bytes(int dtype, int cols);

/* ---- weights ---------------------------------------------------------- */
/* src: rows*cols/32 blocks in .gten order (device copy of the file payload);
 * dst: rows*gten_hip_row_bytes(dtype, cols) bytes, packed layout above.
 * f16 needs no packing (call is a plain copy). */
int gten_hip_pack_weight(const void* src_blocks, int dtype, int rows, int cols, void* dst_packed);

/* ---- the ten operators ------------------------------------------------ */
/* ops::token_embed, gten/ops.h:514-564.  tokens: int32[n] on the device. */
int gten_hip_token_embed(const void* w, int w_dtype, int n_vocab,
                         const int32_t* tokens,
                         void* out, int out_dtype, size_t out_pitch,
                         int n, int d, int start_pos);

/* ops::matmul_2d, gten/ops.h:613-670.  out[r][c] = dot(x[r,:], w[c,:]).
 * (x_dtype,w_dtype) in {(f16,f16),(Q8,Q8),(Q8,Q4)}; out_dtype in {f16,Q8,f32}.
 * The lm_head form (EmbeddingLinear, gten/modules.cpp:70-81) is n rows in, the
 * last row out: pass x = last row, n = 1, start_pos = 0, out = f32[d_out]. */
int gten_hip_matmul_2d(const void* x, int x_dtype, size_t x_pitch,
                       const void* w, int w_dtype,
                       void* out, int out_dtype, size_t out_pitch,
                       int n, int d_in, int d_out, int start_pos);

/* ops::matmul_2d with >= 16 new rows and quantized weights runs on the matrix cores in one of two forms
 * (csrc/gten_mfma.hip): fast (default) -- the block deltas folded into the f16 operands, sums accumulated across K
 * inside the MFMA, results inside the q8 / q4 logit band -- or exact (on != 0) -- one MFMA per quant block = the
 * reference's exact integer block dot, rescaled and added in the scalar build's order (gten/ops.h:296-312), bit for
 * bit.  Process-wide; f16 weights are not affected. */
int gten_hip_set_prefill_exact(int on);
/* The same choice for the decode step of decoders created AFTERWARDS (a decoder's graphs are captured once): exact
 * (on != 0) -- attention probabilities rounded against the statistics of the whole row (two launches per block, the
 * reference's rounding point gten/ops.h:972-997), p.V as separately rounded multiplies and adds, the W.x of 16+ sequences
 * as exact integer block sums (k_dec_mmv) -- or fast (default; DESIGN.md 3.5 lists the deviations: chunk-local rounding
 * of the probabilities, fp16 operand rounding in the wide W.x and p.V).  Up to 8 sequences and contexts <= 256 both forms
 * are the same bytes. */
int gten_hip_set_decode_exact(int on);
/* Decoders of 16+ sequences created AFTERWARDS (Q8 activations, fast forms) keep HEAD-MAJOR SHADOWS of their sequences' K / V
 * caches (on != 0, the default; round 5, csrc/gten_decode_attn_hm.h) or read the cache rows as they lie (0: round 4's kernel).
 * The caches themselves keep the reference's row layout [max_ctx][kv_dim] (gten/modules.cpp:188-201) for every reader and
 * writer; a shadow holds the same bytes per (kv head, 256 positions) as ONE contiguous run in matrix-operand order, is filled
 * from the rows when a sequence (re)starts, follows the decode appends, and is re-imported before the next step whenever ANY
 * call of this library has written into one of the rows since (every writer announces its output range: a stale shadow is
 * not possible through this interface; a caller that writes cache rows with its own kernels restarts the sequence --
 * slot_start / step at another position / generate_multi -- to the same effect). */
int gten_hip_set_kv_head_major(int on);
/* gate | up (with the silu . mul chain) and the lm_head of lanes of 49-64 or 128 rows (q4 / q8 weights, K = 2048) as the streamed
 * kernels of csrc/gten_decode_ffn.h (on != 0, the default) or as k_dec_mmvh (+ k_dec_silumul_rows for q8) like every other shape
 * (0): the same sums bit for bit (tests/test_ffn_streamed_gpu.py), a launch-structure switch only.
 * Takes effect for steps enqueued or captured afterwards. */
int gten_hip_set_ffn_streamed(int on);
/* The o and down projections of f16 decoders of 16+ sequences (fast forms) as EIGHT K planes of 64-feature workgroups
 * (csrc/gten_decode_wxp.h; on != 0, the default) or as k_dec_mmv_f16 in two planes (0).  The two differ in the association of the f32
 * sums only (not in any rounding point of the reference); every wide f16 decoder takes the same form, so sequences agree bit for bit
 * whatever the batch they are decoded in.  Takes effect for steps enqueued or captured afterwards: set it before creating decoders
 * that are to agree. */
int gten_hip_set_wx_planes(int on);
/* host-only self-test of the registry behind that guarantee (which ranges are watched, which writes hit them, whose flag is
 * set): needs no GPU and no gten_hip_init; returns 0 or the number of the first failing case */
int gten_hip_kv_watch_selftest(void);
/* single-sequence q4 decoders created AFTER this call run the step as ONE persistent launch (1) or as the chain of 113
 * launches (0, the default: it is the faster one on MI355X): the same bytes either way
 * (tests/test_persist_gpu.py); csrc/gten_decode_persist.h */
int gten_hip_set_decode_persistent(int on);
/* the persistent step over every live decoder of the process: how many run it, its launches enqueued so far (captured
 * launches count once), the abort code of a poll that gave up (0 = none; cleared by the call: that step's results are
 * invalid), workgroup 0's phase stamps of the newest decoder created with GTEN_HIP_PERSIST_STAMPS=1.  Waits for the stream. */
int gten_hip_persist_status(int* n_decoders, unsigned long long* launches, unsigned* abort_code, unsigned* stamps_host, int n_stamps);

/* ops::rms_norm, gten/ops.h:762-814.  w: f16[d]. */
int gten_hip_rms_norm(const void* x, int dtype, size_t x_pitch, const void* w_f16,
                      void* out, size_t out_pitch, int n, int d, int start_pos);

/* ops::rotary_emb, gten/ops.h:714-760.  In place; position = row index. */
int gten_hip_rotary_emb(void* x, int dtype, size_t pitch, int n, int d, int d_head, int start_pos);

/* ops::silu / silu_inplace (out == x), gten/ops.h:673-711 */
int gten_hip_silu(const void* x, void* out, int dtype, size_t pitch, int n, int d, int start_pos);
/* ops::mul / mul_inplace (out == a), gten/ops.h:816-867 */
int gten_hip_mul(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos);
/* ops::add, gten/ops.h:870-910 */
int gten_hip_add(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos);

/* ops::qkv_attn, gten/ops.h:930-1133.  Causal GQA attention over the K/V
 * caches (= rows [0,n) of k and v).  The reference's `qk` scratch tensor is
 * never read by its callers (gten/modules.cpp:216-221) and is not needed:
 * probabilities are rounded to the activation dtype in flight, exactly as the
 * reference rounds them when it stores a row (gten/ops.h:996-997). */
int gten_hip_qkv_attn(const void* q, const void* k, const void* v, void* out, int dtype,
                      size_t q_pitch, size_t kv_pitch, size_t out_pitch,
                      int n, int n_heads, int n_kv_heads, int d_head, int start_pos);

/* ---- one transformer block over many new rows (prompt processing) ------
 * AttentionBlock::forward (gten/modules.cpp:224-254) for rows [start_pos, n) as ONE call: the same kernels as the
 * operators above, composed so that matrices sharing an input share its conversion and its launch (q | k | v, gate | up),
 * q and k are rotated together, silu and the product are one pass and the residual sums ride in the epilogues of the o and
 * down projections -- 12 launches instead of 23.  Every buffer named here ends with exactly the bytes the module-by-module
 * sequence leaves in it (tests/test_block_rows_gpu.py).  Activation buffers are dense rows in `adtype`; k and v are the
 * K / V caches.  Returns GTEN_HIP_NOT_HANDLED (no error recorded) for configurations this path does not compute (fewer than
 * 16 new rows, d_head != 64, mixed dtypes): the caller then runs the operators. */
#define GTEN_HIP_NOT_HANDLED 1
typedef struct {
    int adtype, wdtype;         /* GTEN_Q8 activations with GTEN_Q8 / GTEN_Q4 weights, or GTEN_F16 with GTEN_F16 */
    int n_embd, n_heads, n_kv_heads, n_ffn;
    const void *attn_norm_w, *wq, *wk, *wv, *wo, *ffn_norm_w, *wgate, *wup, *wdown;
    const void* inp;            /* [.][n_embd]: the block's input rows */
    void *attn_norm_out, *q, *k, *v, *attn_out, *o;
    void* h;                    /* inp + o (Residual) */
    void *ffn_norm_out, *gate, *up, *down;   /* gate ends as silu(gate) * up, in place like the reference's modules */
    void* out;                  /* h + down: the block's output rows */
} gten_hip_block_desc;
int gten_hip_block_rows(const gten_hip_block_desc* b, int n, int start_pos);
/* 0: gten_hip_block_rows answers GTEN_HIP_NOT_HANDLED for everything (the modules run one by one; tests compare the two) */
int gten_hip_set_block_rows(int on);

/* SEVERAL prompts as one row matrix (batched prompt processing; round 3).  The rows of a prompt-sized call are
 * row-independent in every operator except two: the position RoPE rotates by, and the rows attention looks back over.
 * With row segments set -- starts[0] = 0 < starts[1] < ... < starts[n_segments] = the row count (<= 4096), every segment 16 ..
 * 2048 rows --
 * gten_hip_block_rows (start_pos 0, n = starts[n_segments]) treats rows [starts[k], starts[k + 1]) as prompt k: position =
 * row - starts[k], attention inside the segment only; everything else runs once over all rows (one W.x per projection for
 * all prompts).  Per prompt the results do not depend on the other segments: the W.x launches of a segmented call never
 * share a K loop between workgroups (a K split chosen by the row count would), so a prompt gets the same bits alone
 * or beside others.  gten_hip_rotary_emb / gten_hip_qkv_attn REFUSE to run while segments are set (they would rotate
 * and attend across prompts).  n_segments = 0 clears.  The caller copies each prompt's K / V rows from the shared row
 * matrix into its own caches: gten_hip_copy_ranges. */
int gten_hip_set_row_segments(const int32_t* starts, int n_segments);
/* 1 when gten_hip_block_rows computes segmented calls for this configuration (else the modules would run one by one and
 * the two operators above would refuse): ask before setting segments */
int gten_hip_row_segments_ok(int n_embd, int n_ffn, int n_heads, int n_kv_heads, int wdtype, int adtype);
/* up to GTEN_HIP_MAX_COPY_RANGES device-to-device copies (non-overlapping) in ONE launch on the current stream */
#define GTEN_HIP_MAX_COPY_RANGES 64
typedef struct { void* dst; const void* src; size_t bytes; } gten_hip_copy_range;
int gten_hip_copy_ranges(const gten_hip_copy_range* ranges, int n);

/* the greedy sampler's argmax of one row of f32 logits on the device (tinyllama.cpp:416-424: strict >, the first maximum wins):
 * out[0] = the id.  Asynchronous; used by the batched prompt path so that a prompt's first id costs 4 bytes of copy, not 128 KB. */
int gten_hip_argmax_row(const float* logits, int n, int32_t* out);

/* ---- single-token decode fast path -------------------------------------
 * One call = one decoded token = TinyLlama::logits(tokens, start_pos = n-1)
 * (tinyllama.cpp:45-61) plus the greedy argmax of tinyllama.cpp:416-424, for a
 * model whose weights / K,V caches are the SAME device tensors the operators
 * above use (the caches are Linear::acv of the key / value projections,
 * gten/modules.cpp:188-201).  Results are the bytes the operator-by-operator
 * path produces (same rounding points); the work is fused into 5 launches per
 * block and replayed from one hipGraph, with the position n read from device
 * memory.  Rows < n-1 of the caches must already hold the context (from
 * earlier steps or from an operator-path prefill). */
typedef struct {
    const void *wq, *wk, *wv, *wo, *wgate, *wup, *wdown;   /* packed Q8/Q4 or f16 */
    const void *attn_norm, *ffn_norm;                      /* f16 [n_embd] */
    void *kcache, *vcache;                                 /* [max_ctx][kv_dim] activation dtype, dense rows */
} gten_hip_layer_ptrs;

typedef struct {
    int n_vocab, max_ctx, n_embd, n_ffn, n_layers, n_heads, n_kv_heads, wdtype, adtype;
    const void* embed;          /* [n_vocab][n_embd] */
    const void* final_norm;     /* f16 [n_embd] */
    const void* lm_head;        /* [n_vocab][n_embd] */
    float* logits;              /* f32 [n_vocab] output (EmbeddingLinear::acv) */
} gten_hip_decoder_desc;

typedef struct gten_hip_decoder gten_hip_decoder;

/* K/V caches of one (sequence, layer) for multi-sequence decode */
typedef struct { void* kcache; void* vcache; } gten_hip_kv_ptrs;

int gten_hip_decoder_create(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers, gten_hip_decoder** out);
/* Multi-sequence decode: n_seq in {2, 4, 8}, 16/32/48/64 (W.x on the matrix cores, rows =
 * sequences) or 128/192/256/384/512 (LANES: each lane's launch chain is a parallel branch of the step's graph; q8 / q4
 * weights run lanes of 128 sequences -- eight row tiles per W.x workgroup --, f16 weights, the exact forms and 192 lanes of
 * 64, at most four lanes; per sequence the results of a 64-sequence decoder) independent sequences, each with
 * its own K/V caches (kv[seq * n_layers + layer]) and token ids, advance by one token per step and
 * SHARE every weight pass (weights are streamed once per step, not once per sequence).  Per sequence
 * the results are bit-identical to the single-sequence decoder for n_seq <= 8 (n_seq >= 16: the linears follow
 * the matrix-core kernel's block order, i.e. the prefill numerics).  desc->logits is ignored: read a
 * sequence's logits with gten_hip_decoder_logits_seq.  All sequences are at the same position n. */
int gten_hip_decoder_create_multi(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers,
                                  const gten_hip_kv_ptrs* kv, int n_seq, gten_hip_decoder** out);
int gten_hip_decoder_destroy(gten_hip_decoder* dec);
/* token ids (host) for positions [first, first+count): the step for context
 * length n embeds token[n-1]. */
int gten_hip_decoder_set_tokens(gten_hip_decoder* dec, const int32_t* tokens_host, int first, int count);
int gten_hip_decoder_set_tokens_seq(gten_hip_decoder* dec, int seq, const int32_t* tokens_host, int first, int count);
/* asynchronous: computes row n-1, logits and their argmax.  use_graph != 0
 * replays the captured hipGraph (captured on first use). */
int gten_hip_decoder_step(gten_hip_decoder* dec, int n, int use_graph);
/* `count` consecutive steps n_first, n_first + 1, ... (ids already set): asynchronous and free-running -- the position lives
 * on the device and each step's last kernel advances it -- replayed four steps per hipGraph launch (the idle time between
 * two graph replays is then paid once per four tokens).  Same results as `count` calls of gten_hip_decoder_step. */
int gten_hip_decoder_steps(gten_hip_decoder* dec, int n_first, int count, int use_graph);
/* Greedy generation with the sampler on the device (tinyllama.cpp:395-440 without the 128 KB logits copy and the host
 * argmax per token): token ids [0, n_first) must be set (gten_hip_decoder_set_tokens) and the caches hold rows
 * [0, n_first - 1).  Steps n_first, n_first + 1, ... run back to back, each argmax (strict >, first maximum wins) becoming
 * the next input token on the device, until `max_new` ids are produced, the context is full, or `eos` comes up (not
 * stored, as in the reference).  The new ids are copied to out_host[0 .. *n_out).  Single-sequence decoders. */
int gten_hip_decoder_generate(gten_hip_decoder* dec, int n_first, int max_new, int eos, int32_t* out_host, int* n_out);
/* ... and for every sequence of a multi-sequence decoder, sequence q from step n_first[q] (its ids [0, n_first[q]) set,
 * its caches holding rows [0, n_first[q] - 1)): finished sequences are parked while the others go on.  Sequence q produces at
 * most max_new_seq[q] ids (NULL: max_new for all; 0: the sequence is parked from the start); out_host is [n_seq][max_new],
 * n_out [n_seq]. */
int gten_hip_decoder_generate_multi(gten_hip_decoder* dec, const int* n_first, const int* max_new_seq, int max_new, int eos,
                                    int32_t* out_host, int* n_out);
/* ---- continuous batching: the slots of a multi-sequence decoder are started and parked independently --------------
 * A slot (= one sequence's K/V caches and token row) takes a new prompt as soon as its previous sequence has ended,
 * while the other slots go on decoding; every step still streams the weights once for all slots.
 *   slot_start: sequence `seq` generates from step n_first on -- its ids [0, n_first) must be set
 *               (gten_hip_decoder_set_tokens_seq) and its caches hold rows [0, n_first - 1) (operator-path prefill);
 *               each step's argmax becomes its next input token on the device;
 *   slot_start_until: ... and the slot's LAST step is n_last (0: no bound, as slot_start): once it has run, further
 *               shared steps repeat it (the same row, the same bytes into the same places) instead of moving on, so a run
 *               of `steps` need not be cut to the shortest remaining sequence -- the caller reads the ids up to n_last;
 *   slot_park:  the slot stops advancing (its row is still computed by the shared launches; the results are ignored);
 *   run:        `steps` back-to-back graph replays of the whole batch, asynchronous; no live slot without a last step
 *               may pass max_ctx (returns an error instead of launching);
 *   slot_ids:   waits for the stream and copies the argmax ids of steps [n_from, n_from + count) of `seq`.
 * The scheduler that drives these (prompt queue, eos, admission) is host code: TinyLlamaBatch::serve,
 * tinyllama.cpp_amd/host/tinyllama_model.h. */
int gten_hip_decoder_slot_start(gten_hip_decoder* dec, int seq, int n_first);
int gten_hip_decoder_slot_start_until(gten_hip_decoder* dec, int seq, int n_first, int n_last);
int gten_hip_decoder_slot_park(gten_hip_decoder* dec, int seq);
/* a PARKED slot gets another set of caches (kv[layer], n_layers entries; the set it had stays untouched): the caller fills
 * spare sets with the prompts to come while every slot is busy and hands a ready one to the next slot that ends
 * (TinyLlamaBatch::serve).  The next slot_start / slots_apply of the slot makes the shared steps use it. */
int gten_hip_decoder_slot_bind(gten_hip_decoder* dec, int seq, const gten_hip_kv_ptrs* kv);
/* several slots at once, one wait at the end: slot seqs[i] is started at n_first[i] with its last step n_last[i] (0: none) and --
 * when tokens and tokens[i] are given -- its ids [0, n_first[i]) set; n_first[i] == 0 parks it.  A sequence may appear once. */
int gten_hip_decoder_slots_apply(gten_hip_decoder* dec, int count, const int* seqs, const int* n_first, const int* n_last,
                                 const int32_t* const* tokens);
int gten_hip_decoder_run(gten_hip_decoder* dec, int steps);
/* the same run with the lanes whose slots are ALL parked left out when skip_empty_lanes != 0, whatever gten_hip_set_lane_skip
 * says: the tail of a queue, once the caller has moved its last sequences into as few lanes as they fit (slot_bind) */
int gten_hip_decoder_run_lanes(gten_hip_decoder* dec, int steps, int skip_empty_lanes);
/* lanes (round 4): with set_lane_skip(1) a run leaves out every lane whose slots are all parked (its launch chain costs a full
 * lane's time however few slots are live); off by default -- it measured slower on the bench's serving queue, DESIGN.md 3.6 --
 * and the ids never depend on it (tests/test_serving_gpu.py).  lane_info: sequences per lane, lanes, how many the last run took. */
int gten_hip_decoder_lane_info(gten_hip_decoder* dec, int* lane_rows, int* lanes, int* last_run_lanes);
int gten_hip_set_lane_skip(int on);
int gten_hip_decoder_slot_ids(gten_hip_decoder* dec, int seq, int n_from, int count, int32_t* ids_host);
/* ... of EVERY sequence at once: ids_host[q * count + i] = the argmax of step n_from[q] + i of sequence q (count <= 64; a
 * sequence's entries past the steps it ran are unspecified) -- one gather launch and one copy instead of n_seq copies */
int gten_hip_decoder_slot_ids_all(gten_hip_decoder* dec, const int* n_from, int count, int32_t* ids_host);
/* multi-sequence decoders: sequence q decodes row n_per_seq[q] - 1 (continuous batching: sequences of different
 * lengths share the weight passes); results are read per sequence with gten_hip_decoder_result_seq(dec, q, n_per_seq[q]) */
int gten_hip_decoder_step_ragged(gten_hip_decoder* dec, const int* n_per_seq, int use_graph);
/* HIP-event timing of one kernel family of the step (family index as in gten_hip_prof_family_name):
 * `reps` replays of a graph holding only that family's launches at context length n, bracketed by
 * two events on the library's stream; *avg_us = elapsed / (reps * launches per replay). */
int gten_hip_decoder_time_family(gten_hip_decoder* dec, int family, int n, int reps, double* avg_us, int* launches_per_replay);
/* head-major shadows of this decoder: *head_major = whether it keeps them; sequence imports (rows -> shadow) launched so far and
 * the launches they took (tests watch a write into a cache row being followed by a re-import) */
int gten_hip_decoder_kv_info(gten_hip_decoder* dec, int* head_major, unsigned long long* seq_imports, unsigned long long* import_launches);
/* waits for the stream and returns the argmax produced by step n */
int gten_hip_decoder_result(gten_hip_decoder* dec, int n, int32_t* argmax_host);
int gten_hip_decoder_result_seq(gten_hip_decoder* dec, int seq, int n, int32_t* argmax_host);
/* waits for the stream and copies the f32 logits of the last step of `seq` to the host */
int gten_hip_decoder_logits_seq(gten_hip_decoder* dec, int seq, float* logits_host);

#ifdef __cplusplus
}
#endif
#endif
