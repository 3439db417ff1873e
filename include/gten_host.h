/*
 * gten_host.h -- C-ABI of libgten_host.so: the model-level caller of the hot
 * path, i.e. this repository's counterpart of the
 * reference's TinyLlama class, .gten loader and greedy loop
 * (tinyllama.cpp:23-76, 301-440), written in C++ on top of the HBM-backed gten
 * API (tinyllama.cpp_amd/gten/) and exported flat so that tests and bench.py
 * can drive it through ctypes.
 *
 * All pointers here are HOST pointers.  Functions return 0 on success; contract
 * violations inside the gten layer follow the reference's convention instead
 * (red "GTEN ERROR" line on stderr, exit(EXIT_FAILURE), gten/log.h:6-23).
 */
#ifndef GTEN_HOST_H
#define GTEN_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gten_host_model gten_host_model;

/* dims of TinyLLamaParams (tinyllama.cpp:12-20) made configurable so that small
 * parity models can be built; dtypes use the codes of include/gten_hip.h and the
 * pairs chosen by tinyllama.cpp:258-265: (f16,f16), (Q8,Q8), (Q4,Q8). */
typedef struct {
    int n_vocab, max_ctx, n_embd, n_ffn, n_layers, n_heads, n_kv_heads;
    int wdtype, adtype;
} gten_host_config;

void gten_host_default_config(gten_host_config* cfg, int wdtype, int adtype); /* TinyLlama-1.1B */

gten_host_model* gten_host_model_create(const gten_host_config* cfg);
void gten_host_model_free(gten_host_model* m);

/* weights in .gten order (tinyllama.cpp:345-391): embed; per layer q,k,v,o,gate,
 * up,down,input_layernorm,post_attention_layernorm; model.norm; lm_head */
int    gten_host_model_n_weights(const gten_host_model* m);
size_t gten_host_model_weight_bytes(gten_host_model* m, int idx);
int    gten_host_model_set_weight(gten_host_model* m, int idx, const void* bytes, size_t nbytes);
int    gten_host_model_load_gten(gten_host_model* m, const char* path);           /* tinyllama.cpp:336-392 */
int    gten_host_model_load_synthetic(gten_host_model* m, uint64_t seed);         /* host/synth.h */

/* TinyLlama::logits(tokens, start_pos), tinyllama.cpp:45-61: all n token ids are
 * passed, rows [start_pos, n) are computed, logits of the last row (f32[n_vocab])
 * are copied to `logits_out` (may be NULL to skip the copy). */
int gten_host_model_logits(gten_host_model* m, const int32_t* tokens, int n, int start_pos, float* logits_out);

/* greedy loop of tinyllama.cpp:395-440 on token ids: prefill `n_prompt` ids, then
 * append argmax ids (strict >, first maximum wins) until `max_tokens` total or
 * `eos` is produced (pass eos < 0 to never stop).  Returns the total count. */
int gten_host_model_greedy(gten_host_model* m, int32_t* tokens, int n_prompt, int max_tokens, int eos);
/* the same loop with the sampler on the device (gten_hip_decoder_generate): the prompt as above, every later id from
 * back-to-back graph replays whose argmax is the next step's input on the device.  Same ids, same return value. */
int gten_host_model_generate(gten_host_model* m, int32_t* tokens, int n_prompt, int max_tokens, int eos);

/* ---- tokenizer + chat template (tokenizer.h:22-283 of the reference, on its vocabulary file tokenizer.bin).
 * encode: chat_template != 0 gives Tokenizer::encode (tokenizer.h:135-170: [1, 32001] + bpe("user\n" + prompt) +
 * [32002, 29871, 13, 32001, 20255, 13]); 0 gives the plain BPE ids (encode_internal, 172-283).  Returns the id count,
 * or -(count) when `cap` is too small.  decode: Tokenizer::decode (94-110), the piece of `token` after `prev_token`.
 * create returns NULL when the file cannot be opened. */
typedef struct gten_host_tokenizer gten_host_tokenizer;
gten_host_tokenizer* gten_host_tokenizer_create(const char* path, int vocab_size);
void        gten_host_tokenizer_free(gten_host_tokenizer* t);
int         gten_host_tokenizer_encode(gten_host_tokenizer* t, const char* prompt, int chat_template, int32_t* ids_out, int cap);
const char* gten_host_tokenizer_decode(gten_host_tokenizer* t, int prev_token, int token);

/* The fused single-token decode path (include/gten_hip.h) is used by
 * gten_host_model_logits / _greedy whenever exactly one new row is requested;
 * this switch forces the operator-by-operator path instead (on = 0). */
int gten_host_model_set_fast_decode(gten_host_model* m, int on);

/* Throughput-oriented decode: upload token ids once (teacher forcing), then
 * queue steps without waiting.  Step n embeds tokens[n-1], attends over rows
 * [0, n) of the caches and leaves logits + argmax on the device. */
int gten_host_model_decode_begin(gten_host_model* m, const int32_t* tokens, int count);
int gten_host_model_decode_step(gten_host_model* m, int n, int use_graph);          /* asynchronous */
int  gten_host_model_decode_steps(gten_host_model* m, int n_first, int count, int use_graph);  /* asynchronous: steps n_first .. n_first + count - 1, four per graph replay */
int gten_host_model_decode_result(gten_host_model* m, int n, int32_t* argmax_out);  /* waits */

/* ---- several sequences on one GPU sharing one copy of the weights.
 * n_seq in {2, 4, 8} (GEMV kernels, bit-identical to single-sequence decode) or {16, 32, 48, 64} (quantized
 * configurations: every W.x runs as a skinny matrix product on the matrix cores, prefill numerics).  Every sequence is a full model object of the gten API with its own K/V
 * caches; their weight tensors alias sequence 0's storage.  One decode step advances ALL sequences
 * by one token and streams every weight once.  Per sequence the results are bit-identical to
 * gten_host_model_decode_* on a model of its own. */
typedef struct gten_host_batch gten_host_batch;
gten_host_batch* gten_host_batch_create(const gten_host_config* cfg, int n_seq);
void gten_host_batch_free(gten_host_batch* b);
int  gten_host_batch_load_synthetic(gten_host_batch* b, uint64_t seed);
int  gten_host_batch_set_weight(gten_host_batch* b, int idx, const void* bytes, size_t nbytes);  /* idx ascending */
/* prompt of sequence `seq` through the operator path (fills its caches); logits_out may be NULL */
int  gten_host_batch_prefill(gten_host_batch* b, int seq, const int32_t* tokens, int n, float* logits_out);
/* SEVERAL prompts as segments of one row matrix (wide batches, >= 16 sequences; gten_hip_set_row_segments): prompt k =
 * tokens[starts[k] .. starts[k + 1]) (>= 16 ids each, at most 2048 per prompt, 4096 in all, at most 32 prompts) onto the caches of sequence seqs[k] (every
 * sequence at most once: a repeated one returns -1);
 * logits_out, when given, is [n_prompts][n_vocab].  Returns -2 when this batch does not process prompts that way. */
int  gten_host_batch_prefill_many(gten_host_batch* b, const int32_t* seqs, const int32_t* tokens, const int32_t* starts, int n_prompts, float* logits_out);
/* greedy generation of every sequence with the sampler on the device: prompts is [n_seq][max_prompt] (sequence q uses its
 * first n_prompt[q] ids), each prompt is processed on its own caches, then all sequences generate together, each from its
 * own position, until `max_tokens` total ids or `eos`.  out is [n_seq][max_tokens] (prompt + new ids), n_total [n_seq]. */
int  gten_host_batch_generate(gten_host_batch* b, const int32_t* prompts, const int32_t* n_prompt, int max_prompt, int max_tokens, int eos,
                              int32_t* out, int32_t* n_total);
/* Continuous batching: a queue of n_prompts prompts ([n_prompts][max_prompt], prompt j uses its first n_prompt[j] ids) served
 * through the batch's n_seq slots -- a slot whose sequence has ended (eos, `max_tokens` ids in all, or the context) takes the
 * next prompt at once while the other slots go on decoding (TinyLlamaBatch::serve, host/tinyllama_model.h; device side:
 * gten_hip_decoder_slot_start / _run / _slot_ids).  `slice` = shared steps between two looks at the results; max_new > 0
 * additionally bounds the ids generated per prompt (max_new_each, when not NULL: prompt j's own bound).
 * out is [n_prompts][max(max_tokens, max_prompt)] (prompt + new ids), n_total [n_prompts]; stats (may be NULL) receives SIX
 * doubles: {prompt tokens, new tokens, shared steps, admissions, seconds in prompt processing, seconds in shared steps} -- the
 * list this entry point was first published with, and it stays six. */
int  gten_host_batch_serve(gten_host_batch* b, const int32_t* prompts, const int32_t* n_prompt, int n_prompts, int max_prompt,
                           int max_tokens, int eos, int slice, int max_new, const int32_t* max_new_each, int32_t* out, int32_t* n_total, double* stats);
/* ... the same with the caller saying how many doubles `stats` holds: exactly n_stats are written (zeros past the list), so the
 * list can grow without overrunning an older caller.  In order: the six above, lane-steps (shared steps x the lanes each of them
 * ran: a lane without a live slot sits a run out), slots per lane, sequences moved between lanes in the tail of the queue. */
int  gten_host_batch_serve2(gten_host_batch* b, const int32_t* prompts, const int32_t* n_prompt, int n_prompts, int max_prompt,
                            int max_tokens, int eos, int slice, int max_new, const int32_t* max_new_each, int32_t* out, int32_t* n_total, double* stats,
                            int n_stats);
/* tests: k > 0 fixes the admission schedule of gten_host_batch_serve -- exactly k prompts are processed beside every slice
 * (as far as free slots and the queue allow) instead of as many as fit while it runs; 0 restores the default */
int  gten_host_batch_set_serve_schedule(gten_host_batch* b, int k);
/* cache sets beyond the sequences' own that gten_host_batch_serve fills AHEAD of the slots that will take them, so that a slot
 * which ends joins the next slice with a prompt that is already processed (gten_hip_decoder_slot_bind): -1 = a quarter of the
 * slots, at most 64, for batches of 16 sequences and more (the default), 0 = a prompt is processed only once a slot is free */
int  gten_host_batch_set_serve_spares(gten_host_batch* b, int n);
/* percent of the slots that get a processed prompt before a queue's FIRST slice starts (wide batches): 100 (default) fills
 * every slot first -- fewer, fuller shared steps, the first ids later; 0 starts decoding with the first batch of prompts */
int  gten_host_batch_set_serve_ramp(gten_host_batch* b, int percent);
int  gten_host_batch_decode_begin(gten_host_batch* b, int seq, const int32_t* tokens, int count);
int  gten_host_batch_decode_step(gten_host_batch* b, int n, int use_graph);                 /* asynchronous, all sequences */
int  gten_host_batch_decode_steps(gten_host_batch* b, int n_first, int count, int use_graph);   /* asynchronous: count consecutive steps, four per graph replay */
int  gten_host_batch_decode_step_ragged(gten_host_batch* b, const int32_t* n_per_seq, int use_graph);   /* sequence q at its own n */
int  gten_host_batch_decode_result(gten_host_batch* b, int seq, int n, int32_t* argmax_out); /* waits */
int  gten_host_batch_logits(gten_host_batch* b, int seq, float* logits_out);                /* waits; f32[n_vocab] */
int  gten_host_batch_time_family(gten_host_batch* b, int family, int n, int reps, double* avg_us, int* launches);
/* steps n_first .. n_first + steps - 1 of ONE sequence on its own single-sequence decoder (the caches are the ones the shared
 * decoder uses: whatever it appends there, the shared decoder must see -- tests/test_kv_head_major_gpu.py); asynchronous */
int  gten_host_batch_seq_steps(gten_host_batch* b, int seq, const int32_t* tokens, int count, int n_first, int steps);
/* the shared decoder's head-major K / V shadows (gten_hip_decoder_kv_info): kept at all, sequence imports so far, their launches */
int  gten_host_batch_kv_info(gten_host_batch* b, int* head_major, unsigned long long* seq_imports, unsigned long long* import_launches);

/* HIP-event timing of one kernel family of the decode step (see gten_hip_decoder_time_family) */
int gten_host_model_time_family(gten_host_model* m, int family, int n, int reps, double* avg_us, int* launches);

/* synthetic weight tensor `idx` of a model with config `cfg`, in storage layout */
int gten_host_synth_weight(const gten_host_config* cfg, uint64_t seed, int idx, void* out, size_t nbytes);
/* the same weights as a .gten file (tinyllama_to_gten.py:94-201 layout) */
int gten_host_write_gten(const gten_host_config* cfg, uint64_t seed, const char* path);
/* [1] + LCG token ids, SURVEY 8(d) */
void gten_host_synthetic_tokens(int32_t* out, int count, uint32_t seed, int n_vocab);

#ifdef __cplusplus
}
#endif
#endif
