/*
 * gten_oracle.h -- CPU restatement of tinyllama.cpp's gten forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the
 * reported CPU baseline.  The product (tinyllama.cpp_amd/) never links it.
 *
 * Every function names the reference file:line (relative to the upstream
 * repository root) whose behaviour it restates.  Parity of this restatement
 * is pinned against the real reference compiled into oracle/_ref (see
 * oracle/Makefile, tests/test_oracle_vs_ref.py) and against the committed
 * fixtures in tests/golden/ that were produced by that reference build.
 */
#ifndef GTEN_ORACLE_H
#define GTEN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dtype codes follow `enum class Dtype` order, gten/gten_types.h:20-26 */
enum { ORC_I32 = 0, ORC_F16 = 1, ORC_F32 = 2, ORC_Q8 = 3, ORC_Q4 = 4 };

/* 1 (default): float/int partial sums are kept in the same lane structure as
 * the reference's AVX/SSE build (README.md:25 flags).  0: the strictly
 * sequential order of its scalar build (README.md:17). */
void orc_set_simd(int avx_order);
int  orc_get_simd(void);

/* gten/gten_types.h:79-119 */
uint16_t orc_fp32_to_fp16(float f);
float    orc_fp16_to_fp32(uint16_t h);

/* bytes of one row of `cols` elements in storage dtype (gten/tensor.cpp:37-57,
 * gten/tensor.h:97-117).  Q8 rounds a partial last block up. */
size_t orc_row_bytes(int dtype, int cols);

/* activation codec, gten/quants.h:52-143, gten/ops.h:40-96 */
void orc_q8_quantize_row(const float* x, void* out, int n);
void orc_q8_dequantize_row(const void* in, float* out, int n);
void orc_q4_dequantize_row(const void* in, float* out, int n);
void orc_read_row(const void* in, int dtype, float* out, int n);
void orc_write_row(const float* in, void* out, int dtype, int n);

/* offline weight quantizers, tinyllama_to_gten.py:24-148 (round half to even) */
void orc_weight_to_f16(const float* w, size_t numel, void* out);
void orc_weight_quantize_q8(const float* w, int rows, int cols, void* out);
void orc_weight_quantize_q4(const float* w, int rows, int cols, void* out);

/* gten/ops.h:140-512 */
float orc_vec_dot(const void* a, int a_dtype, const void* b, int b_dtype, int n);

/* the ten operator entry points modules.cpp calls, gten/ops.h:554-1133.
 * Pitches are in bytes.  Rows [start_pos, n) are computed. */
void orc_token_embed(const void* w, int w_dtype, size_t w_pitch, const int32_t* tokens,
                     void* out, int out_dtype, size_t out_pitch, int n, int d, int start_pos);
void orc_matmul_2d(const void* x, int x_dtype, size_t x_pitch,
                   const void* w, int w_dtype, size_t w_pitch,
                   void* out, int out_dtype, size_t out_pitch,
                   int n, int d_in, int d_out, int start_pos);
void orc_rms_norm(const void* x, int dtype, size_t x_pitch, const uint16_t* w_f16,
                  void* out, size_t out_pitch, int n, int d, int start_pos);
void orc_rotary_emb(void* x, int dtype, size_t pitch, int n, int d, int d_head, int start_pos);
void orc_silu(const void* x, void* out, int dtype, size_t pitch, int n, int d, int start_pos);
void orc_mul(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos);
void orc_add(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos);
/* q [n][n_heads*d_head], k,v [n][n_kv*d_head], out [n][n_heads*d_head], all `dtype`.
 * The reference's materialised probability tensor (and its stride quirk,
 * gten/ops.h:946-947,1106) is NOT reproduced: probabilities are rounded to the
 * activation dtype row by row exactly as a quirk-free run (max_ctx >= 2n) does. */
void orc_qkv_attn(const void* q, const void* k, const void* v, void* out, int dtype,
                  size_t q_pitch, size_t kv_pitch, size_t out_pitch,
                  int n, int n_heads, int n_kv_heads, int d_head, int start_pos);

/* ---- whole model (tinyllama.cpp:12-61, gten/modules.cpp:11-254) ---- */
typedef struct orc_model orc_model;
typedef struct {
    int n_vocab, max_ctx, n_embd, n_ffn, n_layers, n_heads, n_kv_heads;
    int wdtype, adtype;
} orc_config;

orc_model* orc_model_create(const orc_config* cfg);
void       orc_model_free(orc_model* m);
/* number of weight tensors (1 + 9*L + 2) in .gten order, tinyllama.cpp:345-391 */
int        orc_model_n_weights(const orc_model* m);
size_t     orc_model_weight_bytes(const orc_model* m, int idx);
void       orc_model_weight_shape(const orc_model* m, int idx, int* rows, int* cols, int* dtype);
/* copies `bytes` (already in storage layout) into weight `idx` */
void       orc_model_set_weight(orc_model* m, int idx, const void* bytes, size_t nbytes);
/* returns 0 on success; reads a .gten file (tinyllama.cpp:336-392) */
int        orc_model_load_gten(orc_model* m, const char* path);
/* logits of the last row into out[n_vocab]; same calling convention as
 * TinyLlama::logits(tokens, start_pos) */
void       orc_model_logits(orc_model* m, const int32_t* tokens, int n, int start_pos, float* out);

#ifdef __cplusplus
}
#endif
#endif
