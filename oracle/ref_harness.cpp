// ref_harness.cpp -- thin extern "C" shim around the REAL reference sources.
//
// Built a second time with -DGTEN_DROPIN (oracle/Makefile, target `dropin`): the
// reference's UNMODIFIED tinyllama.cpp (its TinyLlama class, .gten loader and
// module wiring) is then compiled against THIS repository's HBM-backed gten
// headers instead of its own gten/ directory and linked with libgten_hip.so --
// the drop-in boundary exercised by the reference's own caller code
// (tests/test_dropin_gpu.py).  Only the model-level exports exist in that build.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit is compiled only where
// /root/reference exists (oracle/Makefile, target `ref`); it textually includes
// the reference's single translation unit from there (never copied into this
// repository) and exposes its operators / model behind the same C signatures
// as oracle/gten_oracle.h, so tests can run "oracle vs reference" on identical
// buffers and so bench.py can time the reference's own AVX/OpenMP path as the
// CPU baseline (cpu_baseline.kind = "reference").
//
// The reference is a header-only unity build with non-inline definitions
// (gten/ops.h has no include guard), so everything must live in this one TU.

#define main reference_main
#include "tinyllama.cpp"          // resolved with -I/root/reference
#undef main

#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

using gten::Tensor;
using gten::Dtype;

namespace {

Dtype to_dtype(int code)
{
    switch (code) {
    case 0: return gten::kInt32;
    case 1: return gten::kFloat16;
    case 2: return gten::kFloat32;
    case 3: return gten::kQint8;
    case 4: return gten::kQint4;
    }
    std::fprintf(stderr, "ref_harness: bad dtype code %d\n", code);
    std::abort();
}

size_t row_bytes(int dtype, int cols)
{
    switch (dtype) {
    case 0: case 2: return (size_t)cols * 4;
    case 1: return (size_t)cols * 2;
    case 3: return (size_t)((cols + 31) / 32) * sizeof(gten::Q8Block);
    case 4: return (size_t)(cols / 32) * sizeof(gten::Q4Block);
    }
    return 0;
}

void require_dense(size_t pitch, int dtype, int cols, const char* what)
{
    if (pitch != row_bytes(dtype, cols)) {
        std::fprintf(stderr, "ref_harness: %s must be densely packed (pitch %zu != %zu)\n",
                     what, pitch, row_bytes(dtype, cols));
        std::abort();
    }
}

// non-owning view over caller memory (gten/tensor.cpp:74-86)
Tensor view2(const void* p, int rows, int cols, int dtype) { return Tensor(p, {rows, cols}, to_dtype(dtype)); }

} // namespace

extern "C" {

#ifndef GTEN_DROPIN   // ---- operator-level wrappers: need the reference's own gten/ops.h internals

int ref_built_with_avx(void)
{
#if defined(__AVX__) && defined(__F16C__)
    return 1;
#else
    return 0;
#endif
}

uint16_t ref_fp32_to_fp16(float f) { return gten::fp32_to_fp16(f); }
float    ref_fp16_to_fp32(uint16_t h) { return gten::fp16_to_fp32(h); }

void ref_q8_quantize_row(const float* x, void* out, int n)
{
    gten::ops::q8_quantize_row(x, reinterpret_cast<gten::Q8Block*>(out), n);
}
void ref_q8_dequantize_row(const void* in, float* out, int n)
{
    gten::ops::q8_dequantize_row(reinterpret_cast<const gten::Q8Block*>(in), out, n);
}
void ref_q4_dequantize_row(const void* in, float* out, int n)
{
    gten::ops::q4_dequantize_row(reinterpret_cast<const gten::Q4Block*>(in), out, n);
}

float ref_vec_dot(const void* a, int a_dtype, const void* b, int b_dtype, int n)
{
    return gten::ops::vec_dot_product((const char*)a, to_dtype(a_dtype), (const char*)b, to_dtype(b_dtype), n);
}

void ref_token_embed(const void* w, int w_dtype, size_t w_pitch, const int32_t* tokens,
                     void* out, int out_dtype, size_t out_pitch, int n, int d, int start_pos)
{
    require_dense(w_pitch, w_dtype, d, "embedding table");
    require_dense(out_pitch, out_dtype, d, "embedding output");
    int max_tok = 0;
    for (int i = 0; i < n; i++) max_tok = tokens[i] > max_tok ? tokens[i] : max_tok;
    Tensor wt = view2(w, max_tok + 1, d, w_dtype);
    Tensor tk(tokens, {n}, gten::kInt32);
    Tensor ot = view2(out, n, d, out_dtype);
    gten::ops::token_embed(wt, tk, ot, start_pos);
}

void ref_matmul_2d(const void* x, int x_dtype, size_t x_pitch,
                   const void* w, int w_dtype, size_t w_pitch,
                   void* out, int out_dtype, size_t out_pitch,
                   int n, int d_in, int d_out, int start_pos)
{
    require_dense(x_pitch, x_dtype, d_in, "matmul input");
    require_dense(w_pitch, w_dtype, d_in, "matmul weight");
    require_dense(out_pitch, out_dtype, d_out, "matmul output");
    Tensor xt = view2(x, n, d_in, x_dtype);
    Tensor wt = view2(w, d_out, d_in, w_dtype);
    Tensor ot = view2(out, n, d_out, out_dtype);
    gten::ops::matmul_2d(xt, wt, ot, start_pos);
}

void ref_rms_norm(const void* x, int dtype, size_t x_pitch, const uint16_t* w_f16,
                  void* out, size_t out_pitch, int n, int d, int start_pos)
{
    require_dense(x_pitch, dtype, d, "rms_norm input");
    require_dense(out_pitch, dtype, d, "rms_norm output");
    Tensor xt = view2(x, n, d, dtype);
    Tensor wt(w_f16, {d}, gten::kFloat16);
    Tensor ot = view2(out, n, d, dtype);
    gten::ops::rms_norm(xt, wt, ot, start_pos);
}

void ref_rotary_emb(void* x, int dtype, size_t pitch, int n, int d, int d_head, int start_pos)
{
    require_dense(pitch, dtype, d, "rotary input");
    Tensor xt = view2(x, n, d, dtype);
    gten::ops::rotary_emb(xt, d_head, start_pos);
}

void ref_silu(const void* x, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    require_dense(pitch, dtype, d, "silu input");
    Tensor xt = view2(x, n, d, dtype);
    Tensor ot = view2(out, n, d, dtype);
    if (x == out) gten::ops::silu_inplace(xt, start_pos);
    else gten::ops::silu(xt, ot, start_pos);
}

void ref_mul(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    require_dense(pitch, dtype, d, "mul input");
    Tensor at = view2(a, n, d, dtype), bt = view2(b, n, d, dtype), ot = view2(out, n, d, dtype);
    if (a == out) gten::ops::mul_inplace(at, bt, start_pos);
    else gten::ops::mul(at, bt, ot, start_pos);
}

void ref_add(const void* a, const void* b, void* out, int dtype, size_t pitch, int n, int d, int start_pos)
{
    require_dense(pitch, dtype, d, "add input");
    Tensor at = view2(a, n, d, dtype), bt = view2(b, n, d, dtype), ot = view2(out, n, d, dtype);
    gten::ops::add(at, bt, ot, start_pos);
}

// The probability scratch is allocated here with max_ctx >= 2n so that the
// reference's element-stride-as-byte-offset quirk (gten/ops.h:946-947, 996,
// 1106) cannot make rows overlap; that quirk-free behaviour is the oracle.
void ref_qkv_attn(const void* q, const void* k, const void* v, void* out, int dtype,
                  size_t q_pitch, size_t kv_pitch, size_t out_pitch,
                  int n, int n_heads, int n_kv_heads, int d_head, int start_pos)
{
    const int d = n_heads * d_head, kv = n_kv_heads * d_head;
    require_dense(q_pitch, dtype, d, "attention q");
    require_dense(kv_pitch, dtype, kv, "attention k/v");
    require_dense(out_pitch, dtype, d, "attention out");
    int max_ctx = 64;
    while (max_ctx < 2 * n + 64) max_ctx *= 2;
    Tensor qt = view2(q, n, d, dtype), kt = view2(k, n, kv, dtype), vt = view2(v, n, kv, dtype);
    Tensor ot = view2(out, n, d, dtype);
    Tensor qk({n_heads, max_ctx, max_ctx}, to_dtype(dtype));
    qk.resize({n_heads, n, n});
    gten::ops::qkv_attn(qt, kt, vt, qk, ot, max_ctx, start_pos);
}

#endif // !GTEN_DROPIN

// ---- model assembled from the reference's own modules with arbitrary dims
// (TinyLLamaParams is hard-coded, tinyllama.cpp:12-20; the wiring below is the
// one in TinyLlama's ctor and logits(), tinyllama.cpp:30-61).
struct ref_config {
    int n_vocab, max_ctx, n_embd, n_ffn, n_layers, n_heads, n_kv_heads;
    int wdtype, adtype;
};

struct ref_model {
    ref_config c;
    gten::ModuleDtype md;
    gten::Embedding emb;
    gten::RMSNorm norm;
    gten::EmbeddingLinear head;
    std::vector<gten::AttentionBlock> blocks;
    ref_model(const ref_config& cfg)
        : c(cfg), md{to_dtype(cfg.wdtype), to_dtype(cfg.adtype)},
          emb(cfg.n_vocab, cfg.n_embd, cfg.max_ctx, md),
          norm(cfg.n_embd, cfg.max_ctx, {gten::kFloat16, md.adtype}),
          head(cfg.n_embd, cfg.n_vocab, cfg.max_ctx, {md.wdtype, gten::kFloat32})
    {
        blocks.reserve(cfg.n_layers);
        for (int i = 0; i < cfg.n_layers; i++)
            blocks.push_back(gten::AttentionBlock(cfg.n_heads, cfg.n_embd, cfg.n_kv_heads, cfg.n_ffn, cfg.max_ctx, md));
    }
    Tensor& weight(int idx)
    {
        const int last = 1 + 9 * c.n_layers + 1;
        if (idx == 0) return emb.weight;
        if (idx == last) return head.weight;
        if (idx == last - 1) return norm.weight;
        gten::AttentionBlock& b = blocks[(idx - 1) / 9];
        switch ((idx - 1) % 9) {
        case 0: return b.attn.query.weight;
        case 1: return b.attn.key.weight;
        case 2: return b.attn.value.weight;
        case 3: return b.attn.qkv_proj.weight;
        case 4: return b.ffn_gate_proj.weight;
        case 5: return b.ffn_up_proj.weight;
        case 6: return b.ffn_down_proj.weight;
        case 7: return b.attn_norm.weight;
        default: return b.ffn_norm.weight;
        }
    }
};

ref_model* ref_model_create(const ref_config* cfg) { return new ref_model(*cfg); }
void ref_model_free(ref_model* m) { delete m; }
int ref_model_n_weights(const ref_model* m) { return 1 + 9 * m->c.n_layers + 2; }
size_t ref_model_weight_bytes(ref_model* m, int idx) { return m->weight(idx).nbytes(); }

void ref_model_set_weight(ref_model* m, int idx, const void* bytes, size_t nbytes)
{
    Tensor& w = m->weight(idx);
    if (nbytes != w.nbytes()) {
        std::fprintf(stderr, "ref_model_set_weight: weight %d expects %zu bytes, got %zu\n", idx, w.nbytes(), nbytes);
        std::abort();
    }
    std::memcpy(w.data_ptr<char>(), bytes, nbytes);
}

void ref_model_logits(ref_model* m, const int32_t* tokens, int n, int start_pos, float* out)
{
    Tensor tk(tokens, {n}, gten::kInt32);
    Tensor t = m->emb.forward(tk, start_pos);
    for (auto& b : m->blocks) t = b.forward(t, start_pos);
    t = m->norm.forward(t, start_pos);
    t = m->head.forward(t);
    std::memcpy(out, t.data_ptr<float>(), (size_t)m->c.n_vocab * sizeof(float));
}

// The first `n_blocks` blocks only, then the caller LOOKS at the activation row (host read): with the drop-in
// headers this interrupts a recorded single-row forward half way (gten/modules.h), which must then produce what
// the operators produce.  Copies the last row's storage bytes to `out`; returns their count.
size_t ref_model_partial_row(ref_model* m, const int32_t* tokens, int n, int start_pos, int n_blocks, void* out, size_t cap)
{
    Tensor tk(tokens, {n}, gten::kInt32);
    Tensor t = m->emb.forward(tk, start_pos);
    for (int i = 0; i < n_blocks; i++) t = m->blocks[(size_t)i].forward(t, start_pos);
    const size_t pitch = (size_t)t.bstride(0);
    if (pitch > cap) return 0;
    const Tensor& ct = t;
    std::memcpy(out, ct.data_ptr<char>() + (size_t)(n - 1) * pitch, pitch);
    return pitch;
}

// ---- the reference's own TinyLlama class (full-size, tinyllama.cpp:23-76)
struct ref_tinyllama { TinyLlama model; ref_tinyllama(int n_ctx, gten::ModuleDtype md) : model(n_ctx, md) {} };

ref_tinyllama* ref_tl_create(int n_ctx, int wdtype, int adtype)
{
    return new ref_tinyllama(n_ctx, gten::ModuleDtype{to_dtype(wdtype), to_dtype(adtype)});
}
void ref_tl_free(ref_tinyllama* t) { delete t; }
int ref_tl_load(ref_tinyllama* t, const char* path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    t->model.load_from_ckpt(f);
    return 0;
}
void ref_tl_logits(ref_tinyllama* t, const int32_t* tokens, int n, int start_pos, float* out)
{
    Tensor tk(tokens, {n}, gten::kInt32);
    Tensor lg = t->model.logits(tk, start_pos);
    std::memcpy(out, lg.data_ptr<float>(), (size_t)lg.numel() * sizeof(float));
}

#ifdef GTEN_DROPIN
// drop-in build only: the single-row recording of this repository's gten/modules.h on / off (tests compare both)
void ref_set_fused_rows(int on) { gten::detail::fused_rows_enabled() = on != 0; }

// The reference's greedy loop (tinyllama.cpp:395-440) on token ids -- its TinyLlama::logits() per token, the
// logits read through data_ptr<float>() and the argmax on the host, exactly as upstream; only the tokenizer and
// the printing are left out.  Returns the total number of ids in `tokens` (capacity n_predict).
int ref_tl_greedy(ref_tinyllama* t, int32_t* tokens, int n_prompt, int n_predict, int eos)
{
    int n = n_prompt;
    const int max_iters = n_predict - n_prompt;
    for (int i = 0; i < max_iters; i++) {
        Tensor input{tokens, {n}, gten::kInt32};
        const int start_pos = (i == 0) ? 0 : input.numel() - 1;
        Tensor logits = t->model.logits(input, start_pos);
        const int logits_size = logits.numel();
        const float* logits_data = logits.data_ptr<float>();
        float max_prob = -std::numeric_limits<float>::infinity();
        int max_index = 0;
        for (int j = 0; j < logits_size; ++j) {
            const float val = logits_data[j];
            if (val > max_prob) { max_prob = val; max_index = j; }
        }
        if (max_index == eos) break;
        tokens[n++] = max_index;
    }
    return n;
}
#endif

// ---- the reference's own Tokenizer (tokenizer.h, included by tinyllama.cpp): pins for host/tokenizer.h
struct ref_tokenizer { Tokenizer tok; ref_tokenizer(const char* path, int vocab) : tok(path, vocab) {} };
ref_tokenizer* ref_tok_create(const char* path, int vocab_size) { return new ref_tokenizer(path, vocab_size); }
void ref_tok_free(ref_tokenizer* t) { delete t; }
int ref_tok_encode(ref_tokenizer* t, const char* prompt, int32_t* ids_out, int cap)
{
    std::string p(prompt);
    const std::vector<int> ids = t->tok.encode(p);
    if ((int)ids.size() > cap) return -(int)ids.size();
    for (size_t i = 0; i < ids.size(); i++) ids_out[i] = ids[i];
    return (int)ids.size();
}
const char* ref_tok_decode(ref_tokenizer* t, int prev_token, int token) { return t->tok.decode(prev_token, token); }

} // extern "C"
