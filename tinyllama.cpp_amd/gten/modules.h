// modules.h -- the stateful module layer of the gten API, HBM-backed.
//
// Same classes, constructors, forward() signatures and public members as the
// reference's gten/modules.h:11-192 (TinyLlama reaches into public members such
// as block.attn.query.weight and *.exec_time, tinyllama.cpp:346-391, 522-559),
// so a model written against the reference's modules compiles against these.
// Each forward() = Timer + acv.resize + one ops:: call, as gten/modules.cpp:11-254;
// the ops:: calls land in hand-written gfx950 kernels (ops.h).
//
// The K/V cache is, as in the reference, the activation tensor of the key and
// value projections (`attn.key.acv`, `attn.value.acv`: [max_ctx][kv_dim] in the
// activation dtype; rows below start_pos persist between calls).
//
// One new row (decode): when forward() is called with exactly one new row on Embedding, then on each
// AttentionBlock with the previous module's output, then on RMSNorm and EmbeddingLinear -- the sequence of
// TinyLlama::logits (tinyllama.cpp:45-61) -- the calls are RECORDED and the whole row runs as one fused decoder
// step (include/gten_hip.h "single-token decode fast path": 5 launches per block from one hipGraph instead of
// ~16 operator launches), on the same HBM tensors: weights, K/V caches (= attn.key.acv / attn.value.acv) and the
// logits buffer (= EmbeddingLinear::acv).  Same bytes as the operator path for contexts up to 256 rows, f32
// summation-order noise of the chunked softmax beyond (tests/test_dropin_gpu.py).  Any other use -- a different
// call order, several new rows, someone reading an intermediate tensor, a configuration the decoder does not
// cover -- settles the recorded calls operator by operator first, so results are never skipped; the only
// observable difference is that after a fused row the intermediate activation tensors (emb_acv, residuals, ...)
// keep their previous contents.  GTEN_HIP_FAST_DECODE=0 switches the recording off.
//
// Timer semantics: kernel launches are asynchronous, so by default exec_time
// accumulates host launch time only.  Set GTEN_HIP_SYNC_TIMERS=1 to make every
// Timer wait for the GPU before it stops (debug / print_perf use).
#pragma once

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <vector>

#include "ops.h"
#include "tensor.h"

namespace gten {

struct ModuleDtype {
    Dtype wdtype;
    Dtype adtype;
};

// RAII millisecond accumulator (gten/modules.h:170-192).
class Timer {
public:
    explicit Timer(int64_t* time_tracker) : tracker_{time_tracker}, t0_{clock::now()} {}
    ~Timer() { stop(); }
    void stop()
    {
        if (stopped_) return;
        static const bool sync = [] { const char* e = std::getenv("GTEN_HIP_SYNC_TIMERS"); return e && e[0] == '1'; }();
        if (sync) gten_hip_sync();
        const auto t1 = clock::now();
        const int64_t a = std::chrono::time_point_cast<std::chrono::milliseconds>(t0_).time_since_epoch().count();
        const int64_t b = std::chrono::time_point_cast<std::chrono::milliseconds>(t1).time_since_epoch().count();
        *tracker_ += b - a;
        stopped_ = true;
    }

private:
    using clock = std::chrono::high_resolution_clock;
    int64_t* tracker_;
    std::chrono::time_point<clock> t0_;
    bool stopped_ = false;
};

class Embedding;
class RMSNorm;
class AttentionBlock;
class EmbeddingLinear;

namespace detail {

// The recorded single-row forward (see the header comment).  `tail` is the storage the next module of the chain
// must be handed; `replay` re-runs what has been recorded through the operators.
struct PendingRow {
    bool active = false;
    Embedding* emb = nullptr;
    std::vector<AttentionBlock*> blocks;
    RMSNorm* norm = nullptr;
    int n = 0;                               // rows in the context; the new row is n - 1
    int32_t token = 0;                       // id of the new row
    const void* tail = nullptr;
    std::vector<std::function<void()>> replay;
};

inline PendingRow& pending_row()
{
    static PendingRow p;
    return p;
}

inline bool& fused_rows_enabled()
{
    static bool on = [] { const char* e = std::getenv("GTEN_HIP_FAST_DECODE"); return !(e && e[0] == '0'); }();
    return on;
}

// materialise the recorded calls operator by operator (hook of tensor.h's settle_pending)
inline void settle_row()
{
    PendingRow& p = pending_row();
    g_pending_settle = nullptr;
    if (!p.active) return;
    p.active = false;
    std::vector<std::function<void()>> todo;
    todo.swap(p.replay);
    p.blocks.clear();
    p.norm = nullptr;
    for (auto& f : todo) f();
}

inline void record(PendingRow& p, const void* new_tail, std::function<void()> replay)
{
    p.tail = new_tail;
    p.replay.push_back(std::move(replay));
}

inline bool run_fused_row(PendingRow& p, EmbeddingLinear* head);     // defined below the module classes
inline void forget_decoder(const EmbeddingLinear* head);

// A module that is part of the recorded row goes away (a model destroyed between Embedding::forward and lm_head): the
// recording holds its address and closures that call it.  The recorded calls are run NOW, while the module is still
// alive -- nothing is skipped, nothing dangles.  (Copies of modules -- a vector of AttentionBlocks being filled -- are
// never part of a recording: a recording only ever holds the addresses forward() was called on.)
inline void module_gone(const void* m)
{
    PendingRow& p = pending_row();
    if (!p.active) return;
    bool mine = (const void*)p.emb == m || (const void*)p.norm == m;
    for (const AttentionBlock* b : p.blocks) mine = mine || (const void*)b == m;
    if (mine) settle_row();
}

} // namespace detail

/// Embedding table lookup, tokens (n_ctx,) -> (n_ctx, d_embed).  gten/modules.cpp:11-26
class Embedding {
public:
    Embedding() = default;
    Embedding(int n_vocab, int d_embed, int max_ctx, ModuleDtype dtype)
        : weight{Tensor({n_vocab, d_embed}, dtype.wdtype)}, emb_acv{Tensor({max_ctx, d_embed}, dtype.adtype)}
    {
    }
    Embedding(const Embedding&) = default;
    Embedding& operator=(const Embedding&) = default;
    ~Embedding() { detail::module_gone(this); }
    Tensor forward(const Tensor& tokens, const int start_pos = 0)
    {
        detail::settle_pending();
        const int n = tokens.numel();
        if (detail::fused_rows_enabled() && n - start_pos == 1 && tokens.is_host_external() && tokens.dtype() == kInt32) {
            // one new row: start recording (header comment).  The id is copied now; the caller's buffer is not
            // needed again.
            detail::PendingRow& p = detail::pending_row();
            p.emb = this;
            p.n = n;
            p.token = static_cast<const int32_t*>(tokens.host_external_ptr())[n - 1];
            p.blocks.clear();
            p.norm = nullptr;
            p.replay.clear();
            emb_acv.resize({n, weight.dimsize(1)});
            const int32_t id = p.token;
            detail::record(p, emb_acv.storage_id(), [this, n, id, start_pos] {
                std::vector<int32_t> ids((size_t)n, 0);
                ids[(size_t)n - 1] = id;                       // ops::token_embed reads rows [start_pos, n) only
                forward_now(Tensor(ids.data(), {n}, kInt32), start_pos);
            });
            p.active = true;
            detail::g_pending_settle = &detail::settle_row;
            return emb_acv;
        }
        return forward_now(tokens, start_pos);
    }

private:
    Tensor forward_now(const Tensor& tokens, const int start_pos)
    {
        Timer timer{&exec_time};
        emb_acv.resize({tokens.numel(), weight.dimsize(1)});
        ops::token_embed(weight, tokens, emb_acv, start_pos);
        return emb_acv;
    }

public:
    Tensor weight;
    Tensor emb_acv;
    int64_t exec_time{0};
};

/// gten/modules.cpp:83-100; the weight is always fp16.
class RMSNorm {
public:
    RMSNorm(int d_in, int max_ctx, ModuleDtype dtype)
        : weight{Tensor({d_in}, kFloat16)}, acv{Tensor({max_ctx, d_in}, dtype.adtype)}
    {
    }
    RMSNorm(const RMSNorm&) = default;
    RMSNorm& operator=(const RMSNorm&) = default;
    ~RMSNorm() { detail::module_gone(this); }
    Tensor forward(const Tensor& inp, const int start_pos = 0)
    {
        detail::PendingRow& p = detail::pending_row();
        if (p.active && !p.norm && !p.blocks.empty() && inp.storage_id() == p.tail && inp.dimsize(0) == p.n && start_pos == p.n - 1) {
            p.norm = this;                                     // the final norm of the recorded row
            acv.resize({inp.dimsize(0), inp.dimsize(1)});
            Tensor in = inp;
            detail::record(p, acv.storage_id(), [this, in, start_pos] { forward_now(in, start_pos); });
            return acv;
        }
        detail::settle_pending();
        return forward_now(inp, start_pos);
    }

private:
    Tensor forward_now(const Tensor& inp, const int start_pos)
    {
        Timer timer{&exec_time};
        acv.resize({inp.dimsize(0), inp.dimsize(1)});
        ops::rms_norm(inp, weight, acv, start_pos);
        return acv;
    }

public:
    Tensor weight;
    Tensor acv;
    int64_t exec_time{0};
};

/// gten/modules.cpp:28-43
class Residual {
public:
    Residual() = default;
    Residual(int max_ctx, int d_out, Dtype dtype) : acv{Tensor({max_ctx, d_out}, dtype)} {}
    Tensor forward(const Tensor& inp0, const Tensor& inp1, const int start_pos = 0)
    {
        Timer timer{&exec_time};
        acv.resize({inp0.dimsize(0), inp0.dimsize(1)});
        ops::add(inp0, inp1, acv, start_pos);
        return acv;
    }

public:
    Tensor acv;
    int64_t exec_time{0};
};

/// y = x W^T, no bias.  gten/modules.cpp:45-63
class Linear {
public:
    Linear() = default;
    Linear(int d_in, int d_out, int max_ctx, ModuleDtype dtype)
        : weight{Tensor({d_out, d_in}, dtype.wdtype)}, acv{Tensor({max_ctx, d_out}, dtype.adtype)}, max_ctx_{max_ctx}
    {
    }
    Tensor forward(const Tensor& inp, const int start_pos = 0)
    {
        Timer timer{&exec_time};
        acv.resize({inp.dimsize(0), weight.dimsize(0)});
        ops::matmul_2d(inp, weight, acv, start_pos);
        return acv;
    }

    int max_ctx() const { return max_ctx_; }

public:
    Tensor weight;
    Tensor acv;
    int64_t exec_time{0};

private:
    int max_ctx_{0};
    bool has_bias_{false};
};

/// lm_head: logits of the LAST row only, f32 (n_vocab,).  gten/modules.cpp:65-81
class EmbeddingLinear {
public:
    EmbeddingLinear() = default;
    EmbeddingLinear(int n_embd, int n_vocab, int /*max_ctx*/, ModuleDtype dtype)
        : weight{Tensor({n_vocab, n_embd}, dtype.wdtype)}, acv{Tensor({n_vocab}, kFloat32)}
    {
    }
    EmbeddingLinear(const EmbeddingLinear&) = default;
    EmbeddingLinear& operator=(const EmbeddingLinear&) = default;
    ~EmbeddingLinear() { detail::forget_decoder(this); }

    Tensor forward(const Tensor& inp)
    {
        detail::PendingRow& p = detail::pending_row();
        if (p.active && p.norm && inp.storage_id() == p.tail && inp.dimsize(0) == p.n) {
            Timer timer{&exec_time};
            if (detail::run_fused_row(p, this)) return acv;    // the whole row as one decoder step
        }
        detail::settle_pending();
        Timer timer{&exec_time};
        ops::matmul_2d(inp, weight, acv, inp.dimsize(0) - 1);
        return acv;
    }

public:
    Tensor weight;
    Tensor acv;
    int64_t exec_time{0};
};

/// gten/modules.cpp:102-130
class Multiply {
public:
    Multiply() = default;
    Multiply(int max_ctx, int d_out, Dtype dtype, const bool inplace = false) : inplace_{inplace}
    {
        if (!inplace) acv = Tensor({max_ctx, d_out}, dtype);
    }
    Tensor forward(Tensor& inp0, const Tensor& inp1, const int start_pos = 0)
    {
        Timer timer{&exec_time};
        if (inplace_) {
            ops::mul_inplace(inp0, inp1, start_pos);
            return inp0;
        }
        acv.resize({inp0.dimsize(0), inp0.dimsize(1)});
        ops::mul(inp0, inp1, acv, start_pos);
        return acv;
    }

public:
    Tensor acv;
    int64_t exec_time{0};

private:
    bool inplace_{false};
};

/// gten/modules.cpp:132-158
class SiLU {
public:
    SiLU() = default;
    SiLU(int max_ctx, int d_out, Dtype dtype, const bool inplace = false) : inplace_{inplace}
    {
        if (!inplace) acv = Tensor({max_ctx, d_out}, dtype);
    }
    Tensor forward(Tensor& inp, const int start_pos = 0)
    {
        Timer timer{&exec_time};
        if (inplace_) {
            ops::silu_inplace(inp, start_pos);
            return inp;
        }
        acv.resize({inp.dimsize(0), inp.dimsize(1)});
        ops::silu(inp, acv, start_pos);
        return acv;
    }

public:
    Tensor acv;
    bool inplace_{false};
    int64_t exec_time{0};
};

/// In-place only, like the reference (gten/modules.cpp:158-174).
class RotaryEmbedding {
public:
    RotaryEmbedding(const int d_head, const bool inplace = true) : d_head_{d_head}
    {
        GTEN_ASSERTM(inplace, "RotaryEmbedding inplace not implemented.");
    }
    Tensor forward(Tensor& inp, const int start_pos = 0)
    {
        Timer timer{&exec_time};
        ops::rotary_emb(inp, d_head_, start_pos);
        return inp;
    }

public:
    int64_t exec_time{0};

private:
    int d_head_;
};

/// Causal grouped-query self-attention.  gten/modules.cpp:177-222
class SelfAttention {
public:
    SelfAttention(int n_heads, int n_embed, int n_query_groups, int max_ctx, ModuleDtype dtype)
        : query{Linear(n_embed, n_embed, max_ctx, dtype)},
          key{Linear(n_embed, (n_embed / n_heads) * n_query_groups, max_ctx, dtype)},
          value{Linear(n_embed, (n_embed / n_heads) * n_query_groups, max_ctx, dtype)},
          qkv_proj{Linear(n_embed, n_embed, max_ctx, dtype)},
          // kept for API parity; never touched by the kernels, so no HBM is ever
          // allocated for it (storage is created on first device use)
          qk_acv{Tensor({n_heads, max_ctx, max_ctx}, dtype.adtype)},
          qkv_acv{Tensor({max_ctx, n_embed}, dtype.adtype)},
          q_rope{RotaryEmbedding{n_embed / n_heads, /*inplace=*/true}},
          k_rope{RotaryEmbedding{n_embed / n_heads, /*inplace=*/true}},
          n_heads_{n_heads},
          max_ctx_{max_ctx}
    {
    }

    Tensor forward(const Tensor& inp, const int start_pos)
    {
        Tensor q = query.forward(inp, start_pos);
        Tensor k = key.forward(inp, start_pos);      // writes rows [start_pos, n) of the K cache
        q = q_rope.forward(q, start_pos);
        k = k_rope.forward(k, start_pos);            // K is cached post-RoPE
        Tensor v = value.forward(inp, start_pos);    // V cache
        const Tensor qkv = masked_qkv_attn(q, k, v, start_pos);
        return qkv_proj.forward(qkv, start_pos);
    }

public:
    Linear query;
    Linear key;
    Linear value;
    Linear qkv_proj;
    Tensor qk_acv;
    Tensor qkv_acv;
    RotaryEmbedding q_rope;
    RotaryEmbedding k_rope;
    int64_t exec_time_attn{0};

    int n_heads() const { return n_heads_; }
    int max_ctx() const { return max_ctx_; }

private:
    int32_t n_heads_;
    int max_ctx_;

    Tensor masked_qkv_attn(const Tensor& q, const Tensor& k, const Tensor& v, const int start_pos)
    {
        Timer timer{&exec_time_attn};
        const int n_ctx = q.dimsize(0);
        qk_acv.resize({n_heads_, n_ctx, n_ctx});
        qkv_acv.resize({n_ctx, q.dimsize(1)});
        ops::qkv_attn(q, k, v, qk_acv, qkv_acv, max_ctx_, start_pos);
        return qkv_acv;
    }
};

/// One transformer block.  gten/modules.cpp:224-254
class AttentionBlock {
public:
    AttentionBlock(int n_heads, int d_embed, int n_query_groups, int n_mlp, int max_ctx, ModuleDtype dtype)
        : attn_norm{RMSNorm(d_embed, max_ctx, dtype)},
          attn{SelfAttention(n_heads, d_embed, n_query_groups, max_ctx, dtype)},
          inp_res{Residual(max_ctx, d_embed, dtype.adtype)},
          ffn_norm{RMSNorm(d_embed, max_ctx, dtype)},
          ffn_gate_proj{Linear(d_embed, n_mlp, max_ctx, dtype)},
          ffn_up_proj{Linear(d_embed, n_mlp, max_ctx, dtype)},
          ffn_down_proj{Linear(n_mlp, d_embed, max_ctx, dtype)},
          attn_res{Residual(max_ctx, d_embed, dtype.adtype)},
          ffn_mul{Multiply(max_ctx, n_mlp, dtype.adtype, /*inplace=*/true)},
          ffn_silu{SiLU(max_ctx, n_mlp, dtype.adtype, /*inplace=*/true)}
    {
    }
    AttentionBlock(const AttentionBlock&) = default;
    AttentionBlock& operator=(const AttentionBlock&) = default;
    ~AttentionBlock() { detail::module_gone(this); }

    // down( silu(gate(x)) * up(x) ), SiLU and the product in place on the gate buffer
    Tensor ffn_forward(const Tensor& inp, const int start_pos = 0)
    {
        Tensor g = ffn_gate_proj.forward(inp, start_pos);
        const Tensor u = ffn_up_proj.forward(inp, start_pos);
        Tensor sg = ffn_silu.forward(g, start_pos);
        const Tensor prod = ffn_mul.forward(sg, u, start_pos);
        return ffn_down_proj.forward(prod, start_pos);
    }

    Tensor forward(Tensor& inp, const int start_pos)
    {
        detail::PendingRow& p = detail::pending_row();
        if (p.active && !p.norm && inp.storage_id() == p.tail && inp.dimsize(0) == p.n && start_pos == p.n - 1) {
            p.blocks.push_back(this);                          // next block of the recorded row
            attn_res.acv.resize({inp.dimsize(0), inp.dimsize(1)});
            Tensor in = inp;
            detail::record(p, attn_res.acv.storage_id(), [this, in, start_pos]() mutable { forward_now(in, start_pos); });
            return attn_res.acv;
        }
        detail::settle_pending();
        return forward_now(inp, start_pos);
    }

private:
    Tensor forward_now(Tensor& inp, const int start_pos)
    {
        if (inp.dimsize(0) - start_pos >= 16 && forward_rows(inp, start_pos)) return attn_res.acv;
        Tensor h = inp_res.forward(inp, attn.forward(attn_norm.forward(inp, start_pos), start_pos), start_pos);
        return attn_res.forward(h, ffn_forward(ffn_norm.forward(h, start_pos), start_pos), start_pos);
    }

    // Prompt-sized calls: the whole block as one library call (gten_hip_block_rows: the same kernels composed so that
    // matrices sharing an input share its launch, 12 launches instead of 23).  Every activation tensor of the modules --
    // the K / V caches among them -- ends with the bytes the module-by-module sequence above leaves in it.  Returns false
    // when the library does not take the configuration (an unusual head width, ...): the modules then run one by one.
    bool forward_rows(Tensor& inp, const int start_pos)
    {
        const int n = inp.dimsize(0), E = inp.dimsize(1), F = ffn_gate_proj.weight.dimsize(0), H = attn.n_heads();
        const Dtype adt = inp.dtype(), wdt = attn.query.weight.dtype();
        if (!inp.is_2d() || H <= 0 || E % H != 0 || ffn_silu.inplace_ == false) return false;
        const int dh = E / H, KV = attn.key.weight.dimsize(0);
        const Linear* lin[] = {&attn.query, &attn.key, &attn.value, &attn.qkv_proj, &ffn_gate_proj, &ffn_up_proj, &ffn_down_proj};
        for (const Linear* l : lin)
            if (l->weight.dtype() != wdt || l->acv.dtype() != adt) return false;
        if (attn_norm.acv.dtype() != adt || ffn_norm.acv.dtype() != adt || inp_res.acv.dtype() != adt || attn_res.acv.dtype() != adt ||
            attn.qkv_acv.dtype() != adt || KV % dh != 0 || attn.value.weight.dimsize(0) != KV)
            return false;
        if (!attn.query.weight.shape_eq({E, E}) || !attn.key.weight.shape_eq({KV, E}) || !attn.qkv_proj.weight.shape_eq({E, E}) ||
            !ffn_up_proj.weight.shape_eq({F, E}) || !ffn_down_proj.weight.shape_eq({E, F}) || attn_norm.weight.numel() != E || ffn_norm.weight.numel() != E)
            return false;
        if (!((adt == kQint8 && (wdt == kQint8 || wdt == kQint4)) || (adt == kFloat16 && wdt == kFloat16))) return false;
        Timer timer{&attn.qkv_proj.exec_time};
        // the shapes the modules give their outputs
        attn_norm.acv.resize({n, E}); attn.query.acv.resize({n, E}); attn.key.acv.resize({n, KV}); attn.value.acv.resize({n, KV});
        attn.qk_acv.resize({H, n, n}); attn.qkv_acv.resize({n, E}); attn.qkv_proj.acv.resize({n, E}); inp_res.acv.resize({n, E});
        ffn_norm.acv.resize({n, E}); ffn_gate_proj.acv.resize({n, F}); ffn_up_proj.acv.resize({n, F}); ffn_down_proj.acv.resize({n, E});
        attn_res.acv.resize({n, E});
        gten_hip_block_desc b{};
        b.adtype = dtype_code(adt); b.wdtype = dtype_code(wdt);
        b.n_embd = E; b.n_heads = H; b.n_kv_heads = KV / dh; b.n_ffn = F;
        b.attn_norm_w = attn_norm.weight.device_weight(); b.ffn_norm_w = ffn_norm.weight.device_weight();
        b.wq = attn.query.weight.device_weight(); b.wk = attn.key.weight.device_weight(); b.wv = attn.value.weight.device_weight();
        b.wo = attn.qkv_proj.weight.device_weight();
        b.wgate = ffn_gate_proj.weight.device_weight(); b.wup = ffn_up_proj.weight.device_weight(); b.wdown = ffn_down_proj.weight.device_weight();
        b.inp = inp.device_ptr();
        b.attn_norm_out = attn_norm.acv.device_ptr_mut();
        b.q = attn.query.acv.device_ptr_mut(); b.k = attn.key.acv.device_ptr_mut(); b.v = attn.value.acv.device_ptr_mut();
        b.attn_out = attn.qkv_acv.device_ptr_mut(); b.o = attn.qkv_proj.acv.device_ptr_mut(); b.h = inp_res.acv.device_ptr_mut();
        b.ffn_norm_out = ffn_norm.acv.device_ptr_mut();
        b.gate = ffn_gate_proj.acv.device_ptr_mut(); b.up = ffn_up_proj.acv.device_ptr_mut(); b.down = ffn_down_proj.acv.device_ptr_mut();
        b.out = attn_res.acv.device_ptr_mut();
        const int rc = gten_hip_block_rows(&b, n, start_pos);
        if (rc == GTEN_HIP_NOT_HANDLED) return false;
        GTEN_HIP_OK(rc);
        return true;
    }

public:
    RMSNorm attn_norm;
    SelfAttention attn;
    Residual inp_res;
    RMSNorm ffn_norm;
    Linear ffn_gate_proj;
    Linear ffn_up_proj;
    Linear ffn_down_proj;
    Residual attn_res;
    Multiply ffn_mul;
    SiLU ffn_silu;
};

namespace detail {

// One fused decoder per lm_head module, valid for exactly the device tensors it was built on.
struct RowDecoder {
    gten_hip_decoder* dec = nullptr;
    gten_hip_decoder_desc desc{};
    std::vector<gten_hip_layer_ptrs> layers;
    bool unsupported = false;                // the decoder refused this configuration: keep to the operators
};

inline std::map<const EmbeddingLinear*, RowDecoder>& row_decoders()
{
    static auto* m = new std::map<const EmbeddingLinear*, RowDecoder>();     // never destroyed: outlives the HIP runtime's users
    return *m;
}

inline void forget_decoder(const EmbeddingLinear* head)
{
    auto& m = row_decoders();
    auto it = m.find(head);
    if (it == m.end()) return;
    if (it->second.dec) gten_hip_decoder_destroy(it->second.dec);
    m.erase(it);
}

// Every device pointer the decoder works on, from the recorded modules (what TinyLlama::describe collects in
// host/tinyllama_model.h).  Returns false when the chain is not a model the decoder computes.
inline bool describe_row(const PendingRow& p, EmbeddingLinear* head, gten_hip_decoder_desc* d, std::vector<gten_hip_layer_ptrs>* L)
{
    const Tensor& ew = p.emb->weight;
    AttentionBlock& b0 = *p.blocks[0];
    if (!ew.is_2d() || !head->weight.is_2d()) return false;
    const int E = ew.dimsize(1), V = ew.dimsize(0);
    const int H = b0.attn.n_heads(), max_ctx = b0.attn.max_ctx();
    if (H <= 0 || E % H != 0) return false;
    const int dh = E / H;
    const Dtype wdt = b0.attn.query.weight.dtype(), adt = b0.attn.key.acv.dtype();
    *d = gten_hip_decoder_desc{};
    d->n_vocab = V; d->max_ctx = max_ctx; d->n_embd = E; d->n_ffn = b0.ffn_gate_proj.weight.dimsize(0);
    d->n_layers = (int)p.blocks.size(); d->n_heads = H; d->n_kv_heads = b0.attn.key.weight.dimsize(0) / dh;
    d->wdtype = dtype_code(wdt); d->adtype = dtype_code(adt);
    if (ew.dtype() != wdt || head->weight.dtype() != wdt || head->weight.dimsize(0) != V || head->weight.dimsize(1) != E) return false;
    if (p.emb->emb_acv.dtype() != adt || p.norm->acv.dtype() != adt || p.norm->weight.numel() != E) return false;
    if (head->acv.dtype() != kFloat32 || head->acv.numel() != V) return false;
    for (AttentionBlock* b : p.blocks) {
        const bool same = b->attn.n_heads() == H && b->attn.max_ctx() == max_ctx && b->attn.query.weight.dtype() == wdt &&
                          b->attn.key.acv.dtype() == adt && b->attn.query.weight.shape_eq({E, E}) &&
                          b->attn.key.weight.shape_eq({d->n_kv_heads * dh, E}) && b->attn.value.weight.shape_eq({d->n_kv_heads * dh, E}) &&
                          b->attn.qkv_proj.weight.shape_eq({E, E}) && b->ffn_gate_proj.weight.shape_eq({d->n_ffn, E}) &&
                          b->ffn_up_proj.weight.shape_eq({d->n_ffn, E}) && b->ffn_down_proj.weight.shape_eq({E, d->n_ffn}) &&
                          b->attn.key.max_ctx() == max_ctx && b->attn.value.max_ctx() == max_ctx;
        if (!same) return false;
    }
    d->embed = ew.device_weight();
    d->final_norm = p.norm->weight.device_weight();
    d->lm_head = head->weight.device_weight();
    d->logits = static_cast<float*>(head->acv.device_ptr_mut());
    L->assign(p.blocks.size(), gten_hip_layer_ptrs{});
    for (size_t i = 0; i < p.blocks.size(); i++) {
        AttentionBlock& b = *p.blocks[i];
        gten_hip_layer_ptrs& l = (*L)[i];
        l.wq = b.attn.query.weight.device_weight();
        l.wk = b.attn.key.weight.device_weight();
        l.wv = b.attn.value.weight.device_weight();
        l.wo = b.attn.qkv_proj.weight.device_weight();
        l.wgate = b.ffn_gate_proj.weight.device_weight();
        l.wup = b.ffn_up_proj.weight.device_weight();
        l.wdown = b.ffn_down_proj.weight.device_weight();
        l.attn_norm = b.attn_norm.weight.device_weight();
        l.ffn_norm = b.ffn_norm.weight.device_weight();
        l.kcache = b.attn.key.acv.device_ptr_mut();          // the K/V caches ARE these activation tensors; the step
        l.vcache = b.attn.value.acv.device_ptr_mut();        // writes row n-1 (marks the device copy as the newer one)
    }
    return true;
}

// The recorded row is complete (embedding, blocks, final norm, and now lm_head): run it as one decoder step.
// Returns false -- with the recorded calls settled through the operators -- when the decoder does not cover it.
inline bool run_fused_row(PendingRow& p, EmbeddingLinear* head)
{
    // from here on device pointers are fetched for the decoder itself: nothing is pending any more
    g_pending_settle = nullptr;
    p.active = false;
    std::vector<std::function<void()>> recorded;
    recorded.swap(p.replay);
    auto give_up = [&] {
        for (auto& f : recorded) f();
        p.blocks.clear();
        p.norm = nullptr;
        return false;
    };
    RowDecoder& rd = row_decoders()[head];
    if (rd.unsupported) return give_up();
    gten_hip_decoder_desc d;
    std::vector<gten_hip_layer_ptrs> L;
    if (!describe_row(p, head, &d, &L)) return give_up();
    // (field by field: the descriptor has padding between its ints and its pointers; the layer records are pointers only)
    const gten_hip_decoder_desc& o = rd.desc;
    const bool same_desc = d.n_vocab == o.n_vocab && d.max_ctx == o.max_ctx && d.n_embd == o.n_embd && d.n_ffn == o.n_ffn && d.n_layers == o.n_layers &&
                           d.n_heads == o.n_heads && d.n_kv_heads == o.n_kv_heads && d.wdtype == o.wdtype && d.adtype == o.adtype &&
                           d.embed == o.embed && d.final_norm == o.final_norm && d.lm_head == o.lm_head && d.logits == o.logits;
    const bool same = rd.dec && same_desc && L.size() == rd.layers.size() &&
                      std::memcmp(L.data(), rd.layers.data(), L.size() * sizeof(gten_hip_layer_ptrs)) == 0;
    if (!same) {
        if (rd.dec) gten_hip_decoder_destroy(rd.dec);
        rd.dec = nullptr;
        if (gten_hip_decoder_create(&d, L.data(), &rd.dec) != 0) {     // e.g. a head width the fused kernels do not have
            rd.dec = nullptr;
            rd.unsupported = true;
            return give_up();
        }
        rd.desc = d;
        rd.layers = L;
    }
    GTEN_HIP_OK(gten_hip_decoder_set_tokens(rd.dec, &p.token, p.n - 1, 1));
    GTEN_HIP_OK(gten_hip_decoder_step(rd.dec, p.n, /*use_graph=*/1));
    // the step appended row n - 1 to the caches: their tensors say so, as after the modules' own forward (gten/modules.cpp:196-201)
    for (AttentionBlock* b : p.blocks) {
        b->attn.key.acv.resize({p.n, b->attn.key.acv.dimsize(1)});
        b->attn.value.acv.resize({p.n, b->attn.value.acv.dimsize(1)});
    }
    p.blocks.clear();
    p.norm = nullptr;
    return true;
}

} // namespace detail

} // namespace gten
