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
// Timer semantics: kernel launches are asynchronous, so by default exec_time
// accumulates host launch time only.  Set GTEN_HIP_SYNC_TIMERS=1 to make every
// Timer wait for the GPU before it stops (debug / print_perf use).
#pragma once

#include <chrono>
#include <cstdlib>
#include <iostream>

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

/// Embedding table lookup, tokens (n_ctx,) -> (n_ctx, d_embed).  gten/modules.cpp:11-26
class Embedding {
public:
    Embedding() = default;
    Embedding(int n_vocab, int d_embed, int max_ctx, ModuleDtype dtype)
        : weight{Tensor({n_vocab, d_embed}, dtype.wdtype)}, emb_acv{Tensor({max_ctx, d_embed}, dtype.adtype)}
    {
    }
    Tensor forward(const Tensor& tokens, const int start_pos = 0)
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
    Tensor forward(const Tensor& inp, const int start_pos = 0)
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
    Tensor forward(const Tensor& inp)
    {
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
        Tensor h = inp_res.forward(inp, attn.forward(attn_norm.forward(inp, start_pos), start_pos), start_pos);
        return attn_res.forward(h, ffn_forward(ffn_norm.forward(h, start_pos), start_pos), start_pos);
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

} // namespace gten
