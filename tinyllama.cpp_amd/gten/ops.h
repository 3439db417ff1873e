// ops.h -- gten::ops:: operator API of the reference (gten/ops.h:554-1133),
// routed through the C-ABI of include/gten_hip.h into the gfx950 kernels.
//
// Each wrapper performs the same GTEN_ASSERT contract checks as the reference
// operator it replaces (cited per function), translates Tensors into raw HBM
// pointers + byte pitches, and calls one C function.  No arithmetic happens on
// the host.  Calls are asynchronous on the library's stream; reading a result
// through Tensor::data_ptr() synchronises.
#pragma once

#include <vector>

#include "../../include/gten_hip.h"
#include "tensor.h"

namespace gten {
namespace ops {

namespace detail {

// Token ids arrive as a non-owning HOST tensor (tinyllama.cpp:406).  Rows
// [start_pos, n) are staged into one persistent HBM buffer per process.
inline const int32_t* stage_tokens(const Tensor& tokens, int start_pos, int n_vocab)
{
    static void* dev = nullptr;
    static int cap = 0;
    const int n = tokens.numel();
    if (!tokens.is_host_external()) return static_cast<const int32_t*>(tokens.device_ptr());
    gten::detail::ensure_runtime();
    if (n > cap) {
        if (dev) GTEN_HIP_OK(gten_hip_free(dev));
        cap = n < 2048 ? 2048 : n;
        GTEN_HIP_OK(gten_hip_malloc(&dev, (size_t)cap * sizeof(int32_t)));
    }
    const int32_t* host = static_cast<const int32_t*>(tokens.host_external_ptr());
    // the embedding kernel indexes the table with the raw id: an id outside the vocabulary would be an
    // out-of-bounds HBM read, so it is refused here, where the ids are still host memory
    for (int i = start_pos; i < n; i++)
        GTEN_ASSERTM(host[i] >= 0 && host[i] < n_vocab, "token id %d at position %d is outside the vocabulary [0, %d)", host[i], i, n_vocab);
    GTEN_HIP_OK(gten_hip_memcpy_h2d(static_cast<int32_t*>(dev) + start_pos, host + start_pos,
                                    (size_t)(n - start_pos) * sizeof(int32_t)));
    return static_cast<const int32_t*>(dev);
}

} // namespace detail

// gten/ops.h:554-564
inline void token_embed(const Tensor& weight, const Tensor& tokens, Tensor& out, const int start_pos = 0)
{
    GTEN_ASSERT(weight.is_2d());
    GTEN_ASSERT(tokens.is_1d() && tokens.dtype() == kInt32);
    const int n_ctx = tokens.dimsize(0);
    const int n_embd = weight.dimsize(1);
    GTEN_ASSERT(out.shape_eq({n_ctx, n_embd}));
    const int32_t* tok = detail::stage_tokens(tokens, start_pos, weight.dimsize(0));
    GTEN_HIP_OK(gten_hip_token_embed(weight.device_weight(), dtype_code(weight.dtype()), weight.dimsize(0), tok,
                                     out.device_ptr_mut(), dtype_code(out.dtype()), (size_t)out.bstride(0),
                                     n_ctx, n_embd, start_pos));
}

// gten/ops.h:651-670.  A 1-D `out` is the lm_head form (gten/modules.cpp:70-81):
// exactly one new row, written at the start of `out`.
inline void matmul_2d(const Tensor& x, const Tensor& w, Tensor& out, const int start_pos = 0)
{
    const int n_ctx = x.dimsize(0);
    const int n_out = w.dimsize(0);
    const int n_embd = x.dimsize(1);
    GTEN_ASSERT(x.is_2d());
    GTEN_ASSERT(w.is_2d() && w.dimsize(1) == n_embd);
    const char* xp = static_cast<const char*>(x.device_ptr());
    if (out.is_1d()) {
        GTEN_ASSERT(n_ctx - start_pos == 1);
        GTEN_ASSERT(out.shape_eq({n_out}));
        GTEN_HIP_OK(gten_hip_matmul_2d(xp + (size_t)start_pos * x.bstride(0), dtype_code(x.dtype()), (size_t)x.bstride(0),
                                       w.device_weight(), dtype_code(w.dtype()),
                                       out.device_ptr_mut(), dtype_code(out.dtype()),
                                       gten_hip_row_bytes(dtype_code(out.dtype()), n_out), 1, n_embd, n_out, 0));
    } else if (out.is_2d()) {
        GTEN_ASSERT(out.shape_eq({n_ctx, n_out}));
        GTEN_HIP_OK(gten_hip_matmul_2d(xp, dtype_code(x.dtype()), (size_t)x.bstride(0),
                                       w.device_weight(), dtype_code(w.dtype()),
                                       out.device_ptr_mut(), dtype_code(out.dtype()), (size_t)out.bstride(0),
                                       n_ctx, n_embd, n_out, start_pos));
    } else {
        GTEN_ASSERT(false);
    }
}

// gten/ops.h:700-711
inline void silu(const Tensor& inp, Tensor& out, const int start_pos = 0)
{
    GTEN_ASSERT(inp.shape_eq(out.shape()));
    GTEN_ASSERT(inp.dtype() == out.dtype());
    GTEN_HIP_OK(gten_hip_silu(inp.device_ptr(), out.device_ptr_mut(), dtype_code(inp.dtype()), (size_t)inp.bstride(0),
                              inp.dimsize(0), inp.dimsize(1), start_pos));
}
inline void silu_inplace(Tensor& inp, const int start_pos = 0)
{
    void* p = inp.device_ptr_mut();
    GTEN_HIP_OK(gten_hip_silu(p, p, dtype_code(inp.dtype()), (size_t)inp.bstride(0), inp.dimsize(0), inp.dimsize(1), start_pos));
}

// gten/ops.h:757-760
inline void rotary_emb(Tensor& inp, const int d_head, const int start_pos = 0)
{
    GTEN_HIP_OK(gten_hip_rotary_emb(inp.device_ptr_mut(), dtype_code(inp.dtype()), (size_t)inp.bstride(0),
                                    inp.dimsize(0), inp.dimsize(1), d_head, start_pos));
}

// gten/ops.h:806-814
inline void rms_norm(const Tensor& inp, const Tensor& weight, Tensor& out, const int start_pos = 0)
{
    const int n_embd = inp.dimsize(1);
    GTEN_ASSERT(weight.dimsize(0) == n_embd);
    GTEN_ASSERT(inp.is_2d() && inp.dtype() == out.dtype());
    GTEN_ASSERT(weight.is_1d());
    GTEN_ASSERT(inp.shape_eq(out.shape()));
    GTEN_HIP_OK(gten_hip_rms_norm(inp.device_ptr(), dtype_code(inp.dtype()), (size_t)inp.bstride(0), weight.device_weight(),
                                  out.device_ptr_mut(), (size_t)out.bstride(0), inp.dimsize(0), n_embd, start_pos));
}

// gten/ops.h:853-867
inline void mul(const Tensor& inp0, const Tensor& inp1, Tensor& out, const int start_pos = 0)
{
    GTEN_ASSERT(inp0.dtype() == inp1.dtype() && inp1.dtype() == out.dtype());
    GTEN_ASSERT(inp0.shape_eq(inp1.shape()) && inp1.shape_eq(out.shape()));
    GTEN_HIP_OK(gten_hip_mul(inp0.device_ptr(), inp1.device_ptr(), out.device_ptr_mut(), dtype_code(inp0.dtype()),
                             (size_t)inp0.bstride(0), inp0.dimsize(0), inp0.dimsize(1), start_pos));
}
inline void mul_inplace(Tensor& inp0, const Tensor& inp1, const int start_pos = 0)
{
    GTEN_ASSERT(inp0.dtype() == inp1.dtype());
    GTEN_ASSERT(inp0.shape_eq(inp1.shape()));
    void* p = inp0.device_ptr_mut();
    GTEN_HIP_OK(gten_hip_mul(p, inp1.device_ptr(), p, dtype_code(inp0.dtype()), (size_t)inp0.bstride(0),
                             inp0.dimsize(0), inp0.dimsize(1), start_pos));
}

// gten/ops.h:900-910
inline void add(const Tensor& x0, const Tensor& x1, Tensor& out, const int start_pos = 0)
{
    GTEN_ASSERT(x0.is_2d());
    GTEN_ASSERT(x1.is_2d());
    GTEN_ASSERT(out.is_2d());
    GTEN_ASSERT(x0.shape_eq(x1.shape()));
    GTEN_ASSERT(x0.shape_eq(out.shape()));
    GTEN_ASSERT(x0.dtype() == x1.dtype() && x0.dtype() == out.dtype());
    GTEN_HIP_OK(gten_hip_add(x0.device_ptr(), x1.device_ptr(), out.device_ptr_mut(), dtype_code(x0.dtype()),
                             (size_t)x0.bstride(0), x0.dimsize(0), x0.dimsize(1), start_pos));
}

// gten/ops.h:1118-1133.  `qk` (the reference's materialised probability
// tensor) keeps its place in the signature and its shape check, but it is
// neither read nor written here: no caller reads it (gten/modules.cpp:216-221)
// and the kernel rounds probabilities to the activation dtype in flight.
inline void qkv_attn(const Tensor& q, const Tensor& k, const Tensor& v, Tensor& qk, Tensor& qkv, const int max_ctx,
                     const int start_pos = 0)
{
    const int n_ctx = q.dimsize(0);
    const int n_embd = q.dimsize(1);
    const int n_head = qk.dimsize(0);
    GTEN_ASSERT(q.is_2d());
    GTEN_ASSERT(k.is_2d());
    GTEN_ASSERT(v.is_2d());
    GTEN_ASSERT(qk.is_3d() && qk.shape_eq({n_head, n_ctx, n_ctx}));
    GTEN_ASSERT(qkv.is_2d() && qkv.shape_eq({n_ctx, n_embd}));
    GTEN_ASSERT(q.dtype() == k.dtype() && k.dtype() == v.dtype() && v.dtype() == qk.dtype() && qk.dtype() == qkv.dtype());
    GTEN_ASSERT(max_ctx > 0 && max_ctx >= n_ctx);
    const int d_head = n_embd / n_head;
    const int kv_heads = k.dimsize(1) / d_head;
    GTEN_HIP_OK(gten_hip_qkv_attn(q.device_ptr(), k.device_ptr(), v.device_ptr(), qkv.device_ptr_mut(), dtype_code(q.dtype()),
                                  (size_t)q.bstride(0), (size_t)k.bstride(0), (size_t)qkv.bstride(0),
                                  n_ctx, n_head, kv_heads, d_head, start_pos));
}

} // namespace ops
} // namespace gten
