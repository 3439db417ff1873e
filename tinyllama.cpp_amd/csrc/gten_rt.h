// gten_rt.h -- host-side internals shared by the .hip translation units.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/gten_hip.h"

#define GTEN_ROPE_MAX_POS 2048   // TinyLLamaParams::max_ctx, tinyllama.cpp:14
#define GTEN_SEG_MAX_ROWS 4096   // rows of a segmented call (gten_hip_set_row_segments): several prompts, each <= GTEN_ROPE_MAX_POS rows

namespace gtr {

int fail(int code, const char* fmt, ...);
hipStream_t stream();
int stream_index();                    // 0 or 1: which of the library's two streams is selected (gten_hip_select_stream)
void stream_override(hipStream_t s);   // launches go to `s` until it is reset with nullptr (a decoder's lanes, gten_decode.hip)
bool inited();
int rope_table(int d_head, const float2** out);

// ---- watched K / V caches (round 5).  A decoder of 16+ sequences keeps HEAD-MAJOR shadows of its sequences' K / V caches
// (gten_decode_attn_hm.h) and registers every row-layout cache it shadows here, one flag per sequence.  EVERY entry point of
// the C-ABI that writes device memory calls kv_watch_touch on its output range: a write that overlaps a watched cache sets
// the owner's flag, and the owner re-imports that sequence's rows before its next step -- a stale shadow is impossible for
// writes that go through this library (all of its kernels and copies do).  `except`: the owner whose own decode appends keep
// its shadow current.  The registry is host-only state (one calling thread, like the rest of the library).
void kv_watch_add(const void* p, size_t bytes, const void* owner, char* dirty_flag);
void kv_watch_remove(const void* owner, const char* dirty_flag);     // the entries of one flag (dirty_flag null: all of the owner's)
void kv_watch_touch(const void* p, size_t bytes, const void* except = nullptr);
bool kv_watch_any();                   // false: nothing is watched, touches are free
unsigned long long kv_watch_epoch();   // bumped by every add / remove: callers cache "do my caches overlap a foreign watch"
bool kv_watch_overlaps(const void* p, size_t bytes, const void* except);

} // namespace gtr

#define GTR_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return gtr::fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                             __FILE__, __LINE__);                                              \
    } while (0)

#define GTR_NEED_INIT()                                                                        \
    do {                                                                                       \
        if (!gtr::inited()) return gtr::fail(-1, "gten_hip_init() has not been called");       \
    } while (0)

#define GTR_REQUIRE(cond, ...)                                                                 \
    do {                                                                                       \
        if (!(cond)) return gtr::fail(-4, __VA_ARGS__);                                        \
    } while (0)

// launch errors surface here (bad grid, missing code object, ...)
#define GTR_LAUNCHED() GTR_CHECK(hipGetLastError())

// Kernel families for the in-library HIP-event profiler (gten_hip_prof_*): when
// profiling is on, every launch is bracketed by two events recorded on the
// library's stream, so bench.py can report per-kernel average durations that
// are directly comparable with `rocprofv3 --kernel-trace --stats`.
enum {
    KT_PACK = 0, KT_EMBED, KT_MATMUL, KT_RMSNORM, KT_ROPE, KT_ELEMWISE, KT_ATTN,
    KT_DEC_GEMV_QKV, KT_DEC_ATTN_SCORE, KT_DEC_ATTN_PV, KT_DEC_GEMV_O, KT_DEC_GEMV_GATEUP, KT_DEC_GEMV_DOWN,
    KT_DEC_GEMV_HEAD, KT_DEC_ARGMAX, KT_MATMUL_MFMA, KT_DEC_STAGE, KT_ATTN_TILED, KT_DEC_PERSIST, KT_COUNT
};

// gten_mfma.hip: ops::matmul_2d for >= GTEN_MFMA_MIN_ROWS new rows
#define GTEN_MFMA_MIN_ROWS 16
int gten_launch_matmul_mfma(const void* x, int x_dtype, size_t x_pitch, const void* w, int w_dtype,
                            void* out, int out_dtype, size_t out_pitch, int n, int d_in, int d_out, int start_pos);
// ... up to three matrices sharing the input in ONE launch (q | k | v, gate | up), optionally with the residual sum behind a
// single Q8 projection (sum_out = Q8(resid + Q8(W x)), ops::add's arithmetic).  `converted`: the f16 copy of exactly these
// input rows is already in the library's scratch (gten_mfma_convert was the last conversion) -- quantized weights only.
namespace gtr {
struct MfmaMats {
    int n = 1;
    const void* w[3] = {nullptr, nullptr, nullptr};
    void* out[3] = {nullptr, nullptr, nullptr};
    size_t out_pitch[3] = {0, 0, 0};
    int d_out[3] = {0, 0, 0};
    const void* resid = nullptr;
    void* sum_out = nullptr;
    size_t resid_pitch = 0;
    // optional: the RMSNorm that reads sum_out next (weights, its Q8 output rows, the f16 copy for the W.x after it or
    // null).  Taken when the launch shares its K loop and the projection is 2048 wide (the plane sums then run one wave
    // per row and continue into the norm); *norm_done tells the caller whether it was.
    const void* norm_w = nullptr;
    void* norm_out = nullptr;
    size_t norm_out_pitch = 0;
    void* norm_a16 = nullptr;
    bool* norm_done = nullptr;
};
}
int gten_launch_matmul_mfma_multi(const void* x, size_t x_pitch, int w_dtype, const gtr::MfmaMats& m, int out_dtype,
                                  int n, int d_in, int start_pos, bool converted);
int gten_mfma_convert(const void* x, size_t x_pitch, int n, int d_in, int start_pos);
// room for the f16 copy of `rows` rows of up to d_max elements; producers that write the copy themselves fill it from row 0
int gten_mfma_scratch(int rows, int d_max, uint8_t** a16);

// gten_attn_tiled.hip: ops::qkv_attn for >= GTEN_ATTN_TILED_MIN_ROWS new rows (Q8 activations, d_head 64)
#define GTEN_ATTN_TILED_MIN_ROWS 16
// (a16, fast form only: also write the rows as the f16 copy the o projection reads -- gten_mfma_scratch)
int gten_launch_attn_tiled(const void* q, const void* k, const void* v, void* out, size_t q_pitch, size_t kv_pitch,
                           size_t out_pitch, int n, int n_heads, int n_kv_heads, int start_pos, void* a16 = nullptr);
// ... and for f16 activations (f16 MFMA scores: agrees with the row kernel to f32 summation order, not byte for byte)
int gten_launch_attn_tiled_f16(const void* q, const void* k, const void* v, void* out, size_t q_pitch, size_t kv_pitch,
                               size_t out_pitch, int n, int n_heads, int n_kv_heads, int start_pos);

namespace gtr {
const int* row_segments(int* n_segments);   // gten_hip_set_row_segments: starts[0 .. n] or null
bool prefill_exact();          // gten_hip_set_prefill_exact: the exact forms of the prompt-sized kernels (gten_mfma.hip, gten_attn_tiled.hip)
bool prof_on();
void prof_before(int tag);
void prof_after(int tag);
}

#define GTR_LAUNCH(tag, kernel, grid, block, smem, ...)                                        \
    do {                                                                                       \
        gtr::prof_before(tag);                                                                 \
        hipLaunchKernelGGL(kernel, grid, block, smem, gtr::stream(), __VA_ARGS__);             \
        gtr::prof_after(tag);                                                                  \
        GTR_LAUNCHED();                                                                        \
    } while (0)
