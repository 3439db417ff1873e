// gten_runtime.hip -- device selection, stream, HBM allocation and copies
// behind the C-ABI of include/gten_hip.h (replaces the host malloc/free of
// Tensor storage, gten/tensor.cpp:23-25,61, and stages the three points where
// the reference's host code touches tensor bytes, tinyllama.cpp:320,406,414).
#include "gten_rt.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

namespace gtr {

static char g_err[512] = "";
static bool g_inited = false;
static int g_device = -1;
// Two streams: 0 is the default; 1 lets prompt processing be queued beside the decode steps of other sequences
// (gten_hip_select_stream; one calling thread, so `current` is a plain variable)
static hipStream_t g_streams[2] = {nullptr, nullptr};
static int g_cur = 0;
#define g_stream (g_streams[g_cur])
// RoPE tables, one per head width ever asked for: [GTEN_ROPE_MAX_POS][d_head/2] (cos, sin).  Never freed while the
// runtime lives -- decoders keep the pointer in their kernel arguments and in captured hipGraphs, and models of
// different head widths coexist in one process.
static std::vector<std::pair<int, float2*>> g_rope;

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code ? code : -1;
}

static hipStream_t g_override = nullptr;
hipStream_t stream() { return g_override ? g_override : g_streams[g_cur]; }
void stream_override(hipStream_t s) { g_override = s; }
int stream_index() { return g_cur; }
bool inited() { return g_inited; }

// RoPE angles use the host libm exactly as the reference does
// (gten/ops.h:743-746: m * powf(10000, -(2j/d)), then cosf/sinf in f32): device
// fast-math trig is not accurate at angles up to 2047 rad.
int rope_table(int d_head, const float2** out)
{
    for (const auto& e : g_rope)
        if (e.first == d_head) { *out = e.second; return 0; }
    GTR_REQUIRE(d_head >= 2 && d_head <= 1024 && d_head % 2 == 0, "rope_table: head width %d", d_head);
    const int half = d_head / 2;
    std::vector<float2> t((size_t)GTEN_ROPE_MAX_POS * half);
    const float d = (float)d_head;
    for (int m = 0; m < GTEN_ROPE_MAX_POS; m++)
        for (int j = 0; j < half; j++) {
            const float th = (float)m * std::pow(10000.0f, -(2.0f * j / d));
            t[(size_t)m * half + j] = make_float2(std::cos(th), std::sin(th));
        }
    float2* dev = nullptr;
    GTR_CHECK(hipMalloc((void**)&dev, t.size() * sizeof(float2)));
    const hipError_t e = hipMemcpy(dev, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(dev); GTR_CHECK(e); }
    g_rope.emplace_back(d_head, dev);
    *out = dev;
    return 0;
}

// ---- HIP-event profiler -------------------------------------------------
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_ev;          // pairs: 2*i start, 2*i+1 stop
static std::vector<int> g_ev_tag;
static size_t g_ev_used = 0;
static const size_t kMaxPairs = 1 << 16;

bool prof_on() { return g_prof_on; }

void prof_before(int tag)
{
    if (!g_prof_on || g_ev_used >= kMaxPairs) return;
    if (g_ev.size() < 2 * (g_ev_used + 1)) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        g_ev.push_back(a);
        g_ev.push_back(b);
        g_ev_tag.push_back(tag);
    }
    g_ev_tag[g_ev_used] = tag;
    hipEventRecord(g_ev[2 * g_ev_used], stream());
}

void prof_after(int)
{
    if (!g_prof_on || g_ev_used >= kMaxPairs) return;
    hipEventRecord(g_ev[2 * g_ev_used + 1], stream());
    g_ev_used++;
}

} // namespace gtr

using namespace gtr;

extern "C" {

int gten_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int gten_hip_init(int device)
{
    if (g_inited) {
        if (device != g_device) return fail(-2, "gten_hip_init: already bound to device %d", g_device);
        return 0;
    }
    GTR_CHECK(hipSetDevice(device));
    GTR_CHECK(hipStreamCreateWithFlags(&g_streams[0], hipStreamNonBlocking));
    // (stream 1 is created when it is first selected: a process that never serves a queue keeps one hardware queue --
    //  two replicas rehearsed on ONE GPU ran 2.18 instead of 0.6 ms per step with four queues between them)
    g_device = device;
    g_inited = true;
    return 0;
}

const char* gten_hip_last_error(void) { return g_err; }
void* gten_hip_stream(void) { return (void*)g_stream; }

int gten_hip_sync(void)
{
    GTR_NEED_INIT();
    GTR_CHECK(hipStreamSynchronize(g_stream));
    return 0;
}

int gten_hip_malloc(void** dptr, size_t nbytes)
{
    GTR_NEED_INIT();
    if (!dptr) return fail(-3, "gten_hip_malloc: null out pointer");
    GTR_CHECK(hipMalloc(dptr, nbytes ? nbytes : 16));
    return 0;
}

int gten_hip_free(void* dptr)
{
    GTR_NEED_INIT();
    if (!dptr) return 0;
    GTR_CHECK(hipStreamSynchronize(g_streams[0]));    // (work on either stream may still use the buffer)
    if (g_streams[1]) GTR_CHECK(hipStreamSynchronize(g_streams[1]));
    GTR_CHECK(hipFree(dptr));
    return 0;
}

int gten_hip_select_stream(int idx)
{
    GTR_NEED_INIT();
    GTR_REQUIRE(idx == 0 || idx == 1, "gten_hip_select_stream: stream %d (0 or 1)", idx);
    // (measured, serving through 128 slots: the prompt stream at the highest stream priority 29.7k against 29.4k new ids/s;
    //  through 256 slots 19.2k against 11.8k, still behind 128 slots -- not kept; with lanes of 128 rows: 256 slots 36.6k = 36.6k,
    //  128 slots 34.6k = 34.7k, 512 slots 27.8k against 22.1k)
    if (!g_streams[idx]) GTR_CHECK(hipStreamCreateWithFlags(&g_streams[idx], hipStreamNonBlocking));
    g_cur = idx;
    return 0;
}

int gten_hip_stream_wait(int waiter, int on)
{
    GTR_NEED_INIT();
    GTR_REQUIRE((waiter == 0 || waiter == 1) && (on == 0 || on == 1) && waiter != on, "gten_hip_stream_wait: streams %d, %d", waiter, on);
    if (!g_streams[waiter] || !g_streams[on]) return 0;          // a stream that was never selected holds nothing
    static hipEvent_t ev = nullptr;
    if (!ev) GTR_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    GTR_CHECK(hipEventRecord(ev, g_streams[on]));
    GTR_CHECK(hipStreamWaitEvent(g_streams[waiter], ev, 0));
    return 0;
}

int gten_hip_stream_idle(int idx, int* idle)
{
    GTR_NEED_INIT();
    GTR_REQUIRE((idx == 0 || idx == 1) && idle, "gten_hip_stream_idle: stream %d (0 or 1), idle %p", idx, (void*)idle);
    if (!g_streams[idx]) { *idle = 1; return 0; }
    const hipError_t e = hipStreamQuery(g_streams[idx]);
    if (e != hipSuccess && e != hipErrorNotReady) GTR_CHECK(e);
    *idle = e == hipSuccess;
    return 0;
}

int gten_hip_memset(void* dptr, int byte, size_t nbytes)
{
    GTR_NEED_INIT();
    GTR_CHECK(hipMemsetAsync(dptr, byte, nbytes, g_stream));
    return 0;
}

int gten_hip_memcpy_h2d(void* dst, const void* src_host, size_t nbytes)
{
    GTR_NEED_INIT();
    // pageable source: the runtime stages it before returning, so the caller
    // may reuse src_host immediately (the loader reuses one read buffer).
    GTR_CHECK(hipMemcpyAsync(dst, src_host, nbytes, hipMemcpyHostToDevice, g_stream));
    GTR_CHECK(hipStreamSynchronize(g_stream));
    return 0;
}

int gten_hip_memcpy_d2h(void* dst_host, const void* src, size_t nbytes)
{
    GTR_NEED_INIT();
    GTR_CHECK(hipMemcpyAsync(dst_host, src, nbytes, hipMemcpyDeviceToHost, g_stream));
    GTR_CHECK(hipStreamSynchronize(g_stream));
    return 0;
}

int gten_hip_memcpy_d2d(void* dst, const void* src, size_t nbytes)
{
    GTR_NEED_INIT();
    GTR_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, g_stream));
    return 0;
}

int gten_hip_prof_enable(int on)
{
    GTR_NEED_INIT();
    GTR_CHECK(hipStreamSynchronize(g_stream));
    g_prof_on = on != 0;
    if (on) g_ev_used = 0;
    return 0;
}

int gten_hip_prof_read(int family, int* launches, double* total_ms)
{
    GTR_NEED_INIT();
    GTR_CHECK(hipStreamSynchronize(g_stream));
    int n = 0;
    double ms = 0.0;
    for (size_t i = 0; i < g_ev_used; i++) {
        if (g_ev_tag[i] != family) continue;
        float t = 0.f;
        GTR_CHECK(hipEventElapsedTime(&t, g_ev[2 * i], g_ev[2 * i + 1]));
        ms += t;
        n++;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return 0;
}

const char* gten_hip_prof_family_name(int family)
{
    static const char* names[KT_COUNT] = {
        "pack_weight", "token_embed", "matmul_2d", "rms_norm", "rotary_emb", "elementwise", "qkv_attn",
        "decode_gemv_qkv", "decode_attn_score", "decode_attn_pv", "decode_gemv_o", "decode_gemv_gateup",
        "decode_gemv_down", "decode_gemv_head", "decode_argmax", "matmul_2d_mfma", "decode_stage", "qkv_attn_tiled", "decode_persistent"};
    return (family >= 0 && family < KT_COUNT) ? names[family] : nullptr;
}

size_t gten_hip_row_bytes(int dtype, int cols)
{
    switch (dtype) {
    case GTEN_I32: case GTEN_F32: return (size_t)cols * 4;
    case GTEN_F16: return (size_t)cols * 2;
    case GTEN_Q8: return (size_t)((cols + 31) / 32) * 34;
    case GTEN_Q4: return (size_t)(cols / 32) * 18;
    }
    return 0;
}

} // extern "C"
