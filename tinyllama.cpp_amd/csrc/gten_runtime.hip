// gten_runtime.hip -- device selection, stream, HBM allocation and copies
// behind the C-ABI of include/gten_hip.h (replaces the host malloc/free of
// Tensor storage, gten/tensor.cpp:23-25,61, and stages the three points where
// the reference's host code touches tensor bytes, tinyllama.cpp:320,406,414).
#include "gten_rt.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
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

// ---- watched K / V caches (gten_rt.h) -------------------------------------
// Sorted by start address; the caches of one decoder are equally long and disjoint, those of different decoders may
// coincide (two decoders over the same sequences), so a lookup scans every entry that starts inside
// (p - longest entry, p + bytes).
struct KvWatch { uintptr_t lo, hi; const void* owner; char* flag; };
static std::multimap<uintptr_t, KvWatch> g_watch;
static size_t g_watch_longest = 0;
static unsigned long long g_watch_epoch = 1;

void kv_watch_add(const void* p, size_t bytes, const void* owner, char* dirty_flag)
{
    if (!p || !bytes) return;
    const uintptr_t lo = (uintptr_t)p;
    g_watch.emplace(lo, KvWatch{lo, lo + bytes, owner, dirty_flag});
    if (bytes > g_watch_longest) g_watch_longest = bytes;
    g_watch_epoch++;
}

void kv_watch_remove(const void* owner, const char* dirty_flag)
{
    for (auto it = g_watch.begin(); it != g_watch.end();) {
        if (it->second.owner == owner && (!dirty_flag || it->second.flag == dirty_flag)) it = g_watch.erase(it);
        else ++it;
    }
    if (g_watch.empty()) g_watch_longest = 0;
    g_watch_epoch++;
}

template <typename F>
static void kv_watch_scan(const void* p, size_t bytes, const void* except, F&& hit)
{
    if (g_watch.empty() || !p || !bytes) return;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    const uintptr_t from = lo > g_watch_longest ? lo - g_watch_longest : 0;
    for (auto it = g_watch.lower_bound(from); it != g_watch.end() && it->first < hi; ++it) {
        const KvWatch& w = it->second;
        if (w.hi > lo && w.owner != except) hit(w);
    }
}

void kv_watch_touch(const void* p, size_t bytes, const void* except)
{
    kv_watch_scan(p, bytes, except, [](const KvWatch& w) { *w.flag = 1; });
}

bool kv_watch_overlaps(const void* p, size_t bytes, const void* except)
{
    bool any = false;
    kv_watch_scan(p, bytes, except, [&](const KvWatch&) { any = true; });
    return any;
}

bool kv_watch_any() { return !g_watch.empty(); }
unsigned long long kv_watch_epoch() { return g_watch_epoch; }

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
    kv_watch_touch(dptr, nbytes);
    GTR_CHECK(hipMemsetAsync(dptr, byte, nbytes, g_stream));
    return 0;
}

int gten_hip_memcpy_h2d(void* dst, const void* src_host, size_t nbytes)
{
    GTR_NEED_INIT();
    // pageable source: the runtime stages it before returning, so the caller
    // may reuse src_host immediately (the loader reuses one read buffer).
    kv_watch_touch(dst, nbytes);
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
    kv_watch_touch(dst, nbytes);
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

// The watched-cache registry on made-up addresses (nothing is dereferenced): every case of "does this write hit that
// watch" the decoders rely on.  Leaves the registry as it found it.
int gten_hip_kv_watch_selftest(void)
{
    const uintptr_t base = (uintptr_t)1 << 40;
    auto at = [&](uintptr_t off) { return (const void*)(base + off); };
    char fa = 0, fb = 0, fc = 0;
    int owner_a = 0, owner_b = 0;
    const unsigned long long e0 = kv_watch_epoch();
    kv_watch_add(at(0x1000), 0x1000, &owner_a, &fa);        // [0x1000, 0x2000)
    kv_watch_add(at(0x3000), 0x1000, &owner_a, &fb);        // [0x3000, 0x4000)
    kv_watch_add(at(0x1000), 0x1000, &owner_b, &fc);        // the same rows watched by a second owner
    int bad = 0;
    auto expect = [&](int no, bool a, bool b, bool c) {
        if (!bad && ((fa != 0) != a || (fb != 0) != b || (fc != 0) != c)) bad = no;
        fa = fb = fc = 0;
    };
    if (kv_watch_epoch() == e0 || !kv_watch_any()) bad = 100;
    kv_watch_touch(at(0x0), 0x1000);                    expect(1, false, false, false);   // ends where the first watch begins
    kv_watch_touch(at(0x0), 0x1001);                    expect(2, true, false, true);     // one byte into it
    kv_watch_touch(at(0x1fff), 1);                      expect(3, true, false, true);     // its last byte
    kv_watch_touch(at(0x2000), 0x1000);                 expect(4, false, false, false);   // the gap between the two
    kv_watch_touch(at(0x2fff), 2);                      expect(5, false, true, false);    // into the second
    kv_watch_touch(at(0x0), 0x10000);                   expect(6, true, true, true);      // a range that covers everything
    kv_watch_touch(at(0x1800), 8, &owner_a);            expect(7, false, false, true);    // the owner's own appends spare its flags
    kv_watch_touch(at(0x1800), 8, &owner_b);            expect(8, true, false, false);
    kv_watch_touch(at(0x4000), 0x100);                  expect(9, false, false, false);   // just past the end
    kv_watch_touch(nullptr, 16);                        expect(10, false, false, false);
    kv_watch_touch(at(0x1800), 0);                      expect(11, false, false, false);
    if (!bad && (!kv_watch_overlaps(at(0x1800), 8, &owner_a) || kv_watch_overlaps(at(0x3800), 8, &owner_a))) bad = 12;   // only b's watch is foreign to a
    kv_watch_remove(&owner_a, &fb);                                                      // one flag's entries (a slot bound to other rows)
    kv_watch_touch(at(0x3000), 0x1000);                 expect(13, false, false, false);
    kv_watch_touch(at(0x1000), 4);                      expect(14, true, false, true);
    kv_watch_add(at(0x8000), 0x4000, &owner_a, &fb);                                     // ... and its new, longer rows
    kv_watch_touch(at(0xbfff), 1);                      expect(15, false, true, false);
    kv_watch_touch(at(0x1000), 4);                      expect(16, true, false, true);    // the shorter entries are still found
    kv_watch_remove(&owner_a, nullptr);
    kv_watch_touch(at(0x0), 0x10000);                   expect(17, false, false, true);
    kv_watch_remove(&owner_b, nullptr);
    kv_watch_touch(at(0x0), 0x10000);                   expect(18, false, false, false);
    return bad;
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
