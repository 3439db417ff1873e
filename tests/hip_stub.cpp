// not gpu, TEST ONLY: a stand-in for libgten_hip.so that implements every symbol of include/gten_hip.h on host memory,
// so that the host-side C++ above the C-ABI -- the gten modules (recording of single-row calls, the composed block call),
// TinyLlama / TinyLlamaBatch, the continuous-batching scheduler, capi.cpp's marshalling -- can run on a CPU under
// AddressSanitizer / UBSan (tests/test_host_sanitize_cpu.py links tests/host_sanitize_serve.cpp against THIS file instead of
// the HIP library).  It computes no model: operators are no-ops, and the "model" is a fixed rule -- the id that follows
// token t at context length n is (7 t + 13 n + 1) mod n_vocab -- which the operator path (lm_head form of matmul_2d after a
// token_embed) and the decoder entry points both follow, so that every way of generating must produce the same ids.
// Nothing under tinyllama.cpp_amd/ includes or links this file.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/gten_hip.h"

namespace {

char g_err[256] = "";
int g_stream_idx = 0;
int g_block_rows = 1;
std::vector<int32_t> g_tokens;   // ids of the last token_embed (rows [0, n))
std::vector<int32_t> g_stub_seg; // gten_hip_set_row_segments: starts[0 .. n]
std::vector<int32_t> g_rule_seg; // ... as they were when the rows were embedded (the caller clears them before it asks for logits)
const char* g_norm_out = nullptr; // output rows of the last rms_norm (the final norm, when the lm_head form follows) and their pitch
size_t g_norm_pitch = 0;

int fail(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -4;
}

int next_id(int32_t tok, int n, int n_vocab) { return (int)(((long long)tok * 7 + (long long)n * 13 + 1) % n_vocab); }

struct Slot { int n = 1; int advance = 0; int stop = 0; };

}  // namespace

struct gten_hip_decoder {
    gten_hip_decoder_desc d;
    int n_seq = 1;
    std::vector<std::vector<int32_t>> tokens;   // [n_seq][max_ctx + 2]
    std::vector<Slot> slots;
    std::vector<int32_t> last;                  // last result per sequence
};

extern "C" {

int gten_hip_device_count(void) { return 1; }
int gten_hip_init(int) { return 0; }
const char* gten_hip_last_error(void) { return g_err; }
void* gten_hip_stream(void) { return nullptr; }
int gten_hip_sync(void) { return 0; }
int gten_hip_select_stream(int idx)
{
    if (idx != 0 && idx != 1) return fail("select_stream: %d", idx);
    g_stream_idx = idx;
    return 0;
}
int gten_hip_stream_wait(int waiter, int on) { return (waiter == on) ? fail("stream_wait: %d %d", waiter, on) : 0; }
int gten_hip_stream_idle(int idx, int* idle)
{
    static unsigned calls = 0;
    if (!idle || (idx != 0 && idx != 1)) return fail("stream_idle");
    *idle = (++calls % 3) == 0;                 // "sometimes still running": both branches of the scheduler's poll
    return 0;
}
int gten_hip_malloc(void** dptr, size_t nbytes)
{
    *dptr = std::malloc(nbytes ? nbytes : 16);
    return *dptr ? 0 : fail("malloc");
}
int gten_hip_free(void* dptr) { std::free(dptr); return 0; }
int gten_hip_memset(void* dptr, int byte, size_t nbytes) { std::memset(dptr, byte, nbytes); return 0; }
int gten_hip_memcpy_h2d(void* dst, const void* src, size_t nbytes) { std::memcpy(dst, src, nbytes); return 0; }
int gten_hip_memcpy_d2h(void* dst, const void* src, size_t nbytes) { std::memcpy(dst, src, nbytes); return 0; }
int gten_hip_memcpy_d2d(void* dst, const void* src, size_t nbytes) { std::memmove(dst, src, nbytes); return 0; }
int gten_hip_prof_enable(int) { return 0; }
int gten_hip_prof_read(int, int* launches, double* total_ms) { if (launches) *launches = 0; if (total_ms) *total_ms = 0.0; return 0; }
const char* gten_hip_prof_family_name(int) { return nullptr; }
int gten_hip_selftest_q8scale(unsigned long long* a, unsigned long long* b) { if (a) *a = 0; if (b) *b = 0; return 0; }

size_t gten_hip_row_bytes(int dtype, int cols)
{
    switch (dtype) {
    case GTEN_I32: case GTEN_F32: return (size_t)cols * 4;
    case GTEN_F16: return (size_t)cols * 2;
    case GTEN_Q8: return (size_t)(cols / 32) * 34;
    case GTEN_Q4: return (size_t)(cols / 32) * 18;
    }
    return 0;
}
int gten_hip_pack_weight(const void* src, int dtype, int rows, int cols, void* dst)
{
    std::memcpy(dst, src, (size_t)rows * gten_hip_row_bytes(dtype, cols));
    return 0;
}

int gten_hip_token_embed(const void*, int, int n_vocab, const int32_t* tokens, void*, int, size_t, int n, int, int start_pos)
{
    if (n <= 0 || start_pos < 0 || start_pos >= n) return fail("token_embed: rows");
    for (int i = start_pos; i < n; i++)
        if (tokens[i] < 0 || tokens[i] >= n_vocab) return fail("token_embed: id %d", tokens[i]);
    g_tokens.assign(tokens, tokens + n);
    g_rule_seg = g_stub_seg;
    return 0;
}
int gten_hip_matmul_2d(const void* x, int, size_t, const void*, int, void* out, int out_dtype, size_t, int n, int, int d_out, int start_pos)
{
    if (out_dtype == GTEN_F32 && n - start_pos == 1) {       // the lm_head form: one-hot logits of the rule's next id
        float* lg = (float*)out;
        for (int i = 0; i < d_out; i++) lg[i] = 0.f;
        // which row of the last embedded row matrix: the caller hands over a pointer to ONE row of the final norm's output
        // (gten/ops.h, the 1-D form); the last row when that cannot be told.  With row segments set (several prompts in one
        // matrix) the row's context length counts from its prompt's first row.
        if (g_tokens.empty()) return fail("matmul_2d (lm_head): nothing embedded");
        long row = (long)g_tokens.size() - 1;
        if (g_norm_out && g_norm_pitch && (const char*)x >= g_norm_out) {
            const long r = (long)(((const char*)x - g_norm_out) / (long)g_norm_pitch);
            if (r < (long)g_tokens.size()) row = r;
        }
        int first = 0;
        for (size_t k = 0; k + 1 < g_rule_seg.size(); k++)
            if (g_rule_seg[k] <= row) first = g_rule_seg[k];
        lg[next_id(g_tokens[(size_t)row], (int)row + 1 - first, d_out)] = 1.f;
    }
    return 0;
}
int gten_hip_set_prefill_exact(int) { return 0; }
int gten_hip_set_decode_exact(int) { return 0; }
int gten_hip_set_decode_persistent(int) { return 0; }
int gten_hip_argmax_row(const float* logits, int n, int32_t* out)
{
    if (!logits || !out || n <= 0) return fail("argmax_row: arguments");
    int best = 0;
    for (int i = 1; i < n; i++)
        if (logits[i] > logits[best]) best = i;
    out[0] = best;
    return 0;
}
int gten_hip_set_lane_skip(int) { return 0; }
int gten_hip_persist_status(int* n_decoders, unsigned long long* launches, unsigned* abort_code, unsigned*, int)
{
    if (n_decoders) *n_decoders = 0;
    if (launches) *launches = 0;
    if (abort_code) *abort_code = 0;
    return 0;
}
int gten_hip_row_segments_ok(int, int, int, int, int, int) { return g_block_rows; }
int gten_hip_set_row_segments(const int32_t* starts, int n)
{
    if (n > 0 && (!starts || starts[0] != 0)) return fail("set_row_segments: starts[0] must be 0");
    for (int k = 0; k < n; k++)
        if (starts[k + 1] - starts[k] < 16) return fail("set_row_segments: short segment");
    g_stub_seg.assign(starts, starts + (n > 0 ? n + 1 : 0));
    return 0;
}
int gten_hip_copy_ranges(const gten_hip_copy_range* r, int n)
{
    if (!r || n < 0 || n > GTEN_HIP_MAX_COPY_RANGES) return fail("copy_ranges: bad arguments");
    for (int i = 0; i < n; i++) std::memmove(r[i].dst, r[i].src, r[i].bytes);
    return 0;
}
int gten_hip_rms_norm(const void*, int, size_t, const void*, void* out, size_t out_pitch, int, int, int)
{
    g_norm_out = (const char*)out;
    g_norm_pitch = out_pitch;
    return 0;
}
int gten_hip_rotary_emb(void*, int, size_t, int, int, int, int) { return 0; }
int gten_hip_silu(const void*, void*, int, size_t, int, int, int) { return 0; }
int gten_hip_mul(const void*, const void*, void*, int, size_t, int, int, int) { return 0; }
int gten_hip_add(const void*, const void*, void*, int, size_t, int, int, int) { return 0; }
int gten_hip_qkv_attn(const void*, const void*, const void*, void*, int, size_t, size_t, size_t, int, int, int, int, int) { return 0; }
int gten_hip_block_rows(const gten_hip_block_desc* b, int n, int start_pos)
{
    if (!b) return fail("block_rows: null");
    if (!g_block_rows || n - start_pos < 16) return GTEN_HIP_NOT_HANDLED;
    const void* p[] = {b->attn_norm_w, b->wq, b->wk, b->wv, b->wo, b->ffn_norm_w, b->wgate, b->wup, b->wdown, b->inp, b->attn_norm_out, b->q, b->k, b->v,
                       b->attn_out, b->o, b->h, b->ffn_norm_out, b->gate, b->up, b->down, b->out};
    for (const void* q : p)
        if (!q) return fail("block_rows: null pointer");
    // touch the last new row of every activation buffer: a buffer smaller than the modules promise is an ASan report
    const size_t pe = gten_hip_row_bytes(b->adtype, b->n_embd), pf = gten_hip_row_bytes(b->adtype, b->n_ffn);
    const size_t pkv = gten_hip_row_bytes(b->adtype, (b->n_embd / b->n_heads) * b->n_kv_heads);
    void* rows_e[] = {b->attn_norm_out, b->q, b->attn_out, b->o, b->h, b->ffn_norm_out, b->down, b->out};
    for (void* q : rows_e) std::memset((uint8_t*)q + (size_t)(n - 1) * pe, 0, pe);
    std::memset((uint8_t*)b->k + (size_t)(n - 1) * pkv, 0, pkv);
    std::memset((uint8_t*)b->v + (size_t)(n - 1) * pkv, 0, pkv);
    std::memset((uint8_t*)b->gate + (size_t)(n - 1) * pf, 0, pf);
    std::memset((uint8_t*)b->up + (size_t)(n - 1) * pf, 0, pf);
    return 0;
}
int gten_hip_set_block_rows(int on) { g_block_rows = on != 0; return 0; }

static int create_common(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers, int n_seq, gten_hip_decoder** out)
{
    if (!desc || !layers || !out) return fail("decoder_create: null");
    if (desc->n_vocab <= 0 || desc->max_ctx <= 0) return fail("decoder_create: dims");
    gten_hip_decoder* dc = new gten_hip_decoder();
    dc->d = *desc;
    dc->n_seq = n_seq;
    dc->tokens.assign((size_t)n_seq, std::vector<int32_t>((size_t)desc->max_ctx + 2, 0));
    dc->last.assign((size_t)n_seq, 0);
    *out = dc;
    return 0;
}
int gten_hip_decoder_create(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers, gten_hip_decoder** out)
{
    return create_common(desc, layers, 1, out);
}
int gten_hip_decoder_create_multi(const gten_hip_decoder_desc* desc, const gten_hip_layer_ptrs* layers, const gten_hip_kv_ptrs* kv, int n_seq,
                                  gten_hip_decoder** out)
{
    const bool ok = n_seq == 2 || n_seq == 4 || n_seq == 8 || n_seq == 16 || n_seq == 32 || n_seq == 48 || n_seq == 64 || n_seq == 128 || n_seq == 192 || n_seq == 256 || n_seq == 384 || n_seq == 512;
    if (!ok || !kv) return fail("decoder_create_multi: %d sequences", n_seq);
    return create_common(desc, layers, n_seq, out);
}
int gten_hip_decoder_destroy(gten_hip_decoder* dc) { delete dc; return 0; }

int gten_hip_decoder_set_tokens_seq(gten_hip_decoder* dc, int seq, const int32_t* t, int first, int count)
{
    if (!dc || seq < 0 || seq >= dc->n_seq || first < 0 || count < 0 || first + count > dc->d.max_ctx + 1) return fail("set_tokens: range");
    for (int i = 0; i < count; i++) {
        if (t[i] < 0 || t[i] >= dc->d.n_vocab) return fail("set_tokens: id %d", t[i]);
        dc->tokens[(size_t)seq][(size_t)(first + i)] = t[i];
    }
    return 0;
}
int gten_hip_decoder_set_tokens(gten_hip_decoder* dc, const int32_t* t, int first, int count) { return gten_hip_decoder_set_tokens_seq(dc, 0, t, first, count); }

static int one_step(gten_hip_decoder* dc, int seq, int n)
{
    if (n < 1 || n > dc->d.max_ctx) return fail("decoder step: n=%d outside [1, %d]", n, dc->d.max_ctx);
    std::vector<int32_t>& row = dc->tokens[(size_t)seq];
    const int id = next_id(row[(size_t)n - 1], n, dc->d.n_vocab);
    row[(size_t)n] = id;                         // the argmax becomes the next step's input token
    dc->last[(size_t)seq] = id;
    return 0;
}
static void one_hot(const gten_hip_decoder* dc, int seq, float* lg)
{
    for (int i = 0; i < dc->d.n_vocab; i++) lg[i] = 0.f;
    lg[dc->last[(size_t)seq]] = 1.f;
}
int gten_hip_decoder_step(gten_hip_decoder* dc, int n, int)
{
    if (!dc) return fail("decoder_step: null");
    dc->slots.clear();
    for (int q = 0; q < dc->n_seq; q++)
        if (int rc = one_step(dc, q, n)) return rc;
    if (dc->n_seq == 1 && dc->d.logits) one_hot(dc, 0, dc->d.logits);
    return 0;
}
int gten_hip_decoder_steps(gten_hip_decoder* dc, int n_first, int count, int g)
{
    for (int i = 0; i < count; i++)
        if (int rc = gten_hip_decoder_step(dc, n_first + i, g)) return rc;
    return 0;
}
int gten_hip_decoder_step_ragged(gten_hip_decoder* dc, const int* n_per_seq, int)
{
    if (!dc || !n_per_seq || dc->n_seq < 2) return fail("step_ragged: arguments");
    dc->slots.clear();
    for (int q = 0; q < dc->n_seq; q++)
        if (n_per_seq[q] < 1 || n_per_seq[q] > dc->d.max_ctx) return fail("step_ragged: n=%d", n_per_seq[q]);
    for (int q = 0; q < dc->n_seq; q++) one_step(dc, q, n_per_seq[q]);
    return 0;
}
int gten_hip_decoder_generate_multi(gten_hip_decoder* dc, const int* n_first, const int* max_new_seq, int max_new, int eos, int32_t* out, int* n_out)
{
    if (!dc || !n_first || !out || !n_out || max_new < 0) return fail("generate_multi: arguments");
    dc->slots.clear();
    for (int q = 0; q < dc->n_seq; q++) {
        const int room = max_new_seq ? max_new_seq[q] : max_new;
        if (room < 0 || room > max_new) return fail("generate_multi: room");
        int n = n_first[q], made = 0;
        while (made < room && n <= dc->d.max_ctx) {
            if (int rc = one_step(dc, q, n)) return rc;
            const int id = dc->tokens[(size_t)q][(size_t)n];
            if (id == eos) break;
            out[(size_t)q * max_new + made++] = id;
            n++;
        }
        n_out[q] = made;
    }
    return 0;
}
int gten_hip_decoder_generate(gten_hip_decoder* dc, int n_first, int max_new, int eos, int32_t* out, int* n_out)
{
    if (!dc || dc->n_seq != 1) return fail("generate: single-sequence decoders");
    return gten_hip_decoder_generate_multi(dc, &n_first, nullptr, max_new, eos, out, n_out);
}

static void slots_view(gten_hip_decoder* dc)
{
    if (dc->slots.empty()) dc->slots.assign((size_t)dc->n_seq, Slot{});
}
int gten_hip_decoder_slot_start(gten_hip_decoder* dc, int seq, int n_first)
{
    if (!dc || seq < 0 || seq >= dc->n_seq) return fail("slot_start: sequence %d", seq);
    if (n_first < 1 || n_first > dc->d.max_ctx) return fail("slot_start: n_first=%d", n_first);
    slots_view(dc);
    dc->slots[(size_t)seq] = Slot{n_first, 3, 0};
    return 0;
}
int gten_hip_decoder_slot_start_until(gten_hip_decoder* dc, int seq, int n_first, int n_last)
{
    if (int rc = gten_hip_decoder_slot_start(dc, seq, n_first)) return rc;
    if (n_last != 0 && (n_last < n_first || n_last > dc->d.max_ctx)) return fail("slot_start_until: n_last=%d", n_last);
    dc->slots[(size_t)seq].stop = n_last;
    return 0;
}
int gten_hip_decoder_slot_park(gten_hip_decoder* dc, int seq)
{
    if (!dc || seq < 0 || seq >= dc->n_seq) return fail("slot_park: sequence %d", seq);
    slots_view(dc);
    dc->slots[(size_t)seq] = Slot{1, 0, 0};
    return 0;
}
int gten_hip_decoder_run(gten_hip_decoder* dc, int steps);
int gten_hip_decoder_run_lanes(gten_hip_decoder* dc, int steps, int) { return gten_hip_decoder_run(dc, steps); }
int gten_hip_decoder_slot_bind(gten_hip_decoder* dc, int seq, const gten_hip_kv_ptrs* kv)
{
    if (!dc || !kv || seq < 0 || seq >= dc->n_seq) return fail("slot_bind: sequence %d", seq);
    slots_view(dc);
    if (dc->slots[(size_t)seq].advance != 0) return fail("slot_bind: slot %d is not parked", seq);
    return 0;                                               // (the stand-in decodes without caches)
}
int gten_hip_decoder_slots_apply(gten_hip_decoder* dc, int count, const int* seqs, const int* n_first, const int* n_last, const int32_t* const* tokens)
{
    if (!dc || count < 0 || (count > 0 && (!seqs || !n_first || !n_last))) return fail("slots_apply: arguments");
    for (int i = 0; i < count; i++) {
        if (n_first[i] == 0) {
            if (int rc = gten_hip_decoder_slot_park(dc, seqs[i])) return rc;
            continue;
        }
        if (tokens && tokens[i])
            if (int rc = gten_hip_decoder_set_tokens_seq(dc, seqs[i], tokens[i], 0, n_first[i])) return rc;
        if (int rc = gten_hip_decoder_slot_start_until(dc, seqs[i], n_first[i], n_last[i])) return rc;
    }
    return 0;
}
int gten_hip_set_kv_head_major(int) { return 0; }
int gten_hip_set_wx_planes(int) { return 0; }
int gten_hip_set_ffn_streamed(int) { return 0; }
int gten_hip_kv_watch_selftest(void) { return 0; }
int gten_hip_decoder_kv_info(gten_hip_decoder* dc, int* head_major, unsigned long long* seq_imports, unsigned long long* import_launches)
{
    if (!dc) return fail("decoder_kv_info: arguments");
    // (the stand-in reads the cache rows of its fixed next-id rule directly: no shadows)
    if (head_major) *head_major = 0;
    if (seq_imports) *seq_imports = 0;
    if (import_launches) *import_launches = 0;
    return 0;
}
int gten_hip_decoder_lane_info(gten_hip_decoder* dc, int* lane_rows, int* lanes, int* last_run_lanes)
{
    if (!dc) return fail("decoder_lane_info: arguments");
    // (the stand-in has no lanes: one lane of every sequence)
    if (lane_rows) *lane_rows = dc->n_seq;
    if (lanes) *lanes = 1;
    if (last_run_lanes) *last_run_lanes = 1;
    return 0;
}
int gten_hip_decoder_run(gten_hip_decoder* dc, int steps)
{
    if (!dc || steps < 0) return fail("decoder_run: arguments");
    slots_view(dc);
    for (const Slot& s : dc->slots)
        if ((s.advance & 1) && s.stop <= 0 && s.n + steps - 1 > dc->d.max_ctx) return fail("decoder_run: %d steps would take a slot at n=%d past max_ctx %d", steps, s.n, dc->d.max_ctx);
    for (int q = 0; q < dc->n_seq; q++) {
        Slot& s = dc->slots[(size_t)q];
        if (!(s.advance & 1)) continue;
        for (int i = 0; i < steps; i++) {
            one_step(dc, q, s.n);
            if (s.stop <= 0 || s.n < s.stop) s.n++;               // (the last step is repeated, not passed)
        }
    }
    return 0;
}
int gten_hip_decoder_slot_ids(gten_hip_decoder* dc, int seq, int n_from, int count, int32_t* ids)
{
    if (!dc || seq < 0 || seq >= dc->n_seq || n_from < 1 || count < 0 || n_from + count > dc->d.max_ctx + 1 || !ids) return fail("slot_ids: range");
    for (int i = 0; i < count; i++) ids[i] = dc->tokens[(size_t)seq][(size_t)(n_from + i)];
    return 0;
}
int gten_hip_decoder_slot_ids_all(gten_hip_decoder* dc, const int* n_from, int count, int32_t* ids)
{
    if (!dc || !n_from || !ids || count < 0 || count > 64) return fail("slot_ids_all: arguments");
    for (int q = 0; q < dc->n_seq; q++) {
        if (n_from[q] < 0 || n_from[q] > dc->d.max_ctx + 1) return fail("slot_ids_all: sequence %d from step %d", q, n_from[q]);
        for (int i = 0; i < count; i++) {
            const int n = std::min(n_from[q] + i, dc->d.max_ctx);
            ids[(size_t)q * count + i] = dc->tokens[(size_t)q][(size_t)n];
        }
    }
    return 0;
}
int gten_hip_decoder_time_family(gten_hip_decoder*, int, int, int, double* avg_us, int* launches)
{
    if (avg_us) *avg_us = 1.0;
    if (launches) *launches = 1;
    return 0;
}
int gten_hip_decoder_result_seq(gten_hip_decoder* dc, int seq, int n, int32_t* out)
{
    if (!dc || seq < 0 || seq >= dc->n_seq || n < 1 || n > dc->d.max_ctx || !out) return fail("decoder_result: range");
    *out = dc->tokens[(size_t)seq][(size_t)n];
    return 0;
}
int gten_hip_decoder_result(gten_hip_decoder* dc, int n, int32_t* out) { return gten_hip_decoder_result_seq(dc, 0, n, out); }
int gten_hip_decoder_logits_seq(gten_hip_decoder* dc, int seq, float* lg)
{
    if (!dc || seq < 0 || seq >= dc->n_seq || !lg) return fail("logits_seq: range");
    one_hot(dc, seq, lg);
    return 0;
}

}  // extern "C"
