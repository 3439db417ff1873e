#include <algorithm>
// capi.cpp -- flat C exports of the model-level host code (include/gten_host.h).
#include "../../include/gten_host.h"

#include <cstring>
#include <memory>

#include "tinyllama_model.h"
#include "tokenizer.h"

using namespace gten;

struct gten_host_model {
    gten_host_config cfg;
    std::unique_ptr<TinyLlama> model;
};

namespace {

Dtype to_dtype(int code)
{
    switch (code) {
    case GTEN_I32: return kInt32;
    case GTEN_F16: return kFloat16;
    case GTEN_F32: return kFloat32;
    case GTEN_Q8: return kQint8;
    case GTEN_Q4: return kQint4;
    }
    GTEN_ASSERTM(false, "bad dtype code %d", code);
    return kFloat32;
}

TinyLLamaParams to_params(const gten_host_config& c)
{
    TinyLLamaParams p;
    p.n_vocab = c.n_vocab; p.max_ctx = c.max_ctx; p.n_embd = c.n_embd; p.n_ffn = c.n_ffn;
    p.n_layers = c.n_layers; p.n_heads = c.n_heads; p.n_query_groups = c.n_kv_heads;
    return p;
}

} // namespace

extern "C" {

void gten_host_default_config(gten_host_config* cfg, int wdtype, int adtype)
{
    const TinyLLamaParams p;
    cfg->n_vocab = p.n_vocab; cfg->max_ctx = p.max_ctx; cfg->n_embd = p.n_embd; cfg->n_ffn = p.n_ffn;
    cfg->n_layers = p.n_layers; cfg->n_heads = p.n_heads; cfg->n_kv_heads = p.n_query_groups;
    cfg->wdtype = wdtype; cfg->adtype = adtype;
}

gten_host_model* gten_host_model_create(const gten_host_config* cfg)
{
    if (!cfg) return nullptr;
    auto* m = new gten_host_model;
    m->cfg = *cfg;
    m->model.reset(new TinyLlama(cfg->max_ctx, ModuleDtype{to_dtype(cfg->wdtype), to_dtype(cfg->adtype)}, to_params(*cfg)));
    return m;
}

void gten_host_model_free(gten_host_model* m) { delete m; }

int gten_host_model_n_weights(const gten_host_model* m) { return m->model->n_weights(); }

size_t gten_host_model_weight_bytes(gten_host_model* m, int idx) { return m->model->weight(idx).nbytes(); }

int gten_host_model_set_weight(gten_host_model* m, int idx, const void* bytes, size_t nbytes)
{
    if (idx < 0 || idx >= m->model->n_weights()) return -1;
    Tensor& w = m->model->weight(idx);
    if (nbytes != w.nbytes()) return -2;
    std::memcpy(w.data_ptr<char>(), bytes, nbytes);
    w.device_weight();
    return 0;
}

int gten_host_model_load_gten(gten_host_model* m, const char* path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    m->model->load_from_ckpt(f);
    return 0;
}

int gten_host_model_load_synthetic(gten_host_model* m, uint64_t seed)
{
    m->model->load_synthetic(seed);
    return 0;
}

int gten_host_model_logits(gten_host_model* m, const int32_t* tokens, int n, int start_pos, float* logits_out)
{
    if (!tokens || n <= 0 || start_pos < 0 || start_pos >= n) return -1;
    Tensor tk(tokens, {n}, kInt32);
    const Tensor lg = m->model->logits(tk, start_pos);
    if (logits_out) std::memcpy(logits_out, lg.data_ptr<float>(), (size_t)lg.numel() * sizeof(float));
    return 0;
}

int gten_host_model_greedy(gten_host_model* m, int32_t* tokens, int n_prompt, int max_tokens, int eos)
{
    std::vector<int32_t> t(tokens, tokens + n_prompt);
    t.reserve((size_t)max_tokens);
    const int total = greedy_sample(*m->model, t, max_tokens, eos);
    std::memcpy(tokens, t.data(), (size_t)total * sizeof(int32_t));
    return total;
}

int gten_host_model_generate(gten_host_model* m, int32_t* tokens, int n_prompt, int max_tokens, int eos)
{
    std::vector<int32_t> t(tokens, tokens + n_prompt);
    t.reserve((size_t)max_tokens);
    const int total = greedy_generate(*m->model, t, max_tokens, eos);
    std::memcpy(tokens, t.data(), (size_t)total * sizeof(int32_t));
    return total;
}

// ---- tokenizer (host/tokenizer.h): ids and pieces of the reference's tokenizer.h on the same vocabulary file
struct gten_host_tokenizer { Tokenizer tok; gten_host_tokenizer(const char* path, int vocab) : tok(path, vocab) {} };

gten_host_tokenizer* gten_host_tokenizer_create(const char* path, int vocab_size)
{
    if (!path || vocab_size <= 0) return nullptr;
    FILE* f = std::fopen(path, "rb");              // (the class exits the process on a missing file, as the reference does)
    if (!f) return nullptr;
    std::fclose(f);
    return new gten_host_tokenizer(path, vocab_size);
}
void gten_host_tokenizer_free(gten_host_tokenizer* t) { delete t; }
int gten_host_tokenizer_encode(gten_host_tokenizer* t, const char* prompt, int chat_template, int32_t* ids_out, int cap)
{
    if (!t || !prompt || !ids_out) return -1;
    std::string p(prompt);
    const std::vector<int> ids = chat_template ? t->tok.encode(p) : t->tok.encode_plain(p);
    if ((int)ids.size() > cap) return -(int)ids.size();
    for (size_t i = 0; i < ids.size(); i++) ids_out[i] = ids[i];
    return (int)ids.size();
}
const char* gten_host_tokenizer_decode(gten_host_tokenizer* t, int prev_token, int token)
{
    return t ? t->tok.decode(prev_token, token) : "";
}

int gten_host_model_set_fast_decode(gten_host_model* m, int on)
{
    m->model->set_fast_decode(on != 0);
    return 0;
}

int gten_host_model_decode_begin(gten_host_model* m, const int32_t* tokens, int count)
{
    if (!tokens || count <= 0 || count > m->cfg.max_ctx) return -1;
    m->model->decode_set_tokens(tokens, 0, count);
    return 0;
}

int gten_host_model_decode_step(gten_host_model* m, int n, int use_graph)
{
    if (n < 1 || n > m->cfg.max_ctx) return -1;
    m->model->decode_step(n, use_graph != 0);
    return 0;
}

int gten_host_model_decode_steps(gten_host_model* m, int n_first, int count, int use_graph)
{
    m->model->decode_steps(n_first, count, use_graph != 0);
    return 0;
}

int gten_host_model_time_family(gten_host_model* m, int family, int n, int reps, double* avg_us, int* launches)
{
    if (!avg_us) return -1;
    *avg_us = m->model->decode_time_family(family, n, reps, launches);
    return *avg_us < 0.0 ? -4 : 0;
}

int gten_host_model_decode_result(gten_host_model* m, int n, int32_t* argmax_out)
{
    if (!argmax_out) return -1;
    *argmax_out = m->model->decode_result(n);
    return 0;
}

// ---- several sequences sharing one copy of the weights
struct gten_host_batch {
    gten_host_config cfg;
    std::unique_ptr<TinyLlamaBatch> batch;
};

gten_host_batch* gten_host_batch_create(const gten_host_config* cfg, int n_seq)
{
    if (!cfg || !(n_seq == 2 || n_seq == 4 || n_seq == 8 || (n_seq >= 16 && n_seq <= 64 && n_seq % 16 == 0) || (n_seq > 64 && n_seq <= 256 && n_seq % 64 == 0) || n_seq == 384 || n_seq == 512)) return nullptr;
    auto* b = new gten_host_batch;
    b->cfg = *cfg;
    b->batch.reset(new TinyLlamaBatch(n_seq, cfg->max_ctx, ModuleDtype{to_dtype(cfg->wdtype), to_dtype(cfg->adtype)}, to_params(*cfg)));
    return b;
}

void gten_host_batch_free(gten_host_batch* b) { delete b; }

int gten_host_batch_load_synthetic(gten_host_batch* b, uint64_t seed)
{
    b->batch->load_synthetic(seed);
    return 0;
}

int gten_host_batch_set_weight(gten_host_batch* b, int idx, const void* bytes, size_t nbytes)
{
    TinyLlama& m0 = b->batch->seq(0);
    if (idx < 0 || idx >= m0.n_weights()) return -1;
    Tensor& w = m0.weight(idx);
    if (nbytes != w.nbytes()) return -2;
    std::memcpy(w.data_ptr<char>(), bytes, nbytes);
    w.device_weight();
    if (idx == m0.n_weights() - 1) b->batch->share_weights();     // last tensor in: alias them all
    return 0;
}

int gten_host_batch_prefill(gten_host_batch* b, int seq, const int32_t* tokens, int n, float* logits_out)
{
    if (seq < 0 || seq >= b->batch->n_seq() || !tokens || n <= 0) return -1;
    const std::vector<int32_t> prompt(tokens, tokens + n);
    b->batch->prefill(seq, prompt, logits_out);                  // (wide batches: the segmented prompt path, TinyLlamaBatch::prefill)
    return 0;
}

int gten_host_batch_prefill_many(gten_host_batch* b, const int32_t* seqs, const int32_t* tokens, const int32_t* starts, int n_prompts, float* logits_out)
{
    if (!b || !seqs || !tokens || !starts || n_prompts < 1 || n_prompts > TinyLlamaBatch::kPreMax || starts[0] != 0) return -1;
    if (!b->batch->batched_prompts()) return -2;
    std::vector<std::vector<int32_t>> prompts((size_t)n_prompts);
    std::vector<const std::vector<int32_t>*> ps;
    std::vector<int> slots;
    std::vector<float*> lo;
    for (int k = 0; k < n_prompts; k++) {
        const int len = starts[k + 1] - starts[k];
        if (seqs[k] < 0 || seqs[k] >= b->batch->n_seq() || len < 16 || len > b->cfg.max_ctx || starts[k + 1] > TinyLlamaBatch::kPreRows) return -1;
        for (int j = 0; j < k; j++)
            if (seqs[j] == seqs[k]) return -1;        // two prompts onto ONE slot's caches: the copy launch would race
        prompts[(size_t)k].assign(tokens + starts[k], tokens + starts[k + 1]);
        ps.push_back(&prompts[(size_t)k]);
        slots.push_back(seqs[k]);
        lo.push_back(logits_out ? logits_out + (size_t)k * b->cfg.n_vocab : nullptr);
    }
    std::vector<int> first;
    b->batch->prefill_many(slots, ps, &first, &lo);
    return 0;
}

int gten_host_batch_decode_begin(gten_host_batch* b, int seq, const int32_t* tokens, int count)
{
    if (seq < 0 || seq >= b->batch->n_seq() || !tokens || count <= 0 || count > b->cfg.max_ctx) return -1;
    b->batch->decode_set_tokens(seq, tokens, 0, count);
    return 0;
}

int gten_host_batch_decode_step(gten_host_batch* b, int n, int use_graph)
{
    if (n < 1 || n > b->cfg.max_ctx) return -1;
    b->batch->decode_step(n, use_graph != 0);
    return 0;
}

int gten_host_batch_decode_steps(gten_host_batch* b, int n_first, int count, int use_graph)
{
    b->batch->decode_steps(n_first, count, use_graph != 0);
    return 0;
}

int gten_host_batch_decode_step_ragged(gten_host_batch* b, const int32_t* n_per_seq, int use_graph)
{
    if (!n_per_seq) return -1;
    for (int q = 0; q < b->batch->n_seq(); q++)
        if (n_per_seq[q] < 1 || n_per_seq[q] > b->cfg.max_ctx) return -1;
    b->batch->decode_step_ragged(n_per_seq, use_graph != 0);
    return 0;
}

// prompts [n_seq][max_prompt] (sequence q uses its first n_prompt[q] ids): each prompt is processed on its sequence's
// own caches (the reference's logits() call, host argmax of the first new id), then all sequences generate together
// with the sampler on the device.  out is [n_seq][max_tokens]: prompt + new ids; n_total [n_seq].
int gten_host_batch_generate(gten_host_batch* b, const int32_t* prompts, const int32_t* n_prompt, int max_prompt, int max_tokens, int eos,
                             int32_t* out, int32_t* n_total)
{
    if (!prompts || !n_prompt || !out || !n_total || max_tokens <= 0) return -1;
    const int S = b->batch->n_seq();
    std::vector<int> n_first((size_t)S), room((size_t)S);
    int max_new = 0;
    for (int q = 0; q < S; q++) {
        const int P = n_prompt[q];
        if (P <= 0 || P > max_prompt || P >= max_tokens || P >= b->cfg.max_ctx) return -1;
        int32_t* row = out + (size_t)q * max_tokens;
        std::memcpy(row, prompts + (size_t)q * max_prompt, (size_t)P * sizeof(int32_t));
        const int best_i = b->batch->prefill(q, std::vector<int32_t>(row, row + P));
        row[P] = best_i;                                          // (an eos here ends the sequence below)
        n_first[q] = P + 1;
        b->batch->decode_set_tokens(q, row, 0, P + 1);
        // this sequence's own room; none when the prompt's argmax is already eos (the sequence is then parked from the start
        // instead of being decoded for the longest sequence's length)
        room[(size_t)q] = (best_i == eos) ? 0 : max_tokens - (P + 1);
        max_new = std::max(max_new, room[(size_t)q]);
    }
    std::vector<int32_t> gen((size_t)S * (size_t)std::max(max_new, 1));
    std::vector<int> n_out((size_t)S, 0);
    b->batch->decode_generate(n_first.data(), max_new, eos, gen.data(), n_out.data(), room.data());
    for (int q = 0; q < S; q++) {
        int32_t* row = out + (size_t)q * max_tokens;
        int total = n_first[q];
        if (row[total - 1] == eos) { n_total[q] = total - 1; continue; }
        const int room = max_tokens - total;
        const int take = std::min(n_out[q], room);
        std::memcpy(row + total, gen.data() + (size_t)q * max_new, (size_t)take * sizeof(int32_t));
        n_total[q] = total + take;
    }
    return 0;
}

int gten_host_batch_set_serve_schedule(gten_host_batch* b, int k)
{
    if (!b || k < 0) return -4;
    b->batch->set_serve_schedule(k);
    return 0;
}

int gten_host_batch_set_serve_spares(gten_host_batch* b, int n)
{
    if (!b || n < -1 || n > 256) return -4;
    b->batch->set_serve_spares(n);
    return 0;
}

int gten_host_batch_set_serve_ramp(gten_host_batch* b, int percent)
{
    if (!b || percent < 0 || percent > 100) return -4;
    b->batch->set_serve_ramp(percent);
    return 0;
}

int gten_host_batch_serve2(gten_host_batch* b, const int32_t* prompts, const int32_t* n_prompt, int n_prompts, int max_prompt,
                           int max_tokens, int eos, int slice, int max_new, const int32_t* max_new_each, int32_t* out, int32_t* n_total, double* stats,
                           int n_stats)
{
    if (!prompts || !n_prompt || !out || !n_total || n_prompts <= 0 || max_tokens <= 0 || slice <= 0 || n_stats < 0) return -1;
    std::vector<std::vector<int32_t>> ps((size_t)n_prompts), res;
    for (int j = 0; j < n_prompts; j++) {
        if (n_prompt[j] <= 0 || n_prompt[j] > max_prompt || n_prompt[j] > b->cfg.max_ctx) return -1;
        ps[(size_t)j].assign(prompts + (size_t)j * max_prompt, prompts + (size_t)j * max_prompt + n_prompt[j]);
    }
    const TinyLlamaBatch::ServeStats st = b->batch->serve(ps, max_tokens, eos, slice, &res, max_new, max_new_each);
    for (int j = 0; j < n_prompts; j++) {
        const int take = std::min((int)res[(size_t)j].size(), std::max(max_tokens, n_prompt[j]));
        std::memcpy(out + (size_t)j * std::max(max_tokens, max_prompt), res[(size_t)j].data(), (size_t)take * sizeof(int32_t));
        n_total[j] = take;
    }
    if (stats) {
        // (exactly n_stats doubles are written: a caller sized for an older, shorter list stays inside its array)
        const double all[] = {(double)st.prompt_tokens, (double)st.new_tokens, (double)st.steps, (double)st.admissions, st.prefill_s, st.decode_s,
                              (double)st.lane_steps, (double)st.lane_rows, (double)st.moved};
        const int have = (int)(sizeof(all) / sizeof(all[0]));
        for (int i = 0; i < n_stats; i++) stats[i] = i < have ? all[i] : 0.0;
    }
    return 0;
}

/* the entry point as first published: SIX doubles (a caller compiled against that header holds double[6]) */
int gten_host_batch_serve(gten_host_batch* b, const int32_t* prompts, const int32_t* n_prompt, int n_prompts, int max_prompt,
                          int max_tokens, int eos, int slice, int max_new, const int32_t* max_new_each, int32_t* out, int32_t* n_total, double* stats)
{
    return gten_host_batch_serve2(b, prompts, n_prompt, n_prompts, max_prompt, max_tokens, eos, slice, max_new, max_new_each, out, n_total, stats, stats ? 6 : 0);
}

int gten_host_batch_decode_result(gten_host_batch* b, int seq, int n, int32_t* argmax_out)
{
    if (!argmax_out || seq < 0 || seq >= b->batch->n_seq()) return -1;
    *argmax_out = b->batch->decode_result(seq, n);
    return 0;
}

int gten_host_batch_logits(gten_host_batch* b, int seq, float* logits_out)
{
    if (!logits_out || seq < 0 || seq >= b->batch->n_seq()) return -1;
    b->batch->decode_logits(seq, logits_out);
    return 0;
}

int gten_host_batch_time_family(gten_host_batch* b, int family, int n, int reps, double* avg_us, int* launches)
{
    if (!avg_us) return -1;
    *avg_us = b->batch->decode_time_family(family, n, reps, launches);
    return *avg_us < 0.0 ? -4 : 0;
}

int gten_host_batch_seq_steps(gten_host_batch* b, int seq, const int32_t* tokens, int count, int n_first, int steps)
{
    if (seq < 0 || seq >= b->batch->n_seq() || !tokens || count <= 0 || count > b->cfg.max_ctx || n_first < 1 || steps < 0 ||
        n_first + steps - 1 > count)
        return -1;
    TinyLlama& m = b->batch->seq(seq);                 // the sequence's own model object: its single-sequence decoder on the SAME caches
    m.decode_set_tokens(tokens, 0, count);
    m.decode_steps(n_first, steps, true);
    return 0;
}

int gten_host_batch_kv_info(gten_host_batch* b, int* head_major, unsigned long long* seq_imports, unsigned long long* import_launches)
{
    b->batch->kv_info(head_major, seq_imports, import_launches);
    return 0;
}

int gten_host_synth_weight(const gten_host_config* cfg, uint64_t seed, int idx, void* out, size_t nbytes)
{
    std::vector<float> scratch;
    TinyLlama::synth_weight_bytes(to_params(*cfg), ModuleDtype{to_dtype(cfg->wdtype), to_dtype(cfg->adtype)}, seed, idx,
                                  scratch, static_cast<uint8_t*>(out), nbytes);
    return 0;
}

// tinyllama_to_gten.py:94-201: magic, then [len][name][len][name][nbytes][payload]
int gten_host_write_gten(const gten_host_config* cfg, uint64_t seed, const char* path)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) return -1;
    const TinyLLamaParams p = to_params(*cfg);
    const ModuleDtype md{to_dtype(cfg->wdtype), to_dtype(cfg->adtype)};
    const int64_t magic = 0x454c49464e455447LL;
    f.write(reinterpret_cast<const char*>(&magic), 8);
    const int nw = 1 + 9 * p.n_layers + 2;
    static const char* kinds[9] = {"self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj",
                                   "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj", "input_layernorm", "post_attention_layernorm"};
    std::vector<float> scratch;
    std::vector<uint8_t> bytes;
    for (int i = 0; i < nw; i++) {
        std::string name;
        if (i == 0) name = "model.embed_tokens.weight";
        else if (i == nw - 1) name = "lm_head.weight";
        else if (i == nw - 2) name = "model.norm.weight";
        else name = "model.layers." + std::to_string((i - 1) / 9) + "." + kinds[(i - 1) % 9] + ".weight";
        int rows, cols;
        Dtype dt;
        TinyLlama::weight_shape(p, md, i, &rows, &cols, &dt);
        bytes.resize((size_t)rows * synth::row_bytes(dt, cols));
        TinyLlama::synth_weight_bytes(p, md, seed, i, scratch, bytes.data(), bytes.size());
        const int32_t len = (int32_t)name.size(), nbytes = (int32_t)bytes.size();
        for (int rep = 0; rep < 2; rep++) {
            f.write(reinterpret_cast<const char*>(&len), 4);
            f.write(name.data(), len);
        }
        f.write(reinterpret_cast<const char*>(&nbytes), 4);
        f.write(reinterpret_cast<const char*>(bytes.data()), nbytes);
    }
    return f.good() ? 0 : -2;
}

void gten_host_synthetic_tokens(int32_t* out, int count, uint32_t seed, int n_vocab)
{
    const std::vector<int32_t> t = synth::synthetic_tokens(count, seed, n_vocab);
    std::memcpy(out, t.data(), (size_t)count * sizeof(int32_t));
}

} // extern "C"
