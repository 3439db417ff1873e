// tinyllama_model.h -- the caller of the hot path: model wiring, .gten loader
// and greedy loop, written against the HBM-backed gten API.  Counterpart of the
// reference's tinyllama.cpp:12-76 (params, TinyLlama), 301-392 (loader) and
// 395-440 (greedy sampler); CLI, tokenizer and top-k sampling are out of scope
// (SURVEY 8(f) rank 4).
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <fcntl.h>
#include <omp.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <string>
#include <vector>
#include <deque>

#include "../gten/gten.h"
#include "synth.h"

namespace gten {

// tinyllama.cpp:12-20, but runtime-configurable (small parity models)
struct TinyLLamaParams {
    int n_vocab = 32003;
    int max_ctx = 2048;
    int n_embd = 2048;
    int n_ffn = 5632;
    int n_layers = 22;
    int n_heads = 32;
    int n_query_groups = 4;
};

class TinyLlama {
public:
    const TinyLLamaParams params;
    ModuleDtype dtype_;
    int n_ctx_;

public:
    // tinyllama.cpp:30-43
    TinyLlama(const int n_ctx, ModuleDtype dtype, TinyLLamaParams p = TinyLLamaParams{})
        : params{p},
          dtype_{dtype},
          n_ctx_{n_ctx},
          tok_emb_{Embedding(p.n_vocab, p.n_embd, n_ctx, dtype)},
          norm_{RMSNorm(p.n_embd, n_ctx, {kFloat16, dtype.adtype})},
          lm_head_{EmbeddingLinear{p.n_embd, p.n_vocab, n_ctx, {dtype.wdtype, kFloat32}}}
    {
        blocks_.reserve(p.n_layers);
        for (int i = 0; i < p.n_layers; i++)
            blocks_.push_back(AttentionBlock(p.n_heads, p.n_embd, p.n_query_groups, p.n_ffn, n_ctx, dtype));
    }

    // tinyllama.cpp:45-61: the whole token history comes in, rows
    // [start_pos, n) are computed, the result is the f32 logits of the last row.
    Tensor logits(const Tensor& tokens, const int start_pos = 0)
    {
        if (tokens.numel() > n_ctx_) {
            std::cerr << "Number of prompt tokens (" << tokens.numel() << ") exceed provided maximum ctx size (" << n_ctx_ << ")\n";
            std::exit(EXIT_FAILURE);
        }
        // one new row: the fused decode path (same bytes, 6 launches per block)
        if (fast_decode_ && tokens.numel() - start_pos == 1 && tokens.is_host_external()) {
            const int n = tokens.numel();
            const int32_t* ids = static_cast<const int32_t*>(tokens.host_external_ptr());
            decode_set_tokens(ids + (n - 1), n - 1, 1);
            decode_step(n, /*use_graph=*/true);
            (void)lm_head_.acv.device_ptr_mut();      // the step wrote the logits in HBM: host mirror is stale
            return lm_head_.acv;
        }
        // operator by operator (this class drives its own decoder above; with the fast path switched off the
        // modules must not record the row for theirs either -- gten/modules.h)
        struct OpsOnly {
            bool was = detail::fused_rows_enabled();
            OpsOnly() { detail::fused_rows_enabled() = false; }
            ~OpsOnly() { detail::fused_rows_enabled() = was; }
        } ops_only;
        Tensor x = tok_emb_.forward(tokens, start_pos);
        for (auto& block : blocks_) x = block.forward(x, start_pos);
        x = norm_.forward(x, start_pos);
        return lm_head_.forward(x);
    }

    // ---- several prompts as ONE row matrix (TinyLlamaBatch::prefill_many; include/gten_hip.h, gten_hip_set_row_segments):
    // rows [starts[k], starts[k + 1]) of `tokens` are prompt k.  Returns the final-norm rows; logits_of_row gives the
    // logits of one of them.  This object's K / V tensors then hold every prompt's rows, at the prompt's rows.
    Tensor hidden_rows(const Tensor& tokens, const std::vector<int32_t>& starts)
    {
        GTEN_ASSERTM(tokens.numel() <= n_ctx_ && (int)starts.size() >= 2 && starts.back() == tokens.numel(), "hidden_rows: %d rows, context %d",
                     tokens.numel(), n_ctx_);
        struct OpsOnly {
            bool was = detail::fused_rows_enabled();
            OpsOnly() { detail::fused_rows_enabled() = false; }
            ~OpsOnly() { detail::fused_rows_enabled() = was; }
        } ops_only;
        struct Segments {
            explicit Segments(const std::vector<int32_t>& s) { GTEN_HIP_OK(gten_hip_set_row_segments(s.data(), (int)s.size() - 1)); }
            ~Segments() { gten_hip_set_row_segments(nullptr, 0); }
        } segments(starts);
        Tensor x = tok_emb_.forward(tokens, 0);
        for (auto& block : blocks_) x = block.forward(x, 0);
        return norm_.forward(x, 0);
    }
    Tensor logits_of_row(const Tensor& hidden, int row)
    {
        Tensor v = hidden;                                    // (a shallow handle: the same storage, its own shape)
        v.resize({row + 1, params.n_embd});
        return lm_head_.forward(v);                           // EmbeddingLinear computes the last row of what it is given
    }
    AttentionBlock& block(int i) { return blocks_[(size_t)i]; }

    // ---- single-token decode fast path (include/gten_hip.h, "decode fast path")
    void set_fast_decode(bool on) { fast_decode_ = on; }
    bool fast_decode() const { return fast_decode_; }

    // token ids for positions [first, first+count) of the sequence being decoded
    void decode_set_tokens(const int32_t* ids, int first, int count)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_set_tokens(dec_, ids, first, count));
    }
    // asynchronous: row n-1 -> logits (lm_head_.acv, in HBM) and their argmax
    void decode_step(int n, bool use_graph)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_step(dec_, n, use_graph ? 1 : 0));
    }
    // asynchronous: `count` consecutive rows n_first - 1, n_first, ... (ids on the device), four steps per graph replay
    void decode_steps(int n_first, int count, bool use_graph)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_steps(dec_, n_first, count, use_graph ? 1 : 0));
    }
    // device pointers of this model's weights / logits buffer for a decoder (shared code with TinyLlamaBatch)
    void describe(gten_hip_decoder_desc* d, std::vector<gten_hip_layer_ptrs>* L)
    {
        *d = gten_hip_decoder_desc{};
        d->n_vocab = params.n_vocab; d->max_ctx = n_ctx_; d->n_embd = params.n_embd; d->n_ffn = params.n_ffn;
        d->n_layers = params.n_layers; d->n_heads = params.n_heads; d->n_kv_heads = params.n_query_groups;
        d->wdtype = dtype_code(dtype_.wdtype); d->adtype = dtype_code(dtype_.adtype);
        d->embed = tok_emb_.weight.device_weight();
        d->final_norm = norm_.weight.device_weight();
        d->lm_head = lm_head_.weight.device_weight();
        d->logits = static_cast<float*>(lm_head_.acv.device_ptr_mut());
        L->assign((size_t)params.n_layers, gten_hip_layer_ptrs{});
        for (int i = 0; i < params.n_layers; i++) {
            AttentionBlock& b = blocks_[(size_t)i];
            gten_hip_layer_ptrs& p = (*L)[(size_t)i];
            p.wq = b.attn.query.weight.device_weight();
            p.wk = b.attn.key.weight.device_weight();
            p.wv = b.attn.value.weight.device_weight();
            p.wo = b.attn.qkv_proj.weight.device_weight();
            p.wgate = b.ffn_gate_proj.weight.device_weight();
            p.wup = b.ffn_up_proj.weight.device_weight();
            p.wdown = b.ffn_down_proj.weight.device_weight();
            p.attn_norm = b.attn_norm.weight.device_weight();
            p.ffn_norm = b.ffn_norm.weight.device_weight();
            p.kcache = b.attn.key.acv.device_ptr_mut();      // the K/V caches ARE these activation tensors
            p.vcache = b.attn.value.acv.device_ptr_mut();
        }
    }

    // HIP-event timed replay of one kernel family of the decode step (bench.py roofline)
    double decode_time_family(int family, int n, int reps, int* launches)
    {
        ensure_decoder();
        double us = 0.0;
        // (a family this decoder's step does not launch is reported, not fatal: < 0)
        if (gten_hip_decoder_time_family(dec_, family, n, reps, &us, launches) != 0) return -1.0;
        return us;
    }
    int decode_result(int n)
    {
        ensure_decoder();
        int32_t tok = -1;
        GTEN_HIP_OK(gten_hip_decoder_result(dec_, n, &tok));
        return tok;
    }
    // greedy generation with the sampler on the device (gten_hip_decoder_generate): ids [0, n_first) are known, the
    // caches hold rows [0, n_first - 1); returns the number of new ids written to `out`
    int decode_generate(const int32_t* ids, int n_first, int max_new, int eos, int32_t* out)
    {
        decode_set_tokens(ids, 0, n_first);
        int got = 0;
        GTEN_HIP_OK(gten_hip_decoder_generate(dec_, n_first, max_new, eos, out, &got));
        (void)lm_head_.acv.device_ptr_mut();          // the steps wrote logits in HBM: host mirror is stale
        return got;
    }

    ~TinyLlama()
    {
        if (dec_) gten_hip_decoder_destroy(dec_);
    }
    TinyLlama(const TinyLlama&) = delete;
    TinyLlama& operator=(const TinyLlama&) = delete;

    int n_weights() const { return 1 + 9 * params.n_layers + 2; }

    // weight tensors in .gten order (tinyllama.cpp:345-391)
    Tensor& weight(int idx)
    {
        const int last = n_weights() - 1;
        GTEN_ASSERT(idx >= 0 && idx <= last);
        if (idx == 0) return tok_emb_.weight;
        if (idx == last) return lm_head_.weight;
        if (idx == last - 1) return norm_.weight;
        AttentionBlock& b = blocks_[(idx - 1) / 9];
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

    // .gten reader (tinyllama.cpp:301-392): magic, then per tensor
    // [i32 len][name][i32 len][name][i32 nbytes][payload] in fixed order; names
    // are skipped, the payload size must match the tensor.
    void load_from_ckpt(std::ifstream& ckpt)
    {
        Timer load_timer{&load_time};
        int64_t magic = 0;
        ckpt.read(reinterpret_cast<char*>(&magic), sizeof(magic));
        GTEN_ASSERTM(magic == 0x454c49464e455447LL, "Magic number in the binary does not match the expected one.");
        for (int i = 0; i < n_weights(); i++) {
            std::string name;
            for (int rep = 0; rep < 2; rep++) {
                int32_t len = 0;
                ckpt.read(reinterpret_cast<char*>(&len), sizeof(len));
                GTEN_ASSERTM(ckpt.good() && len >= 0 && len < 4096, "Corrupt tensor header in checkpoint.");
                name.resize((size_t)len);
                ckpt.read(name.data(), len);
            }
            int32_t nbytes = 0;
            ckpt.read(reinterpret_cast<char*>(&nbytes), sizeof(nbytes));
            Tensor& w = weight(i);
            GTEN_ASSERTM(static_cast<size_t>(nbytes) == w.nbytes(), "Weight `%s` data size: %d does not match the expected size: %zu.",
                         name.c_str(), nbytes, w.nbytes());
            ckpt.read(w.data_ptr<char>(), nbytes);      // host mirror; staged to HBM below
            GTEN_ASSERTM(ckpt.good(), "Checkpoint ended inside weight `%s`.", name.c_str());
            w.device_weight();                          // upload (+ repack Q8/Q4) and drop the host copy
        }
    }

    // Synthetic weights (host/synth.h): generated, quantized like the reference
    // converter, and staged to HBM tensor by tensor.
    // GTEN_SYNTH_CACHE_DIR (bench.py --gpus N sets it to /dev/shm): the replicas of one node generate the synthetic weights
    // ONCE -- the first process to create the lock file generates them (OpenMP over all cores, seconds) and publishes the
    // file with a rename, the others wait for it and read it -- instead of N OpenMP generations at the same time.
    void load_synthetic(uint64_t seed)
    {
        Timer load_timer{&load_time};
        const char* dir = std::getenv("GTEN_SYNTH_CACHE_DIR");
        if (dir && dir[0] && load_synthetic_cached(dir, seed)) return;
        std::vector<float> f32;
        for (int i = 0; i < n_weights(); i++) {
            Tensor& w = weight(i);
            synth_weight_bytes(params, dtype_, seed, i, f32, w.data_ptr<uint8_t>(), w.nbytes());
            w.device_weight();
        }
    }

    bool load_synthetic_cached(const char* dir, uint64_t seed)
    {
        size_t total = 0;
        for (int i = 0; i < n_weights(); i++) total += weight(i).nbytes();
        char base[512];
        std::snprintf(base, sizeof(base), "%s/gten_synth_v%d_s%llu_w%d_a%d_e%d_f%d_l%d_v%d_h%d_g%d.bin", dir, synth::kFormat, (unsigned long long)seed,
                      (int)dtype_.wdtype, (int)dtype_.adtype, params.n_embd, params.n_ffn, params.n_layers, params.n_vocab, params.n_heads,
                      params.n_query_groups);
        const std::string path = base, lock = path + ".lock", tmp = path + ".tmp";
        auto ready = [&]() { struct stat st; return ::stat(path.c_str(), &st) == 0 && (size_t)st.st_size == total; };
        if (!ready()) {
            // an advisory lock on the lock FILE (never its existence): a generator that dies releases it, nobody waits for a
            // stale file.  Whoever gets the exclusive lock generates; the others block on a shared lock until it is done.
            const int fd = ::open(lock.c_str(), O_CREAT | O_RDWR, 0644);
            if (fd < 0) return false;
            if (::flock(fd, LOCK_EX | LOCK_NB) == 0) {
                if (!ready()) {                                  // (it may have been finished between the check and the lock)
                    const int omp_before = omp_get_max_threads();
                    if (const char* nt = std::getenv("GTEN_SYNTH_GEN_THREADS")) {
                        const int n = std::atoi(nt);
                        if (n > 0) omp_set_num_threads(n);       // the other replicas only wait: every core for this one
                    }
                    std::FILE* f = std::fopen(tmp.c_str(), "wb");
                    bool ok = f != nullptr;
                    std::vector<float> f32;
                    for (int i = 0; i < n_weights(); i++) {
                        Tensor& w = weight(i);
                        synth_weight_bytes(params, dtype_, seed, i, f32, w.data_ptr<uint8_t>(), w.nbytes());
                        if (ok) ok = std::fwrite(w.data_ptr<uint8_t>(), 1, w.nbytes(), f) == w.nbytes();
                        w.device_weight();
                    }
                    if (f) ok = (std::fclose(f) == 0) && ok;
                    if (ok) std::rename(tmp.c_str(), path.c_str());
                    else std::remove(tmp.c_str());
                    omp_set_num_threads(omp_before);             // (the generation is over: this replica's share of the cores again)
                    ::flock(fd, LOCK_UN);
                    ::close(fd);
                    return true;
                }
                ::flock(fd, LOCK_UN);
            } else {
                ::flock(fd, LOCK_SH);                            // blocks while the generator works
                ::flock(fd, LOCK_UN);
            }
            ::close(fd);
            if (!ready()) return false;                          // (the generator could not write the file: everybody for themselves)
        }
        std::FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        bool ok = true;
        for (int i = 0; i < n_weights() && ok; i++) {
            Tensor& w = weight(i);
            ok = std::fread(w.data_ptr<uint8_t>(), 1, w.nbytes(), f) == w.nbytes();
            if (ok) w.device_weight();
        }
        std::fclose(f);
        return ok;                                               // (a short read: the caller generates everything again)
    }

    // shape of tensor idx: rows, cols, storage dtype
    static void weight_shape(const TinyLLamaParams& p, ModuleDtype md, int idx, int* rows, int* cols, Dtype* dt)
    {
        const int E = p.n_embd, F = p.n_ffn, V = p.n_vocab, KV = (E / p.n_heads) * p.n_query_groups;
        const int last = 1 + 9 * p.n_layers + 1;
        *dt = md.wdtype;
        if (idx == 0 || idx == last) { *rows = V; *cols = E; return; }
        if (idx == last - 1) { *rows = 1; *cols = E; *dt = kFloat16; return; }
        switch ((idx - 1) % 9) {
        case 0: case 3: *rows = E; *cols = E; break;
        case 1: case 2: *rows = KV; *cols = E; break;
        case 4: case 5: *rows = F; *cols = E; break;
        case 6: *rows = E; *cols = F; break;
        default: *rows = 1; *cols = E; *dt = kFloat16; break;
        }
    }

    static void synth_weight_bytes(const TinyLLamaParams& p, ModuleDtype md, uint64_t seed, int idx,
                                   std::vector<float>& scratch, uint8_t* out, size_t nbytes)
    {
        int rows, cols;
        Dtype dt;
        weight_shape(p, md, idx, &rows, &cols, &dt);
        GTEN_ASSERTM((size_t)rows * synth::row_bytes(dt, cols) == nbytes, "synthetic weight %d: size mismatch", idx);
        const bool is_norm = (rows == 1 && dt == kFloat16);
        scratch.resize((size_t)rows * cols);
        synth::fill_normal(scratch.data(), scratch.size(), seed, (uint64_t)idx, is_norm ? 1.0f : 0.0f, is_norm ? 0.05f : 0.02f);
        synth::quantize_weight(scratch.data(), rows, cols, dt, out);
    }

    // Lin / Attn / Other split in the spirit of tinyllama.cpp:515-582.  The times
    // are meaningful only with GTEN_HIP_SYNC_TIMERS=1 (see gten/modules.h).
    void print_perf(const int n_pred_tokens)
    {
        int64_t lin = lm_head_.exec_time, attn = 0, other = tok_emb_.exec_time + norm_.exec_time;
        for (auto& b : blocks_) {
            lin += b.attn.query.exec_time + b.attn.key.exec_time + b.attn.value.exec_time + b.attn.qkv_proj.exec_time +
                   b.ffn_gate_proj.exec_time + b.ffn_up_proj.exec_time + b.ffn_down_proj.exec_time;
            attn += b.attn.exec_time_attn;
            other += b.attn_norm.exec_time + b.ffn_norm.exec_time + b.inp_res.exec_time + b.attn_res.exec_time +
                     b.ffn_mul.exec_time + b.ffn_silu.exec_time + b.attn.q_rope.exec_time + b.attn.k_rope.exec_time;
        }
        const int n = n_pred_tokens > 0 ? n_pred_tokens : 1;
        std::cout << "tokens: " << n_pred_tokens << "  ms/tok: linear " << lin / n << "  attention " << attn / n << "  other "
                  << other / n << "  sample " << sample_time / n << "  load " << load_time << " ms  tensor bytes "
                  << G_TensorMemAllocated / 1000000 << " MB\n";
    }

private:
    Embedding tok_emb_;
    RMSNorm norm_;
    EmbeddingLinear lm_head_;
    std::vector<AttentionBlock> blocks_;
    gten_hip_decoder* dec_ = nullptr;
    bool fast_decode_ = [] { const char* e = std::getenv("GTEN_HIP_FAST_DECODE"); return !(e && e[0] == '0'); }();

    // The decoder works on the SAME HBM tensors the modules own: weights (packed
    // at load), the K/V caches (= attn.key.acv / attn.value.acv) and the logits
    // buffer (= lm_head_.acv); it only adds small per-step scratch of its own.
    void ensure_decoder()
    {
        if (dec_) return;
        gten_hip_decoder_desc d;
        std::vector<gten_hip_layer_ptrs> L;
        describe(&d, &L);
        GTEN_HIP_OK(gten_hip_decoder_create(&d, L.data(), &dec_));
    }

public:
    int64_t load_time = 0;
    int64_t sample_time = 0;
};

// S sequences that share ONE copy of the weights.  Each sequence is a full
// TinyLlama object of the gten API -- its own activation tensors, hence its own K/V caches -- whose
// weight Tensors are shallow copies of sequence 0's (gten Tensors share storage on copy,
// gten/tensor.h:24-29), so the weights exist once in HBM.  Prefill goes through each object's
// operator path; single-token steps of all sequences go through one multi-sequence decoder that
// streams every weight once per step.
class TinyLlamaBatch {
public:
    TinyLlamaBatch(int n_seq, int n_ctx, ModuleDtype dtype, TinyLLamaParams p = TinyLLamaParams{}) : n_ctx_{n_ctx}, dtype_{dtype}, params_{p}
    {
        GTEN_ASSERTM(n_seq == 2 || n_seq == 4 || n_seq == 8 || (n_seq >= 16 && n_seq <= 64 && n_seq % 16 == 0) || (n_seq > 64 && n_seq <= 256 && n_seq % 64 == 0) ||
                         n_seq == 384 || n_seq == 512,
                     "TinyLlamaBatch: n_seq %d not in {2, 4, 8, 16, 32, 48, 64, 128, 192, 256, 384, 512}", n_seq);
        for (int i = 0; i < n_seq; i++) {
            seqs_.emplace_back(new TinyLlama(n_ctx, dtype, p));
            seqs_.back()->set_fast_decode(false);
        }
    }
    ~TinyLlamaBatch()
    {
        if (dec_) gten_hip_decoder_destroy(dec_);
    }
    TinyLlamaBatch(const TinyLlamaBatch&) = delete;
    TinyLlamaBatch& operator=(const TinyLlamaBatch&) = delete;

    int n_seq() const { return (int)seqs_.size(); }
    TinyLlama& seq(int i) { return *seqs_[(size_t)i]; }
    // cache sets: 0 .. n_seq - 1 are the sequences' own K / V caches, n_seq .. n_seq + spares - 1 the spare ones that serve()
    // fills ahead of the slots that will take them (ensure_spares); prefill / prefill_many take a cache set
    TinyLlama& cset(int c) { return c < n_seq() ? *seqs_[(size_t)c] : *spares_[(size_t)(c - n_seq())]; }
    int n_sets() const { return n_seq() + (int)spares_.size(); }
    void ensure_spares(int count)
    {
        while ((int)spares_.size() < count) {
            spares_.emplace_back(new TinyLlama(n_ctx_, dtype_, params_));
            TinyLlama& m = *spares_.back();
            for (int w = 0; w < seqs_[0]->n_weights(); w++) m.weight(w) = seqs_[0]->weight(w);
        }
    }

    // call after sequence 0's weights are loaded: every other sequence aliases them
    void share_weights()
    {
        for (size_t i = 1; i < seqs_.size(); i++)
            for (int w = 0; w < seqs_[0]->n_weights(); w++) seqs_[i]->weight(w) = seqs_[0]->weight(w);
        pre_.reset();                                        // (the shared prompt matrix aliases the weights too: rebuilt on next use)
        // the shared decoder holds device pointers to the weights and to every cache set it was ever bound to (the spares among
        // them): it goes before they do and is rebuilt on next use
        if (dec_) { GTEN_HIP_OK(gten_hip_decoder_destroy(dec_)); dec_ = nullptr; }
        spares_.clear();                                     // (... and so do the spare cache sets)
        set_kv_.clear();
    }
    void load_synthetic(uint64_t seed)
    {
        seqs_[0]->load_synthetic(seed);
        share_weights();
    }

    // ---- prompt processing.  Wide batches (>= 16 sequences) process prompts of >= 16 ids as segments of ONE row matrix
    // (a scratch model `pre_` on the shared weights; gten_hip_set_row_segments): every projection is one W.x launch for all
    // the prompts of a call, RoPE and attention run per prompt, and each prompt's K / V rows are then copied from the
    // matrix into its slot's caches.  A prompt's bits do not depend on what shares the matrix with it (no K loop is shared
    // between workgroups in segmented calls), so serve(), generate() and prefill() agree whatever they batch together.
    // Up to 8 sequences keep the per-sequence operator path: there the ids are those of a lone TinyLlama, bit for bit.
    static constexpr int kPreRows = 4096;      // rows of the shared matrix (GTEN_SEG_MAX_ROWS; a prompt has at most max_ctx of them)
    static constexpr int kPreMax = 32;         // prompts per call (two copy ranges per prompt and layer: GTEN_HIP_MAX_COPY_RANGES)
    bool batched_prompts() const
    {
        return n_seq() >= 16 && gten_hip_row_segments_ok(params_.n_embd, params_.n_ffn, params_.n_heads, params_.n_query_groups,
                                                         dtype_code(dtype_.wdtype), dtype_code(dtype_.adtype)) == 1;
    }
    // prompts[k] (>= 16 ids each, kPreRows in all, at most kPreMax) onto the caches of sequence slots[k]; first[k] = argmax of
    // prompt k's logits (strict >, first maximum: tinyllama.cpp:416-424); logits_out[k], when given, receives them
    void prefill_many(const std::vector<int>& slots, const std::vector<const std::vector<int32_t>*>& prompts, std::vector<int>* first,
                      std::vector<float*>* logits_out = nullptr)
    {
        const int K = (int)slots.size();
        GTEN_ASSERTM(K >= 1 && K <= kPreMax && prompts.size() == slots.size(), "prefill_many: %d prompts", K);
        if (!pre_) {
            pre_.reset(new TinyLlama(kPreRows, dtype_, params_));
            pre_->set_fast_decode(false);
            for (int w = 0; w < seqs_[0]->n_weights(); w++) pre_->weight(w) = seqs_[0]->weight(w);
        }
        std::vector<int32_t> ids, starts{0};
        for (const auto* p : prompts) {
            GTEN_ASSERTM((int)p->size() >= 16 && (int)p->size() <= n_ctx_, "prefill_many: a prompt of %zu ids", p->size());
            ids.insert(ids.end(), p->begin(), p->end());
            starts.push_back((int32_t)ids.size());
        }
        GTEN_ASSERTM((int)ids.size() <= kPreRows, "prefill_many: %zu rows (at most %d)", ids.size(), kPreRows);
        Tensor tk(ids.data(), {(int)ids.size()}, kInt32);
        const Tensor hidden = pre_->hidden_rows(tk, starts);
        // every prompt's K / V rows into its own caches: one launch per layer
        std::vector<gten_hip_copy_range> ranges;
        for (int l = 0; l < params_.n_layers; l++) {
            ranges.clear();
            AttentionBlock& src = pre_->block(l);
            const size_t pitch = (size_t)src.attn.key.acv.bstride(0);
            for (int k = 0; k < K; k++) {
                AttentionBlock& dst = cset(slots[(size_t)k]).block(l);
                const size_t off = (size_t)starts[(size_t)k] * pitch, bytes = prompts[(size_t)k]->size() * pitch;
                ranges.push_back({dst.attn.key.acv.device_ptr_mut(), (const uint8_t*)src.attn.key.acv.device_ptr() + off, bytes});
                ranges.push_back({dst.attn.value.acv.device_ptr_mut(), (const uint8_t*)src.attn.value.acv.device_ptr() + off, bytes});
            }
            GTEN_HIP_OK(gten_hip_copy_ranges(ranges.data(), (int)ranges.size()));
        }
        first->assign((size_t)K, 0);
        // a prompt's first id: its logits row stays on the device and so does the sampler (gten_hip_argmax_row, the greedy rule
        // of tinyllama.cpp:416-424) -- the K ids come back in ONE 4 K-byte copy and one wait instead of K copies of 128 KB, K
        // waits and K host loops over the vocabulary (round 4); a caller that wants a prompt's logits gets them as before
        if (!first_ids_) first_ids_.reset(new Tensor({kPreMax}, kInt32));
        int32_t* ids_dev = (int32_t*)first_ids_->device_ptr_mut();
        for (int k = 0; k < K; k++) {
            const Tensor lg = pre_->logits_of_row(hidden, starts[(size_t)k + 1] - 1);
            GTEN_HIP_OK(gten_hip_argmax_row((const float*)lg.device_ptr(), lg.numel(), ids_dev + k));
            if (logits_out && (*logits_out)[(size_t)k])
                std::memcpy((*logits_out)[(size_t)k], lg.data_ptr<float>(), (size_t)lg.numel() * sizeof(float));   // (waits for the stream)
        }
        std::vector<int32_t> got((size_t)K);
        GTEN_HIP_OK(gten_hip_memcpy_d2h(got.data(), ids_dev, (size_t)K * sizeof(int32_t)));
        for (int k = 0; k < K; k++) (*first)[(size_t)k] = got[(size_t)k];
    }
    // one prompt onto sequence seq_i's caches; returns the argmax of its logits (logits_out may be null)
    int prefill(int seq_i, const std::vector<int32_t>& prompt, float* logits_out = nullptr)
    {
        if (batched_prompts() && (int)prompt.size() >= 16 && (int)prompt.size() <= kPreRows) {
            std::vector<int> first;
            std::vector<float*> lo{logits_out};
            prefill_many({seq_i}, {&prompt}, &first, &lo);
            return first[0];
        }
        Tensor tk(prompt.data(), {(int)prompt.size()}, kInt32);
        const Tensor lg = cset(seq_i).logits(tk, 0);                 // operator path on this cache set's own model object
        const float* p = lg.data_ptr<float>();
        int best_i = 0;
        float best = -std::numeric_limits<float>::infinity();
        for (int j = 0; j < lg.numel(); j++)
            if (p[j] > best) { best = p[j]; best_i = j; }
        if (logits_out) std::memcpy(logits_out, p, (size_t)lg.numel() * sizeof(float));
        return best_i;
    }

    void decode_set_tokens(int seq_i, const int32_t* ids, int first, int count)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_set_tokens_seq(dec_, seq_i, ids, first, count));
    }
    // asynchronous: row n-1 of EVERY sequence
    void decode_step(int n, bool use_graph)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_step(dec_, n, use_graph ? 1 : 0));
    }
    void decode_steps(int n_first, int count, bool use_graph)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_steps(dec_, n_first, count, use_graph ? 1 : 0));
    }
    // greedy generation of every sequence with the sampler on the device (gten_hip_decoder_generate_multi): sequence q's
    // ids [0, n_first[q]) are set and its caches hold rows [0, n_first[q] - 1); out is [n_seq][max_new]
    // (max_new_seq: each sequence's own bound, may be null)
    void decode_generate(const int* n_first, int max_new, int eos, int32_t* out, int* n_out, const int* max_new_seq = nullptr)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_generate_multi(dec_, n_first, max_new_seq, max_new, eos, out, n_out));
    }
    // asynchronous: row n[q]-1 of sequence q (continuous batching)
    void decode_step_ragged(const int* n_per_seq, bool use_graph)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_step_ragged(dec_, n_per_seq, use_graph ? 1 : 0));
    }
    int decode_result(int seq_i, int n)
    {
        ensure_decoder();
        int32_t tok = -1;
        GTEN_HIP_OK(gten_hip_decoder_result_seq(dec_, seq_i, n, &tok));
        return tok;
    }
    void decode_logits(int seq_i, float* out)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_logits_seq(dec_, seq_i, out));
    }
    // head-major K / V shadows of the shared decoder (gten_hip_decoder_kv_info)
    void kv_info(int* head_major, unsigned long long* seq_imports, unsigned long long* import_launches)
    {
        ensure_decoder();
        GTEN_HIP_OK(gten_hip_decoder_kv_info(dec_, head_major, seq_imports, import_launches));
    }
    double decode_time_family(int family, int n, int reps, int* launches)
    {
        ensure_decoder();
        double us = 0.0;
        // (a family this decoder's step does not launch is reported, not fatal: < 0)
        if (gten_hip_decoder_time_family(dec_, family, n, reps, &us, launches) != 0) return -1.0;
        return us;
    }

    // ---- continuous batching: a queue of prompts served through the n_seq slots ------------------------------
    // Every prompt is generated greedily to `max_tokens` ids in all (prompt included) or until `eos` (not stored,
    // tinyllama.cpp:425), as greedy_sample does for one sequence; max_new > 0 additionally bounds the new ids per prompt.
    // One host thread keeps two things going that OVERLAP on the GPU:
    //   * a slice of `slice` shared free-running steps of the live slots, queued on stream 0 (gten_hip_decoder_run);
    //   * while it runs: the prompts of the next sequences, one after the other, each on a free slot's own caches on stream 1
    //     (operator path: iteration 0 of the reference's loop incl. the host argmax of its logits).  A parked slot idles on
    //     the decoder's dummy caches (gten_hip_decoder_slot_park), so nothing the shared steps do touches the caches being
    //     filled.  Prompt processing is host-bound (~500 short launches) and the shared steps are GPU-bound: side by side
    //     they hide each other.
    // Between two prompts the slice is polled (gten_hip_stream_idle): once done its ids are read (eos / length -> the slot is
    // parked and becomes free) and the next slice starts at once with the slots whose prompts became ready meanwhile.  Nobody
    // runs past its last step: a slot that ends inside a slice repeats that step until the slice is over
    // (gten_hip_decoder_slot_start_until; cutting the slice to the shortest remaining run instead left 128 slots with
    // slices of one to three steps and prompt batches of two).  Per sequence the ids are those of
    // generating it alone (bit for bit up to 8 slots; tests/test_serving_gpu.py).
    // (lane_steps: shared steps x the lanes each of them ran -- a lane whose slots are all parked is left out of a run; lane_rows:
    //  slots per lane; new ids / (lane_steps x lane_rows) is the share of COMPUTED slot-steps that produced an id)
    struct ServeStats { int64_t prompt_tokens = 0, new_tokens = 0, steps = 0, admissions = 0, lane_steps = 0, moved = 0; int lane_rows = 0; double prefill_s = 0.0, decode_s = 0.0; };
    // (max_new_each, when given, bounds the new ids of prompt j by max_new_each[j] instead of max_new)
    ServeStats serve(const std::vector<std::vector<int32_t>>& prompts, int max_tokens, int eos, int slice,
                     std::vector<std::vector<int32_t>>* out, int max_new = 0, const int32_t* max_new_each = nullptr)
    {
        using clock = std::chrono::steady_clock;
        ensure_decoder();
        const int S = n_seq();
        ServeStats st;
        out->assign(prompts.size(), {});
        // per slot: prompt index (-1: free), next step, last step, the cache set it decodes on (-1: none)
        std::vector<int> job((size_t)S, -1), cur((size_t)S, 0), last((size_t)S, 0), set_of((size_t)S, -1);
        std::vector<char> live((size_t)S, 0);
        // CACHE SETS (round 4).  A prompt is processed onto a free cache set, not onto a free slot: besides the S sets the
        // sequences own there are `spare` more, so that prompts are ready BEFORE the slots that will take them end -- a slot
        // that ends in a harvest gets a ready prompt's set bound (gten_hip_decoder_slot_bind) and joins the very next slice
        // instead of sitting one or two slices out while its replacement is processed.  pool: the free sets; ready: processed
        // prompts waiting for a slot, in queue order.
        struct Ready { int j, set, cur, last; };
        std::deque<Ready> ready;
        const int spare = serve_spares_ >= 0 ? serve_spares_ : (batched_prompts() ? std::min(S / 4, 64) : 0);
        ensure_spares(spare);
        std::vector<int> pool;
        for (int c = n_sets() - 1; c >= 0; c--) pool.push_back(c);           // (taken from the back: the sequences' own sets first)
        size_t next = 0;
        int n_live = 0;
        slice = std::min(std::max(slice, 1), 64);                          // (gten_hip_decoder_slot_ids_all reads up to 64 steps at once)
        int lane_rows = S, n_lanes = 1;
        GTEN_HIP_OK(gten_hip_decoder_lane_info(dec_, &lane_rows, &n_lanes, nullptr));
        st.lane_rows = lane_rows;
        // free slots in the order they are filled: slots of lanes that already run first (a lane without a live slot sits a
        // run out, so a half-empty queue should occupy as few lanes as possible), then by index
        auto free_slots = [&]() {
            std::vector<int> busy((size_t)n_lanes, 0), fq;
            for (int q = 0; q < S; q++)
                if (job[(size_t)q] >= 0) busy[(size_t)(q / lane_rows)]++;
            for (int pass = 0; pass < 2; pass++)
                for (int q = 0; q < S; q++)
                    if (job[(size_t)q] < 0 && (busy[(size_t)(q / lane_rows)] > 0) == (pass == 0)) fq.push_back(q);
            return fq;
        };
        GTEN_HIP_OK(gten_hip_select_stream(0));
        for (int q = 0; q < S; q++) GTEN_HIP_OK(gten_hip_decoder_slot_park(dec_, q));
        // Whatever way this function is left (a failed check ends the process, but an allocation may throw), every slot goes back
        // to its own sequence's caches: a decoder left bound to spare or foreign sets would read and write the wrong rows in the
        // decode_step / generate that follows -- and dangle once share_weights() drops the spares.
        struct Rebind {
            TinyLlamaBatch* b; int S; bool done = false;
            void run()
            {
                if (done) return;
                done = true;
                gten_hip_select_stream(0);
                for (int q = 0; q < S; q++) {
                    gten_hip_decoder_slot_park(b->dec_, q);
                    gten_hip_decoder_slot_bind(b->dec_, q, b->set_kv(q));
                }
            }
            ~Rebind() { run(); }
        } rebind{this, S};
        // the next prompt of the queue onto a free cache set (stream 1); false when the queue is empty
        auto prepare = [&]() {
            while (next < prompts.size() && !pool.empty()) {
                const int j = (int)next++;
                std::vector<int32_t>& row = (*out)[(size_t)j];
                row = prompts[(size_t)j];
                const int P = (int)row.size();
                GTEN_ASSERTM(P >= 1 && P <= n_ctx_, "serve: prompt %d has %d ids (context %d)", j, P, n_ctx_);
                st.prompt_tokens += P;
                const int mn = max_new_each ? max_new_each[j] : max_new;
                const int limit = std::min(std::min(max_tokens, n_ctx_), mn > 0 ? P + mn : n_ctx_);   // ids in all
                if (P >= limit) continue;                                  // no room to generate: returned as is
                const auto t0 = clock::now();
                const int c = pool.back();
                const int best_i = prefill(c, row);                        // this set's caches now hold rows [0, P) (waits for stream 1 only)
                st.prefill_s += std::chrono::duration<double>(clock::now() - t0).count();
                st.admissions++;
                if (best_i == eos) continue;                               // ended at once: the set takes the next prompt
                row.push_back(best_i);
                st.new_tokens++;
                if ((int)row.size() >= limit) continue;
                pool.pop_back();
                ready.push_back(Ready{j, c, (int)row.size(), limit - 1});
                return true;
            }
            return next < prompts.size();
        };
        // Wide batches: the next prompts of the queue -- as many as there are free cache sets, kPreMax at most, kPreRows rows in
        // all, `cap` when the schedule is fixed -- as ONE row matrix (prefill_many) on stream 1.  Prompts under 16 ids (and
        // configurations the segmented call does not compute) go one by one through prepare().
        const bool batched = batched_prompts();
        auto prepare_many = [&](int cap) -> bool {
            if (pool.empty()) return next < prompts.size();
            const size_t room = std::min<size_t>({pool.size(), (size_t)kPreMax, cap > 0 ? (size_t)cap : (size_t)kPreMax});
            std::vector<int> js, limits;
            int rows = 0;
            while (next < prompts.size() && js.size() < room) {
                const int j = (int)next;
                const int P = (int)prompts[(size_t)j].size();
                GTEN_ASSERTM(P >= 1 && P <= n_ctx_, "serve: prompt %d has %d ids (context %d)", j, P, n_ctx_);
                if (!batched || P < 16) break;                               // (short prompt: one by one below)
                if (rows + P > kPreRows) break;
                next++;
                (*out)[(size_t)j] = prompts[(size_t)j];
                st.prompt_tokens += P;
                const int mn = max_new_each ? max_new_each[j] : max_new;
                const int limit = std::min(std::min(max_tokens, n_ctx_), mn > 0 ? P + mn : n_ctx_);
                if (P >= limit) continue;                                    // no room to generate: returned as is
                js.push_back(j); limits.push_back(limit); rows += P;
            }
            if (js.empty()) {
                if (next >= prompts.size()) return false;
                const int P = (int)prompts[next].size();
                if (batched && P >= 16 && P <= kPreRows) return true;        // (only prompts without room were taken: look again)
                return prepare();
            }
            const auto t0 = clock::now();
            std::vector<int> sets(pool.end() - (long)js.size(), pool.end()), first;
            std::reverse(sets.begin(), sets.end());                          // (the order they would be popped in)
            std::vector<const std::vector<int32_t>*> ps;
            for (int j : js) ps.push_back(&prompts[(size_t)j]);
            prefill_many(sets, ps, &first);
            st.prefill_s += std::chrono::duration<double>(clock::now() - t0).count();
            std::vector<int> unused;
            for (size_t k = 0; k < js.size(); k++) {
                std::vector<int32_t>& row = (*out)[(size_t)js[k]];
                st.admissions++;
                bool taken = false;
                if (first[k] != eos) {                                       // (eos: ended at once, the set stays free)
                    row.push_back(first[k]);
                    st.new_tokens++;
                    if ((int)row.size() < limits[k]) { ready.push_back(Ready{js[k], sets[k], (int)row.size(), limits[k] - 1}); taken = true; }
                }
                if (!taken) unused.push_back(sets[k]);
            }
            pool.resize(pool.size() - js.size());
            for (size_t k = unused.size(); k-- > 0;) pool.push_back(unused[k]);
            return true;
        };
        int cnt = 0;                                                       // steps of the slice in flight (0: none)
        bool queue_left = true;
        bool tail = false;                                                 // the queue is empty: the last sequences are kept in as few lanes as they fit
        auto t_slice = clock::now();
        std::vector<int> ap_seq, ap_first, ap_last;                        // gten_hip_decoder_slots_apply's arguments
        std::vector<const int32_t*> ap_tok;
        // ready prompts take the free slots, then the next slice starts (stream 0, asynchronous)
        auto launch_slice = [&]() {
            // THE TAIL.  Once the queue is empty and nothing waits, the live sequences thin out in every lane alike, and a lane
            // costs a full lane's launches however few of its slots are live.  Whenever they would fit into fewer lanes than they
            // occupy, the sequences of the emptiest lane move: parked here, handed over as ready prompts (their cache sets go
            // with them), started again in the free slots of the lanes that stay -- and the runs of the tail leave empty lanes
            // out (gten_hip_decoder_run_lanes).  Same rows, same ids: a sequence does not know its slot.
            if (n_lanes > 1 && serve_schedule_ == 0 && !queue_left && ready.empty() && n_live > 0) {
                std::vector<int> per((size_t)n_lanes, 0);
                for (int q = 0; q < S; q++) per[(size_t)(q / lane_rows)] += live[(size_t)q] ? 1 : 0;
                int occupied = 0, emptiest = -1;
                for (int g = 0; g < n_lanes; g++)
                    if (per[(size_t)g] > 0) { occupied++; if (emptiest < 0 || per[(size_t)g] < per[(size_t)emptiest]) emptiest = g; }
                if (occupied > (n_live + lane_rows - 1) / lane_rows) {
                    ap_seq.clear(); ap_first.clear(); ap_last.clear();
                    for (int q = emptiest * lane_rows; q < (emptiest + 1) * lane_rows; q++) {
                        if (!live[(size_t)q]) continue;
                        ready.push_back(Ready{job[(size_t)q], set_of[(size_t)q], cur[(size_t)q], last[(size_t)q]});
                        ap_seq.push_back(q); ap_first.push_back(0); ap_last.push_back(0);
                        set_of[(size_t)q] = -1; job[(size_t)q] = -1; live[(size_t)q] = 0; n_live--;
                    }
                    GTEN_HIP_OK(gten_hip_decoder_slots_apply(dec_, (int)ap_seq.size(), ap_seq.data(), ap_first.data(), ap_last.data(), nullptr));
                    st.moved += (int64_t)ap_seq.size();
                }
                tail = true;
            }
            ap_seq.clear(); ap_first.clear(); ap_last.clear(); ap_tok.clear();
            if (!ready.empty()) {
                const std::vector<int> fq = free_slots();
                // (the caches of the joining slots were filled on stream 1: the host has waited for that, and the explicit
                //  stream-to-stream dependency makes the next launch on stream 0 acquire what another queue has written)
                if (!fq.empty()) GTEN_HIP_OK(gten_hip_stream_wait(0, 1));
                for (size_t i = 0; i < fq.size() && !ready.empty(); i++) {
                    const int q = fq[i];
                    const Ready r = ready.front();
                    ready.pop_front();
                    const std::vector<int32_t>& row = (*out)[(size_t)r.j];
                    GTEN_ASSERTM((int)row.size() == r.cur, "serve: prompt %d holds %zu ids at step %d", r.j, row.size(), r.cur);
                    GTEN_HIP_OK(gten_hip_decoder_slot_bind(dec_, q, set_kv(r.set)));
                    set_of[(size_t)q] = r.set; job[(size_t)q] = r.j; cur[(size_t)q] = r.cur; last[(size_t)q] = r.last;
                    // (all joining slots in one call below: their ids, step words and cache-table rows go up behind each other, one wait)
                    ap_seq.push_back(q); ap_first.push_back(r.cur); ap_last.push_back(r.last); ap_tok.push_back(row.data());
                    live[(size_t)q] = 1; n_live++;
                }
            }
            if (!ap_seq.empty())
                GTEN_HIP_OK(gten_hip_decoder_slots_apply(dec_, (int)ap_seq.size(), ap_seq.data(), ap_first.data(), ap_last.data(), ap_tok.data()));
            if (n_live == 0) return;
            // (a slot whose run ends inside the slice repeats its last step until the slice is over, slot_start_until: the slice
            //  is cut only when EVERY live slot ends earlier)
            int longest = 0;
            for (int q = 0; q < S; q++)
                if (live[(size_t)q]) longest = std::max(longest, last[(size_t)q] - cur[(size_t)q] + 1);
            cnt = std::min(std::max(slice, 1), longest);
            t_slice = clock::now();
            GTEN_HIP_OK(gten_hip_decoder_run_lanes(dec_, cnt, tail ? 1 : 0));
            st.steps += cnt;
            int ran = n_lanes;
            GTEN_HIP_OK(gten_hip_decoder_lane_info(dec_, nullptr, nullptr, &ran));
            st.lane_steps += (int64_t)cnt * ran;
        };
        // the ids of the finished slice; slots that ended are parked (their cache sets are free for the next prompts)
        std::vector<int> from((size_t)S, 0);
        std::vector<int32_t> all_ids((size_t)S * (size_t)std::max(slice, 1));
        auto harvest = [&]() {
            // every live slot's ids of the slice in ONE gather + copy (waits for stream 0)
            for (int q = 0; q < S; q++) from[(size_t)q] = live[(size_t)q] ? cur[(size_t)q] : 0;
            GTEN_HIP_OK(gten_hip_decoder_slot_ids_all(dec_, from.data(), cnt, all_ids.data()));
            ap_seq.clear(); ap_first.clear(); ap_last.clear();
            for (int q = 0; q < S; q++) {
                if (!live[(size_t)q]) continue;
                const int got = std::min(cnt, last[(size_t)q] - cur[(size_t)q] + 1);                  // (its steps of this slice)
                const int32_t* ids = all_ids.data() + (size_t)q * (size_t)cnt;
                std::vector<int32_t>& row = (*out)[(size_t)job[(size_t)q]];
                bool stop = false;
                for (int i = 0; i < got && !stop; i++) {
                    if (ids[(size_t)i] == eos) stop = true;
                    else { row.push_back(ids[(size_t)i]); st.new_tokens++; }
                }
                cur[(size_t)q] += got;
                if (stop || cur[(size_t)q] > last[(size_t)q]) {
                    ap_seq.push_back(q); ap_first.push_back(0); ap_last.push_back(0);                  // parked, all of them at once below
                    pool.push_back(set_of[(size_t)q]);
                    set_of[(size_t)q] = -1; job[(size_t)q] = -1; live[(size_t)q] = 0; n_live--;
                }
            }
            if (!ap_seq.empty()) GTEN_HIP_OK(gten_hip_decoder_slots_apply(dec_, (int)ap_seq.size(), ap_seq.data(), ap_first.data(), ap_last.data(), nullptr));
            st.decode_s += std::chrono::duration<double>(clock::now() - t_slice).count();
            cnt = 0;
            GTEN_HIP_OK(gten_hip_stream_wait(1, 0));            // (... and the other way round for the cache sets the parked slots leave)
        };
        // One host thread: prompts are processed back to back on stream 1; between two prompt calls the slice on stream 0 is
        // polled, and when it has finished its ids are read and the next slice (with the prompts that became ready) starts.
        int prepared_this_slice = 0;
        for (;;) {
            if (cnt > 0) {
                int idle = 0;
                if (serve_schedule_ > 0) idle = prepared_this_slice >= serve_schedule_;   // fixed schedule (tests): k prompts per slice
                else GTEN_HIP_OK(gten_hip_stream_idle(0, &idle));
                if (!idle && queue_left && !pool.empty()) {                 // slice still running: more prompts beside it
                    GTEN_HIP_OK(gten_hip_select_stream(1));
                    const int before = (int)st.admissions;
                    queue_left = batched ? prepare_many(serve_schedule_ > 0 ? serve_schedule_ - prepared_this_slice : 0) : prepare();
                    GTEN_HIP_OK(gten_hip_select_stream(0));
                    prepared_this_slice += batched ? std::max(1, (int)st.admissions - before) : 1;
                    continue;
                }
                harvest();                                                  // (waits when nothing is left to prepare)
                prepared_this_slice = 0;
            }
            if (n_live == 0 && ready.empty()) {                             // nothing to decode: a prompt first
                if (!queue_left || pool.empty()) break;                     // queue empty, nothing in flight
                GTEN_HIP_OK(gten_hip_select_stream(1));
                queue_left = batched ? prepare_many(serve_schedule_) : prepare();
                // (with nothing live there is nothing to run beside: prompts are processed -- at the prompt kernels' full rate --
                //  until serve_ramp_ percent of the slots have one, and only then does the first slice start: 669 instead of 709
                //  shared steps on the bench's queue, its first 256 prompts 31.3 k instead of 28.6 k new ids/s; the price is the
                //  first ids' latency, ~0.4 s at full size.  0: the first slice starts with the first batch of prompts)
                while (serve_schedule_ == 0 && batched && queue_left && !pool.empty() && (int)ready.size() * 100 < S * serve_ramp_)
                    queue_left = prepare_many(0);
                GTEN_HIP_OK(gten_hip_select_stream(0));
                if (ready.empty()) { if (!queue_left) break; continue; }
            }
            launch_slice();
        }
        // every slot back on its own sequence's caches (all of them are parked now): what follows a serve() -- decode_step,
        // generate -- addresses slot q as sequence q
        for (int q = 0; q < S; q++) GTEN_HIP_OK(gten_hip_decoder_slot_bind(dec_, q, set_kv(q)));
        rebind.done = true;
        return st;
    }

    // Cache sets beyond the sequences' own that serve() fills ahead (-1: a quarter of the slots, at most 64, for wide batches; 0:
    // a prompt is only processed once a slot is free, the behaviour before round 4)
    void set_serve_spares(int n) { serve_spares_ = n; }
    // Percent of the slots that get a processed prompt before a queue's first slice starts (wide batches; default 100)
    void set_serve_ramp(int percent) { serve_ramp_ = std::min(std::max(percent, 0), 100); }

    // Tests: k > 0 fixes the admission schedule -- exactly k prompts are processed beside every slice (as far as slots and
    // queue allow) instead of "as many as fit while the slice runs", so that a run is repeatable slice by slice.
    void set_serve_schedule(int k) { serve_schedule_ = k; }

private:
    std::vector<std::unique_ptr<TinyLlama>> seqs_;
    std::vector<std::unique_ptr<TinyLlama>> spares_;   // spare cache sets of serve()
    std::vector<std::vector<gten_hip_kv_ptrs>> set_kv_;      // per cache set: its layers' cache pointers (gten_hip_decoder_slot_bind)
    std::unique_ptr<TinyLlama> pre_;         // the shared row matrix of batched prompt processing (prefill_many), made on first use
    std::unique_ptr<Tensor> first_ids_;      // [kPreMax] int32 on the device: the prompts' first ids (gten_hip_argmax_row)
    gten_hip_decoder* dec_ = nullptr;
    int n_ctx_;
    ModuleDtype dtype_;
    TinyLLamaParams params_;
    int serve_schedule_ = 0;
    int serve_spares_ = -1;
    int serve_ramp_ = 100;

    const gten_hip_kv_ptrs* set_kv(int c)
    {
        if ((int)set_kv_.size() < n_sets()) set_kv_.resize((size_t)n_sets());
        std::vector<gten_hip_kv_ptrs>& kv = set_kv_[(size_t)c];
        if (kv.empty()) {
            gten_hip_decoder_desc di;
            std::vector<gten_hip_layer_ptrs> Li;
            cset(c).describe(&di, &Li);
            for (auto& p : Li) kv.push_back(gten_hip_kv_ptrs{p.kcache, p.vcache});
        }
        return kv.data();
    }

    void ensure_decoder()
    {
        if (dec_) return;
        gten_hip_decoder_desc d;
        std::vector<gten_hip_layer_ptrs> L, Li;
        seqs_[0]->describe(&d, &L);
        std::vector<gten_hip_kv_ptrs> kv;
        for (auto& m : seqs_) {
            gten_hip_decoder_desc di;
            m->describe(&di, &Li);
            for (auto& p : Li) kv.push_back(gten_hip_kv_ptrs{p.kcache, p.vcache});
        }
        GTEN_HIP_OK(gten_hip_decoder_create_multi(&d, L.data(), kv.data(), (int)seqs_.size(), &dec_));
    }
};

// Greedy loop of tinyllama.cpp:395-440 on token ids (no tokenizer): iteration 0
// is the prefill (start_pos 0), later iterations compute one row; argmax with
// strict '>' so the first maximum wins; stop at `eos`.
inline int greedy_sample(TinyLlama& model, std::vector<int32_t>& tokens, const int n_predict, const int eos)
{
    const int max_iters = n_predict - (int)tokens.size();
    for (int i = 0; i < max_iters; i++) {
        Tensor input{tokens.data(), {(int)tokens.size()}, kInt32};
        const int start_pos = (i == 0) ? 0 : input.numel() - 1;
        Tensor logits = model.logits(input, start_pos);
        Timer sample_timer{&model.sample_time};
        const int n = logits.numel();
        const float* p = const_cast<const Tensor&>(logits).data_ptr<float>();
        float best = -std::numeric_limits<float>::infinity();
        int best_i = 0;
        for (int j = 0; j < n; j++)
            if (p[j] > best) { best = p[j]; best_i = j; }
        if (best_i == eos) break;
        tokens.push_back(best_i);
    }
    return (int)tokens.size();
}

// The same loop with the sampler on the device: the prompt is processed as in the reference (iteration 0 above, host
// argmax of its logits), every later token comes from back-to-back graph replays whose argmax feeds the next step on the
// device -- no logits copy, no host argmax, no host round trip per token.  Same ids as greedy_sample (tested).
inline int greedy_generate(TinyLlama& model, std::vector<int32_t>& tokens, const int n_predict, const int eos)
{
    if ((int)tokens.size() >= n_predict) return (int)tokens.size();
    {
        Tensor input{tokens.data(), {(int)tokens.size()}, kInt32};
        Tensor logits = model.logits(input, 0);
        const int n = logits.numel();
        const float* p = const_cast<const Tensor&>(logits).data_ptr<float>();
        float best = -std::numeric_limits<float>::infinity();
        int best_i = 0;
        for (int j = 0; j < n; j++)
            if (p[j] > best) { best = p[j]; best_i = j; }
        if (best_i == eos) return (int)tokens.size();
        tokens.push_back(best_i);
    }
    const int n_first = (int)tokens.size();
    const int max_new = n_predict - n_first;
    if (max_new <= 0) return n_first;
    std::vector<int32_t> out((size_t)max_new);
    const int got = model.decode_generate(tokens.data(), n_first, max_new, eos, out.data());
    tokens.insert(tokens.end(), out.begin(), out.begin() + got);
    return (int)tokens.size();
}

} // namespace gten
