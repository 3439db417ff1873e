// not gpu: the host-side C++ ABOVE the HIP C-ABI -- gten modules (recording of single-row calls, the composed block call),
// TinyLlama / TinyLlamaBatch, the continuous-batching scheduler, capi.cpp -- run on the CPU against tests/hip_stub.cpp (a
// stand-in for libgten_hip.so whose "model" is a fixed next-id rule) in a binary built with -fsanitize=address,undefined
// (tests/test_host_sanitize_cpu.py).  What is checked besides the sanitizers: every way of generating -- the reference's
// loop through logits(), the device sampler, the fixed batch, the queue through the slots under several admission schedules
// and slice lengths, one by one or (16 slots) several prompts per row matrix -- returns the same ids.  Exits 0 when every check holds.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/gten_hip.h"
#include "../include/gten_host.h"

#define CHECK(c)                                                              \
    do {                                                                      \
        if (!(c)) { std::fprintf(stderr, "host_sanitize_serve: %s failed (line %d)\n", #c, __LINE__); return 1; } \
    } while (0)

int main()
{
    gten_host_config cfg{};
    cfg.n_vocab = 97; cfg.max_ctx = 96; cfg.n_embd = 256; cfg.n_ffn = 512; cfg.n_layers = 2; cfg.n_heads = 4; cfg.n_kv_heads = 2;
    cfg.wdtype = GTEN_Q4; cfg.adtype = GTEN_Q8;
    const int total = cfg.max_ctx;
    const int lengths[] = {5, 40, 1, 17, 60, 9, 33, 2, 64, 90, 95, 96, 12, 7, 21, 16, 31, 48, 3, 70};
    const int NP = (int)(sizeof lengths / sizeof lengths[0]);
    std::vector<std::vector<int32_t>> prompts((size_t)NP);
    for (int j = 0; j < NP; j++) {
        prompts[(size_t)j].resize((size_t)lengths[j]);
        gten_host_synthetic_tokens(prompts[(size_t)j].data(), lengths[j], 1000u + 7u * (unsigned)j, cfg.n_vocab);
    }

    // ---- one sequence: the reference's loop (logits per token + host argmax) and the device sampler
    gten_host_model* m = gten_host_model_create(&cfg);
    CHECK(m);
    CHECK(gten_host_model_load_synthetic(m, 1234) == 0);
    std::vector<std::vector<int32_t>> alone((size_t)NP), alone_eos((size_t)NP);
    for (int j = 0; j < NP; j++) {
        std::vector<int32_t> a((size_t)total, 0), b((size_t)total, 0);
        std::memcpy(a.data(), prompts[(size_t)j].data(), (size_t)lengths[j] * 4);
        std::memcpy(b.data(), prompts[(size_t)j].data(), (size_t)lengths[j] * 4);
        const int na = gten_host_model_greedy(m, a.data(), lengths[j], total, -1);
        const int nb = gten_host_model_generate(m, b.data(), lengths[j], total, -1);
        CHECK(na == total && nb == total);
        CHECK(std::memcmp(a.data(), b.data(), (size_t)total * 4) == 0);
        alone[(size_t)j] = a;
    }
    const int eos = alone[0][30];                                  // an id that comes up: several sequences stop early at it
    int stopped_early = 0;
    for (int j = 0; j < NP; j++) {
        std::vector<int32_t> a((size_t)total, 0);
        std::memcpy(a.data(), prompts[(size_t)j].data(), (size_t)lengths[j] * 4);
        const int na = gten_host_model_generate(m, a.data(), lengths[j], total, eos);
        CHECK(na >= lengths[j] && na <= total);
        a.resize((size_t)na);
        stopped_early += na < total;
        alone_eos[(size_t)j] = a;
    }
    CHECK(stopped_early > 0);
    // single rows through logits(): the recorded chain, then a multi-row call in between (settles what is pending)
    {
        std::vector<float> lg((size_t)cfg.n_vocab);
        std::vector<int32_t> t = prompts[1];
        CHECK(gten_host_model_logits(m, t.data(), (int)t.size(), 0, lg.data()) == 0);
        for (int i = 0; i < 5; i++) {
            int best = 0;
            for (int k = 1; k < cfg.n_vocab; k++)
                if (lg[(size_t)k] > lg[(size_t)best]) best = k;
            CHECK(best == alone[1][t.size()]);
            t.push_back(best);
            CHECK(gten_host_model_logits(m, t.data(), (int)t.size(), (int)t.size() - 1, i == 2 ? nullptr : lg.data()) == 0);
            if (i == 2) CHECK(gten_host_model_logits(m, t.data(), (int)t.size(), 0, lg.data()) == 0);    // the whole context again
        }
    }
    gten_host_model_free(m);

    // ---- the fixed batch and the queue through the slots
    CHECK(gten_host_batch_create(&cfg, 3) == nullptr);
    // (16 slots: prompts of >= 16 ids are processed as segments of ONE row matrix -- TinyLlamaBatch::prefill_many, the
    //  segmented gten_hip_block_rows and the K / V range copies -- mixed with the one-by-one path of the short ones)
    for (int n_seq : {2, 8, 16}) {
        gten_host_batch* b = gten_host_batch_create(&cfg, n_seq);
        CHECK(b);
        CHECK(gten_host_batch_load_synthetic(b, 1234) == 0);
        if (n_seq <= 8) {       // (the fixed batch wants room behind every prompt: the first 8 lengths have it, the 12th fills the context)
            std::vector<int32_t> pr((size_t)n_seq * total, 0), npr((size_t)n_seq), out((size_t)n_seq * total, 0), tot((size_t)n_seq, 0);
            for (int q = 0; q < n_seq; q++) {
                npr[(size_t)q] = lengths[q];
                std::memcpy(pr.data() + (size_t)q * total, prompts[(size_t)q].data(), (size_t)lengths[q] * 4);
            }
            CHECK(gten_host_batch_generate(b, pr.data(), npr.data(), total, total, eos, out.data(), tot.data()) == 0);
            for (int q = 0; q < n_seq; q++) {
                CHECK(tot[(size_t)q] == (int)alone_eos[(size_t)q].size());
                CHECK(std::memcmp(out.data() + (size_t)q * total, alone_eos[(size_t)q].data(), (size_t)tot[(size_t)q] * 4) == 0);
            }
        }
        std::vector<int32_t> pr((size_t)NP * total, 0), npr((size_t)NP), out((size_t)NP * total, 0), tot((size_t)NP, 0);
        for (int j = 0; j < NP; j++) {
            npr[(size_t)j] = lengths[j];
            std::memcpy(pr.data() + (size_t)j * total, prompts[(size_t)j].data(), (size_t)lengths[j] * 4);
        }
        for (int k : {0, 1, 2, 8})
            for (int slice : {16, 5, 3}) {
                CHECK(gten_host_batch_set_serve_schedule(b, k) == 0);
                double stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                CHECK(gten_host_batch_serve2(b, pr.data(), npr.data(), NP, total, total, eos, slice, 0, nullptr, out.data(), tot.data(), stats, 8) == 0);
                long new_ids = 0;
                for (int j = 0; j < NP; j++) {
                    CHECK(tot[(size_t)j] == (int)alone_eos[(size_t)j].size());
                    CHECK(std::memcmp(out.data() + (size_t)j * total, alone_eos[(size_t)j].data(), (size_t)tot[(size_t)j] * 4) == 0);
                    new_ids += tot[(size_t)j] - lengths[j];
                }
                CHECK((long)stats[1] == new_ids);
            }
        // per-prompt budgets
        {
            std::vector<int32_t> each((size_t)NP);
            for (int j = 0; j < NP; j++) each[(size_t)j] = 1 + (7 * j) % 20;
            CHECK(gten_host_batch_set_serve_schedule(b, 0) == 0);
            double six[6];                                  // the first published entry point: exactly six doubles (ASan watches the array's end)
            CHECK(gten_host_batch_serve(b, pr.data(), npr.data(), NP, total, total, eos, 4, 0, each.data(), out.data(), tot.data(), six) == 0);
            for (int j = 0; j < NP; j++) {
                const int want = lengths[j] >= total ? lengths[j] : std::min((int)alone_eos[(size_t)j].size(), lengths[j] + each[(size_t)j]);
                CHECK(tot[(size_t)j] == want);
                CHECK(std::memcmp(out.data() + (size_t)j * total, alone_eos[(size_t)j].data(), (size_t)tot[(size_t)j] * 4) == 0);
            }
        }
        gten_host_batch_free(b);
    }
    std::printf("host_sanitize_serve ok\n");
    return 0;
}
