// tinyllama_cli.cpp -- command line of the reference (tinyllama.cpp:110-298: options, single prompt or chat loop,
// greedy or top-k sampling) on the HBM-backed gten API of this repository.  SURVEY 8(f) rank 4; host code only -- the
// forward path it drives is libgten_hip.so.  Differences from the reference's main(): no model download step (there is
// no network here: --model PATH, default models/tinyllama.<fp16|q8|q4>.gten as there), --tokenizer PATH (default
// tokenizer.bin), --seed for the top-k sampler, --ids to print token ids instead of text (tests); greedy sampling runs
// with the sampler on the device (gten::greedy_generate).
#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <string_view>
#include <vector>

#include "tinyllama_model.h"
#include "tokenizer.h"

using namespace gten;

static const char* usage_message = R"(
USAGE:
./tinyllama_cli [options] -p PROMPT  for a single prompt or
./tinyllama_cli [options] for a chat interface.

Optional args.
-f16 :     Use float-16 model and inference (2.2GB). [default]
-q8  :     Use 8-bit quantized model (1.1GB).
-q4  :     Use 4-bit quantized model (0.62GB).
-greedy :  Greedy sampling (argmax on the device) instead of top-k sampling.
--temp T : Temperature to use during sampling. It must be greater than 0. [default=0.9].
--npred  N : Number of tokens to generate. Minimum is 1 and max is 2048. [default=768].
--topk K : Top tokens to randomly select from during prediction. [default=50].
--model PATH :     .gten checkpoint [default=models/tinyllama.<fp16|q8|q4>.gten].
--tokenizer PATH : vocabulary file [default=tokenizer.bin].
--seed S : seed of the top-k sampler [default: random].
--ids :    print token ids instead of text.
)";

struct Options {
    Dtype model_dtype = kFloat16;
    std::string model_path, tokenizer_path = "tokenizer.bin", prompt;
    int n_predict = 768, topk = 50;
    float temp = 0.9f;
    bool greedy = false, ids = false, seeded = false;
    uint64_t seed = 0;
};

static void emit(const Options& o, Tokenizer& tok, int prev, int id)
{
    if (o.ids) std::cout << id << ' ';
    else std::cerr << tok.decode(prev, id);
}

// greedy: the prompt as in the reference, every later id from the device-side sampler
static void run_greedy(const Options& o, std::string prompt, TinyLlama& model, Tokenizer& tok)
{
    std::vector<int> enc = tok.encode(prompt);
    std::vector<int32_t> tokens(enc.begin(), enc.end());
    const size_t n_prompt = tokens.size();
    greedy_generate(model, tokens, o.n_predict, tok.eos);
    for (size_t i = n_prompt; i < tokens.size(); i++) emit(o, tok, i == n_prompt ? 1 : tokens[i - 1], tokens[i]);
    (o.ids ? std::cout : std::cerr) << '\n';
}

// top-k sampling (tinyllama.cpp:442-507): logits / temp, the k largest, softmax over them, one draw
static void run_topk(const Options& o, std::string prompt, TinyLlama& model, Tokenizer& tok)
{
    std::mt19937 gen(o.seeded ? (uint32_t)o.seed : std::random_device{}());
    std::vector<int> enc = tok.encode(prompt);
    std::vector<int32_t> tokens(enc.begin(), enc.end());
    tokens.reserve((size_t)o.n_predict);
    const int n_vocab = model.params.n_vocab, k = std::min(o.topk, n_vocab);
    std::vector<std::pair<double, int>> cand;
    const int n_new = o.n_predict - (int)tokens.size();
    for (int i = 0; i < n_new; i++) {
        Tensor input{tokens.data(), {(int)tokens.size()}, kInt32};
        Tensor logits = model.logits(input, i == 0 ? 0 : input.numel() - 1);
        const float* lg = const_cast<const Tensor&>(logits).data_ptr<float>();
        cand.clear();
        for (int j = 0; j < n_vocab; j++) cand.emplace_back((double)lg[j] / o.temp, j);
        std::partial_sort(cand.begin(), cand.begin() + k, cand.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        std::vector<double> w((size_t)k);
        for (int j = 0; j < k; j++) w[(size_t)j] = std::exp(cand[(size_t)j].first - cand[0].first);
        std::discrete_distribution<int> dist(w.begin(), w.end());
        const int id = cand[(size_t)dist(gen)].second;
        if (id == tok.eos) break;
        emit(o, tok, i == 0 ? 1 : tokens.back(), id);
        tokens.push_back(id);
    }
    (o.ids ? std::cout : std::cerr) << '\n';
}

int main(int argc, char const* argv[])
{
    Options o;
    std::string model_id = "fp16";
    for (int i = 1; i < argc; i++) {
        const std::string_view arg{argv[i]};
        auto value = [&](const char* what) -> const char* {
            if (i + 1 >= argc) { std::cerr << what << " value is missing.\n"; std::exit(EXIT_FAILURE); }
            return argv[++i];
        };
        if (arg == "--help" || arg == "-h") { std::cout << usage_message << "\n"; return 0; }
        else if (arg == "-f16") { o.model_dtype = kFloat16; model_id = "fp16"; }
        else if (arg == "-q8") { o.model_dtype = kQint8; model_id = "q8"; }
        else if (arg == "-q4") { o.model_dtype = kQint4; model_id = "q4"; }
        else if (arg == "-greedy") o.greedy = true;
        else if (arg == "--ids") o.ids = true;
        else if (arg == "-p") o.prompt = value("prompt");
        else if (arg == "--model") o.model_path = value("model");
        else if (arg == "--tokenizer") o.tokenizer_path = value("tokenizer");
        else if (arg == "--seed") { o.seed = std::strtoull(value("seed"), nullptr, 10); o.seeded = true; }
        else if (arg == "--npred") {
            int v = 0;
            try { v = std::stoi(value("npred")); } catch (...) { std::cerr << "Invalid npred value.\n"; return -1; }
            if (v < 1 || v > 2048) { std::cerr << "npred must be greater than 1 and less than 2048.\n"; return -1; }
            o.n_predict = v;
        } else if (arg == "--temp") {
            float v = 0.f;
            try { v = std::stof(value("temp")); } catch (...) { std::cerr << "Invalid temp value \n"; return -1; }
            if (v <= 0.0f) { std::cerr << "temp value must be greater than zero.\n"; return -1; }
            o.temp = v;
        } else if (arg == "--topk") {
            int v = 0;
            try { v = std::stoi(value("topk")); } catch (...) { std::cerr << "Invalid topk value.\n"; return -1; }
            if (v < 1 || v > 32003) { std::cerr << "topk must be gte 1 and lte " << 32003 << ".\n"; return -1; }
            o.topk = v;
        } else {
            std::cerr << "error: Unknown argument: " << arg << "\n" << usage_message;
            return EXIT_FAILURE;
        }
    }
    if (o.model_path.empty()) o.model_path = "models/tinyllama." + model_id + ".gten";

    std::ifstream checkpoint{o.model_path, std::ios::binary};
    if (!checkpoint.is_open()) {
        std::cerr << "error: cannot open the checkpoint " << o.model_path << " (convert one with `python -m tinyllama.cpp_amd.convert`).\n";
        return EXIT_FAILURE;
    }
    {
        std::ifstream vf{o.tokenizer_path, std::ios::binary};
        if (!vf.is_open()) { std::cerr << "error: cannot open the vocabulary file " << o.tokenizer_path << ".\n"; return EXIT_FAILURE; }
    }
    ModuleDtype dtype;
    dtype.wdtype = o.model_dtype;
    dtype.adtype = (o.model_dtype == kFloat16) ? kFloat16 : kQint8;          // tinyllama.cpp:258-265

    TinyLlama model{o.n_predict, dtype};
    model.load_from_ckpt(checkpoint);
    Tokenizer tokenizer{o.tokenizer_path.c_str(), 32000};

    auto answer = [&](const std::string& prompt) {
        if (o.greedy) run_greedy(o, prompt, model, tokenizer);
        else run_topk(o, prompt, model, tokenizer);
    };
    if (o.prompt.empty()) {
        std::cout << "Chat interface. Write your prompt and press enter to submit. Enter q or press ctrl+c to quit.\n";
        std::string prompt;
        while (true) {
            std::cerr << "\n\n[You]: ";
            if (!std::getline(std::cin, prompt) || prompt == "q") break;
            std::cerr << "\n[Tinyllama-Chat]: \n\n";
            answer(prompt);
        }
    } else {
        answer(o.prompt);
    }
    return 0;
}
