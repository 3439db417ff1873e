// tokenizer.h -- host-side counterpart of the reference's tokenizer.h: the same public
// interface (`Tokenizer{path, vocab_size}`, `encode(std::string&)`, `decode(prev, token)`, `eos`), the same
// ids and pieces, written against the vocabulary file format only (tokenizer.h:49-86: int32 max token length,
// then per entry f32 score, int32 byte length, bytes).
//
// What the reference does, restated:
//   * encode (tokenizer.h:135-170): chat template  [1, 32001] + bpe("user\n" + prompt) + [32002, 29871, 13,
//     32001, 20255, 13]; the prompt string is modified in place ("user\n" is inserted), as there.
//   * bpe (tokenizer.h:172-283, after llama2.c): a leading " " token unless the text is empty; one token per UTF-8
//     code point (lead byte + up to three continuation bytes), bytes of an unknown code point as ids byte + 3;
//     then repeatedly merge the adjacent pair whose concatenation is a vocabulary entry of the HIGHEST score,
//     the LEFTMOST such pair on equal scores, until none is left.
//   * decode (tokenizer.h:94-110): "" past the vocabulary; a leading space is dropped after BOS (id 1); a piece
//     "<0xHH>" is the single byte HH.
// How it is done here: a hash map instead of a sorted array + bsearch, and the merge loop as a doubly linked list
// with a priority queue of candidate pairs (score descending, position ascending, stale entries skipped) instead
// of rescanning the whole sequence per merge -- O(n log n) instead of O(n^2), the same merges in the same order.
#pragma once

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

class Tokenizer {
public:
    const int eos = 32002;

    Tokenizer(const char* path, int vocab_size)
    {
        FILE* f = std::fopen(path, "rb");
        if (!f) { std::fprintf(stderr, "couldn't load %s\n", path); std::exit(EXIT_FAILURE); }      // tokenizer.h:63
        int32_t max_len = 0;
        if (std::fread(&max_len, sizeof(int32_t), 1, f) != 1) fail_read(f);
        vocab_.resize((size_t)vocab_size);
        scores_.resize((size_t)vocab_size);
        for (int i = 0; i < vocab_size; i++) {
            int32_t len = 0;
            if (std::fread(&scores_[(size_t)i], sizeof(float), 1, f) != 1) fail_read(f);
            if (std::fread(&len, sizeof(int32_t), 1, f) != 1 || len < 0) fail_read(f);
            std::string& s = vocab_[(size_t)i];
            s.resize((size_t)len);
            if (len > 0 && std::fread(&s[0], (size_t)len, 1, f) != 1) fail_read(f);
            // the reference compares C strings: a piece ends at its first NUL
            const size_t nul = s.find('\0');
            if (nul != std::string::npos) s.resize(nul);
            index_.emplace(s, i);                       // first entry wins for a repeated piece
        }
        std::fclose(f);
        for (int i = 0; i < 256; i++) { byte_pieces_[i][0] = (char)i; byte_pieces_[i][1] = '\0'; }
    }

    std::vector<int> encode(std::string& prompt)
    {
        prompt.insert(0, "user\n");
        std::vector<int> out = {1, 32001};
        bpe(prompt, out);
        static const int post[] = {32002, 29871, 13, 32001, 20255, 13};
        out.insert(out.end(), post, post + 6);
        return out;
    }

    // ids of a plain text (no template): the reference's encode_internal
    std::vector<int> encode_plain(const std::string& text) const
    {
        std::vector<int> out;
        bpe(text, out);
        return out;
    }

    const char* decode(int prev_token, int token) const
    {
        if (token < 0 || token >= (int)vocab_.size()) return "";
        const char* piece = vocab_[(size_t)token].c_str();
        if (prev_token == 1 && piece[0] == ' ') piece++;
        // "<0xHH>": the raw byte (sscanf("<0x%02hhX>") semantics: one or two hex digits after "<0x")
        if (piece[0] == '<' && piece[1] == '0' && piece[2] == 'x' && std::isxdigit((unsigned char)piece[3])) {
            unsigned v = hexval(piece[3]);
            if (std::isxdigit((unsigned char)piece[4])) v = v * 16 + hexval(piece[4]);
            return byte_pieces_[v & 0xff];
        }
        return piece;
    }

    int vocab_size() const { return (int)vocab_.size(); }

private:
    struct Cand {
        float score;
        int left;           // node index of the left token (nodes keep their order: index order == sequence order)
        int right;
        int id;             // the merged token
        int lver, rver;     // versions of both nodes when the candidate was formed
    };
    struct Worse {
        bool operator()(const Cand& a, const Cand& b) const
        {
            if (a.score != b.score) return a.score < b.score;      // higher score first
            return a.left > b.left;                                // then the leftmost pair
        }
    };

    static unsigned hexval(char c)
    {
        if (c >= '0' && c <= '9') return (unsigned)(c - '0');
        return (unsigned)(std::tolower((unsigned char)c) - 'a' + 10);
    }
    [[noreturn]] static void fail_read(FILE* f)
    {
        std::fprintf(stderr, "failed read\n");
        std::fclose(f);
        std::exit(EXIT_FAILURE);
    }
    int lookup(const std::string& s) const
    {
        const auto it = index_.find(s);
        return it == index_.end() ? -1 : it->second;
    }

    void bpe(const std::string& text, std::vector<int>& out) const
    {
        std::vector<int> tok;
        if (!text.empty() && text[0] != '\0') tok.push_back(lookup(" "));          // add_dummy_prefix
        // one token per UTF-8 code point; unknown code points byte by byte (+3: <unk>, <s>, </s> come first)
        const size_t n = std::strlen(text.c_str());                                  // the reference walks a C string
        std::string cp;
        for (size_t i = 0; i < n; i++) {
            const unsigned char c = (unsigned char)text[i];
            if ((c & 0xC0) != 0x80) cp.clear();                                      // not a continuation byte: a new code point
            cp.push_back((char)c);
            const unsigned char nx = (i + 1 < n) ? (unsigned char)text[i + 1] : 0;
            if ((nx & 0xC0) == 0x80 && cp.size() < 4) continue;
            const int id = lookup(cp);
            if (id != -1) tok.push_back(id);
            else for (unsigned char b : cp) tok.push_back((int)b + 3);
            cp.clear();
        }
        // merges
        const int m = (int)tok.size();
        std::vector<int> prev((size_t)m), next((size_t)m), ver((size_t)m, 0);
        std::vector<char> alive((size_t)m, 1);
        for (int i = 0; i < m; i++) { prev[(size_t)i] = i - 1; next[(size_t)i] = (i + 1 < m) ? i + 1 : -1; }
        std::priority_queue<Cand, std::vector<Cand>, Worse> pq;
        auto consider = [&](int l) {
            if (l < 0) return;
            const int r = next[(size_t)l];
            if (r < 0) return;
            if (tok[(size_t)l] < 0 || tok[(size_t)r] < 0) return;
            const int id = lookup(vocab_[(size_t)tok[(size_t)l]] + vocab_[(size_t)tok[(size_t)r]]);
            if (id != -1) pq.push(Cand{scores_[(size_t)id], l, r, id, ver[(size_t)l], ver[(size_t)r]});
        };
        for (int i = 0; i + 1 < m; i++) consider(i);
        while (!pq.empty()) {
            const Cand c = pq.top();
            pq.pop();
            if (!alive[(size_t)c.left] || !alive[(size_t)c.right] || next[(size_t)c.left] != c.right ||
                ver[(size_t)c.left] != c.lver || ver[(size_t)c.right] != c.rver)
                continue;                                                            // stale
            if (c.score <= -1e10f) continue;                                         // tokenizer.h:251: best_score starts at -1e10
            tok[(size_t)c.left] = c.id;
            ver[(size_t)c.left]++;
            alive[(size_t)c.right] = 0;
            const int rn = next[(size_t)c.right];
            next[(size_t)c.left] = rn;
            if (rn >= 0) prev[(size_t)rn] = c.left;
            consider(prev[(size_t)c.left]);
            consider(c.left);
        }
        for (int i = 0; i < m; i++)
            if (alive[(size_t)i]) out.push_back(tok[(size_t)i]);
    }

    std::vector<std::string> vocab_;
    std::vector<float> scores_;
    std::unordered_map<std::string, int> index_;
    char byte_pieces_[256][2];
};
