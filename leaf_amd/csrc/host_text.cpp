// Native host side of the LEAF search: single-edit string mutation + CLIP BPE tokenisation of the
// B*rho candidates of one search stage, multithreaded, straight into the engine's int32 [N, ctx] wire format.
//
// Semantics restated from the reference (not its code):
//   * mutation      utils_attacks.py:169-213 with alternative = -1 (the only mode attack_text_leaf uses, :318,357):
//                   sentence S viewed as slots/characters "_s0_s1..._"; position z even = slot before character z/2,
//                   odd = character (z-1)/2.  Code point c: c == -1 or c == current cell -> remove the cell (a slot
//                   removal is a no-op), otherwise the cell becomes c (insert on a slot, replace on a character).
//   * tokenisation  src/open_clip/tokenizer.py:66-85,133-265: double html.unescape, strip, whitespace collapse, lower,
//                   regex split, byte->unicode, greedy lowest-rank BPE merges, [SOT] ids [EOT], pad 0 / truncate to ctx.
// Fast path = pure 7-bit ASCII text without '&' (no entities) and without the literal special-token strings; anything
// else is flagged and the Python tokenizer handles that sentence (same results, slower).
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/leaf_hip.h"

namespace {

struct PairHash {
    size_t operator()(const std::pair<std::string, std::string>& p) const {
        return std::hash<std::string>()(p.first) * 1000003u ^ std::hash<std::string>()(p.second);
    }
};

struct Tok {
    std::unordered_map<std::pair<std::string, std::string>, int, PairHash> rank;
    std::unordered_map<std::string, int> encoder;
    int sot = 0, eot = 0;
    std::vector<std::unordered_map<std::string, std::vector<int32_t>>> caches;  // one per worker thread
};

// the GPT-2 byte -> printable-unicode table, as UTF-8 strings
std::vector<std::string> byte_alphabet() {
    std::vector<int> cp(256, -1);
    int extra = 0;
    auto printable = [](int b) { return (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255); };
    for (int b = 0; b < 256; ++b)
        if (printable(b)) cp[b] = b;
    for (int b = 0; b < 256; ++b)
        if (cp[b] < 0) cp[b] = 256 + extra++;
    std::vector<std::string> out(256);
    for (int b = 0; b < 256; ++b) {
        int c = cp[b];
        if (c < 0x80) out[b] = std::string(1, (char)c);
        else { out[b] = std::string(1, (char)(0xC0 | (c >> 6))); out[b] += (char)(0x80 | (c & 0x3F)); }
    }
    return out;
}

std::vector<int> vocab_order() {  // printable bytes first, then the remapped ones (tokenizer.py:31-51)
    std::vector<int> order;
    auto printable = [](int b) { return (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255); };
    for (int b = 0; b < 256; ++b) if (printable(b)) order.push_back(b);
    for (int b = 0; b < 256; ++b) if (!printable(b)) order.push_back(b);
    return order;
}

const std::string END = "</w>";

const std::vector<int32_t>& bpe_word(Tok& tk, std::unordered_map<std::string, std::vector<int32_t>>& cache,
                                     const std::string& word) {
    // The search feeds millions of one-off mutated words through here (12,800 candidates per step); the reference's
    // Python cache grows without bound, this one is dropped when it gets large (callers hold no reference across calls).
    if (cache.size() > (size_t)1 << 19) cache.clear();
    auto it = cache.find(word);
    if (it != cache.end()) return it->second;
    std::vector<std::string> parts;
    parts.reserve(word.size());
    for (size_t i = 0; i < word.size(); ++i) parts.emplace_back(1, word[i]);   // ASCII: one byte = one symbol
    parts.back() += END;
    while (parts.size() > 1) {
        int best = -1, best_rank = 0;
        for (size_t i = 0; i + 1 < parts.size(); ++i) {
            auto r = tk.rank.find({parts[i], parts[i + 1]});
            if (r != tk.rank.end() && (best < 0 || r->second < best_rank)) { best = (int)i; best_rank = r->second; }
        }
        if (best < 0) break;
        const std::string a = parts[best], b = parts[best + 1];
        std::vector<std::string> merged;
        merged.reserve(parts.size());
        for (size_t i = 0; i < parts.size();) {
            if (i + 1 < parts.size() && parts[i] == a && parts[i + 1] == b) { merged.push_back(a + b); i += 2; }
            else { merged.push_back(parts[i]); i += 1; }
        }
        parts.swap(merged);
    }
    std::vector<int32_t> ids;
    ids.reserve(parts.size());
    for (auto& p : parts) {
        auto e = tk.encoder.find(p);
        ids.push_back(e == tk.encoder.end() ? 0 : e->second);
    }
    return cache.emplace(word, std::move(ids)).first->second;
}

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }  // str.split()
inline bool is_letter(unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
inline bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }

bool ascii_fast_path_ok(const char* s, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        unsigned char c = (unsigned char)s[i];
        if (c == '&' || !((c >= 32 && c <= 126) || is_space(c))) return false;   // entities / remapped bytes: Python
    }
    // literal special tokens inside the text take the regex's first alternatives: leave those to Python
    std::string low(s, n);
    for (auto& ch : low) if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
    return low.find("<start_of_text>") == std::string::npos && low.find("<end_of_text>") == std::string::npos;
}

// clean + split + BPE of one ASCII text into a row of the token matrix; returns kept length (eot position + 1)
int encode_row(Tok& tk, std::unordered_map<std::string, std::vector<int32_t>>& cache, const std::string& text, int ctx,
               int32_t* row) {
    // strip + collapse whitespace + lower
    std::string t;
    t.reserve(text.size());
    bool pending_space = false;
    for (unsigned char c : text) {
        if (is_space(c)) { pending_space = !t.empty(); continue; }
        if (pending_space) { t += ' '; pending_space = false; }
        t += (char)((c >= 'A' && c <= 'Z') ? c + 32 : c);
    }
    int n = 0;
    row[n++] = tk.sot;
    auto emit = [&](const std::string& w) {
        const std::vector<int32_t>& ids = bpe_word(tk, cache, w);
        for (int32_t id : ids) { if (n < ctx) row[n] = id; ++n; }
    };
    size_t i = 0;
    const size_t L = t.size();
    while (i < L) {
        unsigned char c = (unsigned char)t[i];
        if (c == ' ') { ++i; continue; }
        if (c == '\'' && i + 1 < L) {   // 's|'t|'re|'ve|'m|'ll|'d  (first alternative that matches)
            size_t len = 0;
            char d = t[i + 1];
            if (d == 's' || d == 't') len = 2;
            else if (d == 'r' && i + 2 < L && t[i + 2] == 'e') len = 3;
            else if (d == 'v' && i + 2 < L && t[i + 2] == 'e') len = 3;
            else if (d == 'm') len = 2;
            else if (d == 'l' && i + 2 < L && t[i + 2] == 'l') len = 3;
            else if (d == 'd') len = 2;
            if (len) { emit(t.substr(i, len)); i += len; continue; }
        }
        size_t j = i;
        if (is_letter(c)) { while (j < L && is_letter((unsigned char)t[j])) ++j; }
        else if (is_digit(c)) { j = i + 1; }
        else { while (j < L && t[j] != ' ' && !is_letter((unsigned char)t[j]) && !is_digit((unsigned char)t[j])) ++j; }
        emit(t.substr(i, j - i));
        i = j;
    }
    if (n < ctx) row[n] = tk.eot;
    ++n;
    if (n > ctx) { n = ctx; row[ctx - 1] = tk.eot; }     // truncate, last id forced to EOT (tokenizer.py:260-262)
    for (int k = n; k < ctx; ++k) row[k] = 0;
    return n;
}

// single edit with alternative = -1 on an ASCII sentence
std::string mutate(const char* s, size_t n, int z, int c) {
    std::string out;
    out.reserve(n + 1);
    if (z & 1) {
        const size_t idx = (size_t)(z - 1) / 2;
        out.append(s, idx);
        if (!(c == -1 || (unsigned char)s[idx] == (unsigned)c)) out += (char)c;
        out.append(s + idx + 1, n - idx - 1);
    } else {
        const size_t idx = (size_t)z / 2;
        out.append(s, idx);
        if (!(c == -1 || c == '_')) out += (char)c;     // a slot holds '_' in the reference's expanded view
        out.append(s + idx, n - idx);
    }
    return out;
}

template <class F>
void parallel_for(int n, int n_threads, F f) {
    if (n_threads <= 1 || n < 2) { for (int i = 0; i < n; ++i) f(i, 0); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t]() {
            for (;;) {
                int i = next.fetch_add(16);
                if (i >= n) break;
                for (int k = i; k < i + 16 && k < n; ++k) f(k, t);
            }
        });
    for (auto& x : th) x.join();
}

}  // namespace

struct leaf_tok : Tok {};

extern "C" int leaf_tok_create(const char* merges, size_t len, leaf_tok_t* out) {
    if (!merges || !out) return 1;
    leaf_tok* tk = new leaf_tok();
    std::vector<std::string> alpha = byte_alphabet();
    std::vector<int> order = vocab_order();
    int id = 0;
    for (int b : order) tk->encoder[alpha[b]] = id++;
    for (int b : order) tk->encoder[alpha[b] + END] = id++;
    // lines 1 .. 49152-256-2 of the merge file: "a b"
    const int n_merges = 49152 - 256 - 2;
    size_t pos = 0;
    int line = 0, r = 0;
    while (pos < len && r < n_merges) {
        size_t e = pos;
        while (e < len && merges[e] != '\n') ++e;
        if (line >= 1) {
            std::string l(merges + pos, e - pos);
            size_t sp = l.find(' ');
            if (sp != std::string::npos) {
                std::string a = l.substr(0, sp), b = l.substr(sp + 1);
                while (!b.empty() && (b.back() == '\r' || b.back() == ' ')) b.pop_back();
                tk->rank[{a, b}] = r++;
                tk->encoder[a + b] = id++;
            }
        }
        pos = e + 1;
        ++line;
    }
    tk->sot = id++;
    tk->eot = id++;
    if (r != n_merges || tk->eot != 49407) { delete tk; return 2; }
    *out = tk;
    return 0;
}

extern "C" void leaf_tok_destroy(leaf_tok_t tk) { delete tk; }

static void ensure_caches(leaf_tok* tk, int n_threads) {
    if ((int)tk->caches.size() < n_threads) tk->caches.resize(n_threads);
}

extern "C" int leaf_tok_encode_batch(leaf_tok_t tk, const char* const* texts, const int32_t* text_len, int n, int ctx,
                                     int32_t* tokens, int32_t* lens, uint8_t* fallback, int n_threads) {
    if (!tk || !texts || !tokens || !lens || !fallback || ctx < 2) return 1;
    if (n_threads < 1) n_threads = 1;
    ensure_caches(tk, n_threads);
    parallel_for(n, n_threads, [&](int i, int t) {
        if (!ascii_fast_path_ok(texts[i], text_len[i])) { fallback[i] = 1; lens[i] = 0; return; }
        fallback[i] = 0;
        lens[i] = encode_row(*tk, tk->caches[t], std::string(texts[i], text_len[i]), ctx, tokens + (size_t)i * ctx);
    });
    return 0;
}

extern "C" int leaf_tok_mutate_encode(leaf_tok_t tk, const char* const* sentences, const int32_t* sent_len, int B,
                                      const int32_t* z, const int32_t* c, int rho, int ctx, int32_t* tokens,
                                      int32_t* lens, uint8_t* fallback, int n_threads) {
    if (!tk || !sentences || !z || !c || !tokens || !lens || !fallback || ctx < 2 || rho < 1) return 1;
    if (n_threads < 1) n_threads = 1;
    ensure_caches(tk, n_threads);
    std::vector<uint8_t> sent_ok(B);
    for (int b = 0; b < B; ++b) sent_ok[b] = ascii_fast_path_ok(sentences[b], sent_len[b]) ? 1 : 0;
    std::atomic<int> bad(0);
    parallel_for(B * rho, n_threads, [&](int i, int t) {   // fallback is per CANDIDATE here
        const int b = i / rho;
        const int zz = z[i], cc = c[i];
        lens[i] = 0;
        if (zz < 0 || zz > 2 * sent_len[b] || cc < -1) { bad = 1; fallback[i] = 1; return; }
        if (!sent_ok[b] || cc > 126 || (cc >= 0 && cc < 32) || cc == '&') { fallback[i] = 1; return; }
        const std::string m = mutate(sentences[b], sent_len[b], zz, cc);
        if (!ascii_fast_path_ok(m.data(), m.size())) { fallback[i] = 1; return; }
        fallback[i] = 0;
        lens[i] = encode_row(*tk, tk->caches[t], m, ctx, tokens + (size_t)i * ctx);
    });
    return bad ? 3 : 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// --constrain (utils_attacks.py:110-143, :321-325, :360-364): a candidate is valid iff it holds STRICTLY FEWER distinct
// dictionary words than its sentence: len(W & set(word_tokenize(lower(candidate)))) < len(W & set(word_tokenize(lower(sentence)))).
// The reference rebuilds the 236k-word set and tokenises all 2 B rho strings in Python on every call.  Here the set is built
// once; a sentence is tokenised once into {dictionary word -> multiplicity}; a single-edit candidate differs from it only
// in the whitespace-delimited window around the edit, so only that window is re-tokenised and the distinct count is updated
// from the multiplicities.  No candidate string is materialised.
//
// Two word tokenizers (the caller says which one its Dictionary uses):
//   kind 0  the regex stand-in  [A-Za-z0-9]+ | [^\sA-Za-z0-9]  (leaf_amd/attacks.py, used when nltk is absent): tokens never
//           span whitespace, so the window logic is exact for every ASCII string;
//   kind 1  nltk.word_tokenize: for text made of letters, digits and whitespace only it is a whitespace split, except for the
//           Treebank contraction words (cannot, gimme, gonna, gotta, lemme, wanna); any other sentence / candidate (punctuation,
//           quotes, periods ...) is declined (fallback = 1) and decided by the caller with the real tokenizer.
namespace {

struct Dict {
    std::unordered_set<std::string> words;
};

inline bool is_alnum(unsigned char c) { return is_letter(c) || is_digit(c); }
inline char lower(unsigned char c) { return (char)((c >= 'A' && c <= 'Z') ? c + 32 : c); }

bool treebank_special(const std::string& w) {
    return w == "cannot" || w == "gimme" || w == "gonna" || w == "gotta" || w == "lemme" || w == "wanna";
}

// tokens of a whitespace-free-bounded piece of text, lower-cased; returns false when the text leaves the fast path of `kind`
template <class F>
bool tokenize_piece(const char* s, size_t n, int kind, F emit) {
    size_t i = 0;
    std::string w;
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        if (is_space(c)) { ++i; continue; }
        if (c >= 128) return false;
        if (is_alnum(c)) {
            w.clear();
            while (i < n && is_alnum((unsigned char)s[i])) { w += lower((unsigned char)s[i]); ++i; }
            if (kind == 1 && treebank_special(w)) return false;
            emit(w);
        } else {
            if (kind == 1) return false;          // punctuation: nltk's rules are not local
            w.assign(1, (char)c);
            emit(w);
            ++i;
        }
    }
    return true;
}

}  // namespace

struct leaf_dict : Dict {};

extern "C" int leaf_dict_create(const char* words, size_t len, leaf_dict_t* out) {
    if (!words || !out) return 1;
    leaf_dict* d = new leaf_dict();
    d->words.reserve(len / 8 + 16);
    size_t pos = 0;
    while (pos < len) {
        size_t e = pos;
        while (e < len && words[e] != '\n') ++e;
        size_t a = pos, b = e;
        while (a < b && is_space((unsigned char)words[a])) ++a;
        while (b > a && is_space((unsigned char)words[b - 1])) --b;
        if (b > a) d->words.emplace(words + a, b - a);
        pos = e + 1;
    }
    *out = d;
    return 0;
}

extern "C" void leaf_dict_destroy(leaf_dict_t d) { delete d; }
extern "C" int64_t leaf_dict_size(leaf_dict_t d) { return d ? (int64_t)d->words.size() : -1; }

extern "C" int leaf_tok_constrain(leaf_dict_t d, int kind, const char* const* sentences, const int32_t* sent_len, int B,
                                  const int32_t* z, const int32_t* c, int rho, uint8_t* valid, uint8_t* fallback, int n_threads) {
    if (!d || !sentences || !sent_len || !z || !c || !valid || !fallback || rho < 1 || (kind != 0 && kind != 1)) return 1;
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> bad(0);
    // sentences are independent: one task per sentence (its rho candidates share the multiplicity map)
    parallel_for(B, n_threads, [&](int b, int) {
        const char* s = sentences[b];
        const int n = sent_len[b];
        uint8_t* v = valid + (size_t)b * rho;
        uint8_t* fb = fallback + (size_t)b * rho;
        std::unordered_map<std::string, int> mult;       // dictionary words of the sentence -> occurrences
        const bool ok = tokenize_piece(s, (size_t)n, kind, [&](const std::string& w) { if (d->words.count(w)) ++mult[w]; });
        if (!ok) { for (int r = 0; r < rho; ++r) { fb[r] = 1; v[r] = 0; } return; }
        const int lo = (int)mult.size();
        std::unordered_map<std::string, int> delta;
        std::string win;
        for (int r = 0; r < rho; ++r) {
            const int zz = z[(size_t)b * rho + r], cc = c[(size_t)b * rho + r];
            fb[r] = 0;
            if (zz < 0 || zz > 2 * n || cc < -1) { bad = 1; fb[r] = 1; v[r] = 0; continue; }
            if (cc > 126 || (cc >= 0 && cc < 32)) { fb[r] = 1; v[r] = 0; continue; }
            // the edit in terms of the original string: characters [e0, e1) are replaced by `ins` (0 or 1 characters)
            int e0, e1;
            char ins = 0;
            bool has_ins = false;
            if (zz & 1) {
                e0 = (zz - 1) / 2; e1 = e0 + 1;
                if (!(cc == -1 || (unsigned char)s[e0] == (unsigned)cc)) { ins = (char)cc; has_ins = true; }
            } else {
                e0 = e1 = zz / 2;
                if (!(cc == -1 || cc == '_')) { ins = (char)cc; has_ins = true; }
            }
            if (e0 == e1 && !has_ins) { v[r] = 0; continue; }          // no-op candidate: same count, never strictly fewer
            // window = the edit extended to whitespace on both sides (tokens of both tokenizers never span whitespace)
            int L = e0, R = e1;
            while (L > 0 && !is_space((unsigned char)s[L - 1])) --L;
            while (R < n && !is_space((unsigned char)s[R])) ++R;
            delta.clear();
            bool okw = tokenize_piece(s + L, (size_t)(R - L), kind, [&](const std::string& w) { if (d->words.count(w)) --delta[w]; });
            win.assign(s + L, (size_t)(e0 - L));
            if (has_ins) win += ins;
            win.append(s + e1, (size_t)(R - e1));
            okw = okw && tokenize_piece(win.data(), win.size(), kind, [&](const std::string& w) { if (d->words.count(w)) ++delta[w]; });
            if (!okw) { fb[r] = 1; v[r] = 0; continue; }
            int cnt = lo;
            for (auto& kv : delta) {
                if (kv.second == 0) continue;
                auto it = mult.find(kv.first);
                const int before = it == mult.end() ? 0 : it->second;
                cnt += (before + kv.second > 0 ? 1 : 0) - (before > 0 ? 1 : 0);
            }
            v[r] = cnt < lo ? 1 : 0;
        }
    });
    return bad ? 3 : 0;
}
