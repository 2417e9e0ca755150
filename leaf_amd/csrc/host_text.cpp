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
#include <pthread.h>
#include <condition_variable>
#include <functional>
#include <mutex>
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

// Persistent worker pool: a search makes ~10 native calls of 1-2 ms each; creating and joining 16 std::threads per call cost
// 0.3-0.5 ms of every one of them.  Workers are created once (grown to the largest thread count asked for), sleep on a
// condition variable between jobs and are joined when the library is unloaded.  One job at a time (the callers are the
// single Python thread of a rank); the calling thread works as worker 0.
class WorkerPool {
public:
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_job_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // runs body(t) for t = 0 .. nt-1 concurrently and returns when all are done
    void run(int nt, const std::function<void(int)>& body) {
        if (nt <= 1) { body(0); return; }
        std::lock_guard<std::mutex> one_job(run_m_);       // callers from different host threads take turns
        std::unique_lock<std::mutex> lk(m_);
        while ((int)workers_.size() < nt - 1) {
            const int id = (int)workers_.size() + 1;
            workers_.emplace_back([this, id]() { loop(id); });
        }
        body_ = &body;
        job_threads_ = nt;
        pending_ = nt - 1;
        ++generation_;
        lk.unlock();
        cv_job_.notify_all();
        body(0);
        lk.lock();
        cv_done_.wait(lk, [this]() { return pending_ == 0; });
        body_ = nullptr;
    }

private:
    void loop(int id) {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_job_.wait(lk, [&]() { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            if (id >= job_threads_) continue;          // this job uses fewer workers
            const std::function<void(int)>* b = body_;
            lk.unlock();
            (*b)(id);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
    std::mutex m_, run_m_;
    std::condition_variable cv_job_, cv_done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* body_ = nullptr;
    unsigned long generation_ = 0;
    int job_threads_ = 0, pending_ = 0;
    bool stop_ = false;
};

// One pool per PROCESS, created on first use and never destroyed.  Never destroyed: a joining destructor at exit / dlclose can hang
// when the interpreter tears the process down while a worker is inside a job, and idle workers own nothing that needs cleaning up.
// Per process: after fork() the child inherits the pool OBJECT but none of its threads (and its locks in whatever state another
// thread of the parent left them), so run() would wait for workers that do not exist; the atfork child handler abandons the
// inherited object -- it is not destroyed: std::thread's destructor would terminate on the phantom joinable threads -- and the
// child's first parallel call builds a pool of its own.
std::atomic<WorkerPool*> g_pool{nullptr};
std::atomic_flag g_pool_lock = ATOMIC_FLAG_INIT;
void pool_after_fork_in_child() {
    g_pool.store(nullptr, std::memory_order_relaxed);
    g_pool_lock.clear();
}
WorkerPool& pool() {
    WorkerPool* p = g_pool.load(std::memory_order_acquire);
    if (p) return *p;
    while (g_pool_lock.test_and_set(std::memory_order_acquire)) std::this_thread::yield();
    p = g_pool.load(std::memory_order_relaxed);
    if (!p) {
        static bool registered = false;
        if (!registered) { pthread_atfork(nullptr, nullptr, pool_after_fork_in_child); registered = true; }
        p = new WorkerPool();
        g_pool.store(p, std::memory_order_release);
    }
    g_pool_lock.clear(std::memory_order_release);
    return *p;
}

template <class F>
void parallel_for(int n, int n_threads, F f) {
    if (n_threads <= 1 || n < 2) { for (int i = 0; i < n; ++i) f(i, 0); return; }
    if (n_threads > n) n_threads = n;
    std::atomic<int> next(0);
    const std::function<void(int)> body = [&](int t) {
        for (;;) {
            const int i = next.fetch_add(16);
            if (i >= n) break;
            for (int k = i; k < i + 16 && k < n; ++k) f(k, t);
        }
    };
    pool().run(n_threads, body);
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
//   kind 1  nltk.word_tokenize: the Treebank substitution pipeline restated below (tb_tokenize_piece), exact on any ASCII text
//           whose tokens cannot depend on Punkt's sentence boundaries; a sentence / candidate in which a lone '.' ends a chunk
//           before the end of the text needs sentence spans: from the caller (leaf_tok_constrain_ranges), from the restated
//           Punkt over the model's tables (leaf_tok_constrain_punkt) -- or it is declined (fallback = 1) and decided by the
//           caller with the real tokenizer.
namespace {

struct Dict {
    std::unordered_set<std::string> words;
};

inline bool is_alnum(unsigned char c) { return is_letter(c) || is_digit(c); }
inline char lower(unsigned char c) { return (char)((c >= 'A' && c <= 'Z') ? c + 32 : c); }


// ---- kind 1: nltk.word_tokenize = Punkt sentence splitting + NLTKWordTokenizer (nltk/tokenize/destructive.py, a third-party
// dependency of the reference: requirements.txt:14).  The Treebank step is a fixed pipeline of regular-expression
// substitutions; it is restated here rule for rule (same order, same non-overlapping left-to-right matching) on a PIECE of the
// lower-cased text that is bounded by whitespace or the text's ends.  Every rule looks at most one character past a space, so a
// piece tokenises as it would inside the whole text once it is told whether it starts at the text's first character
// (`at_start`: the ^" rule), ends at its last (`at_end`: the ([:,])$ rule) and whether only whitespace follows it (`ws_after`:
// the two final-period rules, which end in \s*$).  Checked against nltk 3.6.5's own output (tests/golden/treebank_kat.json)
// and against leaf_amd/treebank.py (tests/test_constrain_native.py).  Texts in which a lone '.' ends a chunk before the text's end
// (tb_punkt_free false) tokenise differently depending on where Punkt ends sentences: they need sentence spans -- the caller's
// (leaf_tok_constrain_ranges), the native Punkt restatement's further down (leaf_tok_constrain_punkt) -- or are declined.
inline bool tb_word(unsigned char c) { return is_alnum(c) || c == '_'; }
inline bool tb_closer(unsigned char c) { return c == ']' || c == ')' || c == '}' || c == '>' || c == '"' || c == '\''; }

// what may directly follow a sentence-final character for Punkt to consider a break there without whitespace
// (nltk/tokenize/punkt.py PunktLanguageVars._re_non_word_chars)
inline bool punkt_nonword(unsigned char c) {
    return c == '?' || c == '!' || c == ')' || c == '"' || c == ';' || c == '}' || c == ']' || c == '*' || c == ':' || c == '@' ||
           c == '\'' || c == '(' || c == '{' || c == '[';
}

// text[0, n): true when sentence boundaries cannot change the tokens (leaf_amd/treebank.py punkt_free): every lone '.' is followed by
// a character that is neither blank nor in Punkt's NONWORD set, or is the text's final period (closers directly behind it, then blanks)
bool tb_punkt_free(const char* t, size_t n) {
    size_t i = 0;
    while (i < n) {
        if (t[i] != '.') { ++i; continue; }
        size_t j = i;
        while (j < n && t[j] == '.') ++j;
        if (j - i == 1 && j < n && (is_space((unsigned char)t[j]) || punkt_nonword((unsigned char)t[j]))) {
            size_t q = j;                                  // the text's final period: closers directly behind it, then only blanks
            while (q < n && tb_closer((unsigned char)t[q])) ++q;
            for (; q < n; ++q)
                if (!is_space((unsigned char)t[q])) return false;
        }
        i = j;
    }
    return true;
}

// final-period rules:  ([^\.])(\.)(CLASS*)\s*$  ->  "\1 \2" + mid + "\3 "   (first form: CLASS includes ' ', mid = " ")
void tb_final_period(std::string& w, bool with_space, bool anchored_end) {
    if (!anchored_end) return;
    size_t q = w.size();
    while (q > 0 && is_space((unsigned char)w[q - 1])) --q;
    size_t r = q;
    while (r > 0 && (tb_closer((unsigned char)w[r - 1]) || (with_space && w[r - 1] == ' '))) --r;
    if (r < 2 || w[r - 1] != '.' || w[r - 2] == '.') return;
    std::string o = w.substr(0, r - 1);
    o += " .";
    if (with_space) o += ' ';
    o.append(w, r, q - r);
    o += ' ';
    w.swap(o);
}

template <class Pred>
void tb_pad_chars(std::string& w, Pred pred) {
    std::string o;
    o.reserve(w.size() + 8);
    for (char c : w) { if (pred((unsigned char)c)) { o += ' '; o += c; o += ' '; } else o += c; }
    w.swap(o);
}

inline bool tb_ieq(const std::string& w, size_t i, const char* lit) {     // case-insensitive literal at w[i..]
    for (size_t k = 0; lit[k]; ++k)
        if (i + k >= w.size() || lower((unsigned char)w[i + k]) != lit[k]) return false;
    return true;
}

// \b(A)(B)\b (case-insensitive) -> " A B "; `need_space`: the (wan)(na)\s form (the blank is consumed)
void tb_contraction(std::string& w, const char* a, const char* b, bool need_space) {
    const size_t la = strlen(a), lb = strlen(b);
    std::string o;
    size_t i = 0;
    bool any = false;
    while (i < w.size()) {
        const bool wb0 = (i == 0 || !tb_word((unsigned char)w[i - 1])) && tb_word((unsigned char)w[i]);
        if (wb0 && tb_ieq(w, i, a) && tb_ieq(w, i + la, b)) {
            const size_t e = i + la + lb;
            const bool ok = need_space ? (e < w.size() && is_space((unsigned char)w[e]))
                                       : (e == w.size() || !tb_word((unsigned char)w[e]));
            if (ok) {
                o += ' '; o.append(w, i, la); o += ' '; o.append(w, i + la, lb); o += ' ';
                i = e + (need_space ? 1 : 0);
                any = true;
                continue;
            }
        }
        o += w[i++];
    }
    if (any) w.swap(o);
}

template <class F>
bool tb_tokenize_piece(const char* s, size_t n, bool at_start, bool at_end, bool ws_after, F emit) {
    std::string w;
    w.reserve(n + 16);
    if (!at_start) w += ' ';
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)s[i];
        if (c >= 128) return false;
        w += lower(c);
    }
    if (!at_end) w += ' ';
    std::string o;
    // ---- STARTING_QUOTES
    {   // ([`]+) -> " \1 "
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (w[i] == '`') { size_t j = i; while (j < w.size() && w[j] == '`') ++j; o += ' '; o.append(w, i, j - i); o += ' '; i = j; }
            else o += w[i++];
        }
        w.swap(o);
    }
    if (at_start && !w.empty() && w[0] == '"') w.replace(0, 1, "``");                       // ^" -> ``
    {   // (``) -> " `` "
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (w[i] == '`' && i + 1 < w.size() && w[i + 1] == '`') { o += " `` "; i += 2; } else o += w[i++];
        }
        w.swap(o);
    }
    {   // ([ \(\[{<])("|'') -> "\1 `` "
        o.clear();
        for (size_t i = 0; i < w.size();) {
            const char c = w[i];
            if ((c == ' ' || c == '(' || c == '[' || c == '{' || c == '<') && i + 1 < w.size()) {
                if (w[i + 1] == '"') { o += c; o += " `` "; i += 2; continue; }
                if (w[i + 1] == '\'' && i + 2 < w.size() && w[i + 2] == '\'') { o += c; o += " `` "; i += 3; continue; }
            }
            o += w[i++];
        }
        w.swap(o);
    }
    {   // (?i)(')(?!re|ve|ll|m|t|s|d|n)(\w)\b -> "\1 \2": an apostrophe before a ONE-character word that is not a clitic letter
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (w[i] == '\'' && i + 1 < w.size() && tb_word((unsigned char)w[i + 1]) &&
                (i + 2 == w.size() || !tb_word((unsigned char)w[i + 2]))) {
                const char c = lower((unsigned char)w[i + 1]);
                if (c != 'm' && c != 't' && c != 's' && c != 'd' && c != 'n') { o += "' "; o += w[i + 1]; i += 2; continue; }
            }
            o += w[i++];
        }
        w.swap(o);
    }
    // ---- PUNCTUATION
    tb_final_period(w, true, at_end || ws_after);
    {   // ([:,])([^\d]) -> " \1 \2"
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if ((w[i] == ':' || w[i] == ',') && i + 1 < w.size() && !is_digit((unsigned char)w[i + 1])) {
                o += ' '; o += w[i]; o += ' '; o += w[i + 1]; i += 2;
            } else o += w[i++];
        }
        w.swap(o);
    }
    if (at_end && !w.empty() && (w.back() == ':' || w.back() == ',')) {                     // ([:,])$ -> " \1 "
        const char c = w.back(); w.pop_back(); w += ' '; w += c; w += ' ';
        // (the text now ends in a blank: the anchored rules below see it, as re.sub on the rewritten text does)
    }
    {   // \.{2,} -> " \g<0> "
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (w[i] == '.') {
                size_t j = i; while (j < w.size() && w[j] == '.') ++j;
                if (j - i >= 2) { o += ' '; o.append(w, i, j - i); o += ' '; } else o += '.';
                i = j;
            } else o += w[i++];
        }
        w.swap(o);
    }
    tb_pad_chars(w, [](unsigned char c) { return c == ';' || c == '@' || c == '#' || c == '$' || c == '%' || c == '&'; });
    tb_final_period(w, false, at_end || ws_after);
    tb_pad_chars(w, [](unsigned char c) { return c == '?' || c == '!'; });
    {   // ([^'])' (blank) -> "\1 ' "
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (w[i] != '\'' && i + 2 < w.size() && w[i + 1] == '\'' && w[i + 2] == ' ') { o += w[i]; o += " ' "; i += 3; }
            else o += w[i++];
        }
        w.swap(o);
    }
    tb_pad_chars(w, [](unsigned char c) { return c == '*'; });
    tb_pad_chars(w, [](unsigned char c) { return c == ']' || c == '[' || c == '(' || c == ')' || c == '{' || c == '}' || c == '<' || c == '>'; });
    {   // -- -> " -- "
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (w[i] == '-' && i + 1 < w.size() && w[i + 1] == '-') { o += " -- "; i += 2; } else o += w[i++];
        }
        w.swap(o);
    }
    w = " " + w + " ";
    // ---- ENDING_QUOTES
    {   // " -> " '' "
        o.clear();
        for (char c : w) { if (c == '"') o += " '' "; else o += c; }
        w.swap(o);
    }
    {   // (\S)('') -> "\1 \2 "
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (!is_space((unsigned char)w[i]) && i + 2 < w.size() && w[i + 1] == '\'' && w[i + 2] == '\'') { o += w[i]; o += " '' "; i += 3; }
            else o += w[i++];
        }
        w.swap(o);
    }
    {   // ([^' ])('[sS]|'[mM]|'[dD]|') (blank) -> "\1 \2 "
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (w[i] != '\'' && w[i] != ' ' && i + 1 < w.size() && w[i + 1] == '\'') {
                const char c = i + 2 < w.size() ? lower((unsigned char)w[i + 2]) : 0;
                if ((c == 's' || c == 'm' || c == 'd') && i + 3 < w.size() && w[i + 3] == ' ') {
                    o += w[i]; o += ' '; o.append(w, i + 1, 2); o += ' '; i += 4; continue;
                }
                if (i + 2 < w.size() && w[i + 2] == ' ') { o += w[i]; o += " ' "; i += 3; continue; }
            }
            o += w[i++];
        }
        w.swap(o);
    }
    {   // ([^' ])('ll|'re|'ve|n't) (blank) -> "\1 \2 "   (upper-case forms: the text is lower-cased)
        o.clear();
        for (size_t i = 0; i < w.size();) {
            if (w[i] != '\'' && w[i] != ' ' && i + 4 < w.size() && w[i + 4] == ' ' &&
                (tb_ieq(w, i + 1, "'ll") || tb_ieq(w, i + 1, "'re") || tb_ieq(w, i + 1, "'ve") || tb_ieq(w, i + 1, "n't"))) {
                // 'lL / 'Ll etc. are not in nltk's alternation, but lower-cased text never holds them
                o += w[i]; o += ' '; o.append(w, i + 1, 3); o += ' '; i += 5; continue;
            }
            o += w[i++];
        }
        w.swap(o);
    }
    // ---- CONTRACTIONS2 / CONTRACTIONS3
    tb_contraction(w, "can", "not", false);
    tb_contraction(w, "d", "'ye", false);
    tb_contraction(w, "gim", "me", false);
    tb_contraction(w, "gon", "na", false);
    tb_contraction(w, "got", "ta", false);
    tb_contraction(w, "lem", "me", false);
    tb_contraction(w, "more", "'n", false);
    tb_contraction(w, "wan", "na", true);
    for (const char* rest : {"is", "was"}) {   // (?i) ('t)(is|was)\b -> " \1 \2 "
        const size_t lr = strlen(rest);
        o.clear();
        bool any = false;
        for (size_t i = 0; i < w.size();) {
            if (w[i] == ' ' && tb_ieq(w, i + 1, "'t") && tb_ieq(w, i + 3, rest) &&
                (i + 3 + lr == w.size() || !tb_word((unsigned char)w[i + 3 + lr]))) {
                o += " 't "; o.append(w, i + 3, lr); o += ' '; i += 3 + lr; any = true; continue;
            }
            o += w[i++];
        }
        if (any) w.swap(o);
    }
    // ---- split
    size_t i = 0;
    std::string tok;
    while (i < w.size()) {
        while (i < w.size() && is_space((unsigned char)w[i])) ++i;
        size_t j = i;
        while (j < w.size() && !is_space((unsigned char)w[j])) ++j;
        if (j > i) { tok.assign(w, i, j - i); emit(tok); }
        i = j;
    }
    return true;
}

// tokens of a piece of text bounded by whitespace or the text's ends, lower-cased; returns false when the text leaves the fast path
// of `kind` (kind 0: never for ASCII; kind 1: non-ASCII).  at_start / at_end / ws_after: where the piece sits (kind 1 only).
template <class F>
bool tokenize_piece(const char* s, size_t n, int kind, F emit, bool at_start = true, bool at_end = true, bool ws_after = true) {
    if (kind == 1) return tb_tokenize_piece(s, n, at_start, at_end, ws_after, emit);
    size_t i = 0;
    std::string w;
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        if (is_space(c)) { ++i; continue; }
        if (c >= 128) return false;
        if (is_alnum(c)) {
            w.clear();
            while (i < n && is_alnum((unsigned char)s[i])) { w += lower((unsigned char)s[i]); ++i; }
            emit(w);
        } else {
            w.assign(1, (char)c);
            emit(w);
            ++i;
        }
    }
    return true;
}

// ---- Punkt: the sentence splitter in front of the Treebank step (nltk/tokenize/punkt.py, Kiss & Strunk 2006; third-party, not
// vendored in the reference).  Its DECISIONS are a fixed algorithm over four tables that the trained model file only fills:
// abbreviation types, collocations, frequent sentence starters, orthographic contexts of word types.  Restated here for ASCII
// text without control characters: PunktLanguageVars' two regular expressions (period contexts, word tokens), PunktToken's
// derived properties, the first (type-based) and second (token-based: collocation, orthographic, sentence-starter heuristics)
// annotation passes, text_contains_sentbreak, _slices_from_text and _realign_boundaries -- i.e. span_tokenize(text).  Pinned by
// tests/golden/punkt_native_kat.json (2 x 2,500 texts split by the real PunktSentenceTokenizer with empty and hand-filled tables);
// at run time leaf_amd/attacks.py compares it with the installed nltk's Punkt on a battery before it is used.
struct Punkt {
    std::unordered_set<std::string> abbrev, starters, colloc;   // colloc keys: first + '\t' + second
    std::unordered_map<std::string, int> ortho;
    // nltk 3.6.6+ finds the period contexts with a rewritten scan (the ReDoS fix, _match_potential_end_contexts) that is meant to be
    // equivalent to the regular expression restated here (3.6.5: \S*[.?!](?=...)) and provably is when a whitespace-delimited chunk
    // holds at most ONE candidate position; with `strict` a text with two or more in one chunk ("what?! yes", "wow!!! nice") is
    // declined instead of decided with the older generation's rule.  The Python side switches strict off after the installed
    // nltk has answered a battery of such texts like this rule (attacks.Dictionary._native_punkt).
    bool strict = true;
};

enum { PK_ORTHO_BEG_UC = 2, PK_ORTHO_MID_UC = 4, PK_ORTHO_UNK_UC = 8, PK_ORTHO_BEG_LC = 16, PK_ORTHO_MID_LC = 32, PK_ORTHO_UNK_LC = 64,
       PK_ORTHO_UC = 14, PK_ORTHO_LC = 112 };

struct PkTok {
    std::string tok, type;       // type: lower-cased, numbers -> ##number##
    bool period_final = false, sentbreak = false, abbr = false, ellipsis = false;
    bool first_upper() const { return tok[0] >= 'A' && tok[0] <= 'Z'; }
    bool first_lower() const { return tok[0] >= 'a' && tok[0] <= 'z'; }
    bool is_ellipsis() const {                                   // \.\.+$ matched from the token's start
        if (tok.size() < 2) return false;
        for (char ch : tok) if (ch != '.') return false;
        return true;
    }
    bool is_initial() const { return tok.size() == 2 && (is_letter((unsigned char)tok[0]) || tok[0] == '_') && tok[1] == '.'; }   // [^\W\d]\.$
    std::string type_no_period() const { return type.size() > 1 && type.back() == '.' ? type.substr(0, type.size() - 1) : type; }
    std::string type_no_sentperiod() const { return sentbreak ? type_no_period() : type; }
};

// _re_multi_char_punct  (?:\-{2,}|\.{2,}|(?:\.\s){2,}\.)  at t[p]: length of the match or 0
size_t pk_multi(const char* t, size_t n, size_t p) {
    if (p >= n) return 0;
    if (t[p] == '-') { size_t q = p; while (q < n && t[q] == '-') ++q; return q - p >= 2 ? q - p : 0; }
    if (t[p] != '.') return 0;
    if (p + 1 < n && t[p + 1] == '.') { size_t q = p; while (q < n && t[q] == '.') ++q; return q - p; }
    size_t q = p, k = 0;                                         // k pairs ". " then a '.': greedy with one step of backtracking
    while (q + 1 < n && t[q] == '.' && is_space((unsigned char)t[q + 1])) { q += 2; ++k; }
    if (k >= 2 && q < n && t[q] == '.') return q + 1 - p;
    if (k >= 3) return q - 2 + 1 - p;
    return 0;
}

// PunktLanguageVars.word_tokenize: findall of  MULTI | (?=WORDSTART)\S+?(?=\s|$|NONWORD|MULTI|,(?=$|\s|NONWORD|MULTI)) | \S
void pk_word_tokens(const char* t, size_t n, std::vector<PkTok>& out) {
    auto end_ok = [&](size_t q) {
        if (q >= n || is_space((unsigned char)t[q]) || punkt_nonword((unsigned char)t[q]) || pk_multi(t, n, q)) return true;
        if (t[q] == ',') {
            const size_t r = q + 1;
            return r >= n || is_space((unsigned char)t[r]) || punkt_nonword((unsigned char)t[r]) || pk_multi(t, n, r) != 0;
        }
        return false;
    };
    size_t p = 0;
    while (p < n) {
        if (is_space((unsigned char)t[p])) { ++p; continue; }
        size_t len = pk_multi(t, n, p);
        if (!len) {
            const char c = t[p];
            const bool wordstart = !(c == '(' || c == '"' || c == '`' || c == '{' || c == '[' || c == ':' || c == ';' || c == '&' ||
                                     c == '#' || c == '*' || c == '@' || c == ')' || c == '}' || c == ']' || c == '-' || c == ',');
            len = 1;
            if (wordstart) while (!end_ok(p + len)) ++len;
        }
        PkTok k;
        k.tok.assign(t + p, len);
        k.period_final = k.tok.back() == '.';
        // type: _RE_NUMERIC  ^-?[\.,]?\d[\d,\.-]*\.?$  -> ##number##
        {
            const std::string& w = k.tok;
            size_t i = 0;
            if (i < w.size() && w[i] == '-') ++i;
            if (i < w.size() && (w[i] == '.' || w[i] == ',')) ++i;
            bool num = i < w.size() && is_digit((unsigned char)w[i]);
            for (size_t j = i + 1; j < w.size() && num; ++j) num = is_digit((unsigned char)w[j]) || w[j] == ',' || w[j] == '.' || w[j] == '-';
            if (num) k.type = "##number##";
            else { k.type.reserve(w.size()); for (char ch : w) k.type += lower((unsigned char)ch); }
        }
        out.push_back(std::move(k));
        p += len;
    }
}

// _ortho_heuristic: 1 = the token starts a sentence, 0 = it does not, -1 = unknown
int pk_ortho(const Punkt& pk, const PkTok& k) {
    if (k.tok.size() == 1 && strchr(";:,.!?", k.tok[0])) return 0;
    auto it = pk.ortho.find(k.type_no_sentperiod());
    const int oc = it == pk.ortho.end() ? 0 : it->second;
    if (k.first_upper() && (oc & PK_ORTHO_LC) && !(oc & PK_ORTHO_MID_UC)) return 1;
    if (k.first_lower() && ((oc & PK_ORTHO_UC) || !(oc & PK_ORTHO_BEG_LC))) return 0;
    return -1;
}

// text_contains_sentbreak(context): a token marked as a sentence break (after both passes) that is not the last token
bool pk_contains_sentbreak(const Punkt& pk, const char* t, size_t n, std::vector<PkTok>& toks) {
    toks.clear();
    pk_word_tokens(t, n, toks);
    for (PkTok& k : toks) {                                      // _first_pass_annotation
        if (k.tok == "." || k.tok == "?" || k.tok == "!") k.sentbreak = true;
        else if (k.is_ellipsis()) k.ellipsis = true;
        else if (k.period_final && !(k.tok.size() >= 2 && k.tok[k.tok.size() - 2] == '.')) {
            std::string base;
            for (size_t i = 0; i + 1 < k.tok.size(); ++i) base += lower((unsigned char)k.tok[i]);
            const size_t dash = base.rfind('-');
            if (pk.abbrev.count(base) || (dash != std::string::npos && pk.abbrev.count(base.substr(dash + 1)))) k.abbr = true;
            else k.sentbreak = true;
        }
    }
    for (size_t i = 0; i + 1 < toks.size(); ++i) {               // _second_pass_annotation(tok i, tok i + 1)
        PkTok& a = toks[i];
        const PkTok& b = toks[i + 1];
        if (!a.period_final) continue;
        const std::string typ = a.type_no_period(), next_typ = b.type_no_sentperiod();
        const bool initial = a.is_initial();
        if (pk.colloc.count(typ + '\t' + next_typ)) { a.sentbreak = false; a.abbr = true; continue; }
        if ((a.abbr || a.ellipsis) && !initial) {
            if (pk_ortho(pk, b) == 1) { a.sentbreak = true; continue; }
            if (b.first_upper() && pk.starters.count(next_typ)) { a.sentbreak = true; continue; }
        }
        if (initial || typ == "##number##") {
            const int st = pk_ortho(pk, b);
            if (st == 0) { a.sentbreak = false; a.abbr = true; continue; }
            if (st == -1 && initial && b.first_upper()) {
                auto it = pk.ortho.find(next_typ);
                if (!((it == pk.ortho.end() ? 0 : it->second) & PK_ORTHO_LC)) { a.sentbreak = false; a.abbr = true; continue; }
            }
        }
    }
    for (size_t i = 0; i + 1 < toks.size(); ++i)
        if (toks[i].sentbreak) return true;
    return false;
}

// PunktSentenceTokenizer.span_tokenize(text) (realign_boundaries = True): (start, end) pairs appended to `out`.  false = the text is
// outside the restated domain (non-ASCII or control characters; with pk.strict two candidate positions in one chunk): the caller asks nltk.
bool punkt_spans(const Punkt& pk, const char* t, size_t n, std::vector<int32_t>& out) {
    for (size_t i = 0; i < n; ++i)
        if ((unsigned char)t[i] < 32 || (unsigned char)t[i] > 126) return false;
    // _slices_from_text.  period_context_re =  \S*[.?!](?=(NONWORD)|\s+(\S+))  : per whitespace-delimited chunk the leftmost match
    // starts at the chunk's first character and (greedy \S*, backtracking from the chunk's end) ends behind its LAST sentence-final
    // character that is followed by a NONWORD character or by blanks and another chunk.
    std::vector<int32_t> sl;                                     // raw slices (start, stop)
    std::vector<PkTok> toks;
    std::string ctx;
    size_t last_break = 0, p = 0;
    while (p < n) {
        if (is_space((unsigned char)t[p])) { ++p; continue; }
        size_t e = p;
        while (e < n && !is_space((unsigned char)t[e])) ++e;
        size_t nx = e;                                           // next chunk [nx, nxe)
        while (nx < n && is_space((unsigned char)t[nx])) ++nx;
        size_t nxe = nx;
        while (nxe < n && !is_space((unsigned char)t[nxe])) ++nxe;
        if (pk.strict) {
            int cands = 0;
            for (size_t i = p; i < e; ++i)
                if ((t[i] == '.' || t[i] == '?' || t[i] == '!') &&
                    ((i + 1 < e && punkt_nonword((unsigned char)t[i + 1])) || (i + 1 == e && nx < n))) ++cands;
            if (cands > 1) return false;
        }
        for (size_t i = e; i-- > p;) {
            if (t[i] != '.' && t[i] != '?' && t[i] != '!') continue;
            const bool nonword = i + 1 < e && punkt_nonword((unsigned char)t[i + 1]);
            const bool spaced = i + 1 == e && nx < n;
            if (!nonword && !spaced) continue;
            ctx.assign(t + p, i + 1 - p);                         // match.group() + after_tok
            if (nonword) ctx += t[i + 1];
            else ctx.append(t + e, nxe - e);
            if (pk_contains_sentbreak(pk, ctx.data(), ctx.size(), toks)) {
                sl.push_back((int32_t)last_break); sl.push_back((int32_t)(i + 1));
                last_break = spaced ? nx : i + 1;
            }
            break;
        }
        p = e;
    }
    size_t end = n;
    while (end > 0 && is_space((unsigned char)t[end - 1])) --end;
    sl.push_back((int32_t)last_break); sl.push_back((int32_t)end);
    // _realign_boundaries: closing quotes / brackets that directly follow a break belong to the sentence in front of them
    // (re_boundary_realignment = ["')\]}]+?(?:\s+|(?=--)|$) matched at the start of the NEXT slice's text)
    const size_t ns = sl.size() / 2;
    int32_t realign = 0;
    for (size_t i = 0; i < ns; ++i) {
        const int32_t s1 = sl[2 * i] + realign, e1 = sl[2 * i + 1];
        if (i + 1 == ns) {
            if (s1 < e1) { out.push_back(s1); out.push_back(e1); }
            break;
        }
        const int32_t s2 = sl[2 * i + 2], e2 = std::max(sl[2 * i + 3], s2);
        int32_t q = s2;
        while (q < e2 && (t[q] == '"' || t[q] == '\'' || t[q] == ')' || t[q] == ']' || t[q] == '}')) ++q;
        bool m = q > s2;
        int32_t mend = q;
        if (m) {
            if (q < e2 && is_space((unsigned char)t[q])) { while (mend < e2 && is_space((unsigned char)t[mend])) ++mend; }
            else if (q + 1 < e2 && t[q] == '-' && t[q + 1] == '-') {}
            else if (q == e2) {}
            else m = false;
        }
        if (m) {
            out.push_back(s1); out.push_back(q);
            realign = mend - s2;
        } else {
            realign = 0;
            if (s1 < e1) { out.push_back(s1); out.push_back(e1); }
        }
    }
    return true;
}

}  // namespace

struct leaf_dict : Dict {};
struct leaf_punkt : Punkt {};

static void punkt_lines(const char* blob, size_t len, const std::function<void(const std::string&)>& f) {
    size_t pos = 0;
    while (blob && pos < len) {
        size_t e = pos;
        while (e < len && blob[e] != '\n') ++e;
        if (e > pos) f(std::string(blob + pos, e - pos));
        pos = e + 1;
    }
}

// The four Punkt tables as '\n'-separated UTF-8 lines: abbreviation types; collocations "first\tsecond"; sentence starters;
// orthographic contexts "type\tflags" (nltk.tokenize.punkt.PunktParameters: abbrev_types, collocations, sent_starters, ortho_context).
extern "C" int leaf_punkt_create(const char* abbrev, size_t abbrev_len, const char* colloc, size_t colloc_len, const char* starters,
                                 size_t starters_len, const char* ortho, size_t ortho_len, leaf_punkt_t* out) {
    if (!out) return 1;
    leaf_punkt* p = new leaf_punkt();
    bool ok = true;
    punkt_lines(abbrev, abbrev_len, [&](const std::string& l) { p->abbrev.insert(l); });
    punkt_lines(colloc, colloc_len, [&](const std::string& l) { if (l.find('\t') == std::string::npos) ok = false; p->colloc.insert(l); });
    punkt_lines(starters, starters_len, [&](const std::string& l) { p->starters.insert(l); });
    punkt_lines(ortho, ortho_len, [&](const std::string& l) {
        const size_t tab = l.rfind('\t');
        if (tab == std::string::npos) { ok = false; return; }
        p->ortho[l.substr(0, tab)] = atoi(l.c_str() + tab + 1);
    });
    if (!ok) { delete p; return 1; }
    *out = p;
    return 0;
}

extern "C" void leaf_punkt_destroy(leaf_punkt_t p) { delete p; }
// strict = 1 (default): decline texts on which nltk generations may differ (struct Punkt); 0: decide everything as nltk 3.6.5 does
extern "C" int leaf_punkt_set_strict(leaf_punkt_t p, int strict) { if (!p) return 1; p->strict = strict != 0; return 0; }

// span_tokenize(text): up to cap_pairs (start, end) pairs into spans; returns 0, 2 = declined (non-ASCII / control characters), 1 = bad
// arguments or overflow
extern "C" int leaf_punkt_spans(leaf_punkt_t p, const char* text, int len, int32_t* spans, int cap_pairs, int32_t* n_pairs) {
    if (!p || !text || !spans || !n_pairs || len < 0 || cap_pairs < 1) return 1;
    std::vector<int32_t> out;
    if (!punkt_spans(*p, text, (size_t)len, out)) return 2;
    if ((int)(out.size() / 2) > cap_pairs) return 1;
    if (!out.empty()) memcpy(spans, out.data(), out.size() * sizeof(int32_t));
    *n_pairs = (int32_t)(out.size() / 2);
    return 0;
}

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

static int constrain_impl(leaf_dict_t d, int kind, const char* const* sentences, const int32_t* sent_len, int B,
                          const int32_t* z, const int32_t* c, int rho, const int32_t* ranges, const int32_t* ranges_off,
                          uint8_t* valid, uint8_t* fallback, int n_threads, const Punkt* pk = nullptr) {
    if (!d || !sentences || !sent_len || !z || !c || !valid || !fallback || rho < 1 || (kind != 0 && kind != 1)) return 1;
    if ((ranges == nullptr) != (ranges_off == nullptr) || (ranges && kind != 1) || (pk && (kind != 1 || ranges))) return 1;
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> bad(0);
    // sentences are independent: one task per sentence (its rho candidates share the multiplicity map)
    parallel_for(B, n_threads, [&](int b, int) {
        const char* s = sentences[b];
        const int n = sent_len[b];
        uint8_t* v = valid + (size_t)b * rho;
        uint8_t* fb = fallback + (size_t)b * rho;
        std::unordered_map<std::string, int> mult;       // dictionary words of the sentence -> occurrences
        // kind 1 with sentence ranges from the caller's Punkt (leaf_tok_constrain_ranges): the caption is tokenised sentence by
        // sentence, as nltk.word_tokenize does, whatever its periods look like
        const int32_t* rg = nullptr;
        int nrg = 0;
        if (ranges_off && ranges_off[b + 1] > ranges_off[b]) { rg = ranges + 2 * (size_t)ranges_off[b]; nrg = ranges_off[b + 1] - ranges_off[b]; }
        bool ok = true;
        // ... or with the Punkt tables themselves (leaf_tok_constrain_punkt): the spans are computed here, for the caption and for
        // every candidate the window logic below cannot decide
        std::vector<int32_t> own_spans, cand_spans;
        // (Punkt sees the LOWER-CASED text, utils_attacks.py:131,139: word_tokenize(o.lower()); the Treebank step lower-cases itself)
        std::string low;
        if (pk && !tb_punkt_free(s, (size_t)n)) {
            low.assign(s, (size_t)n);
            for (char& ch : low) ch = lower((unsigned char)ch);
            ok = punkt_spans(*pk, low.data(), low.size(), own_spans) && !own_spans.empty();
            if (ok) { rg = own_spans.data(); nrg = (int)(own_spans.size() / 2); }
        }
        std::unordered_set<std::string> seen;
        // distinct dictionary words of a whole candidate: Punkt spans, then the Treebank step sentence by sentence; -1 = declined
        auto count_whole = [&](std::string& text) -> int {
            seen.clear();
            for (char& ch : text) ch = lower((unsigned char)ch);
            auto emit = [&](const std::string& w) { if (d->words.count(w)) seen.insert(w); };
            if (tb_punkt_free(text.data(), text.size())) return tokenize_piece(text.data(), text.size(), kind, emit) ? (int)seen.size() : -1;
            cand_spans.clear();
            if (!punkt_spans(*pk, text.data(), text.size(), cand_spans)) return -1;
            int prev_end = 0;
            for (size_t i = 0; i + 1 < cand_spans.size(); i += 2) {
                const int S = cand_spans[i], E = cand_spans[i + 1];
                if (S < prev_end || E < S || E > (int)text.size()) return -1;
                prev_end = E;
                if (!tokenize_piece(text.data() + S, (size_t)(E - S), kind, emit)) return -1;
            }
            return (int)seen.size();
        };
        if (rg) {
            int prev_end = 0;
            for (int i = 0; i < nrg && ok; ++i) {
                const int S = rg[2 * i], E = rg[2 * i + 1];
                ok = S >= prev_end && E >= S && E <= n;
                prev_end = E;
                if (ok) ok = tokenize_piece(s + S, (size_t)(E - S), kind, [&](const std::string& w) { if (d->words.count(w)) ++mult[w]; });
            }
            for (int q = prev_end; q < n && ok; ++q) ok = is_space((unsigned char)s[q]);     // nothing but blanks outside the sentences
        } else {
            ok = kind == 0 || tb_punkt_free(s, (size_t)n);
            ok = ok && tokenize_piece(s, (size_t)n, kind, [&](const std::string& w) { if (d->words.count(w)) ++mult[w]; });
        }
        if (!ok) { for (int r = 0; r < rho; ++r) { fb[r] = 1; v[r] = 0; } return; }
        // kind 1: index of the text's final period, when it has one that ends its chunk (its split depends on what follows it)
        int final_period = -1;
        if (kind == 1) {
            for (int i = n - 1; i >= 0; --i) {
                const unsigned char ch = (unsigned char)s[i];
                if (ch == '.') { if (i == 0 || s[i - 1] != '.') final_period = i; break; }
                if (!tb_closer(ch) && !is_space(ch)) break;
            }
        }
        std::string cand;
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
            bool at_start = true, at_end = true, ws_after = true;
            if (kind == 1 && rg) {
                // Punkt decides a sentence break from the token that carries the period and the token after it: an edit is
                // decided natively only when its window holds no sentence-final character before or after the edit, is not the
                // token right behind one, and lies inside ONE sentence of the caller's segmentation
                auto terminator = [](unsigned char ch) { return ch == '.' || ch == '?' || ch == '!'; };
                bool clean = !(has_ins && terminator((unsigned char)ins));
                for (int q = L; q < R && clean; ++q) clean = !terminator((unsigned char)s[q]);
                int pe = L;                                       // end of the chunk in front of the window
                while (pe > 0 && is_space((unsigned char)s[pe - 1])) --pe;
                int pq = pe;
                while (pq > 0 && tb_closer((unsigned char)s[pq - 1])) --pq;
                if (clean && pq > 0 && terminator((unsigned char)s[pq - 1])) clean = false;
                int si = -1;
                for (int i = 0; i < nrg && clean; ++i)
                    if (L >= rg[2 * i] && R <= rg[2 * i + 1]) { si = i; break; }
                if (!clean || si < 0 || L == R) {
                    int cnt = -1;
                    if (pk) {
                        cand.assign(s, (size_t)e0);
                        if (has_ins) cand += ins;
                        cand.append(s + e1, (size_t)(n - e1));
                        cnt = count_whole(cand);
                    }
                    if (cnt < 0) { fb[r] = 1; v[r] = 0; } else v[r] = cnt < lo ? 1 : 0;
                    continue;
                }
                at_start = L == rg[2 * si]; at_end = R == rg[2 * si + 1]; ws_after = at_end;
            } else if (kind == 1) {
                // the candidate as a whole must stay independent of sentence boundaries, and an edit behind the final period's
                // chunk could change how THAT chunk (outside the window) splits
                cand.assign(s, (size_t)e0);
                if (has_ins) cand += ins;
                cand.append(s + e1, (size_t)(n - e1));
                if (!tb_punkt_free(cand.data(), cand.size()) || (final_period >= 0 && L > final_period)) {
                    const int cnt = pk ? count_whole(cand) : -1;
                    if (cnt < 0) { fb[r] = 1; v[r] = 0; } else v[r] = cnt < lo ? 1 : 0;
                    continue;
                }
                at_start = L == 0; at_end = R == n;
                for (int q = R; q < n && ws_after; ++q) ws_after = is_space((unsigned char)s[q]);
            }
            bool okw = tokenize_piece(s + L, (size_t)(R - L), kind, [&](const std::string& w) { if (d->words.count(w)) --delta[w]; },
                                      at_start, at_end, ws_after);
            win.assign(s + L, (size_t)(e0 - L));
            if (has_ins) win += ins;
            win.append(s + e1, (size_t)(R - e1));
            okw = okw && tokenize_piece(win.data(), win.size(), kind, [&](const std::string& w) { if (d->words.count(w)) ++delta[w]; },
                                        at_start, at_end, ws_after);
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

extern "C" int leaf_tok_constrain(leaf_dict_t d, int kind, const char* const* sentences, const int32_t* sent_len, int B,
                                  const int32_t* z, const int32_t* c, int rho, uint8_t* valid, uint8_t* fallback, int n_threads) {
    return constrain_impl(d, kind, sentences, sent_len, B, z, c, rho, nullptr, nullptr, valid, fallback, n_threads);
}

// The same with the caller's SENTENCE SEGMENTATION for captions whose word tokens depend on it (tokenizer kind 1): sentence b's
// ranges are ranges[2 i], ranges[2 i + 1] (start, end) for i in [ranges_off[b], ranges_off[b + 1]) -- the spans of nltk's Punkt
// (PunktSentenceTokenizer.span_tokenize) on the lower-cased caption; a caption without ranges is handled as by leaf_tok_constrain.
extern "C" int leaf_tok_constrain_ranges(leaf_dict_t d, int kind, const char* const* sentences, const int32_t* sent_len, int B,
                                         const int32_t* z, const int32_t* c, int rho, const int32_t* ranges,
                                         const int32_t* ranges_off, uint8_t* valid, uint8_t* fallback, int n_threads) {
    return constrain_impl(d, kind, sentences, sent_len, B, z, c, rho, ranges, ranges_off, valid, fallback, n_threads);
}

// The same with the Punkt tables (leaf_punkt_create): sentence spans of captions and candidates are computed natively, nothing is
// declined but non-ASCII text and control characters.
extern "C" int leaf_tok_constrain_punkt(leaf_dict_t d, leaf_punkt_t punkt, const char* const* sentences, const int32_t* sent_len, int B,
                                        const int32_t* z, const int32_t* c, int rho, uint8_t* valid, uint8_t* fallback, int n_threads) {
    if (!punkt) return 1;
    return constrain_impl(d, 1, sentences, sent_len, B, z, c, rho, nullptr, nullptr, valid, fallback, n_threads, punkt);
}

// Number of DISTINCT dictionary words of one text (utils_attacks.py:135: len(W & set(word_tokenize(text.lower())))), tokenizer kind as
// above; kind 1 with the caller's sentence spans (n_ranges pairs; 0 = the text must not depend on sentence boundaries).
// Returns 0, 2 when declined (non-ASCII; kind 1 without spans on a sentence-boundary dependent text), 1 on bad arguments.
extern "C" int leaf_tok_count_words(leaf_dict_t d, int kind, const char* text, int len, const int32_t* ranges, int n_ranges,
                                    int32_t* count) {
    if (!d || !text || !count || len < 0 || (kind != 0 && kind != 1) || n_ranges < 0 || (n_ranges && (!ranges || kind != 1))) return 1;
    std::unordered_set<std::string> seen;
    auto emit = [&](const std::string& w) { if (d->words.count(w)) seen.insert(w); };
    bool ok = true;
    if (n_ranges) {
        int prev_end = 0;
        for (int i = 0; i < n_ranges && ok; ++i) {
            const int S = ranges[2 * i], E = ranges[2 * i + 1];
            if (S < prev_end || E < S || E > len) return 1;
            prev_end = E;
            ok = tokenize_piece(text + S, (size_t)(E - S), kind, emit);
        }
    } else {
        if (kind == 1 && !tb_punkt_free(text, (size_t)len)) return 2;
        ok = tokenize_piece(text, (size_t)len, kind, emit);
    }
    if (!ok) return 2;
    *count = (int32_t)seen.size();
    return 0;
}

// Debug / test hook: word tokens of `text` under tokenizer `kind` (0 regex stand-in, 1 nltk.word_tokenize), joined by '\n' into
// out[0, cap).  Returns 0, 2 when the text is declined (kind 1: non-ASCII or sentence-boundary dependent), 1 on bad arguments /
// overflow.
extern "C" int leaf_tok_word_tokens(int kind, const char* text, int len, char* out, int cap, int* out_len) {
    if (!text || !out || !out_len || len < 0 || cap < 1 || (kind != 0 && kind != 1)) return 1;
    if (kind == 1 && !tb_punkt_free(text, (size_t)len)) return 2;
    std::string joined;
    const bool ok = tokenize_piece(text, (size_t)len, kind, [&](const std::string& w) { joined += w; joined += '\n'; });
    if (!ok) return 2;
    if ((int)joined.size() > cap) return 1;
    memcpy(out, joined.data(), joined.size());
    *out_len = (int)joined.size();
    return 0;
}

// dup_of[b * rho + r] = the first r' <= r whose token row equals candidate r's (the tokenizer lower-cases and
// collapses whitespace, so different edits can give identical id rows; only the first needs computing).  Exact: rows with equal
// hashes are compared element-wise.
extern "C" int leaf_tok_duplicate_map(const int32_t* tokens, int B, int rho, int ctx, int32_t* dup_of, int n_threads) {
    if (!tokens || !dup_of || B < 1 || rho < 1 || ctx < 1) return 1;
    if (n_threads < 1) n_threads = 1;
    parallel_for(B, n_threads, [&](int b, int) {
        const int32_t* t = tokens + (size_t)b * rho * ctx;
        int32_t* d = dup_of + (size_t)b * rho;
        std::vector<uint64_t> h(rho);
        for (int r = 0; r < rho; ++r) {
            // position-weighted sum (independent terms: vectorises; a serial FNV chain cost 0.5 ms per stage)
            uint64_t x = 0;
            const int32_t* row = t + (size_t)r * ctx;
            for (int i = 0; i < ctx; ++i) x += (uint64_t)(uint32_t)row[i] * (0x9E3779B97F4A7C15ull * (uint64_t)(2 * i + 1));
            h[r] = x;
            d[r] = r;
            for (int q = 0; q < r; ++q)
                if (h[q] == x && d[q] == q && memcmp(t + (size_t)q * ctx, row, (size_t)ctx * 4) == 0) { d[r] = q; break; }
        }
    });
    return 0;
}
