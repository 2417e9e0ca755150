// CPU-only sanitizer fuzz of the native host pipeline's text entry points: built with -fsanitize=address,undefined together with
// leaf_amd/csrc/host_text.cpp and run by tests/test_host_cpu.py::test_host_text_under_sanitizers (random captions over a punctuation-heavy
// alphabet, control and non-ASCII bytes included; every return code and span range checked).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../include/leaf_hip.h"
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937 rng(7);
    const std::string words = "a\ncat\ndog\nthe\nphoto\nof\nno\ndr\nend\nit\ns\n";
    leaf_dict_t d; if (leaf_dict_create(words.data(), words.size(), &d)) return 1;
    const std::string ab = "dr\nst\ne.g\nno\nj\n", co = "chair\tfree\n##number##\tcat\n", ss = "the\nit\n", oc = "the\t34\ncat\t48\n##number##\t48\n";
    leaf_punkt_t pk; if (leaf_punkt_create(ab.data(), ab.size(), co.data(), co.size(), ss.data(), ss.size(), oc.data(), oc.size(), &pk)) return 1;
    // BPE + mutation entry points too when the decompressed merge table is given (argv[2])
    leaf_tok_t tk = nullptr;
    if (argc > 2) {
        FILE* f = fopen(argv[2], "rb");
        if (!f) { printf("cannot open %s\n", argv[2]); return 1; }
        std::string merges;
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), f)) > 0) merges.append(buf, got);
        fclose(f);
        if (leaf_tok_create(merges.data(), merges.size(), &tk)) { printf("leaf_tok_create failed\n"); return 1; }
    }
    long tokens_total = 0;
    const char alpha[] = "aab  ..?!,;:'\"()[]{}*@-&#`_3/JdDr\t\n\x01\xc3\xa9";
    const int na = sizeof(alpha) - 1;
    long spans_total = 0, declined = 0, valid_total = 0;
    for (int it = 0; it < iters; ++it) {
        leaf_punkt_set_strict(pk, it & 1);
        const int B = 1 + rng() % 4, rho = 1 + rng() % 12;
        std::vector<std::string> S(B);
        std::vector<const char*> ptr(B);
        std::vector<int32_t> len(B);
        for (int b = 0; b < B; ++b) {
            const int n = rng() % 40;
            for (int i = 0; i < n; ++i) S[b] += alpha[rng() % ((it % 3) ? na - 5 : na)];      // a third of the rounds: control / non-ASCII bytes too
            ptr[b] = S[b].data(); len[b] = (int)S[b].size();
            std::vector<int32_t> sp(2 * (S[b].size() / 2 + 2));
            int32_t np = 0;
            const int rc = leaf_punkt_spans(pk, S[b].data(), (int)S[b].size(), sp.data(), (int)sp.size() / 2, &np);
            if (rc == 0) { spans_total += np; for (int i = 0; i < np; ++i) if (sp[2 * i] < 0 || sp[2 * i + 1] > (int)S[b].size()) { printf("span out of range\n"); return 2; } }
            else if (rc == 2) ++declined; else { printf("rc %d\n", rc); return 2; }
        }
        std::vector<int32_t> z(B * rho), c(B * rho);
        for (int b = 0; b < B; ++b) for (int r = 0; r < rho; ++r) {
            z[b * rho + r] = rng() % (2 * len[b] + 1);
            c[b * rho + r] = (rng() % 8 == 0) ? -1 : 32 + rng() % 95;
        }
        std::vector<uint8_t> v(B * rho), fb(B * rho);
        int rc = leaf_tok_constrain_punkt(d, pk, ptr.data(), len.data(), B, z.data(), c.data(), rho, v.data(), fb.data(), 2);
        if (rc != 0 && rc != 3) { printf("constrain_punkt rc %d\n", rc); return 2; }
        for (auto x : v) valid_total += x;
        for (int kind = 0; kind < 2; ++kind) {
            rc = leaf_tok_constrain(d, kind, ptr.data(), len.data(), B, z.data(), c.data(), rho, v.data(), fb.data(), 2);
            if (rc != 0 && rc != 3) { printf("constrain rc %d\n", rc); return 2; }
        }
        if (tk) {
            const int ctx = (it % 7 == 0) ? 8 : 77;              // short contexts: the truncation path
            std::vector<int32_t> toks((size_t)B * rho * ctx), lens(B * rho);
            std::vector<uint8_t> tfb(B * rho);
            rc = leaf_tok_mutate_encode(tk, ptr.data(), len.data(), B, z.data(), c.data(), rho, ctx, toks.data(), lens.data(), tfb.data(), 2);
            if (rc != 0 && rc != 3) { printf("mutate_encode rc %d\n", rc); return 2; }
            for (int i = 0; i < B * rho; ++i) {
                if (tfb[i]) continue;
                if (lens[i] < 2 || lens[i] > ctx || toks[(size_t)i * ctx] != 49406 || toks[(size_t)i * ctx + lens[i] - 1] != 49407) { printf("bad row\n"); return 2; }
                tokens_total += lens[i];
            }
            rc = leaf_tok_encode_batch(tk, ptr.data(), len.data(), B, ctx, toks.data(), lens.data(), tfb.data(), 2);
            if (rc != 0) { printf("encode_batch rc %d\n", rc); return 2; }
            std::vector<int32_t> dup(B * rho);
            rc = leaf_tok_mutate_encode(tk, ptr.data(), len.data(), B, z.data(), c.data(), rho, ctx, toks.data(), lens.data(), tfb.data(), 1);
            if (leaf_tok_duplicate_map(toks.data(), B, rho, ctx, dup.data(), 2)) { printf("duplicate_map failed\n"); return 2; }
            for (int i = 0; i < B * rho; ++i) if (dup[i] < 0 || dup[i] > i % rho) { printf("bad dup\n"); return 2; }
        }
        int32_t cnt = 0;
        leaf_tok_count_words(d, 1, S[0].data(), len[0], nullptr, 0, &cnt);
        char out[512]; int ol = 0;
        leaf_tok_word_tokens(1, S[0].data(), len[0], out, sizeof(out), &ol);
    }
    leaf_punkt_destroy(pk); leaf_dict_destroy(d);
    if (tk) leaf_tok_destroy(tk);
    printf("fuzz ok: %d rounds, %ld spans, %ld texts declined, %ld valid edits, %ld BPE tokens\n", iters, spans_total, declined, valid_total, tokens_total);
    return 0;
}
