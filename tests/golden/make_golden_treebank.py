#!/usr/bin/env python3
"""Known answers for the word tokenizer behind --constrain (utils_attacks.py:110-143 call nltk.word_tokenize on the
lower-cased sentence / candidate): produced by the REAL third-party implementation, nltk 3.6.5's
``nltk.tokenize.destructive.NLTKWordTokenizer`` (the Treebank step of ``word_tokenize``), found in this build container
under /opt/conda (the system interpreter has no nltk; the punkt sentence model is absent everywhere here, so the
sentence-splitting step cannot be run -- the fixture therefore holds only strings on which it cannot matter: no '.' except in
'..' runs, inside a word, or as the text's final period; see leaf_amd/treebank.py).

    /opt/conda/bin/python3.9 tests/golden/make_golden_treebank.py        # writes tests/golden/treebank_kat.json

Strings: hand-written punctuation cases + single-character edits (the search's own mutation: every one of the 96 characters
of V at random slots) of caption-like sentences.
"""
import json
import os
import random
import string

from nltk.tokenize.destructive import NLTKWordTokenizer
import nltk

HERE = os.path.dirname(os.path.abspath(__file__))
V = string.ascii_lowercase + ' ' + string.ascii_uppercase + string.digits + string.punctuation

HAND = [
    "a photo of a cat", "Hello, world!", "it's a dog's life", "don't stop", "i can't, won't and shouldn't", "they'll we're i've i'd i'm",
    "\"quoted\" text", "a \"quoted word\" here", "''double'' single", "say ``hi''", "(parens) [brackets] {braces} <angles>",
    "a--b and c -- d --- e", "wait... what", "one,two, three ,four , five", "1,000 and 2:30 pm", "ratio 3:4, or a:b", "ends with comma,",
    "ends with colon:", "semi;colon @at #hash $dollar %pct &and", "what?! really?", "star*dust * alone", "cannot gimme gonna gotta lemme wanna go",
    "wanna", "more'n that d'ye know", "'tis the season 'twas night", "rock'n'roll", "'a' 'b c' 'quoted'", "o'clock", "the cat's 'hat'",
    "dogs' bones", "a.b.c", "e.g.x", "file.txt here", "the end.", "the end. ", "the end.)", "the end.\"", "the end.'", "(the end.)",
    "x", "", " ", "  two  spaces  ", "tab\tsep", "a_b __init__ snake_case", "50% off!", "#1 fan", "me & you", "a/b\\c|d~e^f=g+h",
    "`tick` ``ticks``", "'", "''", "\"", "' a", "a '", "a' b", "a 's", "n't", "can not", "CANNOT", "Gonna", "i'M", "it'S", "they'LL",
    "x'y", "x'yz", "'em", "'re", "l'm", "'m", "5'6", "'90s", "rock 'n' roll", "a,b", "a,1", "a:1", ",a", ":a", ",,a", "a,,", "::", "a;b;",
    "!?!", "a!b", "a?b", "(a)", "((a))", "a(b)c", "<b>bold</b>", "--", "---", "a--", "--a", "...", "..", "a..b", "a...", "..a",
]

SENTS = ["a photo of a cat", "two dogs playing in the park", "the red car on the street", "a man with a hat, smiling",
         "it's a beautiful day", "people at sunset (beach)", "don't walk", "a cup of coffee & a book", "kids' toys on the floor",
         "the \"best\" pizza in town", "black-and-white photo", "a woman's portrait", "5 o'clock shadow", "what a view!",
         "sale: 50% off", "cannot wait", "i wanna go home"]


def ok(s):
    """strings on which Punkt's sentence splitting cannot change the letter-bearing tokens (leaf_amd.treebank.punkt_free; checked
    against the real Punkt code on 44,000 random punctuation-heavy strings when this rule was written)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from leaf_amd.treebank import punkt_free
    return punkt_free(s.lower())


def mutate(S, z, c):
    if z & 1:
        i = (z - 1) // 2
        return S[:i] + S[i + 1:] if S[i] == c else S[:i] + c + S[i + 1:]
    i = z // 2
    return S[:i] + c + S[i:]


def main():
    tk = NLTKWordTokenizer()
    rng = random.Random(0)
    cases = list(HAND)
    for S in SENTS:
        for _ in range(90):
            z = rng.randrange(2 * len(S) + 1)
            cases.append(mutate(S, z, rng.choice(V)))
        for _ in range(12):                      # two edits: punctuation next to punctuation
            T = mutate(S, rng.randrange(2 * len(S) + 1), rng.choice(string.punctuation))
            cases.append(mutate(T, rng.randrange(2 * len(T) + 1), rng.choice(string.punctuation + " ")))
    seen, out = set(), []
    for s in cases:
        if s in seen or not ok(s):
            continue
        seen.add(s)
        out.append([s, tk.tokenize(s.lower())])
    with open(os.path.join(HERE, "treebank_kat.json"), "w") as f:
        json.dump({"source": f"nltk {nltk.__version__} nltk.tokenize.destructive.NLTKWordTokenizer().tokenize(s.lower())", "cases": out}, f)
    print(len(out), "cases")


if __name__ == "__main__":
    main()
