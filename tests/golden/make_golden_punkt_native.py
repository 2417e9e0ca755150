"""Known answers for the native restatement of nltk's Punkt sentence splitter (leaf_amd/csrc/host_text.cpp: punkt_spans).

Run with an interpreter that has nltk (the build image: /opt/conda/bin/python3.9, nltk 3.6.5):

    /opt/conda/bin/python3.9 tests/golden/make_golden_punkt_native.py

nltk is a third-party dependency of the reference (requirements.txt:14) that is not vendored in it; its Punkt model file
(``punkt`` / ``punkt_tab``) is not in this image either.  Punkt's DECISIONS are a fixed algorithm over four parameter tables
(abbreviation types, collocations, frequent sentence starters, orthographic contexts) -- the model file only fills the tables.
The fixture therefore pins the algorithm with two parameter sets built here: empty tables (what an untrained
``PunktSentenceTokenizer()`` uses) and hand-filled tables that reach every branch of the second annotation pass; the expected
sentence spans come from the REAL ``PunktSentenceTokenizer.span_tokenize``.  Mixed-case text is included although the search only
ever splits lower-cased captions (utils_attacks.py:127,131), so that the upper-case branches are pinned too.
"""
import json
import os
import random

import nltk
from nltk.tokenize.punkt import PunktParameters, PunktSentenceTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))

WORDS = ["a", "cat", "dog", "the", "The", "But", "but", "dr", "Dr", "mr", "e.g", "i.e", "st", "St", "john", "John", "u.s", "vs", "no",
         "3", "3.5", "1,000", "-2", "12", "j", "J", "x", "bach", "Bach", "smith", "Smith", "photo", "of", "and", "in", "it's", "can't",
         "he", "He", "she", "went", "home", "inc", "co", "ph.d", "a.m", "p.m", "and/or", "mid-st", "well-no", "_", "a_b", "42nd"]
PUNCT = [".", ".", ".", ".", "?", "!", ",", ";", ":", "\"", "'", ")", "(", "]", "[", "}", "{", "*", "@", "--", "-", "...", "..", ". . .",
         "&", "#", "`", "''", "?!", "!?", ".)", ".\"", ".'", "?)", ")."]


def manual_params():
    p = PunktParameters()
    p.abbrev_types.update({"dr", "mr", "e.g", "i.e", "st", "u.s", "vs", "no", "inc", "co", "ph.d", "a.m", "p.m", "j"})
    p.collocations.update({("st", "john"), ("##number##", "dog"), ("j", "bach"), ("cat", "dog"), ("x", "##number##"), ("3.5", "cat"),
                           ("dr", "smith"), ("home", "he")})
    p.sent_starters.update({"the", "but", "he"})
    BEG_UC, MID_UC, UNK_UC, BEG_LC, MID_LC, UNK_LC = 2, 4, 8, 16, 32, 64
    for typ, flag in [("the", BEG_UC | MID_LC), ("but", BEG_UC | BEG_LC | MID_LC), ("john", MID_UC | BEG_UC), ("bach", MID_UC),
                      ("smith", UNK_UC), ("cat", MID_LC | BEG_LC), ("dog", MID_LC), ("he", BEG_UC | MID_LC | BEG_LC), ("she", UNK_LC),
                      ("went", MID_LC), ("home", MID_LC | MID_UC), ("a", BEG_LC | MID_LC | BEG_UC), ("##number##", MID_LC | BEG_LC),
                      ("photo", BEG_UC | UNK_LC), ("x", BEG_LC), ("st", MID_UC)]:
        p.add_ortho_context(typ, flag)
    return p


ALPHABET = "aab  ..?!,;:'\"()[]{}*@-&#`_3/J"


def random_text(rng):
    if rng.random() < 0.3:          # character soup: the corners of the two regular expressions
        return "".join(rng.choice(ALPHABET) for _ in range(rng.randint(1, 24)))
    n = rng.randint(1, 9)
    out = []
    for _ in range(n):
        w = rng.choice(WORDS)
        r = rng.random()
        if r < 0.45:
            w = w + rng.choice(PUNCT)
        elif r < 0.55:
            w = rng.choice(PUNCT) + w
        elif r < 0.62:
            w = rng.choice(PUNCT)
        elif r < 0.68:
            w = w + rng.choice(PUNCT) + rng.choice(PUNCT)
        out.append(w)
        out.append(rng.choice([" ", " ", " ", "  ", ""]))
    t = "".join(out)
    if rng.random() < 0.15:
        t = " " + t
    return t


FIXED = [
    "", " ", ".", "a.", "a. b", "a. b.", "what?! yes", "a.b.c. d", "wait... what. no", "hmm.. ok. yes", "dr. smith went home. he slept",
    "(the end.) yes", "she said \"go.\" then left", "she said \"go.\"then left", "go.\"--then", "a?\"b. c", "3. cat", "3. Cat", "j. bach", "J. Bach",
    "j. Bach", "x. 3", "e.g. the cat", "e.g. The cat", "u.s. But no", "st. john. the end", "mid-st. the", "well-no. the", ". . . a", "a . . . b",
    "a. . . b", "a.) b", "a.)b", "a.') b", "a!') b", "a!'--b", "a.  b", "a. )", "a. ) b", "the end.  ", "one. two. three.", "one? two! three.",
    "no. 3", "a.m. the", "ph.d. But", "cat. dog", "home. he", "home. He", "3.5. cat", "-2. the", "1,000. the", "42nd. st. john",
]


def main():
    import sys
    # "--stress N OUT": a larger, uncommitted sample (tests/golden/check_punkt_native.sh compares the native code with it)
    stress = int(sys.argv[2]) if len(sys.argv) > 3 and sys.argv[1] == "--stress" else 0
    rng = random.Random(20260405 + stress)
    texts = list(FIXED)
    while len(texts) < (stress or 2500):
        t = random_text(rng)
        if rng.random() < 0.6:
            t = t.lower()
        texts.append(t)
    out = {"nltk": nltk.__version__, "sets": []}
    for name, params in (("empty", PunktParameters()), ("manual", manual_params())):
        tok = PunktSentenceTokenizer()
        tok._params = params
        cases = [[t, [list(s) for s in tok.span_tokenize(t)]] for t in texts]
        out["sets"].append({
            "name": name,
            "params": {"abbrev_types": sorted(params.abbrev_types), "collocations": sorted(list(c) for c in params.collocations),
                       "sent_starters": sorted(params.sent_starters),
                       "ortho_context": {k: v for k, v in sorted(params.ortho_context.items()) if v}},
            "cases": cases})
        print(name, len(cases), "texts;", sum(len(c[1]) > 1 for c in cases), "with more than one sentence")
    with open(sys.argv[3] if stress else os.path.join(HERE, "punkt_native_kat.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
