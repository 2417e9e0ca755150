#!/usr/bin/env python3
"""Known answers for --constrain on MULTI-sentence captions, produced by the REAL nltk code path: sentence splitting by nltk's
``PunktSentenceTokenizer`` (nltk 3.6.5 under /opt/conda; no trained model is installed anywhere here, so the tokenizer runs with
its default, empty parameters -- the ALGORITHM is the real one: breaks decided from the token that carries the period and the
token after it, initials, numbers, ellipses) followed by the real ``NLTKWordTokenizer`` per sentence, i.e. ``nltk.word_tokenize``
as it runs where a model is installed, minus the learned abbreviation lists.

For every caption: its sentence spans (``span_tokenize`` of the lower-cased text) and, for random single edits (z, c) with all 96
characters of V, whether the candidate is valid under utils_attacks.py:110-143 (strictly fewer distinct dictionary words), computed
by tokenising the WHOLE candidate with Punkt re-run on it.  tests/test_constrain_native.py replays the edits through
leaf_tok_constrain_ranges with the stored spans: what it decides natively must agree.

    /opt/conda/bin/python3.9 tests/golden/make_golden_punkt.py        # writes tests/golden/punkt_kat.json
"""
import json
import os
import random
import string

import nltk
from nltk.tokenize.destructive import NLTKWordTokenizer
from nltk.tokenize.punkt import PunktSentenceTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
V = [-1] + [ord(c) for c in string.ascii_lowercase + ' ' + string.ascii_uppercase + string.digits + string.punctuation]
WORDS = ["a", "photo", "of", "cat", "dog", "the", "on", "table", "red", "car", "at", "an", "is", "man", "two", "people", "in", "park",
         "sun", "set", "sunset", "with", "hat", "to", "do", "go", "no", "chair", "free", "shipping", "end", "it", "s", "new", "yes", "dr", "st"]
VOCAB = WORDS + ["chair.", "shipping.", "dr.", "e.g.", "no.", "st.", "j.", "5", "3.50", "(new)", "what?", "wow!", "cat,", "it's", "\"go.\"", "end.)",
                 "zebra", "42", "don't", "a,b", "wait...", "yes.", "car.", "Dog.", "SALE."]


def apply_edit(S, z, c):
    if z & 1:
        i = (z - 1) // 2
        return S[:i] + S[i + 1:] if (c == -1 or S[i] == chr(c)) else S[:i] + chr(c) + S[i + 1:]
    i = z // 2
    return S if (c == -1 or c == 95) else S[:i] + chr(c) + S[i:]


def main():
    punkt, tb, W = PunktSentenceTokenizer(), NLTKWordTokenizer(), set(WORDS)
    count = lambda t: len(W.intersection(w for s in punkt.tokenize(t.lower()) for w in tb.tokenize(s)))
    rng = random.Random(3)
    cases = []
    for _ in range(160):
        cap = " ".join(rng.choice(VOCAB) for _ in range(rng.randint(3, 10)))
        lo = count(cap)
        edits = []
        for _ in range(30):
            z, c = rng.randrange(2 * len(cap) + 1), rng.choice(V)
            cand = apply_edit(cap, z, c)
            # [z, c, valid, sentence spans of the lower-cased CANDIDATE] -- the spans let the test also replay the run-time fallback
            # (Punkt per declined candidate + native count)
            edits.append([z, c, int(count(cand) < lo), [list(s) for s in punkt.span_tokenize(cand.lower())]])
        cases.append({"caption": cap, "spans": [list(s) for s in punkt.span_tokenize(cap.lower())], "edits": edits})
    # second block: the same with FILLED parameter tables (what a trained model provides: abbreviations, collocations, sentence
    # starters, orthographic contexts), for the native restatement of Punkt itself (leaf_tok_constrain_punkt): [z, c, valid] only
    from nltk.tokenize.punkt import PunktParameters
    params = PunktParameters()
    params.abbrev_types.update({"dr", "st", "e.g", "no", "j"})
    params.collocations.update({("chair", "free"), ("##number##", "cat"), ("car", "dog")})
    params.sent_starters.update({"the", "it"})
    for typ, flag in [("the", 2 | 32), ("cat", 32 | 16), ("dog", 32), ("it", 2 | 16 | 32), ("free", 32), ("a", 16 | 32 | 2), ("new", 4 | 32),
                      ("yes", 16), ("zebra", 64)]:
        params.add_ortho_context(typ, flag)
    punkt2 = PunktSentenceTokenizer()
    punkt2._params = params
    tables = {"abbrev_types": sorted(params.abbrev_types), "collocations": sorted(list(c) for c in params.collocations),
              "sent_starters": sorted(params.sent_starters), "ortho_context": {k: v for k, v in sorted(params.ortho_context.items()) if v}}
    count2 = lambda t: len(W.intersection(w for s in punkt2.tokenize(t.lower()) for w in tb.tokenize(s)))
    rng = random.Random(4)
    cases2 = []
    for _ in range(160):
        cap = " ".join(rng.choice(VOCAB) for _ in range(rng.randint(3, 10)))
        lo = count2(cap)
        edits = []
        for _ in range(30):
            z, c = rng.randrange(2 * len(cap) + 1), rng.choice(V)
            edits.append([z, c, int(count2(apply_edit(cap, z, c)) < lo)])
        cases2.append({"caption": cap, "edits": edits})
    with open(os.path.join(HERE, "punkt_kat.json"), "w") as f:
        json.dump({"source": f"nltk {nltk.__version__}: PunktSentenceTokenizer() (default parameters: no trained model available) + "
                             "NLTKWordTokenizer per sentence", "words": WORDS, "cases": cases, "tables": tables, "cases_tables": cases2}, f)
    print(len(cases2), "captions with filled tables,", sum(e[2] for c in cases2 for e in c["edits"]), "valid edits")
    print(len(cases), "captions,", sum(len(c["edits"]) for c in cases), "edits,",
          sum(len(c["spans"]) > 1 for c in cases), "multi-sentence")


if __name__ == "__main__":
    main()
