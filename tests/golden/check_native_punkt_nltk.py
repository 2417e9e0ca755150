#!/usr/bin/env python3
"""End-to-end check of the native --constrain host path against the REAL nltk objects, run where nltk is installed (the build
image: /opt/conda/bin/python3.9, nltk 3.6.5; nothing is written, the summary line is quoted in DESIGN.md 5b):

    /opt/conda/bin/python3.9 tests/golden/check_native_punkt_nltk.py [n_captions]

1. a Punkt model is TRAINED here with nltk's own PunktTrainer on a synthetic corpus (the image has no ``punkt`` / ``punkt_tab``
   download), which fills the four parameter tables the way the shipped English model does: abbreviations, collocations, sentence
   starters, orthographic contexts;
2. ``Dictionary._native_punkt`` -- the code path ``Dictionary.from_nltk`` runs -- loads that instance's tables into the native
   splitter and self-checks it against the instance;
3. random captions x random single edits over all of V: the native decision (``NativeTokenizer.constrain_mask`` ->
   ``leaf_tok_constrain_punkt``) against utils_attacks.py:110-143 evaluated with the real objects (the trained Punkt +
   ``NLTKWordTokenizer`` on every whole candidate); what the native side declines is decided by ``count_fast`` and compared too.
"""
import os
import random
import string
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np  # noqa: E402

from leaf_amd import _lib, attacks  # noqa: E402
from leaf_amd.native_text import NativeTokenizer  # noqa: E402

_lib.lib()      # before nltk: nltk imports scipy, whose conda build brings an older libstdc++ than the library was linked against

from nltk.tokenize.destructive import NLTKWordTokenizer  # noqa: E402
from nltk.tokenize.punkt import PunktSentenceTokenizer, PunktTrainer  # noqa: E402

WORDS = ["a", "photo", "of", "cat", "dog", "the", "on", "table", "red", "car", "at", "an", "is", "man", "two", "people", "in", "park",
         "sun", "set", "sunset", "with", "hat", "to", "do", "go", "no", "chair", "free", "shipping", "end", "it", "s", "new", "yes", "dr", "st",
         "smith", "john", "street", "he", "she", "we", "they", "left", "home", "inc", "co", "vs"]
VOCAB = WORDS + ["chair.", "shipping.", "dr.", "e.g.", "no.", "st.", "j.", "5", "3.50", "(new)", "what?", "wow!", "cat,", "it's", "\"go.\"", "end.)",
                 "zebra", "42", "don't", "a,b", "wait...", "yes.", "car.", "home.", "left.", "inc.", "vs.", "5.", "p.m.", "u.s."]
V = [-1] + [ord(c) for c in string.ascii_lowercase + ' ' + string.ascii_uppercase + string.digits + string.punctuation]


def corpus(rng):
    subj = ["Dr. Smith", "Mr. Jones", "Mrs. Brown", "The man", "She", "He", "They", "We", "John", "The dog", "J. Bach", "Acme Inc. staff"]
    verb = ["went home", "left at 5 p.m. today", "saw the cat", "bought a chair, e.g. a red one", "lives on Main St. near the park",
            "paid 3.50 for it", "came from the U.S. last year", "played cats vs. dogs", "took photo no. 5 of the set", "sat on the table"]
    out = []
    for _ in range(4000):
        out.append(f"{rng.choice(subj)} {rng.choice(verb)}.")
        if rng.random() < 0.3:
            out.append(rng.choice(["What?", "Wow!", "Yes.", "No."]))
    return " ".join(out)


def main():
    n_caps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = random.Random(11)
    trainer = PunktTrainer()
    trainer.INCLUDE_ALL_COLLOCS = True
    trainer.train(corpus(rng), finalize=True)
    punkt = PunktSentenceTokenizer(trainer.get_params())
    p = punkt._params
    print(f"trained Punkt: {len(p.abbrev_types)} abbreviations {sorted(p.abbrev_types)[:8]}, {len(p.collocations)} collocations, "
          f"{len(p.sent_starters)} sentence starters, {len(p.ortho_context)} orthographic contexts")
    spans = lambda t: list(punkt.span_tokenize(t))
    tb = NLTKWordTokenizer()
    word_tokenize = lambda t: [w for s in punkt.tokenize(t) for w in tb.tokenize(s)]      # nltk.word_tokenize with this model
    native = attacks.Dictionary._native_punkt(punkt, spans)
    assert native is not None, "the native splitter failed its self-check against the trained instance"
    print("strict mode after the start-up check against this nltk:", native.strict)
    D = attacks.Dictionary(WORDS, word_tokenize, kind="nltk")
    D.span_tokenize, D.punkt_native = spans, native
    tok = NativeTokenizer(n_threads=8)
    W = set(WORDS)
    count = lambda t: len(W.intersection(word_tokenize(t.lower())))
    rho = 40
    decided = declined = multi = 0
    for i in range(0, n_caps, 8):
        sents = [" ".join(rng.choice(VOCAB) for _ in range(rng.randint(3, 12))) for _ in range(8)]
        if i % 5 == 0:
            sents[0] = sents[0].title()
        z = np.stack([np.array([rng.randrange(2 * len(S) + 1) for _ in range(rho)]) for S in sents]).astype(np.int32)
        c = np.array([[rng.choice(V) for _ in range(rho)] for _ in sents], dtype=np.int32)
        valid, fb = tok.constrain_mask(D, sents, z, c)
        for b, S in enumerate(sents):
            lo = count(S)
            multi += len(spans(S.lower())) > 1
            for r in range(rho):
                cand = attacks._apply_edit(S, int(z[b, r]), int(c[b, r]))
                want = count(cand) < lo
                if fb[b, r]:
                    declined += 1
                    assert (D.count_fast(cand) < D.count_fast(S)) == want, (S, cand)
                else:
                    decided += 1
                    assert bool(valid[b, r]) == want, (S, int(z[b, r]), int(c[b, r]), cand)
    print(f"{n_caps} captions ({multi} multi-sentence) x {rho} edits: {decided} decided natively, all equal to the real nltk pipeline; "
          f"{declined} declined ({100.0 * declined / (decided + declined):.2f} %), decided by count_fast, all equal")


if __name__ == "__main__":
    main()
