#!/usr/bin/env python3
"""Validation of leaf_amd.treebank.punkt_free against the REAL Punkt code (nltk 3.6.5 under /opt/conda, default parameters: no trained
model is installed here; a trained model only REMOVES breaks, at abbreviations): for 60,000 random punctuation-heavy strings, every
string the rule calls sentence-boundary independent must give the same letter-bearing tokens with and without the Punkt step in front
of NLTKWordTokenizer.  (Quote tokens may differ -- a '?' / '!' sentence end in front of a '"' turns a closing quote into an opening one
-- they are never dictionary words.)  Not part of the test suite (the system interpreter has no nltk):

    /opt/conda/bin/python3.9 tests/golden/check_punkt_free.py         # -> "mismatches: 0"
"""
import sys, random, string
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nltk.tokenize.punkt import PunktSentenceTokenizer
from nltk.tokenize.destructive import NLTKWordTokenizer
from leaf_amd.treebank import punkt_free
p, tb = PunktSentenceTokenizer(), NLTKWordTokenizer()
rng = random.Random(0)
alphabet = "ab c1.,.?!)(\"';:*@[]{}<>-/%&$#  " + ".."
bad = free = 0
for _ in range(60000):
    t = "".join(rng.choice(alphabet) for _ in range(rng.randint(1, 14)))
    if punkt_free(t):
        free += 1
        a = [w for s in p.tokenize(t) for w in tb.tokenize(s)]
        b = tb.tokenize(t)
        a = [w for w in a if any(ch.isalnum() for ch in w)]; b = [w for w in b if any(ch.isalnum() for ch in w)]
        if a != b:
            bad += 1
            if bad < 10: print(repr(t), a, b)
print("punkt_free strings:", free, "mismatches:", bad)
