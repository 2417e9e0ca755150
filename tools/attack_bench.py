#!/usr/bin/env python3
"""Throughput of the REAL (string) search: attack_text on B captions, rho=50, k=1, ViT-L, native vs Python host side,
unconstrained and with --constrain (dictionary = the caption vocabulary + filler words written to a word-list file).
--tokenizer regex: the stand-in word tokenizer; --tokenizer treebank: nltk.word_tokenize's Treebank pipeline (leaf_amd/treebank.py
as the Python side -- nltk itself is not installed for the system interpreter -- and its window-local native restatement),
on captions WITH punctuation (--punct: commas, clitics, quotes, a final period), the case every launch script runs
(--constrain, scripts/train_leaf_vit*.sh)."""
import argparse
import os
import random
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import attacks
from leaf_amd.model import create_model
from leaf_amd.native_text import NativeTokenizer
from leaf_amd.tokenizer import SimpleTokenizer
from leaf_amd.train import _SYN_WORDS

ap = argparse.ArgumentParser()
ap.add_argument("--constrain", action="store_true", help="also time the constrained search")
ap.add_argument("--dict-words", type=int, default=236736, help="size of the word list (nltk's `words` corpus has 236,736 entries)")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--rho", type=int, default=50)
ap.add_argument("--tokenizer", default="regex", choices=["regex", "treebank"])
ap.add_argument("--punct", action="store_true", help="captions with punctuation (commas, 's, quotes, brackets, a final period)")
ap.add_argument("--sentences", type=float, default=0.0,
                help="with --punct: share of captions that get a period in the middle (two sentences, 'vintage chair. free shipping.')")
ap.add_argument("--punkt", choices=["none", "standin", "native"], default="none",
                help="with --tokenizer treebank: sentence splitter in front of the Treebank step. standin = a Python splitter with "
                     "Punkt's cost and kind of dependence (spans asked per caption / per declined candidate, as with an installed nltk "
                     "whose tables failed the native self-check); native = Punkt's algorithm in C++ over a small table set")
ap.add_argument("--no-dedupe", action="store_true")
ap.add_argument("--pipeline", type=int, default=None, help="caption groups interleaved in the search (default: 2 for B >= 64)")
a = ap.parse_args()

B, rho = a.batch, a.rho
rng = random.Random(0)
caps = [" ".join(rng.choice(_SYN_WORDS) for _ in range(rng.randint(4, 16))) for _ in range(B)]
if a.punct:
    def deco(c):
        w = c.split()
        for _ in range(rng.randint(1, 3)):
            i = rng.randrange(len(w))
            w[i] = rng.choice([w[i] + ",", w[i] + "'s", '"' + w[i] + '"', "(" + w[i] + ")", w[i] + "!", w[i] + ":", "don't " + w[i]])
        if len(w) > 3 and rng.random() < a.sentences:
            w[rng.randrange(1, len(w) - 1)] += "."
        return " ".join(w) + rng.choice([".", "", ".", "!"])
    caps = [deco(c) for c in caps]
m = create_model("ViT-L-14-quickgelu", seed=1)
modes = [False, True] if a.constrain else [False]
if a.constrain:
    filler = set(_SYN_WORDS)
    while len(filler) < a.dict_words:
        filler.add("".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(2, 10))))
    path = os.path.join(tempfile.gettempdir(), "leaf_words.txt")
    with open(path, "w") as f:
        f.write("\n".join(sorted(filler)))
    if a.punkt == "none":
        attacks.set_dictionary(attacks.Dictionary.from_file(path, tokenizer=a.tokenizer))
    else:
        import json
        tables = {"abbrev_types": ["dr", "mr", "mrs", "st", "e.g", "i.e", "vs", "inc", "co", "no", "p.m", "a.m"],
                  "collocations": [["##number##", "street"]], "sent_starters": ["the", "a"], "ortho_context": {"the": 34, "a": 50}}
        ppath = os.path.join(tempfile.gettempdir(), "leaf_punkt.json")
        with open(ppath, "w") as f:
            json.dump(tables, f)
        D = attacks.Dictionary.from_file(path, tokenizer="treebank", punkt_params=ppath)
        if a.punkt == "standin":
            # the round-3 path before the native splitter: spans come from a Python call per multi-sentence caption and per declined
            # candidate (here a ctypes call of ~5 us stands in for nltk's PunktSentenceTokenizer.span_tokenize, ~25 us in CPython: a
            # LOWER bound of that path's host time)
            from leaf_amd.native_text import NativePunkt
            free = NativePunkt.from_json(ppath, strict=False)

            def spans(t):
                sp = free.spans(t)
                return sp if sp is not None else [(0, len(t))]
            D.span_tokenize, D.punkt_native = spans, None
        attacks.set_dictionary(D)
for constrain in modes:
    for name, tok in (("python", SimpleTokenizer()), ("native", NativeTokenizer())):
        anchor = m.encode_text(tok.encode_batch(caps))
        ts = []
        for it in range(4):
            np.random.seed(it)
            torch.cuda.synchronize()
            t0 = time.time()
            rows0 = m.rows_scored
            feats, adv = attacks.attack_text_leaf(m, tok, caps, anchor, objective="l2", n=rho, k=1, constrain=constrain,
                                                  dedupe=not a.no_dedupe, pipeline=a.pipeline)
            rows = m.rows_scored - rows0
            torch.cuda.synchronize()
            ts.append(time.time() - t0)
        dt = min(ts[1:])
        changed = sum(x != y for x, y in zip(adv, caps))
        print(f"{name:6s} constrain={int(constrain)} tokenizer={a.tokenizer} punct={int(a.punct)} sentences={a.sentences} punkt={a.punkt} dedupe={int(not a.no_dedupe)} pipeline={a.pipeline}: attack_text B={B} "
              f"rho={rho} k=1: {dt * 1e3:.1f} ms -> {B / dt:.0f} captions/s (search only; {changed}/{B} captions changed; {rows} rows scored)",
              flush=True)
