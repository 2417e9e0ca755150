#!/usr/bin/env python3
"""Where one real-string search (attack_text, B captions, rho = 50, k = 1, ViT-L, native host side) spends its wall time:
host candidate preparation (mutation + BPE [+ constraint]) per stage, GPU scoring per stage (synchronised), the caption K/V
pass.  python tools/attack_profile.py [--constrain --punct]"""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import attacks
from leaf_amd.model import create_model
from leaf_amd.native_text import NativeTokenizer
from leaf_amd.train import _SYN_WORDS

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--rho", type=int, default=50)
a = ap.parse_args()
B, rho = a.batch, a.rho
rng = random.Random(0)
caps = [" ".join(rng.choice(_SYN_WORDS) for _ in range(rng.randint(4, 16))) for _ in range(B)]
m = create_model("ViT-L-14-quickgelu", seed=1)
tok = NativeTokenizer()
anchor = m.encode_text(tok.encode_batch(caps))
T = {}
def timed(name, fn, sync):
    def w(*x, **k):
        if sync: torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*x, **k)
        if sync: torch.cuda.synchronize()
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
for sync in (False, True):
    T.clear()
    o_stage, o_score, o_kv, o_enc = attacks._stage_candidates, m.score_candidates, m.encode_text_kv, tok.encode_batch
    attacks._stage_candidates = timed("host: stage candidates (mutate + BPE)", o_stage, sync)
    m.score_candidates = timed("gpu: score_candidates" + ("" if sync else " (launch only)"), o_score, sync)
    m.encode_text_kv = timed("gpu: caption K/V pass" + ("" if sync else " (launch only)"), o_kv, sync)
    tot = []
    for it in range(4):
        np.random.seed(it)
        if it == 1: T.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        attacks.attack_text(m, tok, caps, anchor, objective="l2", n=rho, k=1)
        torch.cuda.synchronize(); tot.append(time.perf_counter() - t0)
    attacks._stage_candidates, m.score_candidates, m.encode_text_kv = o_stage, o_score, o_kv
    print(("SYNCHRONISED phases" if sync else "ASYNC (as shipped)") + f": total {np.mean(tot[1:]) * 1e3:.1f} ms per search")
    for k_, v in T.items():
        print(f"   {k_:52s} {v / 3 * 1e3:7.2f} ms")
