#!/usr/bin/env python3
"""Throughput of the REAL (string) search: attack_text on B captions, rho=50, k=1, ViT-L, native vs Python host side."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import attacks
from leaf_amd.model import create_model
from leaf_amd.native_text import NativeTokenizer
from leaf_amd.tokenizer import SimpleTokenizer
from leaf_amd.train import _SYN_WORDS
import random

B, rho = 128, 50
rng = random.Random(0)
caps = [" ".join(rng.choice(_SYN_WORDS) for _ in range(rng.randint(4, 16))) for _ in range(B)]
m = create_model("ViT-L-14-quickgelu", seed=1)
for name, tok in (("python", SimpleTokenizer()), ("native", NativeTokenizer())):
    anchor = m.encode_text(tok.encode_batch(caps))
    for it in range(3):
        np.random.seed(it)
        torch.cuda.synchronize()
        t0 = time.time()
        feats, adv = attacks.attack_text(m, tok, caps, anchor, objective="l2", n=rho, k=1)
        torch.cuda.synchronize()
        dt = time.time() - t0
    print(f"{name}: attack_text B={B} rho={rho} k=1: {dt*1e3:.1f} ms  -> {B/dt:.0f} captions/s (search only)", flush=True)
