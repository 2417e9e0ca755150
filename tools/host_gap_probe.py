#!/usr/bin/env python3
"""Host time on the search's critical path: from the moment the stage-1 winners arrive on the host to the moment the stage-2
scoring call has been queued (the GPU idles meanwhile)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import step as S
from leaf_amd.model import LeafCLIPText, create_model, get_config

dev = torch.device("cuda", 0)
cfg = get_config("ViT-L-14-quickgelu")
model = create_model("ViT-L-14-quickgelu", device=dev, dtype="fp16", seed=1, trainable=True)
model.pack()
B, rho = 128, 50
g = torch.Generator().manual_seed(1)
base = torch.zeros(B, 77, dtype=torch.int32)
lens = torch.randint(8, 41, (B,), generator=g)
for i in range(B):
    n = int(lens[i]); base[i, 0] = cfg.vocab_size - 2
    base[i, 1:1 + n] = torch.randint(1, cfg.vocab_size - 2, (n,), generator=g, dtype=torch.int32); base[i, 1 + n] = cfg.vocab_size - 1
base_lens = lens.numpy().astype(np.int32) + 2
base = base.to(dev)
anchor = model.encode_text(base, seq_lens=base_lens)
sc = S.StepConfig(rho=rho, k_adv=1)
T = {}
orig = model._score_prefix
def timed_prefix(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); T.setdefault("score_prefix (plan + launch)", []).append(time.perf_counter() - t0); return r
model._score_prefix = timed_prefix
orig_pos = S.SyntheticCandidates.stage2_positions
def timed_pos(self, pos, b1):
    t0 = time.perf_counter(); r = orig_pos(self, pos, b1); T.setdefault("stage2_positions", []).append(time.perf_counter() - t0); return r
S.SyntheticCandidates.stage2_positions = timed_pos
for it in range(12):
    t0 = time.perf_counter()
    S.search_synthetic(model, anchor, base, sc, it, base_lens=base_lens)
    torch.cuda.synchronize()
    T.setdefault("search total", []).append(time.perf_counter() - t0)
for k, v in T.items():
    v = np.array(v[2 * (len(v) // 12):]) * 1e6
    print(f"{k:32s} median {np.median(v):8.1f} us  min {v.min():8.1f}")
# breakdown of _score_prefix
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(5): S.search_synthetic(model, anchor, base, sc, it, base_lens=base_lens)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(18)
