#!/usr/bin/env python3
"""How long does the GPU wait for the host at the stage boundary of the search?  Event e1 is queued behind the first stage's last launch
(before the host waits for the winners), e2 right in front of the second stage's C call: elapsed(e1, e2) is device time with nothing to
run (the queue is empty when e2 arrives).  Also the host's own wall time for the pieces in between.
  python tools/stage_hole_probe.py [model] [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import step as S
from leaf_amd.model import create_model, get_config

name = sys.argv[1] if len(sys.argv) > 1 else "ViT-L-14-quickgelu"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda", 0)
cfg = get_config(name)
model = create_model(name, device=dev, dtype="fp16", seed=1, trainable=True)
model.pack()
rho = 50
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
ev, host = [], {}
orig_dev = S.SyntheticCandidates.stage2_device
def dev2(self, cur, best1):
    r = orig_dev(self, cur, best1)
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append([e, None]); host["t_sync0"] = time.perf_counter()
    return r
S.SyntheticCandidates.stage2_device = dev2
orig_pos = S.SyntheticCandidates.stage2_positions
def pos2(self, pos, b1):
    host.setdefault("wait for winners", []).append(time.perf_counter() - host["t_sync0"]); host["t_after_sync"] = time.perf_counter()
    return orig_pos(self, pos, b1)
S.SyntheticCandidates.stage2_positions = pos2
lib = model._lib
orig_c = lib.leaf_score_candidates_prefix
class Wrap:
    def __call__(self, *a):
        if ev and ev[-1][1] is None:
            host.setdefault("winners -> C call", []).append(time.perf_counter() - host["t_after_sync"])
            e = torch.cuda.Event(enable_timing=True); e.record(); ev[-1][1] = e
            t0 = time.perf_counter(); r = orig_c(*a); host.setdefault("C call (plan + all launches)", []).append(time.perf_counter() - t0)
            return r
        return orig_c(*a)
class LibProxy:
    def __init__(self, l): object.__setattr__(self, "_l", l)
    def __getattr__(self, k): return Wrap() if k == "leaf_score_candidates_prefix" else getattr(self._l, k)
model._lib = LibProxy(lib)
for it in range(14):
    S.search_synthetic(model, anchor, base, sc, it, base_lens=base_lens)
torch.cuda.synchronize()
holes = np.array([a.elapsed_time(b) * 1e3 for a, b in ev[4:]])
print(f"{name} B={B}: GPU idle at the stage boundary: median {np.median(holes):.0f} us, min {holes.min():.0f}, max {holes.max():.0f}")
for k in ("wait for winners", "winners -> C call", "C call (plan + all launches)"):
    v = np.array(host[k][4:]) * 1e6
    print(f"   host: {k:30s} median {np.median(v):8.0f} us")
