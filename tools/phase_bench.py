"""Per-phase GPU time of one LEAF outer step (HIP events on the current stream; synchronised between phases, so the
sum is slightly above bench.py's step time).  python tools/phase_bench.py [--model ... --batch 128 --rho 50]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from leaf_amd.model import LeafCLIPText, create_model, get_config  # noqa: E402
from leaf_amd.step import StepConfig, search_synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-L-14-quickgelu")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rho", type=int, default=50)
    ap.add_argument("--k-adv", type=int, default=1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=0, help="sequences per pass (row budget = chunk * 77); 0 = engine default")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = get_config(a.model)
    model = create_model(a.model, device=dev, dtype="fp16", seed=1, trainable=True)
    frozen = LeafCLIPText(cfg, device=dev, dtype="fp16").copy_from(model)
    frozen.pack(); model.pack()
    model.set_option('streams', a.streams); frozen.set_option('streams', a.streams)
    if a.chunk:
        model.set_option('chunk', a.chunk)
    sc = StepConfig(rho=a.rho, k_adv=a.k_adv)
    g = torch.Generator().manual_seed(1234)
    B = a.batch
    base = torch.zeros(B, cfg.context_length, dtype=torch.int32)
    lens = torch.randint(8, 41, (B,), generator=g)
    for i in range(B):
        n = int(lens[i])
        base[i, 0] = cfg.vocab_size - 2
        base[i, 1:1 + n] = torch.randint(1, cfg.vocab_size - 2, (n,), generator=g, dtype=torch.int32)
        base[i, 1 + n] = cfg.vocab_size - 1
    base_lens = lens.numpy().astype(np.int32) + 2
    base = base.to(dev)
    acc = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        acc.setdefault(name, []).append(e0.elapsed_time(e1))
        return r

    for it in range(a.iters + 1):
        model.eval()
        anchor = timed("anchor_fwd", lambda: frozen.encode_text(base, seq_lens=base_lens))
        adv = timed("search", lambda: search_synthetic(model, anchor, base, sc, it, base_lens=base_lens))
        model.train()
        feat = timed("train_fwd", lambda: model.forward_train(adv, seq_lens=base_lens))
        model.zero_grad()
        timed("backward", lambda: model.backward(feat, anchor))
        timed("adamw+pack", lambda: (model.adamw_step(1e-5, (0.9, 0.999), 1e-8, 1e-4), model.pack()))
    tot = 0.0
    for k, v in acc.items():
        m = float(np.mean(v[1:]))
        tot += m
        print(f"{k:12s} {m:8.3f} ms")
    print(f"{'sum':12s} {tot:8.3f} ms")


if __name__ == "__main__":
    main()
