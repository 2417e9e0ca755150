#!/usr/bin/env python3
"""Which arithmetic should the frozen model's ANCHOR pass (utils_AT.py:296) run in -- the 16-bit arithmetic the candidates are scored
in ('matched', what the reference does: one arithmetic for both sides, utils_attacks.py:330-348) or the fp32-grade forward of
precise.hip?  Test infrastructure (oracle/ is only the checker).

On the first stage of one BASELINE.json configs[1] search (benchmark model, B = 128 captions, rho = 50 single-edit candidates: the rows
of tests/row_error_census.py, whose PyTorch-CPU fp32 embeddings are re-used when --ref-dir has them) the search's decision is
arg-max_r ||f(cand_r) - anchor||^2.  Against the fp32 decision this prints, for both anchors: arg-max agreement, the worst and mean
REGRET (fp32 loss of the engine's pick / fp32 best), and the error of the loss values themselves.

    python tests/anchor_arith_study.py [--ref-dir tools/diag/census_ref]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import text_oracle as O  # noqa: E402
import row_error_census as RC  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref-dir", default=None)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rho", type=int, default=50)
    a = ap.parse_args()
    import torch
    from leaf_amd.model import create_model
    name = "ViT-L-14-quickgelu"
    B, rho = a.batch, a.rho
    base = O.synthetic_tokens(B, seed=1234)
    c1 = O.synthetic_candidates(base, rho, seed=1235).reshape(-1, 77)
    toks = np.concatenate([base, c1])
    ref = None
    if a.ref_dir and os.path.exists(os.path.join(a.ref_dir, "ref_L.npy")):
        r = np.load(os.path.join(a.ref_dir, "ref_L.npy"))
        if r.shape[0] >= toks.shape[0]:
            ref = r[:toks.shape[0]]                      # census rows: captions, then the stage-1 candidates (same seeds)
    if ref is None:
        cfg = O.CONFIGS[name]
        ref = RC.cpu_reference(O.init_weights(cfg, seed=1), cfg, toks, log=print)
    ref_a, ref_c = ref[:B].astype(np.float64), ref[B:].astype(np.float64).reshape(B, rho, -1)
    loss_o = ((ref_c - ref_a[:, None, :]) ** 2).sum(-1)
    pick_o = loss_o.argmax(-1)
    for mode in ("rowsafe", "fast"):
        m = create_model(name, seed=1)
        m.set_precision(mode)
        f_c = m.encode_text(c1).cpu().numpy().astype(np.float64).reshape(B, rho, -1)
        anchors = {"matched (16-bit, the arithmetic of the candidates)": m.encode_text(base).cpu().numpy().astype(np.float64),
                   "precise (fp32-grade)": m.encode_text(base, precise=True).cpu().numpy().astype(np.float64)}
        print(f"== candidates scored in '{mode}' arithmetic; {B} decisions over {rho} candidates each")
        for label, an in anchors.items():
            loss = ((f_c - an[:, None, :]) ** 2).sum(-1)
            pick = loss.argmax(-1)
            regret = loss_o[np.arange(B), pick] / loss_o.max(-1)
            rel = np.abs(loss - loss_o) / loss_o.max(-1, keepdims=True)
            print(f"   anchor {label:52s}: arg-max agrees {int((pick == pick_o).sum())}/{B}   regret worst {regret.min():.5f} mean {regret.mean():.6f}   "
                  f"|loss - fp32| / max loss: median {np.median(rel):.2e} max {rel.max():.2e}")
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
