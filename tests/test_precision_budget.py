"""Where the 16-bit forward's embedding error comes from, and what a targeted spend would buy.

The oracle's operand-rounding emulation (oracle/text_oracle.py: ``RoundPolicy``) rounds every MFMA operand and every stored 16-bit
tensor of the forward to fp16 -- what the GPU path does, up to summation order -- except at the (layer, site, operand) triples a
policy exempts; an exempted site behaves like fp32 arithmetic there (a hi + lo fp16 split of that operand, or the fp32 MFMA).
Against the reference-generated ViT-L fixture (tests/golden/vitl_quickgelu.npz: open_clip's own ``encode_text`` on these weights)
that attributes the 8.8e-4 rel-L2 to layers and sites.  Findings this test pins (DESIGN.md section 7 quotes the full table, printed
by ``python tests/test_precision_budget.py``):

* weights and activations contribute about equally (6.4e-4 and 5.9e-4 alone);
* the error is made EARLY: block 0 alone accounts for a quarter of the variance... exempting it takes 8.8e-4 to 6.4e-4, block 11 to
  8.7e-4 -- a random-init residual stream grows with depth, so a late block's output is a small part of the row it is added to;
* no cheap spend exists: everything that costs <= 3 % of the step (one GEMM operand of block 0 split in two, the whole trimmed last
  block in fp32) leaves the worst row above 8.8e-4; getting the worst row under 7.5e-4 means block 0 entirely in higher precision
  (three MFMA passes per GEMM plus 32-bit q|k|v / hidden rows there: about +14 % of the search).

The GPU's measured figure for the same kind of rows is in every bench line (``parity_rel_l2``)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import text_oracle as O  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _setup():
    z = np.load(os.path.join(GOLDEN, "vitl_quickgelu.npz"))
    info = json.load(open(os.path.join(GOLDEN, "manifest.json")))["files"]["vitl_quickgelu.npz"]
    cfg = O.CONFIGS["ViT-L-14-quickgelu"]
    return cfg, O.init_weights(cfg, seed=info["weight_seed"]), z["tokens"], z["out"]


def _err(cfg, w, toks, ref, exempt):
    """(global rel-L2, worst row, median row) of the fp16-emulated forward with ``exempt(layer, site, what)`` sites exact; the final
    projection is always exact (the GPU runs it on the fp32 MFMA)."""
    out = O.encode_text(w, cfg, toks, rnd=O.RoundPolicy(O.round_fp16, lambda l, s, wh: s == "final" or exempt(l, s, wh)))
    rows = np.linalg.norm(out - ref, axis=1) / np.linalg.norm(ref, axis=1)
    return float(np.linalg.norm(out - ref) / np.linalg.norm(ref)), float(rows.max()), float(np.median(rows))


CASES = [
    ("fp16 operands everywhere (what the GPU runs)", lambda l, s, wh: False, 0.0),
    ("weights exact (activations / stored rows rounded)", lambda l, s, wh: wh == "w", None),
    ("activations and stored rows exact (weights rounded)", lambda l, s, wh: wh != "w", None),
    ("block 0 exact", lambda l, s, wh: l == 0, 14.0),
    ("blocks 0-1 exact", lambda l, s, wh: l <= 1, 28.0),
    ("block 11 exact (the trimmed last block in fp32: one row per sequence)", lambda l, s, wh: l == 11, 2.0),
    ("block 0: weights of its four GEMMs split hi + lo", lambda l, s, wh: l == 0 and wh == "w", 7.0),
    ("block 0: QKV GEMM operands exact, q|k|v still 16-bit", lambda l, s, wh: l == 0 and s == "qkv" and wh != "out", 3.0),
    ("every block: QKV weights split hi + lo", lambda l, s, wh: s == "qkv" and wh == "w", 20.0),
    ("last two blocks: A operand of out_proj / c_proj split hi + lo", lambda l, s, wh: l >= 10 and s in ("out", "proj") and wh == "a", 3.0),
]
# third column: rough cost in % of the search (an exempted GEMM operand = one more MFMA pass over that GEMM: the four GEMMs of a
# block are 1 / 12 of the search's GEMM work, which is 85 % of it; "exact" blocks also store 32-bit intermediates)


def test_error_is_made_early_and_no_cheap_spend_exists():
    cfg, w, toks, ref = _setup()
    res = {name: _err(cfg, w, toks, ref, ex) for name, ex, _ in (CASES[0], CASES[1], CASES[2], CASES[3], CASES[5], CASES[7], CASES[9])}
    base = res[CASES[0][0]]
    assert 8.0e-4 < base[0] < 9.6e-4 and base[1] < 1.0e-3, base                      # the emulation of the shipped arithmetic
    wts, act = res[CASES[2][0]][0], res[CASES[1][0]][0]                               # weights-only / activations-only error
    assert 0.8 < wts / act < 1.25 and abs((wts ** 2 + act ** 2) ** 0.5 - base[0]) < 0.1 * base[0]   # equal, independent parts
    b0, b11 = res[CASES[3][0]], res[CASES[5][0]]
    assert b0[0] < 0.78 * base[0] and b11[0] > 0.97 * base[0], (b0, b11)             # made early, not late
    for name in (CASES[5][0], CASES[7][0], CASES[9][0]):                              # the <= 3 % spends: worst row stays put
        assert res[name][1] > 8.8e-4, (name, res[name])


if __name__ == "__main__":
    cfg, w, toks, ref = _setup()
    print(f"{'emulated arithmetic (19 captions of the reference fixture, ViT-L QuickGELU)':92s} global   worst row  median   cost")
    for name, ex, cost in CASES:
        g, mx, md = _err(cfg, w, toks, ref, ex)
        print(f"{name:92s} {g:.2e} {mx:.2e}  {md:.2e} {'' if cost is None else ('+%g %%' % cost if cost else 'shipped')}")
    for L in range(cfg.layers):
        g, mx, md = _err(cfg, w, toks, ref, lambda l, s, wh, L=L: l == L)
        print(f"{'block %d exact' % L:92s} {g:.2e} {mx:.2e}  {md:.2e}")
    for site in ("qkv", "attn", "out", "fc", "proj"):
        g, mx, md = _err(cfg, w, toks, ref, lambda l, s, wh, site=site: s == site)
        print(f"{'site %s exact in every block' % site:92s} {g:.2e} {mx:.2e}  {md:.2e}")
