#!/usr/bin/env python3
"""Per-row embedding error of the 16-bit forward on the BASELINE.json towers (north_star: rel-L2 <= 1e-3 per row): the two reference-generated
ViT-L fixtures, and N synthetic captions per tower against the fp32 oracle (test infrastructure: oracle/ is only the checker here).
  python tests/row_error_survey.py [N]

Round 6: SUPERSEDED by tests/row_error_census.py (12,928 rows of one full search + 2,000 rows per other tower against the PyTorch-CPU fp32
forward, per arithmetic variant) -- a 24-row maximum says little about a tight distribution's tail; kept as the quick look it is."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: only tests may import oracle/)
sys.path.insert(0, ROOT)
from oracle import text_oracle as O  # noqa: E402
from leaf_amd.model import create_model  # noqa: E402


def rows(a, b):
    return np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    g = os.path.join(ROOT, "tests", "golden")
    for f, name in (("vitl_gelu", "ViT-L-14"), ("vitl_quickgelu", "ViT-L-14-quickgelu")):
        z = np.load(os.path.join(g, f + ".npz"))
        m = create_model(name, seed=1)
        r = rows(m.encode_text(z["tokens"]).cpu().numpy(), z["out"])
        print(f"{name:22s} reference fixture, {len(r)} captions: row max {r.max():.3e} median {np.median(r):.3e} rows > 1e-3: {(r > 1e-3).sum()}", flush=True)
    for name, seed in (("ViT-L-14-quickgelu", 1), ("ViT-L-14", 1), ("ViT-H-14", 2), ("ViT-bigG-14", 2)):
        cfg = O.CONFIGS[name]
        w = O.init_weights(cfg, seed=seed)
        m = create_model(name, seed=seed)
        toks = O.synthetic_tokens(n, seed=11, min_len=5, max_len=60)
        r = rows(m.encode_text(toks).cpu().numpy(), O.encode_text(w, cfg, toks))
        print(f"{name:22s} oracle, {n} synthetic captions (5-60 tokens): row max {r.max():.3e} median {np.median(r):.3e} rows > 1e-3: {(r > 1e-3).sum()}", flush=True)


if __name__ == "__main__":
    main()
