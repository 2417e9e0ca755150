#!/usr/bin/env python3
"""Census of the per-row embedding error of the forward-only arithmetic (SURVEY 8d gate P1; the reference seam is
``CLIP.encode_text``, src/open_clip/model.py:269-284) -- test infrastructure: oracle/ is only the checker here.

* ViT-L-14-quickgelu, the benchmark model (seed 1): ONE full BASELINE.json configs[1] search = B = 128 synthetic captions, rho = 50,
  both stages (stage 2 edits the position the GPU's own stage-1 arg-max picked): 12,800 candidate rows + the 128 captions;
* ViT-L-14 (erf GELU), ViT-H-14, ViT-bigG-14: N rows each (captions + single-edit candidates of them).

Every row is compared with the plain PyTorch CPU fp32 forward of the same weights (oracle/torch_cpu_harness.py, sequences cut after the
batch's longest EOT: causal attention makes that exact).  For each arithmetic variant (residual stream 16 + 8 bits / fp32 rows,
split_blocks 0 / 1 / ..., the fp32-grade ``precise`` mode) it prints the histogram, P50 / P99 / P99.9 / max and the number of rows
above 1e-3, and stores the per-row errors (npz) beside the text.

    python tests/row_error_census.py [--rows-other 2000] [--out gpurun_out/census] [--towers L,Lerf,H,G] [--modes ...]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: only tests may import oracle/)
sys.path.insert(0, ROOT)
from oracle import text_oracle as O  # noqa: E402
from oracle import torch_cpu_harness as H  # noqa: E402


def cpu_reference(w, cfg, toks, chunk=128, log=None):
    import torch
    torch.set_num_threads(H.usable_cores())
    tower = H.TorchTextTower(w, cfg)
    L = int(toks.argmax(-1).max()) + 1
    t = torch.from_numpy(np.ascontiguousarray(toks[:, :L]).astype(np.int64))
    out = []
    t0 = time.time()
    with torch.no_grad():
        for s in range(0, t.shape[0], chunk):
            out.append(tower.encode_text(t[s:s + chunk]).numpy())
            if log and (s // chunk) % 10 == 0:
                log(f"    cpu reference {s + chunk}/{t.shape[0]} rows, {time.time() - t0:.0f} s")
    return np.concatenate(out)


def row_err(got, ref):
    got = got.astype(np.float64)
    ref = ref.astype(np.float64)
    return np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)


def summarize(r):
    q = lambda p: float(np.quantile(r, p))
    return {"rows": int(r.size), "p50": q(0.5), "p90": q(0.9), "p99": q(0.99), "p999": q(0.999), "max": float(r.max()), "min": float(r.min()),
            "mean": float(r.mean()), "std": float(r.std()), "rows_above_1e-3": int((r > 1e-3).sum()), "rows_above_9.5e-4": int((r > 9.5e-4).sum())}


def histogram(r, lo=None, hi=None, bins=20):
    lo = float(r.min()) if lo is None else lo
    hi = float(r.max()) if hi is None else hi
    if hi <= lo:
        hi = lo * 1.0001 + 1e-12
    h, e = np.histogram(r, bins=bins, range=(lo, hi))
    width = max(int(h.max()), 1)
    return "\n".join(f"      [{e[i]:.3e}, {e[i + 1]:.3e})  {h[i]:6d}  {'#' * int(40 * h[i] / width)}" for i in range(bins))


MODES = {
    # label: (compact_resid, split masks per leading block [bit 0 QKV, 1 out_proj w, 2 c_fc, 3 c_proj w], precise)
    "rowsafe": (1, None, False),                 # the DEFAULT (leaf_amd.model.PRECISION_MODES), 16 + 8-bit stream
    "rowsafe fp32 resid": (0, None, False),
    "fast": (1, (), False),                      # rounds 1-5: no split GEMM, 16 + 8-bit stream
    "fast fp32 resid": (0, (), False),
    "split1": (1, (15,), False),                 # all four GEMMs of block 0
    "split2": (1, (15, 15), False),
    "precise": (0, (), True),
}


def mode_of(label):
    """a MODES name, or 'm' + dash-separated masks (m3-1 = QKV + out_proj of block 0, QKV of block 1), optional suffix '/f32'"""
    if label in MODES:
        return MODES[label]
    body, _, suf = label.partition("/")
    return (0 if suf == "f32" else 1, tuple(int(x) for x in body[1:].split("-")), False)


def gpu_rows(m, toks, mode, chunk=3200):
    import torch
    from leaf_amd.model import PRECISION_MODES
    compact, masks, precise = mode_of(mode)
    m.set_option("compact_resid", compact)
    m.set_split_masks((PRECISION_MODES["rowsafe"] if masks is None else masks)[:m.cfg.layers - 1])
    outs = []
    for s in range(0, toks.shape[0], chunk):
        t = toks[s:s + chunk]
        outs.append(m.encode_text(t, precise=precise).cpu().numpy())
    torch.cuda.synchronize()
    m.set_option("compact_resid", 1)
    m.set_precision("rowsafe")
    return np.concatenate(outs)


def search_rows(m, B, rho, seed):
    """captions + both stages' candidates of one configs[1] search (stage 2 at the GPU's stage-1 winners' positions)"""
    base = O.synthetic_tokens(B, seed=seed)
    c1 = O.synthetic_candidates(base, rho, seed=seed + 1)
    anchor = m.encode_text(base)
    idx, _ = m.score_candidates(c1.reshape(-1, 77), anchor, rho, "l2")
    win = c1[np.arange(B), idx.cpu().numpy()]
    pos = np.array([int(np.nonzero(win[b] != base[b])[0][0]) if np.any(win[b] != base[b]) else 1 for b in range(B)])
    c2 = O.synthetic_candidates(win, rho, seed=seed + 2, fixed_pos=pos)
    return np.concatenate([base, c1.reshape(-1, 77), c2.reshape(-1, 77)])


def other_rows(n, seed):
    """n rows: captions of 5..60 tokens and single-edit candidates of them (9 per caption)"""
    nb = (n + 9) // 10
    base = O.synthetic_tokens(nb, seed=seed, min_len=5, max_len=60)
    c = O.synthetic_candidates(base, 9, seed=seed + 1)
    return np.concatenate([base, c.reshape(-1, 77)])[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-other", type=int, default=2000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "census"))
    ap.add_argument("--towers", default="L,Lerf,H,G")
    ap.add_argument("--modes", default="rowsafe,fast,rowsafe fp32 resid,precise")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rho", type=int, default=50)
    ap.add_argument("--weight-seed", type=int, default=None, help="override the towers' weight seeds (default: 1 / 1 / 2 / 2, the benchmark's and the fixtures')")
    ap.add_argument("--caption-seed", type=int, default=1234, help="seed of the synthetic captions / candidates")
    ap.add_argument("--search-towers", default="L", help="towers (comma list) whose rows are one full search instead of --rows-other rows")
    ap.add_argument("--ref-dir", default=None, help="re-use ref_<tower>.npy of an earlier census of the same rows (saves the CPU minutes)")
    args = ap.parse_args()
    from leaf_amd.model import create_model
    os.makedirs(args.out, exist_ok=True)
    txt = open(os.path.join(args.out, "row_error_census.txt"), "a")

    def log(s=""):
        print(s, flush=True)
        txt.write(s + "\n")
        txt.flush()

    towers = {"L": ("ViT-L-14-quickgelu", 1, "search"), "Lerf": ("ViT-L-14", 1, "other"), "H": ("ViT-H-14", 2, "other"),
              "G": ("ViT-bigG-14", 2, "other")}
    modes = [s for s in args.modes.split(",") if s]
    summary = {}
    for key in args.towers.split(","):
        name, seed, kind = towers[key]
        kind = "search" if key in args.search_towers.split(",") else "other"
        seed = seed if args.weight_seed is None else args.weight_seed
        cfg = O.CONFIGS[name]
        w = O.init_weights(cfg, seed=seed)
        m = create_model(name, seed=seed)
        m.set_precision("fast")        # the row set is the one of profiles/r06_row_error_census.txt (stage 2 follows the round-5 arithmetic's winners)
        toks = search_rows(m, args.batch, args.rho, args.caption_seed) if kind == "search" else other_rows(args.rows_other, 11 + args.caption_seed - 1234)
        log(f"== {name} (random init, seed {seed}): {toks.shape[0]} rows"
            + (f" = one configs[1] search: {args.batch} captions + 2 x {args.batch} x {args.rho} candidates" if kind == "search" else
               " (captions of 5..60 tokens + single-edit candidates)"))
        t0 = time.time()
        ref_path = os.path.join(args.ref_dir or args.out, f"ref_{key}.npy")
        if args.ref_dir and os.path.exists(ref_path) and np.load(ref_path).shape[0] == toks.shape[0]:
            ref = np.load(ref_path)
            log(f"   PyTorch CPU fp32 reference: re-used from {ref_path} (same seeded rows)")
        else:
            ref = cpu_reference(w, cfg, toks, log=log)
            log(f"   PyTorch CPU fp32 reference: {time.time() - t0:.0f} s on {H.usable_cores()} cores")
            np.save(os.path.join(args.out, f"ref_{key}.npy"), ref)
        for mode in modes:
            got = gpu_rows(m, toks, mode)
            r = row_err(got, ref)
            s = summarize(r)
            s["global_rel_l2"] = float(np.linalg.norm(got.astype(np.float64) - ref) / np.linalg.norm(ref.astype(np.float64)))
            summary[f"{name}|{mode}"] = s
            np.save(os.path.join(args.out, f"rows_{key}_{mode.replace(' ', '_')}.npy"), r.astype(np.float32))
            log(f"   -- {mode}: rows {s['rows']}  global {s['global_rel_l2']:.3e}  P50 {s['p50']:.3e}  P99 {s['p99']:.3e}  P99.9 {s['p999']:.3e}  "
                f"max {s['max']:.3e}  rows > 1e-3: {s['rows_above_1e-3']}  rows > 9.5e-4: {s['rows_above_9.5e-4']}")
            log(histogram(r))
        del m
    with open(os.path.join(args.out, "row_error_census.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
