"""Plain PyTorch-CPU fp32 harness of the LEAF text step -- TEST / BASELINE INFRASTRUCTURE ONLY.

Used by ``bench.py``'s ``cpu_baseline`` leg (and checked against the numpy oracle in tests/test_oracle_golden.py); nothing
under ``leaf_amd/`` imports it.  It is what SURVEY.md 8d / BASELINE.md section 4 plan as the CPU line: the same ops the
reference executes through torch (``nn.Embedding`` gather, ``F.layer_norm``, packed in-proj ``F.linear``, causal
``scaled_dot_product_attention``, QuickGELU / erf-GELU MLP, EOT arg-max pooling, projection -- src/open_clip/model.py:269-284,
transformer.py:210-265 -- then the search arithmetic of utils_attacks.py:330-348, the TextFARE loss of utils_AT.py:321-322,
``backward()`` and ``torch.optim.AdamW`` with the reference's two groups), written here from the oracle's restatement, dense
77-row sequences, no work skipping, all host cores.
"""
from __future__ import annotations

import os
import time
from typing import Dict

import numpy as np


def usable_cores() -> int:
    """Cores this process may really use: scheduler affinity, capped by a cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


class TorchTextTower:
    def __init__(self, weights: Dict[str, np.ndarray], cfg, requires_grad: bool = False):
        import torch
        self.torch, self.cfg = torch, cfg
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(requires_grad) for k, v in weights.items()}

    def encode_text(self, tokens):
        torch = self.torch
        F = torch.nn.functional
        cfg, p = self.cfg, self.p
        N, L = tokens.shape
        d, H = cfg.width, cfg.heads
        x = p["token_embedding.weight"][tokens] + p["positional_embedding"][:L]
        for i in range(cfg.layers):
            q = f"transformer.resblocks.{i}."
            xn = F.layer_norm(x, (d,), p[q + "ln_1.weight"], p[q + "ln_1.bias"], cfg.eps)
            qkv = F.linear(xn, p[q + "attn.in_proj_weight"], p[q + "attn.in_proj_bias"]).view(N, L, 3, H, d // H)
            qh, kh, vh = (qkv[:, :, j].transpose(1, 2) for j in range(3))
            o = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True).transpose(1, 2).reshape(N, L, d)
            x = x + F.linear(o, p[q + "attn.out_proj.weight"], p[q + "attn.out_proj.bias"])
            xn = F.layer_norm(x, (d,), p[q + "ln_2.weight"], p[q + "ln_2.bias"], cfg.eps)
            h = F.linear(xn, p[q + "mlp.c_fc.weight"], p[q + "mlp.c_fc.bias"])
            h = h * torch.sigmoid(1.702 * h) if cfg.quick_gelu else F.gelu(h)
            x = x + F.linear(h, p[q + "mlp.c_proj.weight"], p[q + "mlp.c_proj.bias"])
        x = F.layer_norm(x, (d,), p["ln_final.weight"], p["ln_final.bias"], cfg.eps)
        return x[torch.arange(N), tokens.argmax(-1)] @ p["text_projection"]


def time_step(weights, cfg, base: np.ndarray, make_candidates, rho: int, k: int, budget_s: float = 20.0, chunk: int = 100):
    """One full step on ``base`` [B,77] (anchor forward, 2k scoring stages, training forward/backward, AdamW), fp32, dense.
    The candidate forwards are identical dense work, so when they exceed ``budget_s`` the remaining ones are extrapolated
    from the measured sequences/s.  Returns a dict with samples/s and what was measured."""
    import torch
    cores = usable_cores()
    torch.set_num_threads(cores)
    B = base.shape[0]
    frozen = TorchTextTower(weights, cfg)
    model = TorchTextTower(weights, cfg, requires_grad=True)
    tb = torch.from_numpy(base.astype(np.int64))
    t_all = time.perf_counter()
    with torch.no_grad():
        anchor = frozen.encode_text(tb)
    t_anchor = time.perf_counter() - t_all
    cur = base
    n_total = 2 * k * B * rho
    n_done, t_cand = 0, 0.0
    extrapolated = False
    for it in range(k):
        pos = None
        for stage in range(2):
            cand = make_candidates(cur, rho, 100 + 2 * it + stage, pos)        # [B,rho,77]
            flat = torch.from_numpy(cand.reshape(B * rho, -1).astype(np.int64))
            feats = []
            t0 = time.perf_counter()
            with torch.no_grad():
                for s in range(0, B * rho, chunk):
                    if t_cand + (time.perf_counter() - t0) > budget_s and n_done + s >= 2 * chunk:
                        extrapolated = True
                        break
                    feats.append(model.encode_text(flat[s:s + chunk]))
            t_cand += time.perf_counter() - t0
            got = sum(f.shape[0] for f in feats)
            n_done += got
            if extrapolated:
                break
            f = torch.cat(feats).view(B, rho, -1)
            loss = ((f - anchor[:, None, :]) ** 2).sum(-1)
            idx = loss.argmax(-1).numpy()
            if stage == 0:
                win = cand[np.arange(B), idx]
                pos = np.array([int(np.nonzero(win[b] != cur[b])[0][0]) if np.any(win[b] != cur[b]) else 1 for b in range(B)])
            else:
                cur = cand[np.arange(B), idx]
        if extrapolated:
            break
    seq_per_s = n_done / t_cand
    t_search = n_total / seq_per_s
    t0 = time.perf_counter()
    feat = model.encode_text(torch.from_numpy(cur.astype(np.int64)))
    loss = torch.nn.functional.mse_loss(anchor, feat, reduction="none").sum(-1).mean()
    loss.backward()
    excl = lambda n, q: q.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n
    opt = torch.optim.AdamW([{"params": [q for n, q in model.p.items() if excl(n, q)], "weight_decay": 0.0},
                             {"params": [q for n, q in model.p.items() if not excl(n, q)], "weight_decay": 1e-4}], lr=1e-5)
    opt.step()
    t_train = time.perf_counter() - t0
    total = t_anchor + t_search + t_train
    return {"samples_per_s": B / total, "cores": cores, "seq_per_s": seq_per_s, "measured_candidate_forwards": int(n_done),
            "total_candidate_forwards": int(n_total), "extrapolated": extrapolated, "t_anchor_s": t_anchor,
            "t_search_s": t_search, "t_train_s": t_train, "wall_s": time.perf_counter() - t_all, "loss": float(loss.detach())}
