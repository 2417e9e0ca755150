#!/usr/bin/env python3
"""Golden fixture for the OPTIONAL embedding-space PGD mode, produced by the REFERENCE's own code.

No reference path runs this attack on text, so the fixture composes the reference pieces the row cites:
  * the reference ``CLIP.encode_text`` (src/open_clip/model.py:269-284) with an additive perturbation delta injected at
    the output of ``token_embedding`` through a forward hook (the embedding-input form of src/pez/open_clip_pez/model.py:210-228),
  * the 'l2' objective ``((anchor - f)**2).sum()`` of the continuous attack loop (utils_attacks.py:680-697),
  * the update ``delta <- project_perturbation(delta + alpha * normalize_grad(grad, norm), eps, norm)`` with both functions
    imported from src/robust_vlm/train/utils.py:96-114 (linf: sign / clamp; l2: per-sample F.normalize / torch.renorm).
Runs only in the build container.   python tests/golden/make_golden_pgd.py  ->  tests/golden/tiny_pgd.npz
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stubs, TINY config, weight loading)


def main():
    MG.install_stubs()
    import importlib.machinery
    import types
    wb = types.ModuleType("wandb")            # absent here; only init_wandb (unused) touches it
    wb.__spec__ = importlib.machinery.ModuleSpec("wandb", None)
    wb.init = MagicMock()
    sys.modules["wandb"] = wb
    from open_clip.model import CLIP
    import robust_vlm.train.utils as RU
    from robust_vlm.train.utils import normalize_grad, project_perturbation
    if not hasattr(RU, "F"):   # the module calls F.normalize (:113) without importing torch.nn.functional as F
        RU.F = torch.nn.functional
    from oracle import text_oracle as O

    torch.set_num_threads(8)
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=12)
    model = CLIP(**MG.TINY, quick_gelu=True).float().eval()
    MG.load_np_state(model, w)
    for p in model.parameters():
        p.requires_grad = False
    toks = O.synthetic_tokens(6, seed=21, min_len=3, max_len=30)
    text = torch.from_numpy(toks.astype(np.int64))
    rng = np.random.default_rng(22)
    with torch.no_grad():
        clean = model.encode_text(text).numpy()
    anchor = (clean + 0.5 * rng.standard_normal(clean.shape)).astype(np.float32)
    eot = toks.argmax(-1)
    keep = (np.arange(77)[None, :] <= eot[:, None])[:, :, None]          # positions after EOT never reach the output
    out = dict(tokens=toks.astype(np.int32), anchor=anchor, clean=clean)
    for norm, eps, alpha in (("linf", 0.05, 0.02), ("l2", 2.0, 0.8)):
        d0 = (eps * (2 * rng.random((6, 77, 128)) - 1)).astype(np.float32) * keep      # utils_attacks.py:680 init, masked
        if norm == "l2":
            d0 = project_perturbation(torch.from_numpy(d0), eps, norm).numpy()
        delta = torch.from_numpy(d0.copy())
        out[f"{norm}_eps"], out[f"{norm}_alpha"], out[f"{norm}_delta0"] = np.float32(eps), np.float32(alpha), d0
        for k in range(3):
            delta.requires_grad_(True)
            hook = model.token_embedding.register_forward_hook(lambda m, i, o: o + delta)
            feat = model.encode_text(text)
            hook.remove()
            loss = ((torch.from_numpy(anchor) - feat) ** 2).sum()
            (grad,) = torch.autograd.grad(loss, delta)
            delta = project_perturbation(delta.detach() + alpha * normalize_grad(grad, norm), eps, norm).detach()
            out[f"{norm}_loss{k}"] = np.float32(loss.item())
            out[f"{norm}_feat{k}"] = feat.detach().numpy()
            out[f"{norm}_grad{k}"] = grad.numpy().copy()
            out[f"{norm}_delta{k + 1}"] = delta.numpy().copy()
            assert np.abs(grad.numpy() * (1 - keep)).max() == 0.0
    np.savez_compressed(os.path.join(HERE, "tiny_pgd.npz"), **out)
    print("written tiny_pgd.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim > 1})


if __name__ == "__main__":
    main()
