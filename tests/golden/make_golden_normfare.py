#!/usr/bin/env python3
"""Golden fixture for --normalize_fare (utils_AT.py:296,319: anchor and adversarial features are L2-normalised before
the TextFARE loss), produced by the REFERENCE's own CLIP.encode_text(normalize=True) + torch.autograd on the tiny config.
Runs only in the build container.   python tests/golden/make_golden_normfare.py -> tests/golden/tiny_normfare.npz"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    MG.install_stubs()
    from open_clip.model import CLIP
    from oracle import text_oracle as O

    torch.set_num_threads(8)
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=12)
    model = CLIP(**MG.TINY, quick_gelu=True).float()
    MG.load_np_state(model, w)
    for p_ in model.visual.parameters():
        p_.requires_grad = False
    toks = O.synthetic_tokens(8, seed=31, min_len=3, max_len=40)
    text = torch.from_numpy(toks.astype(np.int64))
    rng = np.random.default_rng(32)
    with torch.no_grad():
        clean = model.encode_text(text).numpy()
    a = clean + 0.5 * rng.standard_normal(clean.shape)
    anchor = (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)   # the frozen model's normalised features
    model.train()
    feat = model.encode_text(text, normalize=True)
    loss = Fn.mse_loss(torch.from_numpy(anchor), feat, reduction='none').sum(dim=-1).mean()
    loss.backward()
    grads = {n_: p_.grad.detach().numpy().copy() for n_, p_ in model.named_parameters()
             if p_.requires_grad and p_.grad is not None and n_ in w}
    rows = np.unique(toks)
    np.savez_compressed(os.path.join(HERE, "tiny_normfare.npz"), tokens=toks.astype(np.int32), anchor=anchor,
                        feat=feat.detach().numpy(), loss=np.float32(loss.item()),
                        **{"g:" + k: v for k, v in grads.items() if k != "token_embedding.weight"},
                        g_tok_rows=grads["token_embedding.weight"][rows], tok_rows=rows.astype(np.int32))
    print("written tiny_normfare.npz loss", loss.item())


if __name__ == "__main__":
    main()
