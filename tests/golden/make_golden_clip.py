#!/usr/bin/env python3
"""Golden fixture for --grad-clip-norm (utils_AT.py:348-357: torch.nn.utils.clip_grad_norm_(model.parameters(), c, 2.0)
between the backward and the AdamW step), produced by the REFERENCE's CLIP on the tiny config with the same torch calls.
Runs only in the build container.   python tests/golden/make_golden_clip.py -> tests/golden/tiny_clip.npz"""
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
    model = CLIP(**MG.TINY, quick_gelu=True).float().train()
    MG.load_np_state(model, w)
    for p_ in model.visual.parameters():
        p_.requires_grad = False
    z = np.load(os.path.join(HERE, "tiny_quickgelu.npz"))
    toks, anchor = z["tokens"][:8], z["anchor"]
    feat = model.encode_text(torch.from_numpy(toks.astype(np.int64)))
    loss = Fn.mse_loss(torch.from_numpy(anchor), feat, reduction='none').sum(dim=-1).mean()
    loss.backward()
    max_norm = 0.5
    total = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm, norm_type=2.0)
    exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or 'logit_scale' in n
    params = [(n_, p_) for n_, p_ in model.named_parameters() if p_.requires_grad]
    opt = torch.optim.AdamW([
        {"params": [p_ for n_, p_ in params if exclude(n_, p_)], "weight_decay": 0.},
        {"params": [p_ for n_, p_ in params if not exclude(n_, p_)], "weight_decay": 0.2}],
        lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    opt.step()
    after = MG.state_to_np(model)
    rows = np.unique(toks)
    np.savez_compressed(os.path.join(HERE, "tiny_clip.npz"), max_norm=np.float32(max_norm), total_norm=np.float32(total.item()),
                        loss=np.float32(loss.item()),
                        **{"after:" + k: v for k, v in after.items() if k != "token_embedding.weight" and k in w},
                        after_tok_rows=after["token_embedding.weight"][rows], tok_rows=rows.astype(np.int32))
    print("written tiny_clip.npz total_norm", total.item(), "loss", loss.item())


if __name__ == "__main__":
    main()
