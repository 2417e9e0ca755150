#!/usr/bin/env python3
"""Golden fixture for --grad-clip-norm TOGETHER WITH --accum-freq 2 without a GradScaler (utils_AT.py:327-362, the
``scaler is None`` branch: ``backward(loss / accum_freq)``, then ``clip_grad_norm_`` after EVERY micro-batch -- i.e. on the running
sum -- and the gradient ``optimizer.step()`` then sees), produced by the REFERENCE's CLIP on the tiny config with the same torch calls.
Runs only in the build container.   python tests/golden/make_golden_microclip.py -> tests/golden/tiny_microclip.npz"""
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
    accum, max_norm = 2, 40.0
    norms, losses = [], []
    for j in range(accum):
        sl = slice(4 * j, 4 * j + 4)
        feat = model.encode_text(torch.from_numpy(toks[sl].astype(np.int64)))
        loss = Fn.mse_loss(torch.from_numpy(anchor[sl]), feat, reduction='none').sum(dim=-1).mean()
        (loss / accum).backward()                                                   # utils_AT.py:327,337
        norms.append(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm, norm_type=2.0).item())   # :359-360
        losses.append(loss.item())
    # the gradient the optimizer step sees (text tower, oracle key names)
    grads = {k: p_.grad.detach().numpy().copy() for k, p_ in model.named_parameters() if p_.grad is not None and k in w}
    rows = np.unique(toks)
    tok_grad = dict(model.named_parameters())["token_embedding.weight"].grad.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "tiny_microclip.npz"), max_norm=np.float32(max_norm), norms=np.float32(norms),
                        losses=np.float32(losses), final_grad_norm=np.float32(np.sqrt(sum(float((g.astype(np.float64) ** 2).sum())
                                                                                          for g in grads.values()))),
                        **{"grad:" + k: g for k, g in grads.items() if k != "token_embedding.weight"},
                        grad_tok_rows=tok_grad[rows], tok_rows=rows.astype(np.int32))
    print("written tiny_microclip.npz norms", norms, "losses", losses)


if __name__ == "__main__":
    main()
