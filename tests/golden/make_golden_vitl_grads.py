#!/usr/bin/env python3
"""Golden fixture for the TextFARE loss + backward at PRODUCTION shape (ViT-L/14 text tower, both activations), produced
by the REFERENCE's CLIP + torch.autograd on CPU in fp32 (utils_AT.py:317-337).  Weights are oracle.init_weights(seed 1)
(regenerated on the GPU box from the seed, checked by abs-sums in manifest.json); the fixture stores, for EVERY trainable
text-tower tensor, the gradient's L2 norm and a fixed strided sample of its elements, plus the loss and features -- small
enough to commit, wide enough to catch a wrong tensor, a wrong scale or a transposed weight gradient.

(The optimizer / checkpoint structure fixture, ckpt_structure.json, moved to make_golden_ckpt.py in round 3: it is produced by
executing the reference's own statements in the reference's order.  No optimizer is built here.)

Runs only in the build container.   python tests/golden/make_golden_vitl_grads.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

N_SAMPLE = 257   # elements sampled per tensor (prime stride pattern below)


def sample_index(numel: int) -> np.ndarray:
    """Deterministic sample of flat indices (shared with the test): a coprime stride walk over the tensor."""
    n = min(N_SAMPLE, numel)
    stride = max(1, numel // n) | 1
    return (np.arange(n, dtype=np.int64) * stride * 7 + 3) % numel


def main():
    MG.install_stubs()
    import open_clip
    from open_clip.model import CLIP
    from oracle import text_oracle as O

    torch.set_num_threads(8)
    for mname, tag in (("ViT-L-14-quickgelu", "quickgelu"), ("ViT-L-14", "gelu")):
        cfg = O.CONFIGS[mname]
        w = O.init_weights(cfg, seed=1)
        model = open_clip.create_model(mname, pretrained=None, precision="fp32", device="cpu").train()
        MG.load_np_state(model, w)
        for p_ in model.visual.parameters():
            p_.requires_grad = False
        toks = O.synthetic_tokens(6, seed=21)
        with torch.no_grad():
            f0 = model.encode_text(torch.from_numpy(toks.astype(np.int64))).numpy()
        # anchor ~||f|| away from f: a well-conditioned gradient comparison (as tiny_*.npz)
        anchor = (f0 + 0.5 * np.abs(f0).mean() / 0.8 * np.random.default_rng(5).standard_normal(f0.shape)).astype(np.float32)
        feat = model.encode_text(torch.from_numpy(toks.astype(np.int64)))
        loss = Fn.mse_loss(torch.from_numpy(anchor), feat, reduction='none').sum(dim=-1).mean()
        loss.backward()
        out = dict(tokens=toks.astype(np.int32), anchor=anchor, feat=feat.detach().numpy(), loss=np.float32(loss.item()))
        for n_, p_ in model.named_parameters():
            if n_ not in w or p_.grad is None:
                continue
            g = p_.grad.detach().numpy().astype(np.float32).ravel()
            out["n:" + n_] = np.float64(np.linalg.norm(g.astype(np.float64)))
            out["s:" + n_] = g[sample_index(g.size)]
        rows = np.unique(toks)
        out["tok_rows"] = rows.astype(np.int32)
        out["g_tok_rows"] = model.token_embedding.weight.grad.detach().numpy()[rows].astype(np.float16)   # 26 x 768: fp16 keeps it small
        np.savez_compressed(os.path.join(HERE, f"vitl_grads_{tag}.npz"), **out)
        print(mname, "loss", loss.item(), "tensors", sum(1 for k in out if k.startswith("n:")))
        del model


if __name__ == "__main__":
    main()
