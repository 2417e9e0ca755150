#!/usr/bin/env python3
"""Golden fixture for the TextFARE loss + backward at PRODUCTION shape (ViT-L/14 text tower, both activations), produced
by the REFERENCE's CLIP + torch.autograd on CPU in fp32 (utils_AT.py:317-337).  Weights are oracle.init_weights(seed 1)
(regenerated on the GPU box from the seed, checked by abs-sums in manifest.json); the fixture stores, for EVERY trainable
text-tower tensor, the gradient's L2 norm and a fixed strided sample of its elements, plus the loss and features -- small
enough to commit, wide enough to catch a wrong tensor, a wrong scale or a transposed weight gradient.

Also writes ckpt_structure.json: the reference's AdamW grouping (train_AT_text_only.py:326-341) and
``optimizer.state_dict()`` / ``model.state_dict()`` structure for the tiny CLIP (key order, shapes, group membership) --
what ``--resume`` (train_AT_text_only.py:351-372) expects to find in epoch_latest.pt.

Runs only in the build container.   python tests/golden/make_golden_vitl_grads.py
"""
import json
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

    # ---- checkpoint / optimizer structure of the reference (tiny CLIP, same grouping lambda)
    model = CLIP(**MG.TINY, quick_gelu=True).float().train()
    for p_ in model.visual.parameters():   # train_AT_text_only.py:489-490
        p_.requires_grad = False
    exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or 'logit_scale' in n
    named = list(model.named_parameters())
    gain_or_bias = [(n, p) for n, p in named if exclude(n, p) and p.requires_grad]
    rest = [(n, p) for n, p in named if not exclude(n, p) and p.requires_grad]
    opt = torch.optim.AdamW([{"params": [p for _, p in gain_or_bias], "weight_decay": 0.},
                             {"params": [p for _, p in rest], "weight_decay": 1e-4}], lr=1e-5, betas=(0.9, 0.999), eps=1e-8)
    toks = torch.from_numpy(MG_tokens())
    f = model.encode_text(toks)
    (f ** 2).sum().backward()
    opt.step()
    osd = opt.state_dict()
    struct = {
        "state_dict_keys": [[k, list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()],
        "group_names": [[n for n, _ in gain_or_bias], [n for n, _ in rest]],
        "param_groups": [{k: (v if k != "params" else list(v)) for k, v in g.items()} for g in osd["param_groups"]],
        "state_entry_keys": sorted(next(iter(osd["state"].values())).keys()),
        "state_ids": sorted(osd["state"].keys()),
        "step_dtype": str(next(iter(osd["state"].values()))["step"].dtype),
        "step_shape": list(next(iter(osd["state"].values()))["step"].shape),
        "no_grad_names": [n for n, p in named if p.requires_grad and p.grad is None],
    }

    def jsonable(o):
        if isinstance(o, dict):
            return {k: jsonable(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [jsonable(v) for v in o]
        if isinstance(o, (bool, int, float, str)) or o is None:
            return o
        return str(o)
    with open(os.path.join(HERE, "ckpt_structure.json"), "w") as fjs:
        json.dump(jsonable(struct), fjs)
    print("ckpt_structure.json:", len(struct["state_dict_keys"]), "state_dict keys;", len(struct["state_ids"]), "optimizer states;",
          "params without grad:", struct["no_grad_names"])


def MG_tokens():
    from oracle import text_oracle as O
    return O.synthetic_tokens(4, seed=2).astype(np.int64)


if __name__ == "__main__":
    main()
