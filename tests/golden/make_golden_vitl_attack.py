#!/usr/bin/env python3
"""BASELINE.json configs[0] at its STATED size, traced from the REFERENCE itself (VERDICT r4 missing-3).

ViT-L/14 text tower (QuickGELU lineage), 8 captions, rho = 50, k = 1, ``constrain=True`` (stub word list -- the real nltk
corpus is absent), objective 'l2': the reference's own ``utils_attacks.attack_text`` (utils_attacks.py:297-393,
646-647) runs on CPU in fp32 with the weights ``oracle.init_weights(cfg, seed=1)`` loaded into the reference's ``CLIP`` (as the
vitl_* fixtures do).  Recorded: the candidate strings of both stages (a spy tokenizer), the reference's OWN ``loss[B, rho]``
and arg-max of both stages (a spy on ``torch.argmax``, the call at :348 / :386), the anchor, the adversarial sentences and the
returned features.  Only inputs and outputs are stored; no reference source travels.  Runs only in the build container.

    python tests/golden/make_golden_vitl_attack.py      # ~2 min on 8 cores; writes attack_vitl_k1_c1.{json,npz}
"""
import json
import os
import sys

import numpy as np
import torch
import transformers  # noqa: F401  (before the torchvision stub, as in make_golden.py)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REPO, V_DEFAULT, install_stubs, load_np_state  # noqa: E402

SENTENCES = ["a photo of a cat", "A Photo of a DOG on the table!", "two people in the park at sunset", "the red car",
             "a man with a hat", "I'm sure it's 42 degrees", "an old man is at the table with two people", "go to the sun set"]
SEED, RHO, K = 321, 50, 1


def main():
    install_stubs()
    import open_clip
    import utils_attacks
    from oracle import text_oracle as O

    torch.set_num_threads(8)
    tokenizer = open_clip.get_tokenizer("ViT-L-14")
    mname = "ViT-L-14-quickgelu"
    cfg = O.CONFIGS[mname]
    w = O.init_weights(cfg, seed=1)
    model = open_clip.create_model(mname, pretrained=None, precision="fp32", device="cpu").eval()
    load_np_state(model, w)

    with torch.no_grad():
        anchor = model.encode_text(tokenizer(SENTENCES))
    cand_log, loss_log, pick_log = [], [], []

    class SpyTok:
        def __call__(self, texts, context_length=None):
            cand_log.append(list(texts) if not isinstance(texts, str) else [texts])
            return tokenizer(texts, context_length)

    real_argmax = torch.argmax

    def spy_argmax(x, *a, **kw):
        r = real_argmax(x, *a, **kw)
        if x.dim() == 2 and tuple(x.shape) == (len(SENTENCES), RHO):
            loss_log.append(x.detach().numpy().copy())
            pick_log.append(r.numpy().copy())
        return r

    np.random.seed(SEED)
    torch.argmax = spy_argmax
    try:
        with torch.no_grad():
            feats, adv = utils_attacks.attack_text(model, SpyTok(), list(SENTENCES), anchor.clone(), "cpu", objective="l2",
                                                   n=RHO, k=K, V=V_DEFAULT, constrain=True)
    finally:
        torch.argmax = real_argmax
    assert len(cand_log) == len(loss_log) == len(pick_log) == 2 * K
    rejected = [sum(c == s for c in cand_log[st][i * RHO:(i + 1) * RHO]) for st in range(2 * K) for i, s in enumerate(SENTENCES)]
    with open(os.path.join(HERE, "attack_vitl_k1_c1.json"), "w") as f:
        json.dump(dict(model=mname, weight_seed=1, sentences=SENTENCES, seed=SEED, rho=RHO, k=K, constrain=True,
                       stage_candidates=cand_log, adv=adv, candidates_equal_to_caption=rejected,
                       source="utils_attacks.attack_text (utils_attacks.py:297-393,646-647), reference CLIP fp32 on CPU"), f)
    np.savez_compressed(os.path.join(HERE, "attack_vitl_k1_c1.npz"), anchor=anchor.numpy(), feats=feats.numpy(),
                        loss=np.stack(loss_log), picks=np.stack(pick_log).astype(np.int32))
    print("adv:", adv)
    print("candidates equal to their caption (rejected or no-op):", rejected)
    print("written under", HERE, "| repo", REPO)


if __name__ == "__main__":
    main()
