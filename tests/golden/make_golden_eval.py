#!/usr/bin/env python3
"""Fixture for ``eval_textfare.py --per-sentence``: the reference's evaluation loop order (eval_textfare.py:113-141 -- ONE sentence
per ``attack_text_leaf`` call, so the global numpy RNG is consumed sentence by sentence) executed with the reference's own
``attack_text_leaf`` / tokenizer / ``CLIP.encode_text`` on two tiny models (clean = seed 12, fine-tuned stand-in = seed 13).

Runs only in the build container (needs /root/reference; same stub recipe as make_golden.py).  Stores inputs and outputs only:
sentences, the candidate strings of every stage of every sentence, adversarial sentences, the two TextFARE columns.

    python tests/golden/make_golden_eval.py     # writes tests/golden/eval_per_sentence.json
"""
import json
import os
import sys

import numpy as np
import torch
import transformers  # noqa: F401  (before torchvision is stubbed)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, V_DEFAULT, install_stubs, load_np_state  # noqa: E402

SENTENCES = ["a photo of a cat", "Stocks rally as oil prices fall", "two people in the park at sunset", "the red car"]


def main():
    install_stubs()
    import open_clip
    from open_clip.model import CLIP
    import utils_attacks
    from oracle import text_oracle as O

    torch.set_num_threads(4)
    tokenizer = open_clip.get_tokenizer("ViT-L-14")
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    clean_model = CLIP(**TINY, quick_gelu=True).float().eval()
    load_np_state(clean_model, O.init_weights(cfg, seed=12))
    model = CLIP(**TINY, quick_gelu=True).float().eval()
    load_np_state(model, O.init_weights(cfg, seed=13))
    out = {}
    for k, rho in ((1, 20), (2, 20)):
        log = []

        class SpyTok:
            def __call__(self, texts, context_length=None):
                log.append(list(texts) if not isinstance(texts, str) else [texts])
                return tokenizer(texts, context_length)
        spy = SpyTok()
        rows = []
        np.random.seed(5)
        with torch.no_grad():
            for text in SENTENCES:                        # one attack call per sentence: the loop order of eval_textfare.py:113-141 (attack_name == 'leaf')
                ids = tokenizer([text])
                f_clean = clean_model.encode_text(ids, normalize=False)
                f_model = model.encode_text(ids, normalize=False)
                del log[:]
                _, adv = utils_attacks.attack_text_leaf(model, spy, [text], f_model, "cpu", objective='l2', n=rho, k=k, V=V_DEFAULT,
                                                        debug=False, constrain=False)
                f_adv = model.encode_text(tokenizer([adv[0]]), normalize=False)
                rows.append(dict(sentence=text, adv_sentence=adv[0], stage_candidates=[list(c) for c in log],
                                 textfare_clean=float(((f_clean - f_model) ** 2).sum()), textfare_adv=float(((f_clean - f_adv) ** 2).sum())))
        out[f"k{k}"] = dict(seed=5, k=k, rho=rho, rows=rows)
    with open(os.path.join(HERE, "eval_per_sentence.json"), "w") as f:
        json.dump(dict(clean_seed=12, model_seed=13, model="tiny-test-quickgelu", cases=out), f)
    print("written", os.path.join(HERE, "eval_per_sentence.json"))


if __name__ == "__main__":
    main()
