#!/usr/bin/env python3
"""TextFARE clean / adversarial loss of a fine-tuned text encoder on local captions.

Equivalent of the reference's ``eval_textfare.py:112-149`` for the LEAF attack: for every sentence,
``textfare_clean = ||f_clean(s) - f_model(s)||^2`` and ``textfare_adv = ||f_clean(s) - f_model(s_adv)||^2`` where
``s_adv = attack_text_leaf(model, ..., anchor=f_model(s), n=rho, k=k)``; results go to
``results_textfare/<name>_leaf_k{k}_rho_{rho}[_constrained].csv`` with the reference's four columns.  Differences:
the captions come from a local text file (one per line; the reference downloads AG-News), sentences are attacked in
batches (the LEAF search is per-sentence independent, so results do not change), models are local checkpoints.

    python eval_textfare.py --model ViT-L-14-quickgelu --clean /path/clean.bin --robust /path/epoch_latest.pt \
        --texts captions.txt --k 1 --rho 50 --n-test 100
"""
import argparse
import csv
import os

from leaf_amd import configure_runtime

configure_runtime()   # HIP_FORCE_DEV_KERNARG=1, before torch loads the HIP runtime (leaf_amd/__init__.py says why)
import string
import sys

import numpy as np
import torch

from leaf_amd.attacks import Dictionary, attack_text_leaf, set_dictionary
from leaf_amd.model import create_model
from leaf_amd.tokenizer import get_tokenizer


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-L-14-quickgelu")
    ap.add_argument("--clean", default=None, help="checkpoint of the original (clean) text encoder; default: random init seed 1")
    ap.add_argument("--robust", default=None, help="checkpoint of the fine-tuned encoder; default: same as --clean")
    ap.add_argument("--texts", required=True)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--rho", type=int, default=50)
    ap.add_argument("--n-test", type=int, default=100)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--constrain", action="store_true")
    ap.add_argument("--dictionary-file", default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out-dir", default="results_textfare")
    ap.add_argument("--random-init", action="store_true", help="allow --clean to be omitted (seeded random weights)")
    a = ap.parse_args(argv)
    if not a.clean and not a.random_init and not a.model.startswith("tiny-test"):
        raise SystemExit("--clean is empty: pass the clean checkpoint, or --random-init to evaluate seeded random weights on purpose")
    V = [-1] + [ord(c) for c in string.ascii_lowercase + ' ' + string.ascii_uppercase + string.digits + string.punctuation]
    np.random.seed(a.seed)
    clean = create_model(a.model, pretrained=a.clean, seed=1)
    model = create_model(a.model, pretrained=a.robust or a.clean, seed=1)
    tok = get_tokenizer(a.model)
    if a.constrain:
        set_dictionary(Dictionary.from_file(a.dictionary_file) if a.dictionary_file else Dictionary.from_nltk())
    with open(a.texts) as f:
        sentences = [l.strip() for l in f if l.strip()][:a.n_test]
    os.makedirs(a.out_dir, exist_ok=True)
    name = os.path.basename((a.robust or a.clean or a.model).rstrip("/")).split(".")[0]
    path = os.path.join(a.out_dir, f"{name}_{os.path.basename(a.texts).split('.')[0]}_leaf_k{a.k}_rho_{a.rho}" +
                        ("_constrained" if a.constrain else "") + ".csv")
    rows = []
    for i in range(0, len(sentences), a.batch_size):
        batch = sentences[i:i + a.batch_size]
        ids = tok.encode_batch(batch)
        f_clean = clean.encode_text(ids)
        f_orig = model.encode_text(ids)
        _, adv = attack_text_leaf(model, tok, batch, f_orig.clone(), objective="l2", n=a.rho, k=a.k, V=V, constrain=a.constrain)
        f_adv = model.encode_text(tok.encode_batch(adv))
        lc = ((f_clean - f_orig) ** 2).sum(-1).cpu().numpy()
        la = ((f_clean - f_adv) ** 2).sum(-1).cpu().numpy()
        rows += [dict(sentence=s, adv_sentence=t, textfare_clean=float(x), textfare_adv=float(y)) for s, t, x, y in zip(batch, adv, lc, la)]
        with open(path, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["sentence", "adv_sentence", "textfare_clean", "textfare_adv"])
            w.writeheader()
            w.writerows(rows)
    print(f"leaf k={a.k} rho={a.rho}: TextFARE clean {np.mean([r['textfare_clean'] for r in rows]):.5f} "
          f"adv {np.mean([r['textfare_adv'] for r in rows]):.5f}  ({len(rows)} sentences) -> {path}")
    return rows


if __name__ == "__main__":
    main()
