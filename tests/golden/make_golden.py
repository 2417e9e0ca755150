#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never runs this).
Recipe = SURVEY.md section 8c: import torch/transformers first, register stub modules for
the reference's absent third-party imports (torchvision, timm, ftfy, nltk, torchmetrics,
webdataset, braceexpand), put the reference on sys.path, then call its own
``open_clip.create_model`` / ``get_tokenizer`` / ``utils_attacks.attack_text`` on CPU, fp32.

Only inputs and outputs are stored (arrays + JSON); no reference source travels.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import json
import math
import os
import string
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch
import transformers  # noqa: F401  (must be imported before torchvision is stubbed)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

# small stub dictionary used for the --constrain rule (the real nltk corpus is absent)
STUB_WORDS = ["a", "photo", "of", "cat", "dog", "the", "on", "table", "red", "car", "at", "an", "is", "man",
              "two", "people", "in", "park", "sun", "set", "sunset", "with", "hat", "to", "do", "go", "no"]


def install_stubs():
    for name in ["torchvision", "torchvision.ops", "torchvision.ops.misc", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchvision.datasets",
                 "torchmetrics", "torchmetrics.multimodal", "torchmetrics.multimodal.clip_score",
                 "timm", "timm.models", "timm.models.layers", "timm.layers",
                 "webdataset", "webdataset.filters", "webdataset.tariterators", "braceexpand"]:
        sys.modules[name] = MagicMock()
    ftfy = types.ModuleType("ftfy")
    ftfy.fix_text = lambda s: s  # exact for ASCII input
    sys.modules["ftfy"] = ftfy
    nltk = types.ModuleType("nltk")
    nltk.download = lambda *a, **k: True
    tok = types.ModuleType("nltk.tokenize")
    import re as _re
    tok.word_tokenize = lambda s: _re.findall(r"[A-Za-z0-9]+|[^\sA-Za-z0-9]", s)
    corpus = types.ModuleType("nltk.corpus")
    words = types.SimpleNamespace(words=lambda: list(STUB_WORDS))
    corpus.words = words
    nltk.tokenize, nltk.corpus = tok, corpus
    sys.modules.update({"nltk": nltk, "nltk.tokenize": tok, "nltk.corpus": corpus})
    sys.path.insert(0, os.path.join(REF, "src"))
    sys.path.insert(0, REF)


V_DEFAULT = [-1] + [ord(c) for c in string.ascii_lowercase + ' ' + string.ascii_uppercase + string.digits + string.punctuation]

TINY = dict(embed_dim=64, vision_cfg=dict(image_size=32, layers=1, width=64, patch_size=16),
            text_cfg=dict(context_length=77, vocab_size=49408, width=128, heads=2, layers=2))


def state_to_np(model):
    keep = ("token_embedding", "positional_embedding", "transformer.", "ln_final", "text_projection")
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in model.state_dict().items()
            if k.startswith(keep)}


def load_np_state(model, w):
    sd = model.state_dict()
    for k, v in w.items():
        sd[k] = torch.from_numpy(v.copy())
    model.load_state_dict(sd)


CAPTIONS = ["a photo of a cat", "A Photo of a DOG on the table!", "two people in the park at sunset",
            "the red car", "a man with a hat", "an   extra   spaced &amp; html &lt;b&gt; caption",
            "I'm sure it's 42 degrees, isn't it?", "x"]


def main():
    install_stubs()
    import open_clip
    from open_clip.model import CLIP
    import utils_attacks
    from oracle import text_oracle as O

    torch.set_num_threads(8)
    tokenizer = open_clip.get_tokenizer("ViT-L-14")
    manifest = {"torch": torch.__version__, "numpy": np.__version__, "files": {}}

    # ---- 1. tokenizer known answers (a2)
    kat_texts = CAPTIONS + ["", "hello world " * 60, "UPPER lower MiXeD", "tab\tand\nnewline", "emoji-free ascii ~ ` ^",
                            "a photo of a cat ", " a photo of a cat", "a ph oto of a cat", "aphoto of a cat",
                            "1234567890", "don't won't they'll we've I'd"]
    ids = tokenizer(kat_texts).numpy().astype(np.int32)
    with open(os.path.join(HERE, "tokenizer_kat.json"), "w") as f:
        json.dump({"texts": kat_texts, "ids": ids.tolist()}, f)
    manifest["files"]["tokenizer_kat.json"] = "SimpleTokenizer.__call__ (src/open_clip/tokenizer.py:226-265)"

    # ---- 2. generate_sentence / generate_all_sentences known answers (a3)
    V = [-1] + [ord(c) for c in string.ascii_lowercase + ' ' + string.ascii_uppercase + string.digits + string.punctuation]
    gs = []
    for S in ["cat", "a b", "x", "hello world"]:
        for z in range(2 * len(S) + 1):
            for u in [0, 1, 27, 28, 60, 95]:
                for alt in [None, -1, ord('q')]:
                    gs.append(dict(S=S, z=z, u=u, alt=alt, out=utils_attacks.generate_sentence(S, z, u, V, 1, alternative=alt)))
    space = []
    for S in ["cat", "a b", " ab ", "hello world"]:
        space.append(dict(S=S, out=utils_attacks.generate_all_sentences(S, [ord(' ')], subset_z=None, alternative=-1)))
    np.random.seed(7)
    rnd = [dict(S=S, z=z, n=n, out=utils_attacks.generate_random_sentences_at_z(S, z, V, n, alternative=-1))
           for S, z, n in [("hello world", 3, 10), ("hello world", 4, 120), ("cat", 0, 5)]]
    valid = utils_attacks.valid_sentence_batched(["a photo of a cat", "the red car"],
                                                 [["a photo of a ca t", "a photo of acat", "a photo of a cat", "a phot o of a cat"],
                                                  ["thered car", "the red ca r", "the re d car", "the red car"]])
    with open(os.path.join(HERE, "mutation_kat.json"), "w") as f:
        json.dump({"V": V, "generate_sentence": gs, "space_all": space, "random_at_z_seed7": rnd,
                   "stub_words": STUB_WORDS, "valid_batched": valid}, f)
    manifest["files"]["mutation_kat.json"] = "utils_attacks.py:110-143,169-236,275-295"

    # ---- 3. tiny encode_text, both activations (a6) + converter canonical input
    for qg in (False, True):
        name = "tiny_quickgelu" if qg else "tiny_gelu"
        cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=qg)
        w = O.init_weights(cfg, seed=11 + int(qg))
        model = CLIP(**TINY, quick_gelu=qg).float().eval()
        load_np_state(model, w)
        toks = tokenizer(CAPTIONS).numpy()
        canon = np.array([[49406] + list(range(1, 77))], dtype=np.int64)  # conversion/convert_2.py:237
        canon[0, -1] = 49407
        syn = O.synthetic_tokens(6, seed=3)
        full = np.concatenate([toks, canon, syn], 0)
        with torch.no_grad():
            out = model.encode_text(torch.from_numpy(full)).numpy()
            outn = model.encode_text(torch.from_numpy(full), normalize=True).numpy()
        # loss + grads + AdamW step (a8, a9)
        anchor = (out[:8] + 0.5 * np.random.default_rng(5).standard_normal(out[:8].shape)).astype(np.float32)  # ||f-a|| ~ ||f||: well-conditioned gradient comparison
        model.train()
        for p_ in model.visual.parameters():
            p_.requires_grad = False
        feat = model.encode_text(torch.from_numpy(full[:8]))
        import torch.nn.functional as Fn
        loss = Fn.mse_loss(torch.from_numpy(anchor), feat, reduction='none').sum(dim=-1).mean()
        (loss / 2.0).backward()   # accum_freq = 2
        named = [(n_, p_) for n_, p_ in model.named_parameters() if p_.requires_grad and p_.grad is not None]
        grads = {n_: p_.grad.detach().numpy().copy() for n_, p_ in named if n_ in w}
        exclude = lambda n, p: p.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or 'logit_scale' in n
        params = [(n_, p_) for n_, p_ in model.named_parameters() if p_.requires_grad]
        opt = torch.optim.AdamW([
            {"params": [p_ for n_, p_ in params if exclude(n_, p_)], "weight_decay": 0.},
            {"params": [p_ for n_, p_ in params if not exclude(n_, p_)], "weight_decay": 0.2}],
            lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
        opt.step()
        after = state_to_np(model)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), tokens=full.astype(np.int32), out=out, out_norm=outn,
                            anchor=anchor, loss=np.float32(loss.item()),
                            **{"g:" + k: v for k, v in grads.items() if k != "token_embedding.weight"},
                            g_tok_rows=grads["token_embedding.weight"][np.unique(full[:8])],
                            g_tok_other_abs_sum=np.float32(np.abs(np.delete(grads["token_embedding.weight"], np.unique(full[:8]), axis=0)).sum()),
                            **{"after:" + k: v for k, v in after.items() if k != "token_embedding.weight"},
                            after_tok_rows=after["token_embedding.weight"][np.unique(full[:8])],
                            tok_rows=np.unique(full[:8]).astype(np.int32))
        manifest["files"][name + ".npz"] = dict(cfg=dict(width=128, heads=2, layers=2, embed_dim=64, quick_gelu=qg),
                                                weight_seed=11 + int(qg), accum_freq=2,
                                                adamw=dict(lr=1e-3, betas=[0.9, 0.98], eps=1e-6, wd=0.2))

    # ---- 4. attack_text_leaf trace on the tiny model (a3, a5, a7)
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=12)
    model = CLIP(**TINY, quick_gelu=True).float().eval()
    load_np_state(model, w)
    trace = {}
    for k_adv, constrain in [(1, False), (2, False), (1, True)]:
        sents = CAPTIONS[:5] + ["I'm sure it's 42 degrees", "hello world", "x"]
        with torch.no_grad():
            anchor = model.encode_text(tokenizer(sents))
        log = []
        orig_tok = tokenizer.__call__

        class SpyTok:
            def __call__(self, texts, context_length=None):
                log.append(list(texts) if not isinstance(texts, str) else [texts])
                return tokenizer(texts, context_length)
        np.random.seed(123)
        with torch.no_grad():
            feats, adv = utils_attacks.attack_text(model, SpyTok(), list(sents), anchor.clone(), "cpu", objective='l2',
                                                   n=50, k=k_adv, V=V, constrain=constrain)
        key = f"k{k_adv}_c{int(constrain)}"
        trace[key] = dict(sentences=sents, seed=123, rho=50, k=k_adv, constrain=constrain,
                          stage_candidates=log, adv=adv)
        np.savez_compressed(os.path.join(HERE, f"attack_{key}.npz"), anchor=anchor.numpy(), feats=feats.numpy())
    with open(os.path.join(HERE, "attack_trace.json"), "w") as f:
        json.dump(trace, f)
    manifest["files"]["attack_trace.json"] = "utils_attacks.attack_text (utils_attacks.py:297-393,646-647), tiny quickgelu model seed 12"

    # ---- 5. ViT-L-shape outputs (weights regenerated from the seed by oracle.init_weights)
    for mname, qg in (("ViT-L-14", False), ("ViT-L-14-quickgelu", True)):
        cfg = O.CONFIGS[mname]
        w = O.init_weights(cfg, seed=1)
        model = open_clip.create_model(mname, pretrained=None, precision="fp32", device="cpu").eval()
        load_np_state(model, w)
        base = O.synthetic_tokens(4, seed=0)
        cand = O.synthetic_candidates(base, 3, seed=1).reshape(-1, 77)
        full = np.concatenate([base, cand, tokenizer(CAPTIONS[:3]).numpy()], 0)
        with torch.no_grad():
            out = model.encode_text(torch.from_numpy(full)).numpy()
        chk = {k: float(np.abs(v).sum(dtype=np.float64)) for k, v in list(w.items())[:4]}
        np.savez_compressed(os.path.join(HERE, f"vitl_{'quickgelu' if qg else 'gelu'}.npz"), tokens=full.astype(np.int32), out=out)
        manifest["files"][f"vitl_{'quickgelu' if qg else 'gelu'}.npz"] = dict(model=mname, weight_seed=1, weight_abs_sums=chk)
        del model

    # ---- 6. cosine_lr known answers (a9)
    from open_clip_train.scheduler import cosine_lr

    class FakeOpt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0}]
    fo = FakeOpt()
    sched = cosine_lr(fo, 1e-5, 1400, 18750)
    lrs = {}
    for s in [0, 1, 699, 1399, 1400, 1401, 5000, 18749]:
        sched(s)
        lrs[str(s)] = fo.param_groups[0]["lr"]
    manifest["cosine_lr"] = dict(base_lr=1e-5, warmup=1400, steps=18750, values=lrs)

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
