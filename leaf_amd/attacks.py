"""LEAF character-level search (host side) driving the HIP ``score_candidates`` kernel chain.

Mirrors the reference's ``utils_attacks.py``: same function names, argument meaning, candidate order and
numpy global-RNG consumption, so that with the same ``np.random.seed`` the same candidate strings are
produced (checked against traces of the reference in tests/golden/attack_trace.json).

* ``generate_sentence``               utils_attacks.py:169-213
* ``generate_all_sentences(_at_z)``   utils_attacks.py:215-224,275-295
* ``generate_random_sentences_at_z``  utils_attacks.py:226-236
* ``valid_sentence_batched``          utils_attacks.py:110-143
* ``attack_text_leaf`` / ``attack_text`` utils_attacks.py:297-393,646-647

Difference by design: the reference computes ``encode_text`` -> loss -> ``argmax`` as separate torch ops;
here one C-ABI call (``leaf_score_candidates``) runs the forward of all B*rho candidates, the loss, the
first-index arg-max and the winner gather on the GPU and only B indices come back to the host.
"""
from __future__ import annotations

import re
import string
from typing import Callable, List, Optional, Sequence

import numpy as np

DEFAULT_V = [-1] + [ord(c) for c in string.ascii_lowercase + ' ' + string.ascii_uppercase + string.digits + string.punctuation]

_OBJ = {"l2": 0, "negl2": 1, "dissim": 2, "sim": 3}


# ----------------------------------------------------------------------------- string mutation
def generate_sentence(S: str, z, u, V: Sequence[int], k: int = 1, alternative: Optional[int] = None) -> str:
    """One edit of ``S``.  The sentence is viewed as k gap slots before every character plus k trailing
    ones (slot text is '_'); position ``z`` indexes that expanded view.  ``V[u] == -1`` deletes, a
    character equal to the current one deletes (alternative == -1) or becomes ``alternative``."""
    step = k + 1
    n_pos = step * len(S) + k
    cells = ['_'] * n_pos
    keep = [False] * n_pos
    for i, ch in enumerate(S):
        cells[step * i + k] = ch
        keep[step * i + k] = True
    if isinstance(z, list):
        for p, c in zip(z, u):
            if V[c] != -1:
                cells[p], keep[p] = chr(V[c]), True
            else:
                cells[p], keep[p] = '_', False
    else:
        c = V[u]
        if c == -1:
            cells[z], keep[z] = '_', False
        elif cells[z] == chr(c) and alternative is not None:
            if alternative != -1:
                cells[z], keep[z] = chr(alternative), True
            else:
                cells[z], keep[z] = '_', False
        else:
            cells[z], keep[z] = chr(c), True
    return ''.join(ch for ch, m in zip(cells, keep) if m)


def generate_all_sentences_at_z(S, z, V, k=1, alternative=-1):
    return [generate_sentence(S, z, u, V, k, alternative=alternative) for u in range(len(V))]


def generate_random_sentences_at_z(S, z, V, n, k=1, alternative=-1):
    picks = np.random.choice(range(len(V)), size=n, replace=(n > len(V)))
    return [generate_sentence(S, z, u, V, k, alternative=alternative) for u in picks]


def generate_all_sentences(S, V, subset_z=None, k=1, alternative=None):
    if subset_z is None:
        subset_z = range((k + 1) * len(S) + k)
    out = []
    for z in subset_z:
        out += generate_all_sentences_at_z(S, z, V, k, alternative=alternative)
    return out


# ----------------------------------------------------------------------------- dictionary constraint
class Dictionary:
    """Word list + word tokenizer behind ``--constrain``.  The reference uses nltk (``words`` corpus,
    ``word_tokenize``) and rebuilds the set on every call (utils_attacks.py:125); here it is built once."""

    def __init__(self, words: Sequence[str], tokenize: Optional[Callable[[str], List[str]]] = None):
        self.words = frozenset(words)
        self.tokenize = tokenize or (lambda s: re.findall(r"[A-Za-z0-9]+|[^\sA-Za-z0-9]", s))

    @classmethod
    def from_nltk(cls):
        import nltk  # noqa: F401  (absent in the build image; present where the reference runs)
        from nltk.corpus import words
        from nltk.tokenize import word_tokenize
        return cls(words.words(), word_tokenize)

    @classmethod
    def from_file(cls, path: str):
        with open(path) as f:
            return cls([w.strip() for w in f if w.strip()])

    def count(self, sentence: str) -> int:
        return len(self.words.intersection(self.tokenize(sentence.lower())))


_dictionary: Optional[Dictionary] = None


def set_dictionary(d: Optional[Dictionary]):
    global _dictionary
    _dictionary = d


def get_dictionary() -> Dictionary:
    global _dictionary
    if _dictionary is None:
        _dictionary = Dictionary.from_nltk()
    return _dictionary


def valid_sentence_batched(original, attacked, debug=False):
    """valid iff the attacked sentence has STRICTLY FEWER dictionary words than the original
    (utils_attacks.py:143)."""
    if isinstance(attacked, str):
        attacked = [[attacked]]
    if isinstance(attacked[0], str):
        attacked = [attacked]
    if isinstance(original, str):
        original = [original]
    D = get_dictionary()
    lo = [D.count(o) for o in original]
    return [[D.count(a) < l for a in AS] for l, AS in zip(lo, attacked)]


# ----------------------------------------------------------------------------- the search
def attack_text_leaf(model, tokenizer, sentences, anchor_features, device=None, objective="l2", n=10, k=1,
                     V=DEFAULT_V, constrain=False, debug=False, return_trace: Optional[list] = None):
    """LEAF attack on a batch of sentences.  ``model`` is a ``leaf_amd.model.LeafCLIPText`` (anything with
    ``score_candidates``); ``anchor_features`` a float32 CUDA tensor [B, D].  Returns
    ``(best_features [B,D], adversarial sentences)`` like the reference."""
    import torch
    sentences = list(sentences)
    B = len(sentences)
    if objective in ("dissim", "sim"):
        anchor_features /= anchor_features.norm(dim=-1, keepdim=True)   # in place, as the reference does
    space = [ord(' ')]
    best_feat = None
    for _ in range(k):
        # stage 1: rho random positions, insert / replace-with / delete a space
        positions, SS = [], []
        for S in sentences:
            positions.append(np.random.choice(range(2 * len(S) + 1), size=n, replace=n > 2 * len(S) + 1))
            SS.append(generate_all_sentences(S, space, subset_z=positions[-1], alternative=-1))
        if constrain:
            valid = valid_sentence_batched(sentences, SS)
            SS = [[c if ok else S for c, ok in zip(row, vrow)] for S, row, vrow in zip(sentences, SS, valid)]
        flat = [c for row in SS for c in row]
        if return_trace is not None:
            return_trace.append(flat)
        ids_best, _ = model.score_candidates(tokenizer.encode_batch(flat), anchor_features, n, objective,
                                             want_features=False)
        ids_best = ids_best.cpu().numpy()
        best_pos = [positions[row][i] for row, i in enumerate(ids_best)]
        # stage 2: rho random characters at the chosen position
        SS = [generate_random_sentences_at_z(S, best_pos[i], V, n, alternative=-1) for i, S in enumerate(sentences)]
        if constrain:
            valid = valid_sentence_batched(sentences, SS)
            SS = [[c if ok else S for c, ok in zip(row, vrow)] for S, row, vrow in zip(sentences, SS, valid)]
        flat = [c for row in SS for c in row]
        if return_trace is not None:
            return_trace.append(flat)
        ids_best, best_feat = model.score_candidates(tokenizer.encode_batch(flat), anchor_features, n, objective,
                                                     want_features=True)
        ids_best = ids_best.cpu().numpy()
        sentences = [flat[row * n + int(i)] for row, i in enumerate(ids_best)]
        if debug:
            print(sentences[0])
    return best_feat, sentences


def attack_text(model, tokenizer, sentences, image_features, device=None, objective="l2", n=10, k=1, V=DEFAULT_V,
                constrain=False, debug=False):
    return attack_text_leaf(model, tokenizer, sentences, image_features, device, objective, n, k, V, constrain, debug)
