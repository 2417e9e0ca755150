"""Native --constrain (leaf_tok_constrain, leaf_amd/csrc/host_text.cpp) against the Python restatement of
utils_attacks.py:110-143 and the reference-generated known answers (tests/golden/mutation_kat.json:valid_batched).
CPU only: the host pipeline needs no GPU."""
import json
import os
import random

import numpy as np
import pytest

from leaf_amd import attacks
from leaf_amd.native_text import NativeTokenizer


@pytest.fixture(scope="module")
def tok():
    return NativeTokenizer(n_threads=4)


@pytest.fixture(scope="module")
def kat(golden_dir):
    with open(os.path.join(golden_dir, "mutation_kat.json")) as f:
        return json.load(f)


def _edit_for(S, cand):
    """(z, c) with generate_sentence(S, z, c, alternative=-1) == cand, by search over the single edits."""
    for z in range(2 * len(S) + 1):
        for c in attacks.DEFAULT_V:
            if attacks._apply_edit(S, z, c) == cand:
                return z, c
    raise AssertionError((S, cand))


def test_reference_known_answers_through_the_native_path(tok, kat):
    D = attacks.Dictionary(kat["stub_words"])
    originals = ["a photo of a cat", "the red car"]
    cands = [["a photo of a ca t", "a photo of acat", "a photo of a cat", "a phot o of a cat"],
             ["thered car", "the red ca r", "the re d car", "the red car"]]
    zc = np.array([[_edit_for(S, c_) for c_ in row] for S, row in zip(originals, cands)], dtype=np.int32)
    valid, fb = tok.constrain_mask(D, originals, np.ascontiguousarray(zc[:, :, 0]), np.ascontiguousarray(zc[:, :, 1]))
    assert not fb.any()
    assert valid.tolist() == kat["valid_batched"], "reference valid_sentence_batched (utils_attacks.py:110-143) known answers"


def test_native_equals_python_on_random_edits(tok, kat):
    D = attacks.Dictionary(kat["stub_words"])
    rng = random.Random(1)
    words = kat["stub_words"] + ["zebra", "qwerty", "x1", "42", "don't", "end.", "(hi)", "a,b", "CAT", "The"]
    total = 0
    for trial in range(120):
        sents = [" ".join(rng.choice(words) for _ in range(rng.randint(1, 9))) for _ in range(5)]
        if trial % 5 == 0:
            sents[0] = "  " + sents[0] + "\t "
        if trial % 11 == 0:
            sents[1] = "x"
        rho = 40
        z = np.stack([np.array([rng.randrange(2 * len(S) + 1) for _ in range(rho)]) for S in sents]).astype(np.int32)
        c = np.array([[rng.choice(attacks.DEFAULT_V) for _ in range(rho)] for _ in sents], dtype=np.int32)
        valid, fb = tok.constrain_mask(D, sents, z, c)
        assert not fb.any(), "the regex tokenizer is local: every ASCII candidate is decided natively"
        for b, S in enumerate(sents):
            lo = D.count(S)
            for r in range(rho):
                want = D.count(attacks._apply_edit(S, int(z[b, r]), int(c[b, r]))) < lo
                assert bool(valid[b, r]) == want, (S, int(z[b, r]), int(c[b, r]))
                total += 1
    assert total == 120 * 5 * 40


def test_nltk_mode_decides_only_what_it_can_decide_exactly(tok, kat):
    """kind 'nltk': nltk.word_tokenize is a whitespace split for letters / digits / whitespace (Treebank contraction words
    excepted); everything else must come back as fallback for the caller's real tokenizer."""
    split = lambda s: s.split()
    D = attacks.Dictionary(kat["stub_words"] + ["can", "not"], tokenize=split, kind="nltk")
    sents = ["a photo of a cat", "the red car", "we cannot go", "a photo, of a cat", "gonna do it"]
    rho = 2 * max(len(s) for s in sents) + 1
    z = np.stack([np.arange(rho) % (2 * len(S) + 1) for S in sents]).astype(np.int32)
    for ch in (ord(' '), ord('x'), -1, ord('!')):
        c = np.full(z.shape, ch, dtype=np.int32)
        valid, fb = tok.constrain_mask(D, sents, z, c)
        assert fb[2].all() and fb[3].all() and fb[4].all(), "contraction words / punctuation in the sentence: declined"
        for b in (0, 1):
            S = sents[b]
            for r in range(rho):
                cand = attacks._apply_edit(S, int(z[b, r]), ch)
                if ch == ord('!') and cand != S:
                    assert fb[b, r], "a candidate with punctuation is declined"
                    continue
                assert not fb[b, r]
                assert bool(valid[b, r]) == (D.count(cand) < D.count(S)), (S, cand)


def test_stage_candidates_constrained_native_equals_python(tok, kat):
    """attacks._stage_candidates with --constrain: native mask + native mutate/BPE == the all-Python path (same tokens, same
    no-op replacement of invalid candidates)."""
    from leaf_amd.tokenizer import SimpleTokenizer
    attacks.set_dictionary(attacks.Dictionary(kat["stub_words"]))
    try:
        sents = ["a photo of a cat", "two people in the park at sunset", "the red car", "x"]
        rng = np.random.default_rng(3)
        rho = 50
        z = np.stack([rng.integers(0, 2 * len(S) + 1, rho) for S in sents]).astype(np.int32)
        c = np.array(attacks.DEFAULT_V, dtype=np.int32)[rng.integers(0, len(attacks.DEFAULT_V), (len(sents), rho))]
        z1, c1, z2, c2 = z.copy(), c.copy(), z.copy(), c.copy()
        t_nat, _ = attacks._stage_candidates(tok, sents, z1, c1, True, None)
        t_py, _ = attacks._stage_candidates(SimpleTokenizer(), sents, z2, c2, True, None)
        assert np.array_equal(t_nat, t_py) and np.array_equal(z1, z2) and np.array_equal(c1, c2)
        assert (z1 == 0).sum() > 10, "some candidates must have been rejected for the test to mean anything"
    finally:
        attacks.set_dictionary(None)
