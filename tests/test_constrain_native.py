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


@pytest.fixture(scope="module")
def tb_kat(golden_dir):
    with open(os.path.join(golden_dir, "treebank_kat.json")) as f:
        return json.load(f)


def _native_tokens(text, kind=1):
    import ctypes as C
    from leaf_amd import _lib
    b = text.encode()
    out, n = C.create_string_buffer(8 * len(b) + 64), C.c_int()
    rc = _lib.lib().leaf_tok_word_tokens(kind, b, len(b), out, len(out), C.byref(n))
    if rc == 2:
        return None
    assert rc == 0
    r = out.raw[:n.value].decode()
    return r.split("\n")[:-1] if r else []


def test_treebank_restatements_reproduce_nltk(tb_kat):
    """The word tokenizer of --constrain is nltk.word_tokenize (utils_attacks.py:135,139).  Both restatements of its Treebank
    step -- leaf_amd/treebank.py (regex pipeline) and the native one in host_text.cpp -- against 1,800 strings tokenised by the
    REAL nltk 3.6.5 (tests/golden/make_golden_treebank.py): punctuation, quotes, clitics, contractions, brackets, single-character
    edits of captions with every character of V."""
    from leaf_amd.treebank import punkt_free, treebank_tokenize
    assert "nltk 3.6.5" in tb_kat["source"] and len(tb_kat["cases"]) > 1500
    for s, want in tb_kat["cases"]:
        assert punkt_free(s.lower()), s
        assert treebank_tokenize(s.lower()) == want, s
        assert _native_tokens(s) == want, s
    # texts whose tokens depend on the Punkt sentence model are declined, not guessed
    from leaf_amd.treebank import word_tokenize
    for s in ("a cat. a dog", "e.g. this", "the end.) x", "one. two."):
        assert not punkt_free(s) and _native_tokens(s) is None
        with pytest.raises(ValueError):
            word_tokenize(s)
    assert _native_tokens("caf\u00e9 au lait") is None, "non-ASCII: the Python path"


def test_nltk_mode_native_equals_the_treebank_tokenizer_on_random_edits(tok, kat):
    """kind 'nltk': leaf_tok_constrain re-tokenises only the window around the edit with the restated Treebank pipeline (told
    whether the window touches the text's ends).  On 30,000 random single edits -- all 96 characters of V, captions with commas,
    quotes, clitics, brackets, a final period -- every candidate it decides must agree with tokenising the WHOLE candidate, and it
    may decline only what really depends on sentence boundaries (plus edits behind the text's final period)."""
    from leaf_amd.treebank import punkt_free, treebank_tokenize
    words = kat["stub_words"] + ["can", "not", "it", "s", "do", "wan", "na", "end", "hi", "b", "t", "is"]
    D = attacks.Dictionary(words, tokenize=lambda s: treebank_tokenize(s), kind="nltk")
    rng = random.Random(5)
    vocab = kat["stub_words"] + ["zebra", "x1", "42", "don't", "it's", "(hi)", "a,b", "1,000", "cannot", "wanna", "\"cat\"", "dog's", "dogs'",
                                 "'tis", "rock'n'roll", "a--b", "wait...", "what?!", "50%", "me&you", "'a'", "[the]", "e.g", "CAT", "The", "a:b", "2:30"]
    decided = declined = 0
    for trial in range(150):
        sents = [" ".join(rng.choice(vocab) for _ in range(rng.randint(1, 8))) for _ in range(5)]
        if trial % 3 == 0:
            sents[0] += "."                       # a final period
        if trial % 7 == 0:
            sents[1] = "  " + sents[1] + ". "
        if trial % 10 == 0:
            sents[2] = sents[2] + ". and more"    # sentence-boundary dependent: the whole caption is declined
        rho = 40
        z = np.stack([np.array([rng.randrange(2 * len(S) + 1) for _ in range(rho)]) for S in sents]).astype(np.int32)
        c = np.array([[rng.choice(attacks.DEFAULT_V) for _ in range(rho)] for _ in sents], dtype=np.int32)
        valid, fb = tok.constrain_mask(D, sents, z, c)
        for b, S in enumerate(sents):
            if not punkt_free(S.lower()):
                assert fb[b].all(), S
                declined += rho
                continue
            lo = D.count(S)
            for r in range(rho):
                cand = attacks._apply_edit(S, int(z[b, r]), int(c[b, r]))
                if fb[b, r]:
                    declined += 1
                    last_period = S.rstrip(" \t])}>\"'").rfind(".")
                    behind = S.rstrip(" \t])}>\"'").endswith(".") and (int(z[b, r]) // 2) > last_period
                    assert not punkt_free(cand.lower()) or behind, (S, cand)
                    continue
                decided += 1
                assert bool(valid[b, r]) == (D.count(cand) < lo), (S, cand, int(z[b, r]), int(c[b, r]))
    print("nltk-mode native constraint: decided", decided, "declined", declined)
    assert decided > 25000 and declined < 0.15 * (decided + declined)


_ABBR = {"e.g", "dr", "mr", "vs", "no", "st"}


def _fake_punkt_spans(text):
    """A stand-in for nltk's Punkt with the same KIND of dependence: whether a period (+ closing quotes / brackets) that ends a
    chunk also ends a sentence depends on the token that carries it (abbreviation list) and on the token after it (no break in
    front of a digit or an opening bracket) -- and '?' / '!' in front of a blank always end one.  Spans as span_tokenize gives them."""
    closers = "])}>\"'"
    n = len(text.rstrip())
    spans, start, i = [], 0, 0
    while i < n:
        ch = text[i]
        if ch in ".?!":
            j = i + 1
            while j < n and text[j] == '.':
                j += 1
            lone = ch != '.' or (j - i == 1 and (i == 0 or text[i - 1] != '.'))
            k = j
            while k < n and text[k] in closers:
                k += 1
            if lone and k < n and text[k].isspace():
                q = k
                while q < n and text[q].isspace():
                    q += 1
                tok0 = text[:i].split()[-1] if text[:i].split() else ""
                nxt = text[q] if q < n else ""
                brk = ch in "?!" or (tok0.strip(closers + "([{<") not in _ABBR and not nxt.isdigit() and nxt not in "([{<")
                if brk and q < n:
                    spans.append((start, k))
                    start = q
            i = max(j, i + 1)
        else:
            i += 1
    spans.append((start, n))
    return spans


def test_multi_sentence_captions_are_decided_natively_from_the_callers_sentence_spans(tok, kat):
    """Captions whose word tokens depend on where sentences end ("vintage chair. free shipping."): nltk's Punkt is asked ONCE per
    caption for its sentence spans and leaf_tok_constrain_ranges tokenises sentence by sentence; a candidate is decided natively
    only when its edit cannot change a sentence-break decision (no '.', '?', '!' in its window, not the token right behind one).
    Checked against tokenising the WHOLE candidate with the splitter re-run on it, for a stand-in splitter of Punkt's kind."""
    from leaf_amd.treebank import punkt_free, spans_word_tokenize
    words = kat["stub_words"] + ["can", "not", "it", "s", "do", "end", "hi", "b", "chair", "free", "shipping", "dr", "st", "no"]
    D = attacks.Dictionary(words, tokenize=lambda s: spans_word_tokenize(s, _fake_punkt_spans(s)), kind="nltk")
    D.span_tokenize = _fake_punkt_spans
    rng = random.Random(11)
    vocab = kat["stub_words"] + ["chair.", "shipping.", "dr.", "e.g.", "no.", "st.", "5", "(new)", "what?", "wow!", "cat,", "it's", "\"go.\"", "end.)",
                                 "zebra", "42", "don't", "a,b", "free", "hi"]
    decided = declined = multi = 0
    for trial in range(150):
        sents = [" ".join(rng.choice(vocab) for _ in range(rng.randint(2, 9))) for _ in range(5)]
        rho = 40
        z = np.stack([np.array([rng.randrange(2 * len(S) + 1) for _ in range(rho)]) for S in sents]).astype(np.int32)
        c = np.array([[rng.choice(attacks.DEFAULT_V) for _ in range(rho)] for _ in sents], dtype=np.int32)
        valid, fb = tok.constrain_mask(D, sents, z, c)
        for b, S in enumerate(sents):
            multi += not punkt_free(S.lower())
            lo = D.count(S)
            for r in range(rho):
                if fb[b, r]:
                    declined += 1
                    continue
                decided += 1
                cand = attacks._apply_edit(S, int(z[b, r]), int(c[b, r]))
                assert bool(valid[b, r]) == (D.count(cand) < lo), (S, cand, int(z[b, r]), int(c[b, r]))
    print("multi-sentence constraint: captions needing spans", multi, "of", 150 * 5, "decided", decided, "declined", declined)
    assert multi > 300 and decided > 0.6 * (decided + declined)


def test_multi_sentence_known_answers_from_the_real_punkt_code(tok, golden_dir):
    """The same path against the REAL nltk code: tests/golden/punkt_kat.json holds, for 160 captions (119 multi-sentence), the spans of
    nltk's PunktSentenceTokenizer and the validity of 4,800 random single edits computed by re-running Punkt + NLTKWordTokenizer on
    every whole candidate (tests/golden/make_golden_punkt.py).  Given the spans of the CAPTION only, everything the native code
    decides must agree; what it declines goes to the real tokenizer at run time."""
    with open(os.path.join(golden_dir, "punkt_kat.json")) as f:
        k = json.load(f)
    spans = {c["caption"].lower(): [tuple(s) for s in c["spans"]] for c in k["cases"]}
    D = attacks.Dictionary(k["words"], tokenize=lambda s: (_ for _ in ()).throw(AssertionError("no whole-string tokenisation here")), kind="nltk")
    D.span_tokenize = lambda t: spans[t]
    decided = declined = 0
    for i in range(0, len(k["cases"]), 8):
        batch = k["cases"][i:i + 8]
        sents = [c["caption"] for c in batch]
        z = np.array([[e[0] for e in c["edits"]] for c in batch], dtype=np.int32)
        cc = np.array([[e[1] for e in c["edits"]] for c in batch], dtype=np.int32)
        valid, fb = tok.constrain_mask(D, sents, z, cc)
        for b, c in enumerate(batch):
            for r, e in enumerate(c["edits"]):
                if fb[b, r]:
                    declined += 1
                else:
                    decided += 1
                    assert bool(valid[b, r]) == bool(e[2]), (c["caption"], e)
    # the run-time fallback for what was declined (and, here, for every edit): Punkt's spans of the CANDIDATE + a native count
    cand_spans = {}
    for c in k["cases"]:
        for e in c["edits"]:
            cand_spans[attacks._apply_edit(c["caption"], e[0], e[1]).lower()] = [tuple(s) for s in e[3]]
    D.span_tokenize = lambda t: spans[t] if t in spans else cand_spans[t]
    for c in k["cases"]:
        lo = D.count_fast(c["caption"])
        for e in c["edits"]:
            assert (D.count_fast(attacks._apply_edit(c["caption"], e[0], e[1])) < lo) == bool(e[2]), (c["caption"], e[:3])
    print("real-Punkt known answers: decided", decided, "declined", declined)
    assert decided > 0.5 * (decided + declined)     # (half of this vocabulary's tokens carry a period: far more than real captions)


def _punkt_from(params, strict):
    from leaf_amd.native_text import NativePunkt
    return NativePunkt(params["abbrev_types"], [tuple(c) for c in params["collocations"]], params["sent_starters"], params["ortho_context"],
                       strict=strict)


def test_native_punkt_reproduces_the_real_sentence_splitter(golden_dir):
    """host_text.cpp punkt_spans against nltk's PunktSentenceTokenizer.span_tokenize (tests/golden/punkt_native_kat.json: 2 x 2,500
    texts, empty and hand-filled parameter tables; make_golden_punkt_native.py --stress compared 2 x 100,000 more).  strict mode
    (the default) may decline a text -- two candidate break positions in one chunk -- but never answers differently."""
    with open(os.path.join(golden_dir, "punkt_native_kat.json")) as f:
        k = json.load(f)
    for st in k["sets"]:
        free, strict = _punkt_from(st["params"], False), _punkt_from(st["params"], True)
        declined = 0
        for text, exp in st["cases"]:
            exp = [tuple(e) for e in exp]
            assert free.spans(text) == exp, (st["name"], text)
            got = strict.spans(text)
            declined += got is None
            assert got is None or got == exp, (st["name"], text)
        assert declined < 0.25 * len(st["cases"])
    assert strict.spans("what?! yes") is None and free.spans("what?! yes") is not None
    assert strict.spans("caf\u00e9. yes") is None and strict.spans("a.\tb") is None          # outside the restated domain


def test_native_punkt_start_up_check_against_the_installed_splitter(golden_dir):
    """Dictionary._native_punkt (the from_nltk code path) with stand-ins for the installed Punkt instance: tables are taken from
    the instance; the native splitter is used only if it reproduces the instance on the battery, and texts with several
    sentence-end candidates in one chunk are decided natively only if the instance answers them like the restated rule."""
    import types
    with open(os.path.join(golden_dir, "punkt_kat.json")) as f:
        tables = json.load(f)["tables"]
    params = types.SimpleNamespace(abbrev_types=set(tables["abbrev_types"]), collocations={tuple(c) for c in tables["collocations"]},
                                   sent_starters=set(tables["sent_starters"]), ortho_context=dict(tables["ortho_context"]))
    punkt = types.SimpleNamespace(_params=params, _lang_vars=None)
    same = _punkt_from(tables, False)
    native = attacks.Dictionary._native_punkt(punkt, lambda t: same.spans(t))                 # an instance that follows the 3.6.5 rule
    assert native is not None and native.strict is False and native.spans("what?! yes") is not None
    from leaf_amd.treebank import PUNKT_MULTI_CHECK_STRINGS
    other = lambda t: [(0, len(t.rstrip()))] if t in PUNKT_MULTI_CHECK_STRINGS else same.spans(t)    # differs on the ambiguous class only
    native = attacks.Dictionary._native_punkt(punkt, other)
    assert native is not None and native.strict is True and native.spans("what?! yes") is None and native.spans("a cat. a dog") is not None
    assert attacks.Dictionary._native_punkt(punkt, lambda t: [(0, len(t))]) is None            # a different splitter altogether
    punkt_de = types.SimpleNamespace(_params=params, _lang_vars=types.SimpleNamespace(
        sent_end_chars=(".", "?", "!", ";"), internal_punctuation=",:;", _re_word_start="x", _re_multi_char_punct="y"))
    assert attacks.Dictionary._native_punkt(punkt_de, lambda t: same.spans(t)) is None        # non-default language variables


@pytest.mark.parametrize("block", ["cases", "cases_tables"])
def test_constrain_with_the_native_sentence_splitter(tok, golden_dir, block):
    """leaf_tok_constrain_punkt: Punkt's tables on the native side, sentence spans of captions and candidates computed in C++.
    Known answers from the real nltk code (whole-candidate Punkt + NLTKWordTokenizer, make_golden_punkt.py), with empty tables
    ("cases") and with filled ones ("cases_tables").  With strict off NOTHING is declined; with strict on what is decided agrees."""
    with open(os.path.join(golden_dir, "punkt_kat.json")) as f:
        k = json.load(f)
    params = k["tables"] if block == "cases_tables" else {"abbrev_types": [], "collocations": [], "sent_starters": [], "ortho_context": {}}
    for strict in (False, True):
        D = attacks.Dictionary(k["words"], tokenize=lambda s: (_ for _ in ()).throw(AssertionError("no Python tokenisation here")), kind="nltk")
        D.punkt_native = _punkt_from(params, strict)
        decided = declined = 0
        for i in range(0, len(k[block]), 8):
            batch = k[block][i:i + 8]
            sents = [c["caption"] for c in batch]
            z = np.array([[e[0] for e in c["edits"]] for c in batch], dtype=np.int32)
            cc = np.array([[e[1] for e in c["edits"]] for c in batch], dtype=np.int32)
            valid, fb = tok.constrain_mask(D, sents, z, cc)
            for b, c in enumerate(batch):
                for r, e in enumerate(c["edits"]):
                    if fb[b, r]:
                        declined += 1
                    else:
                        decided += 1
                        assert bool(valid[b, r]) == bool(e[2]), (strict, c["caption"], e[:3])
        print(block, "strict", strict, "decided", decided, "declined", declined)
        assert declined == 0 if not strict else declined < 0.2 * (decided + declined)


def test_word_list_dictionary_with_exported_punkt_tables(tok, golden_dir, tmp_path):
    """``--dictionary-file words.txt --dictionary-tokenizer treebank --punkt-params tables.json`` (tools/export_punkt_params.py): the
    reference's constraint without nltk.  Both of its paths -- the Python one (``Dictionary.count``: native sentence spans +
    restated Treebank step) and the native one behind ``_stage_candidates`` -- against the real-nltk known answers."""
    with open(os.path.join(golden_dir, "punkt_kat.json")) as f:
        k = json.load(f)
    (tmp_path / "words.txt").write_text("\n".join(k["words"]))
    (tmp_path / "punkt.json").write_text(json.dumps(k["tables"]))
    D = attacks.Dictionary.from_file(str(tmp_path / "words.txt"), tokenizer="treebank", punkt_params=str(tmp_path / "punkt.json"))
    assert D.punkt_native is not None and D.kind == "nltk"
    for c in k["cases_tables"][:60]:
        lo = D.count(c["caption"])
        for e in c["edits"]:
            assert (D.count(attacks._apply_edit(c["caption"], e[0], e[1])) < lo) == bool(e[2]), (c["caption"], e)
    attacks.set_dictionary(D)
    try:
        batch = k["cases_tables"][:16]
        sents = [c["caption"] for c in batch]
        z = np.array([[e[0] for e in c["edits"]] for c in batch], dtype=np.int32)
        cc = np.array([[e[1] for e in c["edits"]] for c in batch], dtype=np.int32)
        z0 = z.copy()
        attacks._stage_candidates(tok, sents, z, cc, True, None)          # rejected candidates become the no-op edit (0, -1)
        for b, c in enumerate(batch):
            for r, e in enumerate(c["edits"]):
                if not e[2]:
                    assert z[b, r] == 0 and cc[b, r] == -1, (c["caption"], e)
                else:
                    assert z[b, r] == z0[b, r]
    finally:
        attacks.set_dictionary(None)


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
