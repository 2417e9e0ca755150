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

    def __init__(self, words: Sequence[str], tokenize: Optional[Callable[[str], List[str]]] = None, kind: Optional[str] = None):
        self.words = frozenset(words)
        self.tokenize = tokenize or (lambda s: re.findall(r"[A-Za-z0-9]+|[^\sA-Za-z0-9]", s))
        # which word tokenizer the native constraint (leaf_tok_constrain) has to reproduce: 'regex' = the stand-in above,
        # 'nltk' = nltk.word_tokenize (native only for letters/digits/whitespace), None = custom callable (Python only)
        self.kind = kind if kind is not None else ("regex" if tokenize is None else None)
        self._native = None
        # kind 'nltk' only: span_tokenize(text) -> [(start, end), ...] of the sentence splitter that `tokenize` runs in front of
        # the Treebank step (nltk's Punkt); None = unknown (captions whose tokens depend on it are decided string by string)
        self.span_tokenize = None
        self._span_cache = {}
        # kind 'nltk' only: native_text.NativePunkt built from that splitter's tables (sentence spans of captions AND candidates are
        # then computed in C++: nothing but non-ASCII text goes back to Python); None = not available / failed its self-check
        self.punkt_native = None

    @classmethod
    def from_nltk(cls):
        """The reference's dictionary: nltk's ``words`` corpus + ``word_tokenize`` (utils_attacks.py:7-9,125-139).  The native
        constraint restates word_tokenize's Treebank step (leaf_amd/treebank.py, host_text.cpp); it is switched on only after the
        restatement has reproduced the INSTALLED nltk on a fixed battery of strings -- a different nltk generation falls back to
        calling nltk itself for every candidate."""
        import logging
        import nltk  # noqa: F401  (absent from the build image's system interpreter; present where the reference runs)
        from nltk.corpus import words
        from nltk.tokenize import word_tokenize
        from .treebank import self_check
        ok = self_check(word_tokenize)
        if not ok:
            logging.warning(f"nltk {getattr(nltk, '__version__', '?')}: word_tokenize differs from the restated Treebank pipeline on the "
                            "self-check strings -- --constrain is decided with nltk itself for every candidate (slow host path)")
        d = cls(words.words(), word_tokenize, kind="nltk" if ok else None)
        if ok:
            try:    # the Punkt instance word_tokenize uses (nltk >= 3.8.2: _get_punkt_tokenizer; before: the pickled model)
                try:
                    from nltk.tokenize import _get_punkt_tokenizer
                    punkt = _get_punkt_tokenizer("english")
                except ImportError:
                    punkt = nltk.data.load("tokenizers/punkt/english.pickle")
                spans = lambda t: list(punkt.span_tokenize(t))
                from .treebank import SPAN_CHECK_STRINGS, treebank_tokenize
                # sentence by sentence through the restated Treebank step must reproduce word_tokenize on multi-sentence text
                if all([w for a, b in spans(t) for w in treebank_tokenize(t[a:b])] == word_tokenize(t) for t in SPAN_CHECK_STRINGS):
                    d.span_tokenize = spans
                    d.punkt_native = cls._native_punkt(punkt, spans)
                else:
                    logging.warning("nltk's sentence spans + the restated Treebank step do not reproduce word_tokenize on the self-check "
                                    "strings: multi-sentence captions are decided with nltk itself, string by string")
            except Exception as e:      # no Punkt model installed, a different API: the slower path, never a wrong answer
                logging.warning(f"nltk Punkt sentence spans unavailable ({e}): multi-sentence captions are decided string by string")
        return d

    @staticmethod
    def _native_punkt(punkt, spans):
        """The installed Punkt instance's tables behind the native restatement of its algorithm -- used only when it reproduces
        the instance's own ``span_tokenize`` on the self-check battery (a text the native code declines is not a failure)."""
        import logging
        try:
            from .native_text import NativePunkt
            from .treebank import PUNKT_CHECK_STRINGS
            native = NativePunkt.from_nltk(punkt)
            got = [(t, native.spans(t)) for t in PUNKT_CHECK_STRINGS]
            wrong = [t for t, g in got if g is not None and g != [(int(a), int(b)) for a, b in spans(t)]]
            if wrong or sum(g is not None for _, g in got) < len(got) // 2:
                logging.warning(f"the native Punkt restatement differs from the installed nltk on {wrong[:3]!r}...: sentence spans are "
                                "asked from nltk (one call per multi-sentence caption and per candidate that touches a sentence end)")
                return None
            # texts with several candidate positions in one chunk: decided natively only if THIS nltk answers them like the restated
            # rule (otherwise the native splitter keeps declining them: strict mode)
            from .treebank import PUNKT_MULTI_CHECK_STRINGS
            native.set_strict(False)
            if not all(native.spans(t) == [(int(a), int(b)) for a, b in spans(t)] for t in PUNKT_MULTI_CHECK_STRINGS):
                native.set_strict(True)
            logging.info("--constrain: nltk's Punkt tables loaded into the native sentence splitter (abbreviations %d, collocations %d, "
                         "sentence starters %d, orthographic contexts %d)" % native.sizes
                         + ("" if not native.strict else "; texts with several sentence-end candidates in one chunk go to nltk"))
            return native
        except Exception as e:
            logging.warning(f"native Punkt unavailable ({e}): sentence spans are asked from nltk")
            return None

    def native_handle(self):
        """leaf_dict_t of this word set (built once), or None when the tokenizer has no native restatement."""
        if self.kind is None:
            return None
        if self._native is None:
            import ctypes as C
            from . import _lib
            blob = "\n".join(self.words).encode("utf-8")
            h = C.c_void_p()
            _lib.check(_lib.lib().leaf_dict_create(blob, len(blob), C.byref(h)), "leaf_dict_create")
            self._native = h
        return self._native

    def __del__(self):
        try:
            if self._native is not None:
                from . import _lib
                _lib.lib().leaf_dict_destroy(self._native)
                self._native = None
        except Exception:
            pass

    @classmethod
    def from_file(cls, path: str, tokenizer: str = "regex", punkt_params: Optional[str] = None):
        """A word-list file (one word per line).  ``tokenizer='treebank'``: tokenise like nltk.word_tokenize WITHOUT nltk: the
        restated Treebank pipeline.  With ``punkt_params`` (the tables of nltk's trained Punkt model, exported once where nltk is
        installed: tools/export_punkt_params.py) sentences are split by the native restatement of Punkt first, as
        ``word_tokenize`` does; without it a text whose tokens would depend on the sentence model (a lone '.' ending a chunk
        inside the text) is tokenised as ONE sentence -- ``word_tokenize(text, preserve_line=True)``."""
        with open(path) as f:
            ws = [w.strip() for w in f if w.strip()]
        if tokenizer == "treebank":
            from .treebank import treebank_tokenize
            if punkt_params is None:
                return cls(ws, treebank_tokenize, kind="nltk")
            from .native_text import NativePunkt
            native = NativePunkt.from_json(punkt_params)
            strict_off = NativePunkt.from_json(punkt_params, strict=False)    # no nltk to ask: every text is decided (3.6.5 rules)

            def spans(t):
                sp = strict_off.spans(t)
                return sp if sp is not None else [(0, len(t))]

            d = cls(ws, lambda t: [w for a, b in spans(t) for w in treebank_tokenize(t[a:b])], kind="nltk")
            d.span_tokenize = spans
            d.punkt_native = native
            return d
        return cls(ws)

    def sentence_spans(self, sentences):
        """For the native constraint: per caption the sentence spans of its LOWER-CASED text, for the captions whose word tokens
        depend on where sentences end (treebank.punkt_free is False) and [] for the others; None when no sentence splitter is
        known.  One splitter call per such caption (cached: both stages of an edit ask for the same captions)."""
        if self.span_tokenize is None:
            return None
        from .treebank import punkt_free
        out = []
        for s in sentences:
            t = s.lower()
            sp = self._span_cache.get(t)
            if sp is None:
                sp = [] if punkt_free(t) else [(int(a), int(b)) for a, b in self.span_tokenize(t)]
                if len(self._span_cache) > 65536:
                    self._span_cache.clear()
                self._span_cache[t] = sp
            out.append(sp)
        return out

    def count_fast(self, sentence: str) -> int:
        """``count`` for the strings the native constraint declines: with a sentence splitter known (nltk's Punkt) the string costs one
        splitter call and a native count (leaf_tok_count_words) instead of a whole ``word_tokenize``; otherwise ``count``."""
        if self.kind != "nltk" or self.span_tokenize is None:
            return self.count(sentence)
        import ctypes as C
        from . import _lib
        from .treebank import punkt_free
        t = sentence.lower()
        try:
            raw = t.encode("ascii")
        except UnicodeEncodeError:
            return self.count(sentence)
        if punkt_free(t):
            sp = []
        else:
            sp = self.punkt_native.spans(t) if self.punkt_native is not None else None
            if sp is None:
                sp = self.span_tokenize(t)
            sp = [int(x) for se in sp for x in se]
        arr = (C.c_int32 * len(sp))(*sp) if sp else None
        n = C.c_int32()
        rc = _lib.lib().leaf_tok_count_words(self.native_handle(), 1, raw, len(raw), arr, len(sp) // 2, C.byref(n))
        return n.value if rc == 0 else self.count(sentence)

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
def _apply_edit(S: str, z: int, c: int) -> str:
    """generate_sentence(S, z, u, V, alternative=-1) with c = V[u] (code point or -1)."""
    if z & 1:
        i = (z - 1) // 2
        return S[:i] + S[i + 1:] if (c == -1 or S[i] == chr(c)) else S[:i] + chr(c) + S[i + 1:]
    i = z // 2
    return S if (c == -1 or c == 95) else S[:i] + chr(c) + S[i:]     # a slot holds '_' (95) in the expanded view


def _stage_candidates(tokenizer, sentences, z, c, constrain, trace):
    """Token ids / kept lengths of the B x rho single-edit candidates (z, c) of ``sentences``.
    With a ``NativeTokenizer`` the mutation + BPE of all candidates run in C++ threads
    (leaf_amd/csrc/host_text.cpp); strings are only materialised for --constrain, traces and fast-path misses."""
    B, rho = z.shape
    native = hasattr(tokenizer, "mutate_encode")
    native_constrain = constrain and native and hasattr(tokenizer, "constrain_mask") and get_dictionary().kind is not None
    if native_constrain:
        # --constrain without materialising strings: the C++ side re-tokenises only the word(s) around each edit and
        # compares distinct dictionary-word counts (leaf_tok_constrain); what it declines is decided here in Python
        D = get_dictionary()
        valid, fb = tokenizer.constrain_mask(D, sentences, z, c)
        lo = {}                                           # dictionary words of a sentence: counted once, not once per candidate
        for i in np.nonzero(fb.reshape(-1))[0]:
            b, r = divmod(int(i), rho)
            if b not in lo:
                lo[b] = D.count_fast(sentences[b])
            valid[b, r] = D.count_fast(_apply_edit(sentences[b], int(z[b, r]), int(c[b, r]))) < lo[b]
        z[~valid], c[~valid] = 0, -1                      # the no-op edit: candidate == original sentence
    need_strings = (constrain and not native_constrain) or trace is not None or not native
    SS = None
    if need_strings:
        SS = [[_apply_edit(S, int(z[b, r]), int(c[b, r])) for r in range(rho)] for b, S in enumerate(sentences)]
        if constrain and not native_constrain:
            valid = valid_sentence_batched(sentences, SS)
            for b, S in enumerate(sentences):
                for r in range(rho):
                    if not valid[b][r]:
                        SS[b][r] = S
                        z[b, r], c[b, r] = 0, -1          # the no-op edit: candidate == original sentence
        if trace is not None:
            trace.append([x for row in SS for x in row])
    if native:
        toks, lens = tokenizer.mutate_encode(sentences, z, c, lambda b, r: _apply_edit(sentences[b], int(z[b, r]), int(c[b, r])))
    else:
        toks = tokenizer.encode_batch([x for row in SS for x in row])
        lens = None
    return toks, lens


def _choice_range(pop: int, size: int, replace: bool) -> np.ndarray:
    """``np.random.choice(range(pop), size=size, replace=replace)`` on the GLOBAL numpy RNG, drawing exactly the same numbers from
    exactly the same stream: numpy's legacy ``RandomState.choice`` without ``p`` is ``permutation(pop)[:size]`` (no replacement) or
    ``randint(0, pop, size)`` (with), and indexing ``range(pop)`` with the result is the identity.  The direct forms skip the
    range -> array conversion, which costs more than the draw (2 x 128 calls per search: several ms of a 33-ms search);
    ``tests/test_host_cpu.py::test_rng_draws_are_stream_identical`` compares the streams, the reference traces replay through it."""
    if replace:
        return np.random.randint(0, pop, size=size)
    return np.random.permutation(pop)[:size]


def duplicate_map(toks: np.ndarray, B: int, n: int) -> np.ndarray:
    """dup_of[b, r] = the first r' <= r whose token row equals candidate r's (r itself when the row is new).  The CLIP
    tokenizer lower-cases and collapses whitespace (src/open_clip/tokenizer.py:83-85,139), so stage 2's 'a' / 'A', a space
    inserted next to a space and with-replacement draws (utils_attacks.py:317: replace = rho > 2 len + 1) all give identical
    id rows -- and identical losses: only the first needs computing."""
    from . import _lib
    t = np.ascontiguousarray(toks.reshape(B, n, -1), dtype=np.int32)
    dup = np.empty((B, n), dtype=np.int32)
    rc = _lib.lib().leaf_tok_duplicate_map(t.ctypes.data, B, n, t.shape[-1], dup.ctypes.data, 1)
    if rc != 0:
        raise _lib.LeafHipError(f"leaf_tok_duplicate_map failed ({rc})")
    return dup.astype(np.int64)


def attack_text_leaf(model, tokenizer, sentences, anchor_features, device=None, objective="l2", n=10, k=1,
                     V=DEFAULT_V, constrain=False, debug=False, return_trace: Optional[list] = None,
                     return_picks: Optional[list] = None, dedupe: bool = True, pipeline: Optional[int] = None,
                     anchor_ready=None):
    """LEAF attack on a batch of sentences.  ``model`` is a ``leaf_amd.model.LeafCLIPText`` (anything with
    ``score_candidates``); ``anchor_features`` a float32 CUDA tensor [B, D].  Returns
    ``(best_features [B,D], adversarial sentences)`` like the reference.  The numpy global RNG is consumed exactly as
    the reference does (one draw per sentence and stage, utils_attacks.py:317,236).

    ``pipeline`` (default 2 for B >= 64 with prefix reuse, else 1): the captions are handled in that many contiguous groups
    whose stages are interleaved -- while the GPU scores one group's candidates the host mutates, tokenises and constrains the
    next group's (the GPU used to idle for every stage's host preparation: a quarter of a 33-ms search).  Captions are
    independent and every row has the same bits whichever launch computes it, so the result does not depend on the grouping
    (``tests/test_gpu_forward.py::test_token_identical_candidates_are_computed_once``); all random draws of an edit are made up
    front in the reference's order (stage-1 positions for every sentence, then stage-2 characters for every sentence: the
    reference draws nothing in between).  ``anchor_ready``: a CUDA event after which ``anchor_features`` is valid (the caller
    computed it on a side stream); the first scoring launch waits for it, the clean captions' K/V pass before it does not."""
    import torch
    sentences = list(sentences)
    B = len(sentences)
    def wait_anchor():
        """the caller computed ``anchor_features`` on a side stream: order the current stream behind it, once"""
        nonlocal anchor_ready
        if anchor_ready is not None:
            torch.cuda.current_stream().wait_event(anchor_ready)
            anchor_ready = None

    if objective in ("dissim", "sim"):
        wait_anchor()
        anchor_features /= anchor_features.norm(dim=-1, keepdim=True)   # in place, as the reference does
    Varr = np.asarray(V, dtype=np.int32)
    reuse = hasattr(model, "encode_text_kv") and getattr(model, "trim_rows", False)
    dedupe = dedupe and reuse
    if pipeline is None:
        pipeline = 2 if (reuse and B >= 64) else 1
    if return_trace is not None or not reuse:
        pipeline = 1                      # a trace lists a stage's candidates of ALL sentences in one piece
    pipeline = max(1, min(int(pipeline), B))
    bounds = np.linspace(0, B, pipeline + 1).astype(int)
    groups = [np.arange(bounds[g], bounds[g + 1]) for g in range(pipeline)]
    dev = anchor_features.device

    def prefix_lens(toks, base, Bg):
        """leading positions where a candidate's ids equal its clean caption's (>= 1: SOT)"""
        neq = toks.reshape(Bg, n, -1) != base[:, None, :]
        first = neq.argmax(-1)
        first[~neq.any(-1)] = toks.shape[-1]
        return first.reshape(-1)

    def launch(toks, lens, base, kv, anchor_g, want_features):
        """Queue one stage's scoring of one group; returns finish() -> (winner indices [Bg] on the host, winner features or None).
        The indices travel to pinned host memory right behind the scoring launches and finish() waits for THAT copy's event
        only, so work queued later (the next group's scoring) does not delay it.  With ``dedupe`` a candidate whose id row
        repeats an earlier candidate's of the same caption is not computed again: its slot is handed to the no-op edit (the
        clean caption: ONE row under prefix reuse), the loss of its first occurrence is copied into it and the arg-max runs over
        the completed [Bg, rho] losses -- first index wins, so the first occurrence beats its copies exactly as in torch.argmax
        over the reference's full loss matrix (utils_attacks.py:348,386)."""
        wait_anchor()
        Bg = anchor_g.shape[0]
        pl = prefix_lens(toks, base, Bg) if reuse else None
        dup = duplicate_map(toks, Bg, n) if dedupe else None
        has_dup = dup is not None and bool((dup != np.arange(n)[None, :]).any())
        if not has_dup:
            ids, feat = model.score_candidates(toks, anchor_g, n, objective, want_features=want_features, seq_lens=lens,
                                               prefix_lens=pl, kv=kv)
            both = ids.to(torch.int64)[None, :]
        else:
            is_dup = (dup != np.arange(n)[None, :]).reshape(-1)
            toks = toks.copy().reshape(Bg * n, -1)
            rep = np.repeat(np.arange(Bg), n)[is_dup]
            toks[is_dup] = base[rep]
            if lens is not None:
                lens = lens.copy()
                lens[is_dup] = (base.argmax(-1) + 1)[rep]
            pl = pl.copy()
            pl[is_dup] = toks.shape[-1]                          # nothing differs from the clean caption
            dup_dev = torch.from_numpy(dup).pin_memory().to(dev, non_blocking=True)
            ids_k, feat, loss = model.score_candidates(toks, anchor_g, n, objective, want_features=want_features, want_loss=True,
                                                       seq_lens=lens, prefix_lens=pl, kv=kv)
            ids = loss.gather(1, dup_dev).argmax(-1)             # first maximum wins
            both = torch.stack([ids.to(torch.int64), ids_k.to(torch.int64)])
        host = torch.empty(both.shape, dtype=torch.int64, pin_memory=True)
        host.copy_(both, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()

        def finish():
            nonlocal feat
            ev.synchronize()
            h = host.numpy()
            ids_h = h[0].copy()
            if has_dup and want_features and (ids_h != h[1]).any():
                # (only when the clean caption itself out-scored every candidate in a slot it was lent: re-encode the winners;
                # rows are bit-identical whichever launch computes them)
                feat = model.encode_text(toks.reshape(Bg, n, -1)[np.arange(Bg), ids_h])
            return ids_h, feat
        return finish

    best_feat = None
    for _ in range(k):
        base_all = tokenizer.encode_batch(sentences) if reuse else None
        # every draw of this edit, in the reference's order: stage-1 positions (utils_attacks.py:317), stage-2 characters (:236)
        z1 = np.stack([_choice_range(2 * len(S) + 1, n, n > 2 * len(S) + 1) for S in sentences]).astype(np.int32)
        u2 = np.stack([_choice_range(len(V), n, n > len(V)) for _ in sentences])
        kvs, st1 = [], []
        for gi, g in enumerate(groups):
            sg = [sentences[b] for b in g]
            base = base_all[g] if reuse else None
            # clean captions once per edit: their per-layer K/V serve both stages' candidates (queued first: it runs while the
            # host prepares the first stage)
            kv = model.encode_text_kv(base, slot=gi) if reuse else None
            kvs.append((sg, base, kv))
            # stage 1: rho random positions, insert / replace-with / delete a space
            z = z1[g].copy()
            c = np.full((len(g), n), ord(' '), dtype=np.int32)
            positions = z.copy()        # the reference reads the winner's position from the sampled ones (:351-353)
            toks, lens = _stage_candidates(tokenizer, sg, z, c, constrain, return_trace)
            st1.append((positions, launch(toks, lens, base, kv, anchor_features[g[0]:g[-1] + 1], False)))
        st2, picks1 = [], []
        for gi, g in enumerate(groups):
            sg, base, kv = kvs[gi]
            positions, fin = st1[gi]
            ids_best, _ = fin()
            picks1.append(ids_best)
            best_pos = positions[np.arange(len(g)), ids_best]
            # stage 2: rho random characters at the chosen position
            c = Varr[u2[g]]
            z = np.repeat(best_pos[:, None], n, axis=1).astype(np.int32)
            toks, lens = _stage_candidates(tokenizer, sg, z, c, constrain, return_trace)
            st2.append((z, c, launch(toks, lens, base, kv, anchor_features[g[0]:g[-1] + 1], True)))
        if return_picks is not None:
            return_picks.append(np.concatenate(picks1))
        new_sentences, feats, picks2 = [], [], []
        for gi, g in enumerate(groups):
            sg = kvs[gi][0]
            z, c, fin = st2[gi]
            ids_best, f = fin()
            picks2.append(ids_best)
            feats.append(f)
            new_sentences += [_apply_edit(S, int(z[b, ids_best[b]]), int(c[b, ids_best[b]])) for b, S in enumerate(sg)]
        if return_picks is not None:
            return_picks.append(np.concatenate(picks2))
        sentences = new_sentences
        best_feat = feats[0] if len(feats) == 1 else torch.cat(feats, 0)
        if debug:
            print(sentences[0])
    wait_anchor()      # k == 0: nothing was scored, but the caller's next use of the anchor is on this stream
    return best_feat, sentences


def attack_text(model, tokenizer, sentences, image_features, device=None, objective="l2", n=10, k=1, V=DEFAULT_V,
                constrain=False, debug=False, anchor_ready=None):
    return attack_text_leaf(model, tokenizer, sentences, image_features, device, objective, n, k, V, constrain, debug,
                            anchor_ready=anchor_ready)


# ----------------------------------------------------------------------------- optional embedding-space PGD (a12)
def attack_embedding_pgd(model, tokens, anchor_features, eps: float, alpha: float, k: int = 1, norm: str = "linf",
                         delta0=None, seq_lens=None, seed: Optional[int] = None):
    """Continuous attack on the token EMBEDDINGS: k steps of  delta <- project(delta + alpha * normalize_grad(grad))
    maximising sum((anchor - f(tokens; delta))**2).  NOT part of the reference's text trainer;
    the loop is the reference's continuous attack (utils_attacks.py:680-697) with the embedding-input forward of
    src/pez/open_clip_pez/model.py:210-228 and the normalise / project pair of src/robust_vlm/train/utils.py:96-114.
    ``delta0``: packed fp32 [rows, width] start (default eps * U(-1, 1), utils_attacks.py:680, projected for 'l2').
    Returns (features of the perturbed captions [B, D], delta) - delta stays resident on the device throughout."""
    import torch
    if seq_lens is None:
        arr = tokens if isinstance(tokens, np.ndarray) else tokens.cpu().numpy()
        seq_lens = arr.reshape(-1, arr.shape[-1]).argmax(-1) + 1
    lens = np.asarray(seq_lens).reshape(-1)
    rows = int(lens.sum()) if getattr(model, "trim_rows", False) else lens.size * model.cfg.context_length
    if delta0 is None:
        gen = torch.Generator(device=model.device)
        gen.manual_seed(0 if seed is None else seed)
        delta = eps * (2 * torch.rand(rows, model.cfg.width, device=model.device, generator=gen) - 1)
    else:
        delta = delta0.to(device=model.device, dtype=torch.float32).contiguous().clone()
    was_training = getattr(model, "training", False)
    model.eval()
    feat = model.forward_train(tokens, seq_lens=seq_lens, delta=delta)
    if delta0 is None and norm == "l2":      # bring the random start into the ball: a zero-gradient step projects only
        model.pgd_step(delta, torch.zeros_like(delta), 0.0, eps, "l2")
        feat = model.forward_train(tokens, seq_lens=seq_lens, delta=delta)
    for _ in range(k):
        _, grad = model.input_grad(feat, anchor_features)
        model.pgd_step(delta, grad, alpha, eps, norm)
        feat = model.forward_train(tokens, seq_lens=seq_lens, delta=delta)
    if was_training:
        model.train()
    return feat, delta
