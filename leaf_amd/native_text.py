"""ctypes wrapper of the native host pipeline (leaf_amd/csrc/host_text.cpp): multithreaded CLIP BPE and fused
single-edit mutation + tokenisation of a whole search stage.  Inputs outside the native fast path (non-ASCII, '&'
entities) are routed to the Python tokenizer, so results are always those of ``SimpleTokenizer``."""
from __future__ import annotations

import ctypes as C
import gzip
import os
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from .tokenizer import _BPE_PATH, SimpleTokenizer


class NativeTokenizer(SimpleTokenizer):
    """Drop-in ``SimpleTokenizer`` whose batch paths run in C++ threads."""

    def __init__(self, bpe_path: str = _BPE_PATH, context_length: int = 77, n_threads: int = None):
        super().__init__(bpe_path, context_length)
        self._lib = _lib.lib()
        with gzip.open(bpe_path) as f:
            text = f.read()
        h = C.c_void_p()
        rc = self._lib.leaf_tok_create(text, len(text), C.byref(h))
        if rc != 0:
            raise _lib.LeafHipError(f"leaf_tok_create failed ({rc})")
        self._h = h
        # per RANK: the ranks of a node share its cores (LOCAL_WORLD_SIZE is set by torch.distributed.run)
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        share = max(1, usable // max(int(os.environ.get("LOCAL_WORLD_SIZE", "1")), 1))
        self.n_threads = n_threads or int(os.environ.get("LEAF_HOST_THREADS", str(min(32, share))))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.leaf_tok_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def _c_strings(texts: Sequence[str]):
        raw = [t.encode("utf-8") for t in texts]
        arr = (C.c_char_p * len(raw))(*raw)
        lens = np.array([len(r) for r in raw], dtype=np.int32)
        return raw, arr, lens

    def encode_batch(self, texts, context_length: int = None) -> np.ndarray:
        return self.encode_batch_lens(texts, context_length)[0]

    def encode_batch_lens(self, texts, context_length: int = None) -> Tuple[np.ndarray, np.ndarray]:
        """tokens int32 [n, ctx] and kept lengths int32 [n] (EOT position + 1)."""
        if isinstance(texts, str):
            texts = [texts]
        L = context_length or self.context_length
        n = len(texts)
        raw, arr, blen = self._c_strings(texts)
        toks = np.zeros((n, L), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        fb = np.zeros(n, dtype=np.uint8)
        rc = self._lib.leaf_tok_encode_batch(self._h, arr, blen.ctypes.data, n, L, toks.ctypes.data, lens.ctypes.data,
                                             fb.ctypes.data, self.n_threads)
        if rc != 0:
            raise _lib.LeafHipError(f"leaf_tok_encode_batch failed ({rc})")
        for i in np.nonzero(fb)[0]:
            toks[i] = SimpleTokenizer.encode_batch(self, [texts[i]], L)[0]
            lens[i] = int(toks[i].argmax()) + 1
        return toks, lens

    def mutate_encode(self, sentences: Sequence[str], z: np.ndarray, c: np.ndarray, make_candidate) -> Tuple[np.ndarray, np.ndarray]:
        """Tokens/lengths of the B x rho candidates ``generate_sentence(S_b, z[b,r], ., alternative=-1)`` with
        replacement code points ``c[b,r]`` (-1 = delete).  ``make_candidate(b, r)`` builds the string in Python for
        the candidates the native fast path declines."""
        B, rho = z.shape
        L = self.context_length
        raw, arr, blen = self._c_strings(sentences)
        ascii_len_ok = np.array([len(s) == len(r) for s, r in zip(sentences, raw)])
        toks = np.zeros((B * rho, L), dtype=np.int32)
        lens = np.zeros(B * rho, dtype=np.int32)
        fb = np.zeros(B * rho, dtype=np.uint8)
        zz = np.ascontiguousarray(z, dtype=np.int32)
        cc = np.ascontiguousarray(c, dtype=np.int32)
        rc = self._lib.leaf_tok_mutate_encode(self._h, arr, blen.ctypes.data, B, zz.ctypes.data, cc.ctypes.data, rho, L,
                                              toks.ctypes.data, lens.ctypes.data, fb.ctypes.data, self.n_threads)
        if rc not in (0, 3):
            raise _lib.LeafHipError(f"leaf_tok_mutate_encode failed ({rc})")
        for i in np.nonzero(fb)[0]:
            toks[i] = SimpleTokenizer.encode_batch(self, [make_candidate(i // rho, i % rho)], L)[0]
            lens[i] = int(toks[i].argmax()) + 1
        return toks, lens

    def constrain_mask(self, dictionary, sentences: Sequence[str], z: np.ndarray, c: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """--constrain validity of the B x rho single-edit candidates (strictly fewer distinct dictionary words than the
        sentence, utils_attacks.py:110-143) computed natively: (valid bool [B, rho], fallback bool [B, rho]); entries with
        fallback set were declined by the native path (non-ASCII, or text outside nltk's whitespace-split regime) and must
        be decided with ``dictionary.count``."""
        B, rho = z.shape
        raw, arr, blen = self._c_strings(sentences)
        valid = np.zeros((B, rho), dtype=np.uint8)
        fb = np.zeros((B, rho), dtype=np.uint8)
        zz = np.ascontiguousarray(z, dtype=np.int32)
        cc = np.ascontiguousarray(c, dtype=np.int32)
        kind = {"regex": 0, "nltk": 1}[dictionary.kind]
        punkt = getattr(dictionary, "punkt_native", None) if kind == 1 else None
        if punkt is not None:
            # the splitter's tables are on the native side: spans of captions and candidates are computed there
            rc = self._lib.leaf_tok_constrain_punkt(dictionary.native_handle(), punkt._h, arr, blen.ctypes.data, B, zz.ctypes.data,
                                                    cc.ctypes.data, rho, valid.ctypes.data, fb.ctypes.data, self.n_threads)
            if rc not in (0, 3):
                raise _lib.LeafHipError(f"leaf_tok_constrain_punkt failed ({rc})")
            for b, (s_, r_) in enumerate(zip(sentences, raw)):
                if len(s_) != len(r_):
                    fb[b] = 1
            return valid.astype(bool), fb.astype(bool)
        spans = dictionary.sentence_spans(sentences) if kind == 1 and hasattr(dictionary, "sentence_spans") else None
        if spans is not None and any(spans):
            # captions whose tokens depend on where sentences end: nltk's Punkt was asked once per caption (not once per
            # candidate); the native code tokenises them sentence by sentence
            off = np.zeros(B + 1, dtype=np.int32)
            np.cumsum([len(sp) for sp in spans], out=off[1:])
            flat = np.ascontiguousarray([x for sp in spans for se in sp for x in se], dtype=np.int32)
            rc = self._lib.leaf_tok_constrain_ranges(dictionary.native_handle(), kind, arr, blen.ctypes.data, B, zz.ctypes.data,
                                                     cc.ctypes.data, rho, flat.ctypes.data, off.ctypes.data, valid.ctypes.data,
                                                     fb.ctypes.data, self.n_threads)
        else:
            rc = self._lib.leaf_tok_constrain(dictionary.native_handle(), kind, arr, blen.ctypes.data, B, zz.ctypes.data, cc.ctypes.data,
                                              rho, valid.ctypes.data, fb.ctypes.data, self.n_threads)
        if rc not in (0, 3):
            raise _lib.LeafHipError(f"leaf_tok_constrain failed ({rc})")
        for b, (s_, r_) in enumerate(zip(sentences, raw)):      # byte offsets == character offsets only for ASCII sentences
            if len(s_) != len(r_):
                fb[b] = 1
        return valid.astype(bool), fb.astype(bool)


class NativePunkt:
    """nltk's Punkt sentence splitter restated in C++ (host_text.cpp ``punkt_spans``): ``span_tokenize`` over the four parameter
    tables of a trained model (``PunktParameters``: abbrev_types, collocations, sent_starters, ortho_context).  ASCII text without
    control characters; ``spans`` returns None for anything else (ask nltk)."""

    def __init__(self, abbrev_types=(), collocations=(), sent_starters=(), ortho_context=None, strict: bool = True):
        """strict (default): texts on which nltk generations may differ -- two candidate break positions inside one
        whitespace-delimited chunk, "what?! yes" -- are declined; False decides them as nltk 3.6.5 does."""
        self._lib = _lib.lib()
        blobs = ["\n".join(sorted(abbrev_types)), "\n".join(f"{a}\t{b}" for a, b in sorted(collocations)), "\n".join(sorted(sent_starters)),
                 "\n".join(f"{k}\t{int(v)}" for k, v in sorted((ortho_context or {}).items()) if v)]
        entries = list(abbrev_types) + [x for c in collocations for x in c] + list(sent_starters) + list(ortho_context or {})
        if any("\n" in x or "\t" in x for x in entries):
            raise ValueError("Punkt table entries with tabs / newlines")
        raw = [b.encode("utf-8") for b in blobs]
        h = C.c_void_p()
        rc = self._lib.leaf_punkt_create(raw[0], len(raw[0]), raw[1], len(raw[1]), raw[2], len(raw[2]), raw[3], len(raw[3]), C.byref(h))
        if rc != 0:
            raise _lib.LeafHipError(f"leaf_punkt_create failed ({rc})")
        self._h = h
        self._lib.leaf_punkt_set_strict(h, int(strict))
        self.strict = bool(strict)
        self.sizes = (len(abbrev_types), len(collocations), len(sent_starters), len(ortho_context or {}))

    def set_strict(self, strict: bool):
        self._lib.leaf_punkt_set_strict(self._h, int(strict))
        self.strict = bool(strict)

    @classmethod
    def from_nltk(cls, punkt):
        """From a ``PunktSentenceTokenizer`` (the instance nltk.sent_tokenize uses).  Raises when the instance does not use the stock
        language variables (another language's subclass changes the regular expressions this restates)."""
        lv = getattr(punkt, "_lang_vars", None)
        if lv is not None and (tuple(lv.sent_end_chars) != (".", "?", "!") or lv.internal_punctuation != ",:;"
                               or lv._re_word_start != r"[^\(\"\`{\[:;&\#\*@\)}\]\-,]" or lv._re_multi_char_punct != r"(?:\-{2,}|\.{2,}|(?:\.\s){2,}\.)"):
            raise ValueError("non-default PunktLanguageVars")
        p = punkt._params
        return cls(p.abbrev_types, p.collocations, p.sent_starters, dict(p.ortho_context))

    @classmethod
    def from_json(cls, path: str, strict: bool = True):
        """Tables exported with tools/export_punkt_params.py (``{"abbrev_types": [...], "collocations": [[a, b], ...],
        "sent_starters": [...], "ortho_context": {type: flags}}``): nltk's sentence splitter where nltk is not installed."""
        import json
        with open(path) as f:
            d = json.load(f)
        return cls(d["abbrev_types"], [tuple(c) for c in d["collocations"]], d["sent_starters"], d["ortho_context"], strict=strict)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.leaf_punkt_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def spans(self, text: str):
        try:
            raw = text.encode("ascii")
        except UnicodeEncodeError:
            return None
        cap = len(raw) // 2 + 2
        out = np.zeros(2 * cap, dtype=np.int32)
        n = C.c_int32()
        rc = self._lib.leaf_punkt_spans(self._h, raw, len(raw), out.ctypes.data, cap, C.byref(n))
        if rc == 2:
            return None
        if rc != 0:
            raise _lib.LeafHipError(f"leaf_punkt_spans failed ({rc})")
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n.value)]
