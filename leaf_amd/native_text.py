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
