"""CLIP byte-pair tokenizer, host side of the LEAF text path.

Drop-in for the reference's ``SimpleTokenizer`` (src/open_clip/tokenizer.py:133-265): same cleaning
(``ftfy.fix_text`` when available, double ``html.unescape``, whitespace collapse, lower-case :66-85),
same regex split (:160-163), same greedy lowest-rank-first merges (:172-211), same framing
``[SOT] + ids + [EOT]``, zero padding to 77, truncation with the last id forced to EOT (:256-263).
Written independently around an explicit rank table; reads the OpenAI merge file in ``leaf_amd/data``.
"""
from __future__ import annotations

import gzip
import html
import os
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import numpy as np
import regex

try:  # the reference cleans with ftfy; it is an identity on ASCII text, which is all we can check here
    import ftfy as _ftfy
except Exception:  # pragma: no cover - ftfy is absent in the build image
    _ftfy = None

DEFAULT_CONTEXT_LENGTH = 77
_BPE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bpe_simple_vocab_16e6.txt.gz")
_END = "</w>"


def _byte_alphabet() -> Dict[int, str]:
    """The GPT-2 reversible byte <-> printable-unicode table (tokenizer.py:31-51)."""
    printable = [b for rng in ((33, 126), (161, 172), (174, 255)) for b in range(rng[0], rng[1] + 1)]
    table, extra = {}, 0
    for b in printable:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    # the vocabulary order is: printable bytes first, then the remapped ones in byte order
    order = printable + [b for b in range(256) if b not in printable]
    return {b: table[b] for b in order}


class SimpleTokenizer:
    def __init__(self, bpe_path: str = _BPE_PATH, context_length: int = DEFAULT_CONTEXT_LENGTH):
        self.byte_encoder = _byte_alphabet()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        with gzip.open(bpe_path) as f:
            lines = f.read().decode("utf-8").split("\n")
        n_merges = 49152 - 256 - 2
        merges: List[Tuple[str, str]] = [tuple(l.split()) for l in lines[1:1 + n_merges]]
        symbols = list(self.byte_encoder.values())
        vocab = symbols + [s + _END for s in symbols] + ["".join(m) for m in merges]
        vocab += ["<start_of_text>", "<end_of_text>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.vocab_size = len(vocab)
        self.sot_token_id = self.encoder["<start_of_text>"]
        self.eot_token_id = self.encoder["<end_of_text>"]
        self.all_special_ids = [self.sot_token_id, self.eot_token_id]
        self.context_length = context_length
        self._word_cache: Dict[str, Tuple[int, ...]] = {
            "<start_of_text>": (self.sot_token_id,), "<end_of_text>": (self.eot_token_id,)}
        self._splitter = regex.compile(
            r"<start_of_text>|<end_of_text>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
            regex.IGNORECASE)

    # ------------------------------------------------------------------ cleaning
    @staticmethod
    def clean(text: str) -> str:
        if _ftfy is not None:
            text = _ftfy.fix_text(text)
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).strip().lower()

    # ------------------------------------------------------------------ BPE on one pre-token
    def _merge_word(self, word: str) -> Tuple[int, ...]:
        hit = self._word_cache.get(word)
        if hit is not None:
            return hit
        parts = list(word[:-1]) + [word[-1] + _END]
        rank = self.rank
        while len(parts) > 1:
            best, best_rank = None, None
            for pair in zip(parts, parts[1:]):
                r = rank.get(pair)
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = pair, r
            if best is None:
                break
            a, b = best
            merged, i, n = [], 0, len(parts)
            while i < n:
                if i + 1 < n and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        ids = tuple(self.encoder[p] for p in parts)
        if len(self._word_cache) > (1 << 20):      # the search produces millions of one-off mutated words: keep it bounded
            self._word_cache = {"<start_of_text>": (self.sot_token_id,), "<end_of_text>": (self.eot_token_id,)}
        self._word_cache[word] = ids
        return ids

    def encode(self, text: str) -> List[int]:
        out: List[int] = []
        benc = self.byte_encoder
        for piece in self._splitter.findall(self.clean(text)):
            out.extend(self._merge_word("".join(benc[b] for b in piece.encode("utf-8"))))
        return out

    def decode(self, tokens: Iterable[int]) -> str:
        text = "".join(self.decoder[int(t)] for t in tokens)
        raw = bytearray(self.byte_decoder[c] for c in text)
        return raw.decode("utf-8", errors="replace").replace(_END, " ")

    # ------------------------------------------------------------------ batches
    def encode_batch(self, texts: Union[str, Sequence[str]], context_length: int = None) -> np.ndarray:
        """int32 [n, context_length] -- the wire format of the HIP engine."""
        if isinstance(texts, str):
            texts = [texts]
        L = context_length or self.context_length
        out = np.zeros((len(texts), L), dtype=np.int32)
        for i, t in enumerate(texts):
            ids = [self.sot_token_id] + self.encode(t) + [self.eot_token_id]
            if len(ids) > L:
                ids = ids[:L]
                ids[-1] = self.eot_token_id
            out[i, :len(ids)] = ids
        return out

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = None):
        """Reference signature: LongTensor [n, 77] on the CPU (tokenizer.py:226)."""
        import torch
        return torch.from_numpy(self.encode_batch(texts, context_length).astype(np.int64))


_default = None


def get_tokenizer(model_name: str = "", context_length: int = DEFAULT_CONTEXT_LENGTH) -> SimpleTokenizer:
    """open_clip.get_tokenizer equivalent for the CLIP text towers in scope (all use the same BPE)."""
    global _default
    if _default is None or _default.context_length != context_length:
        _default = None
        if os.environ.get("LEAF_NATIVE_HOST", "1") != "0":
            # C++ threads for the batch paths (leaf_amd/csrc/host_text.cpp); same results.  No silent fallback: if the
            # library is missing or fails to load this raises (LEAF_NATIVE_HOST=0 selects the Python tokenizer on purpose).
            from .native_text import NativeTokenizer
            _default = NativeTokenizer(context_length=context_length)
        else:
            _default = SimpleTokenizer(context_length=context_length)
    return _default
