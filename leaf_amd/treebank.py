"""The word tokenizer behind ``--constrain``: ``nltk.word_tokenize`` (utils_attacks.py:110-143 apply it to the lower-cased
sentence and to every candidate).

``nltk`` is a third-party dependency of the reference that is NOT vendored in it (requirements.txt:14, unpinned; the reference
downloads the ``punkt_tab`` model at import, utils_attacks.py:7-9, i.e. it expects nltk >= 3.8.2).  ``word_tokenize`` is
``sent_tokenize`` (the trained Punkt sentence splitter) followed, per sentence, by ``NLTKWordTokenizer.tokenize``: a fixed
pipeline of regular-expression substitutions (nltk/tokenize/destructive.py; Penn-Treebank conventions plus unicode quotes,
MacIntyre's contractions) and a final ``str.split()``.  This module restates that published pipeline -- rule for rule, in the
order nltk applies them -- for two purposes:

* it is the checker of the native, window-local implementation in ``csrc/host_text.cpp`` (``leaf_tok_constrain`` kind 1);
* where nltk is not installed it stands in for ``word_tokenize`` together with a word-list file (``Dictionary.from_file(...,
  tokenizer="treebank")``).

Pinned by ``tests/golden/treebank_kat.json``: 1,800 strings tokenised by the REAL nltk (3.6.5, found in the build container under
/opt/conda; ``tests/golden/make_golden_treebank.py``), and re-checked at run time against whichever nltk is installed where
training runs (``self_check``; a mismatch switches the native path off).

Punkt, the sentence splitter in front of this step, is restated natively (csrc/host_text.cpp ``punkt_spans``: the algorithm is
fixed, the trained model only fills four parameter tables, which ``native_text.NativePunkt`` takes from the installed nltk or from
a file exported by tools/export_punkt_params.py).  Its decisions can change the tokens only through the rules that are anchored at
the end of a sentence string -- the final-period rules -- i.e. only for a lone '.' that ends a whitespace-delimited chunk somewhere
INSIDE the text ("... cat. a dog ..."): whether "cat." stays one token depends on whether a sentence ends there.
``punkt_free(text)`` is False exactly for such texts; only they need sentence spans at all (``word_tokenize`` below raises for
them: it has no splitter).  '?' and '!' are always split off, wherever sentences end.
"""
from __future__ import annotations

import re
from typing import List

# --- the substitution pipeline of nltk.tokenize.destructive.NLTKWordTokenizer (3.6.5 ... 3.9.x: the rules that decide
# letter-bearing tokens are the same; later versions only reorder the two closing-quote rules and add a whitespace collapse)
_STARTING_QUOTES = [
    (re.compile("([«“‘„]|[`]+)", re.U), r" \1 "),
    (re.compile(r"^\""), r"``"),
    (re.compile(r"(``)"), r" \1 "),
    (re.compile(r"([ \(\[{<])(\"|\'{2})"), r"\1 `` "),
    (re.compile(r"(?i)(\')(?!re|ve|ll|m|t|s|d|n)(\w)\b", re.U), r"\1 \2"),
]
_PUNCTUATION = [
    (re.compile(r'([^\.])(\.)([\]\)}>"\'' "»”’ " r"]*)\s*$", re.U), r"\1 \2 \3 "),
    (re.compile(r"([:,])([^\d])"), r" \1 \2"),
    (re.compile(r"([:,])$"), r" \1 "),
    (re.compile(r"\.{2,}", re.U), r" \g<0> "),
    (re.compile(r"[;@#$%&]"), r" \g<0> "),
    (re.compile(r'([^\.])(\.)([\]\)}>"\']*)\s*$'), r"\1 \2\3 "),
    (re.compile(r"[?!]"), r" \g<0> "),
    (re.compile(r"([^'])' "), r"\1 ' "),
    (re.compile(r"[*]", re.U), r" \g<0> "),
]
_PARENS_BRACKETS = (re.compile(r"[\]\[\(\)\{\}\<\>]"), r" \g<0> ")
_DOUBLE_DASHES = (re.compile(r"--"), r" -- ")
_ENDING_QUOTES = [
    (re.compile("([»”’])", re.U), r" \1 "),
    (re.compile(r'"'), " '' "),
    (re.compile(r"(\S)(\'\')"), r"\1 \2 "),
    (re.compile(r"([^' ])('[sS]|'[mM]|'[dD]|') "), r"\1 \2 "),
    (re.compile(r"([^' ])('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) "), r"\1 \2 "),
]
_CONTRACTIONS2 = [re.compile(p) for p in (
    r"(?i)\b(can)(?#X)(not)\b", r"(?i)\b(d)(?#X)('ye)\b", r"(?i)\b(gim)(?#X)(me)\b", r"(?i)\b(gon)(?#X)(na)\b",
    r"(?i)\b(got)(?#X)(ta)\b", r"(?i)\b(lem)(?#X)(me)\b", r"(?i)\b(more)(?#X)('n)\b", r"(?i)\b(wan)(?#X)(na)\s")]
_CONTRACTIONS3 = [re.compile(p) for p in (r"(?i) ('t)(?#X)(is)\b", r"(?i) ('t)(?#X)(was)\b")]


def treebank_tokenize(text: str) -> List[str]:
    """``NLTKWordTokenizer().tokenize(text)``: one sentence string in, tokens out."""
    for rx, sub in _STARTING_QUOTES:
        text = rx.sub(sub, text)
    for rx, sub in _PUNCTUATION:
        text = rx.sub(sub, text)
    text = _PARENS_BRACKETS[0].sub(_PARENS_BRACKETS[1], text)
    text = _DOUBLE_DASHES[0].sub(_DOUBLE_DASHES[1], text)
    text = " " + text + " "
    for rx, sub in _ENDING_QUOTES:
        text = rx.sub(sub, text)
    for rx in _CONTRACTIONS2:
        text = rx.sub(r" \1 \2 ", text)
    for rx in _CONTRACTIONS3:
        text = rx.sub(r" \1 \2 ", text)
    return text.split()


_CLOSERS = "])}>\"'"
# what may directly follow a sentence-final character for Punkt to consider a sentence break there WITHOUT whitespace
# (nltk/tokenize/punkt.py, PunktLanguageVars._re_non_word_chars: (?:[?!)";}\]\*:@'\({\[]))
_PUNKT_NONWORD = "?!)\";}]*:@'({["


def punkt_free(text: str) -> bool:
    """True when Punkt's sentence boundaries cannot change ``word_tokenize(text)`` (module docstring).  Punkt looks for breaks at
    ``\\S*[.?!](?=NONWORD | \\s+\\S+)``; a break changes the Treebank tokens only through a lone '.', so the text is safe when every
    lone '.' (not part of a '..' run) is either followed by a character that is neither blank nor in Punkt's NONWORD set (inside a
    word: "3.50", "a.b"), or is the final period of the whole text (closing brackets / quotes directly behind it, then only blanks).  (Where a '?' or '!' ends
    a sentence in front of a quote, Punkt can also turn a closing quote token into an opening one -- never a letter-bearing token.)"""
    n = len(text)
    i = 0
    while i < n:
        if text[i] != '.':
            i += 1
            continue
        j = i
        while j < n and text[j] == '.':
            j += 1
        if j - i == 1 and j < n and (text[j].isspace() or text[j] in _PUNKT_NONWORD):
            # ... unless it is the text's final period: closers directly behind it, then nothing but blanks (closers AFTER a blank do
            # not count: Treebank would turn a detached '"' into an opening quote before it looks for the final period)
            k = j
            while k < n and text[k] in _CLOSERS:
                k += 1
            if text[k:].strip() != "":
                return False
        i = j
    return True


def word_tokenize(text: str) -> List[str]:
    """``nltk.word_tokenize(text)`` for texts on which Punkt cannot matter; raises for the others (use the real nltk)."""
    if not punkt_free(text):
        raise ValueError("sentence-boundary dependent text: needs nltk's Punkt model")
    return treebank_tokenize(text)


SELF_CHECK_STRINGS = [
    "a photo of a cat", "hello, world!", "it's a dog's life", "i can't, won't and shouldn't", "they'll we're i've i'd i'm",
    "a \"quoted word\" here", "''double'' single", "(parens) [brackets] {braces} <angles>", "a--b and c -- d --- e", "wait... what",
    "one,two, three ,four , five", "1,000 and 2:30 pm", "ends with comma,", "ends with colon:", "semi;colon @at #hash $dollar %pct &and",
    "what?! really?", "star*dust * alone", "cannot gimme gonna gotta lemme wanna go", "more'n that d'ye know", "'tis the season 'twas night",
    "rock'n'roll", "'a' 'b c' 'quoted'", "dogs' bones", "a.b.c", "the end.", "the end.)", "(the end.)", "a_b snake_case", "50% off!",
    "x'y", "'em", "5'6", "a,1", ",,a", "a..b", "`tick` ``ticks``", "kids' toys (new)", "the \"best\" pizza, in town: yes!",
]


# multi-sentence texts for the run-time check of "Punkt spans + restated Treebank step == word_tokenize" (Dictionary.from_nltk)
SPAN_CHECK_STRINGS = [
    "a cat. a dog.", "the end. (really.) yes", "dr. smith went home. he slept", "what? yes! no. maybe", "e.g. this one, and that. done",
    "first sentence here. second one, with a comma. third!", "she said \"go.\" then left", "price: 3.50 dollars. cheap",
]


# the run-time check of the NATIVE Punkt restatement (host_text.cpp punkt_spans) against the installed nltk's Punkt instance, with
# the installed model's tables: abbreviations, initials, ordinals, ellipses, closers behind a break, breaks without whitespace
PUNKT_CHECK_STRINGS = SPAN_CHECK_STRINGS + [
    "a. b", "a. b.", "the end.  ", "one. two. three.", "one? two! three.", "wait... what. no", "hmm.. ok. yes", "(the end.) yes",
    "she said \"go.\"then left", "go.\"--then", "3. cat", "j. bach", "x. 3", "e.g. the cat", "i.e. a dog. yes", "u.s. but no",
    "st. john. the end", "mr. smith and mrs. jones. they left", "no. 3 is here. ok", "at 5 p.m. we left. then home", "a.) b", "a.') b",
    "a!') b", "a. ) b", ". . . a", "a . . . b", "vs. the world. fin", "inc. and co. ltd. closed", "fig. 2 shows. it", "1,000. the",
    "a photo of a cat. a photo of a dog.", "is it? yes. no! ok", "mid-st. the", "ph.d. but", "the u.s.a. is big. yes", "jan. 5. feb",
    "Dr. Smith went home. He slept.", "J. Bach wrote it. Then left", "e.g. The cat", "3. Cat", "It is 5 p.m. Then we go",
]


# texts with TWO OR MORE candidate break positions inside one whitespace-delimited chunk: nltk 3.6.6 rewrote the scan that finds the
# period contexts (its ReDoS fix); where the installed generation answers these like the 3.6.5 rule the native splitter restates,
# its strict mode (decline such texts) is switched off (attacks.Dictionary._native_punkt)
PUNKT_MULTI_CHECK_STRINGS = [
    "what?! yes", "wow!!! nice", "very bad acting!!! i promise.", "a?\"b. c", "hm?!) ok", "no!? really. yes", "go!!\" she said. ok",
    "a.) b.) c", "x?'y. z", "really?!?! no way. yes", "(what?!) he said", "stop!!!", "one!!! two??? three...", "e.g.?! no", "3.?) cat",
    "wait...?! what", "\"no!\"? yes. ok", "a!b?c. d", "end.\"?) next", "ok?!.. fine",
]


def spans_word_tokenize(text: str, spans) -> List[str]:
    """``nltk.word_tokenize(text)`` given the sentence spans its Punkt step produces: the Treebank step sentence by sentence."""
    return [w for a, b in spans for w in treebank_tokenize(text[a:b])]


def self_check(real_word_tokenize) -> bool:
    """Compare this restatement with the installed nltk on a fixed battery (run once when a Dictionary is built from nltk):
    False -> the caller must not use the native / restated tokenizer (a different nltk generation)."""
    try:
        return all(real_word_tokenize(s) == treebank_tokenize(s) for s in SELF_CHECK_STRINGS)
    except Exception:
        return False
