#!/usr/bin/env python3
"""Export what --constrain needs from an nltk installation, so that the trainer can run the reference's exact constraint
(utils_attacks.py:110-143: nltk ``words`` corpus + ``word_tokenize``) on a machine WITHOUT nltk:

    python tools/export_punkt_params.py --punkt punkt_english.json --words words.txt
    python train_AT_text_only.py ... --constrain --dictionary-file words.txt --dictionary-tokenizer treebank --punkt-params punkt_english.json

``--punkt``: the four parameter tables of the trained English Punkt model that ``nltk.sent_tokenize`` uses (abbreviation types,
collocations, frequent sentence starters, orthographic contexts) as JSON; leaf_amd.native_text.NativePunkt runs nltk's algorithm
over them (host_text.cpp ``punkt_spans``).  ``--words``: the ``words`` corpus, one entry per line.
Needs nltk with its ``punkt`` / ``punkt_tab`` and ``words`` downloads (the reference fetches them at import, utils_attacks.py:7-9).
"""
import argparse
import json


def load_punkt():
    import nltk
    try:
        from nltk.tokenize import _get_punkt_tokenizer        # nltk >= 3.8.2 (punkt_tab)
        return _get_punkt_tokenizer("english")
    except ImportError:
        return nltk.data.load("tokenizers/punkt/english.pickle")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--punkt", help="output JSON for the Punkt tables")
    ap.add_argument("--words", help="output text file for the words corpus")
    a = ap.parse_args()
    if a.punkt:
        p = load_punkt()._params
        with open(a.punkt, "w") as f:
            json.dump({"abbrev_types": sorted(p.abbrev_types), "collocations": sorted(list(c) for c in p.collocations),
                       "sent_starters": sorted(p.sent_starters), "ortho_context": {k: int(v) for k, v in sorted(p.ortho_context.items()) if v}}, f)
        print(f"{a.punkt}: {len(p.abbrev_types)} abbreviations, {len(p.collocations)} collocations, {len(p.sent_starters)} sentence starters, "
              f"{len(p.ortho_context)} orthographic contexts")
    if a.words:
        from nltk.corpus import words
        ws = words.words()
        with open(a.words, "w") as f:
            f.write("\n".join(ws))
        print(f"{a.words}: {len(ws)} entries")


if __name__ == "__main__":
    main()
