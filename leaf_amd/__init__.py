"""leaf_amd -- MI355X-native engine for the LEAF text-encoder adversarial fine-tuning step.

Only the hot path of LIONS-EPFL/LEAF is built here: anchor forward, the LEAF character search
scored by a forward-only CLIP text transformer, TextFARE loss forward+backward, AdamW -- as hand-written HIP
kernels for gfx950 behind the C ABI in include/leaf_hip.h.  Importing the package does not load the shared
library; constructing a model does, and fails loudly when it is missing.
"""
from .tokenizer import SimpleTokenizer, get_tokenizer  # noqa: F401

__all__ = ["SimpleTokenizer", "get_tokenizer", "create_model", "LeafCLIPText", "attack_text"]


def __getattr__(name):
    if name in ("create_model", "LeafCLIPText", "TextConfig", "get_config", "MODEL_CONFIGS"):
        from . import model
        return getattr(model, name)
    if name in ("attack_text", "attack_text_leaf"):
        from . import attacks
        return getattr(attacks, name)
    raise AttributeError(name)
