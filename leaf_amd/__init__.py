"""leaf_amd -- MI355X-native engine for the LEAF text-encoder adversarial fine-tuning step.

Only the hot path of LIONS-EPFL/LEAF is built here: anchor forward, the LEAF character search
scored by a forward-only CLIP text transformer, TextFARE loss forward+backward, AdamW -- as hand-written HIP
kernels for gfx950 behind the C ABI in include/leaf_hip.h.  Importing the package does not load the shared
library; constructing a model does, and fails loudly when it is missing.
"""
import os as _os

# Kernel arguments straight in device memory (read by the HIP runtime when it initialises, i.e. at the first HIP call of the process): a
# step is ~530 mostly small, dependent launches, and this takes 0.6-0.7 ms off a 50-ms step (round 4, same-box A/B: 50.45-50.78 ->
# 49.82-50.0 ms).  A value the user has set is left alone.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .tokenizer import SimpleTokenizer, get_tokenizer  # noqa: F401,E402

__all__ = ["SimpleTokenizer", "get_tokenizer", "create_model", "LeafCLIPText", "attack_text"]


def __getattr__(name):
    if name in ("create_model", "LeafCLIPText", "TextConfig", "get_config", "MODEL_CONFIGS"):
        from . import model
        return getattr(model, name)
    if name in ("attack_text", "attack_text_leaf"):
        from . import attacks
        return getattr(attacks, name)
    raise AttributeError(name)
