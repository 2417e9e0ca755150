"""leaf_amd -- MI355X-native engine for the LEAF text-encoder adversarial fine-tuning step.

Only the hot path of LIONS-EPFL/LEAF is built here: anchor forward, the LEAF character search
scored by a forward-only CLIP text transformer, TextFARE loss forward+backward, AdamW -- as hand-written HIP
kernels for gfx950 behind the C ABI in include/leaf_hip.h.  Importing the package does not load the shared
library; constructing a model does, and fails loudly when it is missing.
"""
import os as _os
import sys as _sys


def configure_runtime(dev_kernarg: bool = True, warn: bool = True) -> dict:
    """Process-wide HIP runtime settings this engine benefits from -- an EXPLICIT call, made by the entry points
    (train_AT_text_only.py, eval_textfare.py, bench.py) before their first ``import torch``; importing ``leaf_amd`` changes nothing.

    ``HIP_FORCE_DEV_KERNARG=1``: kernel arguments straight in device memory.  The HIP runtime reads it once, when it initialises
    (the first HIP call of the process: ``torch.cuda.is_available()``, a tensor on the device, ...).  A step is ~530 mostly small,
    dependent launches and this takes 0.6-1.1 ms off a 50-ms step (round 4, same-box A/B: 50.45-50.78 -> 49.82-50.0 ms).  A value
    the user has exported is left alone.  If HIP is already initialised the setting cannot take effect any more: with ``warn`` that
    is said (once) instead of silently losing ~2 %.  Returns what was done: {"HIP_FORCE_DEV_KERNARG": value in effect for a runtime
    that starts now, "applied": bool, "hip_initialised": bool}."""
    torch = _sys.modules.get("torch")
    started = bool(torch is not None and torch.cuda.is_initialized())
    applied = False
    if dev_kernarg and "HIP_FORCE_DEV_KERNARG" not in _os.environ:
        if started:
            if warn:
                import warnings
                warnings.warn("leaf_amd.configure_runtime(): the HIP runtime is already initialised, HIP_FORCE_DEV_KERNARG=1 can no "
                              "longer take effect (0.6-1.1 ms of a 50-ms step); call it before the first `import torch` / HIP call, "
                              "or export the variable", RuntimeWarning, stacklevel=2)
        else:
            _os.environ["HIP_FORCE_DEV_KERNARG"] = "1"
            applied = True
    return {"HIP_FORCE_DEV_KERNARG": _os.environ.get("HIP_FORCE_DEV_KERNARG"), "applied": applied, "hip_initialised": started}


from .tokenizer import SimpleTokenizer, get_tokenizer  # noqa: F401,E402

__all__ = ["SimpleTokenizer", "get_tokenizer", "create_model", "LeafCLIPText", "attack_text", "configure_runtime"]


def __getattr__(name):
    if name in ("create_model", "LeafCLIPText", "TextConfig", "get_config", "MODEL_CONFIGS"):
        from . import model
        return getattr(model, name)
    if name in ("attack_text", "attack_text_leaf"):
        from . import attacks
        return getattr(attacks, name)
    raise AttributeError(name)
