"""Checkpoint key maps for the text tower: OpenCLIP <-> HF ``CLIPTextModel`` / ``CLIPModel``.

The map restates what the reference's converters do tensor by tensor
(conversion/convert_2.py:37-99, conversion/convert_to_openclip.py:78-121): q/k/v are the three row chunks
of ``in_proj_*``; ``text_projection.weight`` (HF, nn.Linear) is ``text_projection`` (OpenCLIP, applied as
``x @ P``) transposed; ``fc1/fc2`` are ``c_fc/c_proj``; ``layer_norm1/2`` are ``ln_1/ln_2``.
The training checkpoint dict layout is the reference's (train_AT_text_only.py:516-525).
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

_TEXT_PREFIXES = ("token_embedding.", "positional_embedding", "transformer.", "ln_final.", "text_projection")


def _t(v) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v


def hf_to_openclip(sd: Dict[str, torch.Tensor], cfg) -> Dict[str, torch.Tensor]:
    pre = "text_model."
    out = {"token_embedding.weight": _t(sd[pre + "embeddings.token_embedding.weight"]),
           "positional_embedding": _t(sd[pre + "embeddings.position_embedding.weight"]),
           "ln_final.weight": _t(sd[pre + "final_layer_norm.weight"]),
           "ln_final.bias": _t(sd[pre + "final_layer_norm.bias"])}
    for i in range(cfg.layers):
        h, o = f"{pre}encoder.layers.{i}.", f"transformer.resblocks.{i}."
        out[o + "attn.in_proj_weight"] = torch.cat([_t(sd[h + f"self_attn.{n}_proj.weight"]) for n in "qkv"], 0)
        out[o + "attn.in_proj_bias"] = torch.cat([_t(sd[h + f"self_attn.{n}_proj.bias"]) for n in "qkv"], 0)
        for a, b in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"),
                     ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
            out[o + b + ".weight"] = _t(sd[h + a + ".weight"])
            out[o + b + ".bias"] = _t(sd[h + a + ".bias"])
    if "text_projection.weight" in sd:
        out["text_projection"] = _t(sd["text_projection.weight"]).t().contiguous()
    elif cfg.embed_dim == cfg.width:
        # CLIPTextModel has no projection: its embedding is the pooled EOT state (utils_attacks.py:49-53)
        out["text_projection"] = torch.eye(cfg.width)
    else:
        raise KeyError("HF checkpoint without text_projection.weight and embed_dim != width")
    if "logit_scale" in sd:
        out["logit_scale"] = _t(sd["logit_scale"])
    return out


def openclip_to_hf(sd: Dict[str, torch.Tensor], cfg, with_projection: bool = True) -> Dict[str, torch.Tensor]:
    pre = "text_model."
    out = {pre + "embeddings.token_embedding.weight": sd["token_embedding.weight"],
           pre + "embeddings.position_embedding.weight": sd["positional_embedding"],
           pre + "final_layer_norm.weight": sd["ln_final.weight"], pre + "final_layer_norm.bias": sd["ln_final.bias"]}
    for i in range(cfg.layers):
        h, o = f"{pre}encoder.layers.{i}.", f"transformer.resblocks.{i}."
        for j, n in enumerate("qkv"):
            out[h + f"self_attn.{n}_proj.weight"] = sd[o + "attn.in_proj_weight"].chunk(3, 0)[j].contiguous()
            out[h + f"self_attn.{n}_proj.bias"] = sd[o + "attn.in_proj_bias"].chunk(3, 0)[j].contiguous()
        for a, b in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"),
                     ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
            out[h + a + ".weight"] = sd[o + b + ".weight"]
            out[h + a + ".bias"] = sd[o + b + ".bias"]
    if with_projection:
        out["text_projection.weight"] = sd["text_projection"].t().contiguous()
    return out


def to_openclip_text_keys(sd, cfg) -> Dict[str, torch.Tensor]:
    """Accept OpenCLIP (optionally 'module.'-prefixed, full CLIP incl. visual.*), a training checkpoint dict
    ({'state_dict': ...}) or HF keys; return the OpenCLIP text-tower subset."""
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if any(k.startswith("module.") for k in sd):
        sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}
    if any(k.startswith("text_model.") for k in sd):
        return hf_to_openclip(sd, cfg)
    out = {k: _t(v) for k, v in sd.items() if k.startswith(_TEXT_PREFIXES) and not k.startswith("visual.")}
    if "logit_scale" in sd:
        out["logit_scale"] = _t(sd["logit_scale"])
    return out


def load_checkpoint_file(path: str) -> Dict[str, torch.Tensor]:
    """open_clip_pytorch_model.bin / epoch_latest.pt / *.safetensors / a directory holding one of them."""
    if os.path.isdir(path):
        for name in ("open_clip_pytorch_model.bin", "model.safetensors", "pytorch_model.bin", "epoch_latest.pt"):
            if os.path.exists(os.path.join(path, name)):
                path = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"no checkpoint file found in {path}")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu", weights_only=False)


def save_training_checkpoint(path: str, epoch: int, name: str, model, optimizer_state: dict):
    """Atomic write of the reference's checkpoint dict (train_AT_text_only.py:516-525): tmp file + os.replace."""
    ck = {"epoch": epoch, "name": name, "state_dict": {k: v.cpu() for k, v in model.state_dict().items()},
          "optimizer": optimizer_state}
    tmp = os.path.join(os.path.dirname(path) or ".", "tmp.pt")
    torch.save(ck, tmp)
    os.replace(tmp, path)
