"""Checkpoint key maps for the text tower: OpenCLIP <-> HF ``CLIPTextModel`` / ``CLIPModel``.

The map restates what the reference's converters do tensor by tensor
(conversion/convert_2.py:37-99, conversion/convert_to_openclip.py:78-121): q/k/v are the three row chunks
of ``in_proj_*``; ``text_projection.weight`` (HF, nn.Linear) is ``text_projection`` (OpenCLIP, applied as
``x @ P``) transposed; ``fc1/fc2`` are ``c_fc/c_proj``; ``layer_norm1/2`` are ``ln_1/ln_2``.
The training checkpoint dict layout is the reference's (train_AT_text_only.py:516-525): ``optimizer`` is a
``torch.optim.AdamW.state_dict()`` with the reference's two parameter groups in its ``named_parameters`` order (:326-341) --
over the WHOLE CLIP, image tower included, because the reference freezes ``model.visual`` only after it has built the optimizer
(:489-490); text-only groups with ``--lock-image`` or when the run started from a checkpoint without an image tower.
"""
from __future__ import annotations

import re
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


def non_text_tensors(sd) -> Dict[str, torch.Tensor]:
    """Every tensor of an OpenCLIP checkpoint that is NOT part of the text tower (visual.*, logit_bias, ...), on the host; {} for
    HF checkpoints.  Carried through training so that the written state_dict stays loadable by the reference."""
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    if any(k.startswith("text_model.") for k in sd):
        return {}
    out = {}
    for k, v in sd.items():
        k2 = k[len("module."):] if k.startswith("module.") else k
        if not k2.startswith(_TEXT_PREFIXES) and k2 != "logit_scale" and hasattr(v, "shape"):
            out[k2] = _t(v).detach().cpu()
    return out


# ----------------------------------------------------------------------------- reference AdamW state_dict layout
_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked", "attn_mask")


def _exclude(name: str, ndim: int) -> bool:
    """The reference's ``exclude`` lambda (train_AT_text_only.py:326): no weight decay for gains, biases, logit_scale."""
    return ndim < 2 or "bn" in name or "ln" in name or "bias" in name or "logit_scale" in name


# Parameter names of the image towers of open_clip's CLIP class whose parameter / buffer split is known here: VisionTransformer
# (src/open_clip/transformer.py:468-537 + the residual blocks :224-237) and ModifiedResNet (src/open_clip/modified_resnet.py:
# 17-36,61-65,109-128).  A key outside these patterns may be a buffer of another tower (a timm trunk's relative_position_index,
# ...) that named_parameters() would not list: the ids of a full-CLIP optimizer layout built from it would be shifted.
_KNOWN_VISUAL = re.compile(
    r"^(logit_bias|visual\.(class_embedding|positional_embedding|proj|conv1\.weight|ln_(pre|post)\.(weight|bias)"
    r"|transformer\.resblocks\.\d+\.(ln_[12]\.(weight|bias)|ls_[12]\.gamma|attn\.(in_proj_(weight|bias)|out_proj\.(weight|bias))"
    r"|mlp\.(c_fc|c_proj)\.(weight|bias))"
    r"|attn_pool(_contrastive)?\.(query|ln_[qk]\.(weight|bias)|attn\.(in_proj_(weight|bias)|[qkv]_proj_weight|out_proj\.(weight|bias)))"
    r"|(conv[123]\.weight|bn[123]\.(weight|bias))"
    r"|layer[1-4]\.\d+\.(conv[123]\.weight|bn[123]\.(weight|bias)|downsample\.(0\.weight|1\.(weight|bias)))"
    r"|attnpool\.(positional_embedding|[kqvc]_proj\.(weight|bias)))"
    r"|.*\.(running_mean|running_var|num_batches_tracked)|.*attn_mask)$")


def image_tower_is_known(extra) -> bool:
    """True when every carried-through non-text tensor is a parameter or buffer of an image tower whose named_parameters()
    order can be reproduced from the checkpoint keys alone; {} / None (no image tower) counts as known."""
    return all(_KNOWN_VISUAL.match(k) for k in (extra or {}))


def non_text_parameters(extra) -> list:
    """(name, ndim) of the carried-through non-text tensors that are PARAMETERS of the reference's CLIP, in checkpoint order:
    buffers (BatchNorm statistics of the ResNet towers, a persistent attn_mask of old checkpoints) are not in
    ``named_parameters()`` and so not in the optimizer."""
    return [(k, int(v.ndim)) for k, v in (extra or {}).items() if not k.endswith(_BUFFER_SUFFIXES)]


def reference_param_groups(layers: int, extra=None):
    """Names of the reference's two AdamW groups in ITS order: train_AT_text_only.py:326-341 runs over
    ``model.named_parameters()`` of the full CLIP while the image tower still requires grad -- it is frozen only at :489-490 --
    so unless ``--lock-image`` was given (:286-290, before the optimizer) the groups hold every ``visual.*`` parameter too.
    Group 0 = excluded from weight decay (p.ndim < 2 or 'bn' / 'ln' / 'bias' / 'logit_scale' in the name), group 1 = the rest.

    ``extra``: the non-text tensors of the start checkpoint (``LeafCLIPText.extra_state``: ``visual.*``, ``logit_bias``) in their
    checkpoint order, or None / {} for the TEXT-ONLY layout (``--lock-image``, or a start from an HF / text-only checkpoint, where
    there is no image tower to name).  ``named_parameters()`` order of open_clip's CLIP: its own parameters
    (positional_embedding, text_projection, logit_scale[, logit_bias]), then the children in registration order: visual.*,
    transformer.*, token_embedding, ln_final.  Pinned by tests/golden/ckpt_structure.json (both cases; generated by executing the
    reference's statements in the reference's order, tests/golden/make_golden_ckpt.py)."""
    order = [("positional_embedding", 2), ("text_projection", 2), ("logit_scale", 0)]
    order += non_text_parameters(extra)
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        order += [(p + n, 1 if n.endswith("bias") or n.startswith("ln_") else 2) for n in
                  ("ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                   "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                   "mlp.c_proj.weight", "mlp.c_proj.bias")]
    order += [("token_embedding.weight", 2), ("ln_final.weight", 1), ("ln_final.bias", 1)]
    return [n for n, d in order if _exclude(n, d)], [n for n, d in order if not _exclude(n, d)]


_GROUP_DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                       decoupled_weight_decay=True)


def optimizer_state_to_torch(layout, layers, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, lrs=None, extra=None):
    """Flat fp32 moments (engine layout: name -> (offset, shape)) -> torch.optim.AdamW.state_dict() as the reference's optimizer
    writes it (train_AT_text_only.py:516-525): param ids number group 0 then group 1 over the groups of ``reference_param_groups``
    (``extra`` as there); parameters that never receive a gradient on this path -- ``logit_scale`` and every ``visual.*`` id --
    have no state entry, exactly as in the reference; every text parameter carries the common ``step`` as a 0-d fp32 tensor."""
    g0, g1 = reference_param_groups(layers, extra)
    names = g0 + g1
    state = {}
    for pid, n in enumerate(names):
        if n not in layout:
            continue
        off, shape = layout[n]
        numel = int(np.prod(shape))
        if step > 0:
            state[pid] = {"step": torch.tensor(float(step), dtype=torch.float32),
                          "exp_avg": exp_avg[off:off + numel].reshape(shape).detach().cpu().clone(),
                          "exp_avg_sq": exp_avg_sq[off:off + numel].reshape(shape).detach().cpu().clone()}
    lrs = lrs or (lr, lr)
    groups = [dict(lr=lrs[0], betas=tuple(betas), eps=eps, weight_decay=0.0, **_GROUP_DEFAULTS, params=list(range(len(g0)))),
              dict(lr=lrs[1], betas=tuple(betas), eps=eps, weight_decay=weight_decay, **_GROUP_DEFAULTS,
                   params=list(range(len(g0), len(names))))]
    return {"state": state, "param_groups": groups}


def optimizer_state_layout(osd, layers, extra=None) -> str:
    """Which of the two layouts a torch AdamW state_dict has: 'full' (groups over the whole CLIP: what the reference writes
    without --lock-image; needs the non-text names, ``extra``) or 'text-only' (--lock-image, or written from a text-only start).
    Raises when it has neither."""
    pg = osd["param_groups"]
    sizes = [len(g["params"]) for g in pg]
    t0, t1 = reference_param_groups(layers)
    if sizes == [len(t0), len(t1)]:
        return "text-only"
    if extra:
        f0, f1 = reference_param_groups(layers, extra)
        if sizes == [len(f0), len(f1)]:
            return "full"
        hint = f"or {len(f0)} + {len(f1)} with the {len(non_text_parameters(extra))} non-text parameters of this checkpoint"
    else:
        hint = "and the checkpoint's state_dict names no image tower to account for more"
    raise ValueError(f"optimizer state has groups of {sizes} parameters: the reference's two groups hold {len(t0)} + {len(t1)} text "
                     f"parameters, {hint}")


def optimizer_state_from_torch(osd, layout, layers, exp_avg, exp_avg_sq, extra=None) -> int:
    """Inverse of ``optimizer_state_to_torch``; also reads what the REFERENCE wrote (``extra`` = the non-text tensors of the SAME
    checkpoint's state_dict, which name the ``visual.*`` ids between the text ids).  Copies every text parameter's moments into
    the flat buffers and returns the step count; state of non-text ids (there is none on this path) is ignored."""
    layout_kind = optimizer_state_layout(osd, layers, extra)
    g0, g1 = reference_param_groups(layers, extra if layout_kind == "full" else None)
    names = g0 + g1
    pg = osd["param_groups"]
    ids = list(pg[0]["params"]) + list(pg[1]["params"])
    step = 0
    exp_avg.zero_()
    exp_avg_sq.zero_()
    for pid, n in zip(ids, names):
        st = osd["state"].get(pid)
        if st is None or n not in layout:
            continue
        off, shape = layout[n]
        numel = int(np.prod(shape))
        if tuple(st["exp_avg"].shape) != tuple(shape):
            raise ValueError(f"optimizer state of {n} (id {pid}): shape {tuple(st['exp_avg'].shape)} != {tuple(shape)}")
        exp_avg[off:off + numel].copy_(st["exp_avg"].reshape(-1))
        exp_avg_sq[off:off + numel].copy_(st["exp_avg_sq"].reshape(-1))
        step = max(step, int(float(st["step"])))
    return step


# ----------------------------------------------------------------------------- open_clip_config.json / HF export
def read_open_clip_config(path: str):
    """``open_clip_config.json`` next to a hub checkpoint (src/open_clip/factory.py:200-207 ``_get_hf_config``):
    {"model_cfg": {"embed_dim", "quick_gelu"?, "text_cfg": {"context_length", "vocab_size", "width", "heads", "layers"}}, ...}.
    ``path`` may be the json, the checkpoint file beside it or the directory.  Returns a TextConfig or None."""
    import json
    from .model import TextConfig
    if os.path.isdir(path):
        cand = os.path.join(path, "open_clip_config.json")
    elif os.path.basename(path) == "open_clip_config.json":
        cand = path
    else:
        cand = os.path.join(os.path.dirname(path) or ".", "open_clip_config.json")
    if not os.path.exists(cand):
        return None
    with open(cand) as f:
        cfg = json.load(f)
    mc = cfg.get("model_cfg", cfg)
    t = mc["text_cfg"]
    return TextConfig(width=t.get("width", 512), heads=t.get("heads", 8), layers=t.get("layers", 12), embed_dim=mc["embed_dim"],
                      context_length=t.get("context_length", 77), vocab_size=t.get("vocab_size", 49408),
                      quick_gelu=bool(mc.get("quick_gelu", False)))


def write_hf_text_model(out_dir: str, sd: Dict[str, torch.Tensor], cfg, with_projection: bool = True):
    """The release format of the reference (README.md:98; conversion/convert_2.py:37-99,133-203): a HuggingFace
    ``CLIPTextModelWithProjection`` (``CLIPTextModel`` when ``with_projection`` is False: its embedding is the pooled EOT
    state, utils_attacks.py:49-53) as ``model.safetensors`` + ``config.json``; ``hidden_act`` = quick_gelu for the OpenAI-lineage
    B/L towers, gelu for H/g/bigG (convert_2.py:133,154,185,203)."""
    import json
    from safetensors.torch import save_file
    os.makedirs(out_dir, exist_ok=True)
    hf = openclip_to_hf({k: _t(v).detach().cpu().float() for k, v in sd.items()}, cfg, with_projection=with_projection)
    save_file({k: v.contiguous() for k, v in hf.items()}, os.path.join(out_dir, "model.safetensors"), metadata={"format": "pt"})
    config = {"architectures": ["CLIPTextModelWithProjection" if with_projection else "CLIPTextModel"], "model_type": "clip_text_model",
              "hidden_size": cfg.width, "intermediate_size": 4 * cfg.width, "num_attention_heads": cfg.heads,
              "num_hidden_layers": cfg.layers, "hidden_act": "quick_gelu" if cfg.quick_gelu else "gelu",
              "projection_dim": cfg.embed_dim, "max_position_embeddings": cfg.context_length, "vocab_size": cfg.vocab_size,
              "layer_norm_eps": cfg.ln_eps, "attention_dropout": 0.0, "initializer_range": 0.02, "initializer_factor": 1.0,
              "pad_token_id": 1, "bos_token_id": cfg.vocab_size - 2, "eos_token_id": cfg.vocab_size - 1, "torch_dtype": "float32"}
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(config, f, indent=2)
    return out_dir


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
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ck = {"epoch": epoch, "name": name, "state_dict": sd, "optimizer": optimizer_state}
    tmp = os.path.join(os.path.dirname(path) or ".", "tmp.pt")
    torch.save(ck, tmp)
    os.replace(tmp, path)
