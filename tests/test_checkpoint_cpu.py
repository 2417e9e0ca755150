"""Checkpoint interchange with the reference, all on the CPU:

* the AdamW ``state_dict`` layout the reference writes / ``--resume`` reads (train_AT_text_only.py:326-341,351-372,516-525),
  pinned by tests/golden/ckpt_structure.json (generated from the reference's own CLIP + torch.optim.AdamW);
* the HuggingFace ``CLIPTextModel(WithProjection)`` release format (README.md:98, conversion/convert_2.py), checked by loading
  the written directory with ``transformers`` and comparing its embeddings with the oracle;
* ``open_clip_config.json`` (src/open_clip/factory.py:200-207) and the carry-through of non-text tensors.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from leaf_amd import _lib
from leaf_amd import checkpoint as CK
from leaf_amd.model import MODEL_CONFIGS, TextConfig
from oracle import text_oracle as O


def _layout(cfg: TextConfig):
    """The engine's flat parameter layout (host-only C calls; no GPU needed)."""
    lib = _lib.lib()
    c = _lib.TextCfgC(cfg.layers, cfg.width, cfg.heads, cfg.embed_dim, cfg.context_length, cfg.vocab_size, int(cfg.quick_gelu), cfg.ln_eps)
    h = C.c_void_p()
    _lib.check(lib.leaf_text_create(C.byref(c), 1, C.byref(h)), "create")
    out, name = {}, C.create_string_buffer(128)
    off, rows, cols = C.c_size_t(), C.c_int64(), C.c_int64()
    for i in range(lib.leaf_text_num_tensors(h)):
        lib.leaf_text_param_info(h, i, name, 128, C.byref(off), C.byref(rows), C.byref(cols))
        out[name.value.decode()] = (off.value, (rows.value, cols.value) if cols.value else (rows.value,))
    n, nd = lib.leaf_text_param_count(h), lib.leaf_text_decay_count(h)
    lib.leaf_text_destroy(h)
    return out, n, nd


def _fixture(golden_dir):
    with open(os.path.join(golden_dir, "ckpt_structure.json")) as f:
        return json.load(f)


def _extra_from_fixture(s):
    """The non-text tensors a run started from the reference's (tiny) CLIP checkpoint carries through, in checkpoint order."""
    text = ("token_embedding.", "positional_embedding", "transformer.", "ln_final.", "text_projection", "logit_scale")
    return {k: torch.zeros(shape) for k, shape, _ in s["state_dict_keys"] if not k.startswith(text)}


def test_fixture_generator_ran_the_reference_steps_in_the_reference_order(golden_dir):
    """round 2's generator froze model.visual BEFORE building the optimizer and so pinned a layout the
    reference never writes.  The generator now executes the reference's own statements and records where they stand in the
    reference and what the model looked like when the optimizer was built; a re-ordered generator fails here."""
    s = _fixture(golden_dir)
    d, l = s["cases"]["default"], s["cases"]["lock_image"]
    for c in (d, l):
        ln = c["reference_lines"]
        assert ln["lock_image"] < ln["optimizer"] < ln["freeze_visual"] < ln["checkpoint_dict"]
        assert c["checkpoint_keys"] == ["epoch", "name", "optimizer", "state_dict"]
    assert d["visual_trainable_when_optimizer_was_built"] is True and l["visual_trainable_when_optimizer_was_built"] is False
    vis = [n for g in d["group_names"] for n in g if n.startswith("visual.")]
    assert len(vis) == sum(1 for n, _ in s["named_parameters"] if n.startswith("visual.")) > 0
    assert not any(n.startswith("visual.") for g in l["group_names"] for n in g)
    # no state for what never receives a gradient on this path: logit_scale and the whole image tower
    assert set(d["no_state_names"]) == {"logit_scale", *vis} and l["no_state_names"] == ["logit_scale"]
    # the gradient fixture's generator builds no optimizer at all any more
    src = open(os.path.join(golden_dir, "make_golden_vitl_grads.py")).read()
    assert "optim." not in src and "AdamW" not in src


@pytest.mark.parametrize("case", ["default", "lock_image"])
def test_adamw_groups_match_the_reference_fixture(golden_dir, case):
    s = _fixture(golden_dir)
    c = s["cases"][case]
    extra = _extra_from_fixture(s) if case == "default" else None
    g0, g1 = CK.reference_param_groups(2, extra)
    assert [g0, g1] == c["group_names"]
    cfg = MODEL_CONFIGS["tiny-test-quickgelu"]
    layout, n, nd = _layout(cfg)
    # the engine's decay split must be the reference's: the text tensors of group 1 == the tensors below n_decay
    assert sorted(k for k, (off, _) in layout.items() if off < nd) == sorted(k for k in g1 if k in layout)
    m, v = torch.arange(n, dtype=torch.float32) * 1e-3, torch.arange(n, dtype=torch.float32) * 1e-6 + 1.0
    osd = CK.optimizer_state_to_torch(layout, cfg.layers, m, v, 1, 1e-5, (0.9, 0.999), 1e-8, 1e-4, extra=extra)
    assert sorted(osd["state"].keys()) == c["state_ids"], "the ids WITH state are exactly the reference's (text parameters)"
    names = g0 + g1
    assert [names[i] for i in sorted(osd["state"])] == c["state_names"]
    first = osd["state"][c["state_ids"][0]]
    assert sorted(first.keys()) == c["state_entry_keys"]
    assert str(first["step"].dtype) == c["step_dtype"] and list(first["step"].shape) == c["step_shape"] and float(first["step"]) == c["step_value"]
    for got, want in zip(osd["param_groups"], c["param_groups"]):
        assert got["params"] == want["params"] and set(want) <= set(got)
        assert got["weight_decay"] == want["weight_decay"] and list(got["betas"]) == want["betas"] and got["eps"] == want["eps"]
    shapes = {k: v_ for k, v_, _ in s["state_dict_keys"]}
    for pid, name in enumerate(names):
        if pid in osd["state"]:
            assert list(osd["state"][pid]["exp_avg"].shape) == shapes[name], name
    # round trip into fresh flat buffers
    m2, v2 = torch.zeros(n), torch.zeros(n)
    assert CK.optimizer_state_layout(osd, cfg.layers, extra) == ("full" if extra else "text-only")
    assert CK.optimizer_state_from_torch(osd, layout, cfg.layers, m2, v2, extra=extra) == 1
    assert torch.equal(m, m2) and torch.equal(v, v2)


def _reference_ordered_adamw(s, lock_image):
    """torch.optim.AdamW built the way the reference builds it (train_AT_text_only.py:326-341) over parameters with the
    fixture's names / shapes in named_parameters order: over ALL of them unless --lock-image froze the image tower first."""
    shapes = {k: sh for k, sh, _ in s["state_dict_keys"]}
    named = [(n, torch.nn.Parameter(torch.zeros(shapes[n]))) for n, _ in s["named_parameters"]]
    if lock_image:
        for n, p_ in named:
            if n.startswith("visual."):
                p_.requires_grad = False
    exclude = lambda n, p_: p_.ndim < 2 or "bn" in n or "ln" in n or "bias" in n or "logit_scale" in n
    gb = [p_ for n, p_ in named if exclude(n, p_) and p_.requires_grad]
    rest = [p_ for n, p_ in named if not exclude(n, p_) and p_.requires_grad]
    opt = torch.optim.AdamW([{"params": gb, "weight_decay": 0.0}, {"params": rest, "weight_decay": 0.2}], lr=1e-3)
    return dict(named), opt


@pytest.mark.parametrize("lock_image", [False, True])
def test_optimizer_state_interchange_with_a_reference_ordered_adamw(golden_dir, lock_image):
    """Both directions of --resume (train_AT_text_only.py:364-366, 516-525): the reference's optimizer.load_state_dict() takes
    what we write (torch checks the group sizes and copies the state by id), and we read what torch's AdamW -- i.e. the
    reference -- writes, image-tower ids included; ids and shapes are checked per parameter."""
    s = _fixture(golden_dir)
    cfg = MODEL_CONFIGS["tiny-test-quickgelu"]
    layout, n, _ = _layout(cfg)
    extra = None if lock_image else _extra_from_fixture(s)
    params, opt = _reference_ordered_adamw(s, lock_image)
    rng = torch.Generator().manual_seed(0)
    m, v = torch.randn(n, generator=rng), torch.rand(n, generator=rng)
    ours = CK.optimizer_state_to_torch(layout, cfg.layers, m, v, 5, 3e-4, (0.9, 0.98), 1e-6, 0.2, extra=extra)
    opt.load_state_dict(ours)                        # raises on any group-size mismatch
    for name, (off, shp) in layout.items():
        st = opt.state[params[name]]
        numel = int(np.prod(shp))
        assert float(st["step"]) == 5 and torch.equal(st["exp_avg"], m[off:off + numel].view(shp)), name
        assert torch.equal(st["exp_avg_sq"], v[off:off + numel].view(shp)), name
    assert params["logit_scale"] not in opt.state
    assert not any(k.startswith("visual.") and p_ in opt.state for k, p_ in params.items())
    assert opt.param_groups[1]["lr"] == 3e-4 and opt.param_groups[1]["weight_decay"] == 0.2 and opt.param_groups[0]["weight_decay"] == 0.0
    # the other direction: one more step of the text parameters in torch, then read torch's state_dict back
    for name, p_ in params.items():
        p_.grad = torch.ones_like(p_) if name in layout else None
    opt.step()
    written = opt.state_dict()
    assert [len(g["params"]) for g in written["param_groups"]] == [len(g) for g in s["cases"]["lock_image" if lock_image else "default"]["group_names"]]
    m2, v2 = torch.zeros(n), torch.zeros(n)
    assert CK.optimizer_state_from_torch(written, layout, cfg.layers, m2, v2, extra=extra) == 6
    for name, (off, shp) in layout.items():      # (the flat buffers have alignment gaps between tensors)
        sl = slice(off, off + int(np.prod(shp)))
        assert torch.allclose(m2[sl], 0.9 * m[sl] + 0.1, atol=1e-6) and torch.allclose(v2[sl], 0.98 * v[sl] + 0.02, atol=1e-6), name


def test_optimizer_state_with_unknown_groups_is_refused(golden_dir):
    """A full-CLIP optimizer state cannot be mapped without the image tower's parameter names: loud, never a silent mis-assignment."""
    s = _fixture(golden_dir)
    cfg = MODEL_CONFIGS["tiny-test-quickgelu"]
    layout, n, _ = _layout(cfg)
    full = CK.optimizer_state_to_torch(layout, cfg.layers, torch.zeros(n), torch.ones(n), 1, 1e-5, (0.9, 0.999), 1e-8, 1e-4,
                                       extra=_extra_from_fixture(s))
    with pytest.raises(ValueError, match="names no image tower"):
        CK.optimizer_state_from_torch(full, layout, cfg.layers, torch.zeros(n), torch.zeros(n), extra=None)
    # BatchNorm buffers of a ResNet image tower are not parameters: they take no id
    extra = {"visual.bn1.weight": torch.zeros(4), "visual.bn1.running_mean": torch.zeros(4), "visual.bn1.num_batches_tracked": torch.zeros(())}
    assert CK.non_text_parameters(extra) == [("visual.bn1.weight", 1)]


@pytest.mark.parametrize("name,with_proj", [("tiny-test-quickgelu", True), ("tiny-test", True), ("tiny-test", False)])
def test_hf_export_loads_in_transformers_and_matches_the_oracle(tmp_path, name, with_proj):
    transformers = pytest.importorskip("transformers")
    cfg = MODEL_CONFIGS[name]
    ocfg = O.TextCfg(cfg.width, cfg.heads, cfg.layers, cfg.embed_dim, quick_gelu=cfg.quick_gelu)
    w = O.init_weights(ocfg, seed=5)
    CK.write_hf_text_model(str(tmp_path / "hf"), {k: torch.from_numpy(v) for k, v in w.items()}, cfg, with_projection=with_proj)
    conf = json.load(open(tmp_path / "hf" / "config.json"))
    assert conf["hidden_act"] == ("quick_gelu" if cfg.quick_gelu else "gelu") and conf["projection_dim"] == cfg.embed_dim
    toks = O.synthetic_tokens(5, seed=4)
    want = O.encode_text(w, ocfg, toks)
    cls = transformers.CLIPTextModelWithProjection if with_proj else transformers.CLIPTextModel
    hf = cls.from_pretrained(str(tmp_path / "hf")).eval()
    with torch.no_grad():
        out = hf(input_ids=torch.from_numpy(toks))
    if with_proj:
        got = out.text_embeds.numpy()
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-5
    else:   # CLIPTextModel: the pooled EOT state before the projection (utils_attacks.py:49-53)
        got = out.pooler_output.numpy() @ w["text_projection"]
        assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-5
    # and the exported keys map back onto the engine's layout
    from safetensors.torch import load_file
    back = CK.hf_to_openclip(load_file(str(tmp_path / "hf" / "model.safetensors")), cfg) if with_proj else None
    if back is not None:
        for k, v in w.items():
            assert np.array_equal(back[k].numpy(), v), k


def test_open_clip_config_json(tmp_path):
    cfg = {"model_cfg": {"embed_dim": 768, "quick_gelu": True, "vision_cfg": {"image_size": 224, "layers": 24, "width": 1024, "patch_size": 14},
                         "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": 768, "heads": 12, "layers": 12}},
           "preprocess_cfg": {"mean": [0.48, 0.45, 0.40], "std": [0.26, 0.26, 0.27]}}
    (tmp_path / "open_clip_config.json").write_text(json.dumps(cfg))
    (tmp_path / "open_clip_pytorch_model.bin").write_bytes(b"")
    for p in (tmp_path, tmp_path / "open_clip_pytorch_model.bin", tmp_path / "open_clip_config.json"):
        got = CK.read_open_clip_config(str(p))
        assert got == TextConfig(768, 12, 12, 768, 77, 49408, quick_gelu=True)
    assert CK.read_open_clip_config(str(tmp_path / "elsewhere" / "x.bin")) is None


def test_non_text_tensors_are_carried_through():
    sd = {"module.visual.conv1.weight": torch.ones(4, 3, 2, 2), "module.visual.proj": torch.zeros(4, 2), "module.logit_scale": torch.tensor(2.0),
          "module.token_embedding.weight": torch.zeros(8, 4), "module.transformer.resblocks.0.ln_1.weight": torch.ones(4)}
    extra = CK.non_text_tensors({"epoch": 3, "state_dict": sd})
    assert sorted(extra) == ["visual.conv1.weight", "visual.proj"]
    assert CK.non_text_tensors({"text_model.embeddings.token_embedding.weight": torch.zeros(2, 2)}) == {}


def test_unknown_image_tower_is_not_guessed_into_the_optimizer_layout():
    """which carried-through tensors are parameters can only be told for open_clip's own ViT / ModifiedResNet towers; a
    timm trunk's buffers (relative_position_index, ...) would shift the ids of a full-CLIP optimizer layout, so such a
    checkpoint gets the text-only layout (with a warning) instead of a silently wrong one."""
    from leaf_amd.checkpoint import image_tower_is_known
    vit = ["visual.class_embedding", "visual.positional_embedding", "visual.proj", "visual.conv1.weight", "visual.ln_pre.weight",
           "visual.transformer.resblocks.23.attn.in_proj_weight", "visual.transformer.resblocks.0.mlp.c_proj.bias",
           "visual.transformer.resblocks.3.ls_1.gamma", "visual.ln_post.bias", "logit_bias"]
    rn = ["visual.conv2.weight", "visual.bn1.running_var", "visual.bn3.num_batches_tracked", "visual.layer3.5.conv2.weight",
          "visual.layer1.0.downsample.0.weight", "visual.layer4.0.downsample.1.bias", "visual.attnpool.c_proj.bias",
          "visual.attnpool.positional_embedding"]
    assert image_tower_is_known({}) and image_tower_is_known(None)
    assert image_tower_is_known({k: 0 for k in vit + rn})
    for k in ("visual.trunk.blocks.0.attn.relative_position_index", "visual.trunk.patch_embed.proj.weight", "visual.head.proj.weight"):
        assert not image_tower_is_known({**{v: 0 for v in vit}, k: 0}), k
