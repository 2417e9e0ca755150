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


def test_adamw_groups_match_the_reference_fixture(golden_dir):
    with open(os.path.join(golden_dir, "ckpt_structure.json")) as f:
        s = json.load(f)
    g0, g1 = CK.reference_param_groups(2)
    assert [g0, g1] == s["group_names"]
    cfg = MODEL_CONFIGS["tiny-test-quickgelu"]
    layout, n, nd = _layout(cfg)
    # the engine's decay split must be the reference's: group 1 == the tensors below n_decay
    assert sorted(k for k, (off, _) in layout.items() if off < nd) == sorted(g1)
    m, v = torch.arange(n, dtype=torch.float32) * 1e-3, torch.arange(n, dtype=torch.float32) * 1e-6 + 1.0
    osd = CK.optimizer_state_to_torch(layout, cfg.layers, m, v, 7, 1e-5, (0.9, 0.999), 1e-8, 1e-4)
    assert sorted(osd["state"].keys()) == s["state_ids"], "logit_scale (id 0) has no state, every other parameter has"
    first = osd["state"][1]
    assert sorted(first.keys()) == s["state_entry_keys"]
    assert str(first["step"].dtype) == s["step_dtype"] and list(first["step"].shape) == s["step_shape"] and float(first["step"]) == 7.0
    for got, want in zip(osd["param_groups"], s["param_groups"]):
        assert got["params"] == want["params"] and set(want) <= set(got)
        assert got["weight_decay"] == want["weight_decay"] and list(got["betas"]) == want["betas"] and got["eps"] == want["eps"]
    # shapes per id = the reference's state_dict shapes
    shapes = {k: v_ for k, v_, _ in s["state_dict_keys"]}
    for pid, name in enumerate(g0 + g1):
        if pid in osd["state"]:
            assert list(osd["state"][pid]["exp_avg"].shape) == shapes[name], name
    # round trip into fresh flat buffers
    m2, v2 = torch.zeros(n), torch.zeros(n)
    assert CK.optimizer_state_from_torch(osd, layout, cfg.layers, m2, v2) == 7
    assert torch.equal(m, m2) and torch.equal(v, v2)


def test_written_optimizer_state_loads_into_torch_adamw():
    """What the reference does on --resume: optimizer.load_state_dict(checkpoint['optimizer']) on an AdamW built with its two
    groups (train_AT_text_only.py:333-341,366) -- torch checks group sizes and casts the state to the parameters."""
    cfg = MODEL_CONFIGS["tiny-test"]
    layout, n, _ = _layout(cfg)
    g0, g1 = CK.reference_param_groups(cfg.layers)
    shape = lambda k: layout[k][1] if k in layout else ()
    params = {k: torch.nn.Parameter(torch.zeros(shape(k))) for k in g0 + g1}
    opt = torch.optim.AdamW([{"params": [params[k] for k in g0], "weight_decay": 0.0},
                             {"params": [params[k] for k in g1], "weight_decay": 0.2}], lr=1e-3)
    rng = torch.Generator().manual_seed(0)
    m, v = torch.randn(n, generator=rng), torch.rand(n, generator=rng)
    opt.load_state_dict(CK.optimizer_state_to_torch(layout, cfg.layers, m, v, 5, 3e-4, (0.9, 0.98), 1e-6, 0.2))
    st = opt.state[params["transformer.resblocks.1.mlp.c_fc.weight"]]
    off, shp = layout["transformer.resblocks.1.mlp.c_fc.weight"]
    assert float(st["step"]) == 5 and torch.equal(st["exp_avg"], m[off:off + shp[0] * shp[1]].view(shp))
    assert params["logit_scale"] not in opt.state
    assert opt.param_groups[1]["lr"] == 3e-4 and opt.param_groups[1]["weight_decay"] == 0.2 and opt.param_groups[0]["weight_decay"] == 0.0
    # and the other direction: a state_dict written by torch's AdamW (= by the reference) fills the flat buffers
    for p_ in params.values():
        p_.grad = torch.ones_like(p_) if p_.ndim else None
    opt.step()
    m2, v2 = torch.zeros(n), torch.zeros(n)
    assert CK.optimizer_state_from_torch(opt.state_dict(), layout, cfg.layers, m2, v2) == 6
    assert torch.allclose(m2[off:off + 4], 0.9 * m[off:off + 4] + 0.1)


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
