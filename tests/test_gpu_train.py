"""Training half on the GPU: stash forward == inference forward, TextFARE loss, gradients, AdamW, one full step."""
import json
import os

import numpy as np
import pytest

from oracle import text_oracle as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("name,seed,model,qg", [("tiny_gelu", 11, "tiny-test", False),
                                                 ("tiny_quickgelu", 12, "tiny-test-quickgelu", True)])
def test_loss_and_grads_vs_golden(torch_mod, golden_dir, name, seed, model, qg):
    from leaf_amd.model import create_model
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = create_model(model, seed=seed, trainable=True)
    toks = z["tokens"][:8]
    feat = m.forward_train(toks)
    # the training forward keeps explicit LayerNorm outputs in its stash; the inference forward folds LayerNorm into the GEMMs
    # (lnfold.h): same maths, different roundings.  With folding off the two are the same instruction sequence bit for bit.
    assert rel_l2(feat.cpu().numpy(), m.encode_text(toks).cpu().numpy()) < 1.5e-3
    m.set_option("ln_fold", 0)
    assert np.array_equal(feat.cpu().numpy(), m.encode_text(toks).cpu().numpy()), "stash forward must equal the unfolded inference forward"
    m.set_option("ln_fold", 1)
    m.zero_grad()
    loss = m.backward(feat, torch_mod.from_numpy(z["anchor"]).cuda(), accum_scale=0.5)
    torch_mod.cuda.synchronize()
    assert abs(float(loss) - float(z["loss"])) < 2e-3 * abs(float(z["loss"]))     # P3
    worst = {}
    for k, (off, shape) in m.layout.items():
        g = m.grads[off: off + int(np.prod(shape))].view(shape).cpu().numpy()
        if k == "token_embedding.weight":
            rows = z["tok_rows"]
            r = rel_l2(g[rows], z["g_tok_rows"])
            assert np.abs(np.delete(g, rows, axis=0)).sum() == 0.0
        else:
            r = rel_l2(g, z["g:" + k])
        worst[k] = r
    tol = 1.2e-2 if os.environ.get("LEAF_GRAD_DTYPE", "").lower().startswith("b") else 6e-3
    bad = {k: v for k, v in worst.items() if v > tol}
    print("max grad rel-L2", max(worst.values()))
    # P4: default gradient path = fp16 operands + per-step power-of-two loss scale (11 significand bits, the
    # reference's fp16-autocast regime): measured per-tensor rel-L2 3.5-4.0e-3 vs the fp32 reference (bound 6e-3);
    # bf16 (LEAF_GRAD_DTYPE=bf16) measures 7-8e-3 (bound 1.2e-2).  The fixture's anchor sits ~||f|| away from f so the
    # forward's own 1e-3 error is not amplified by cancellation in (f - anchor).
    assert not bad, bad
    # accumulate: a second backward doubles the gradient
    g1 = m.grads.clone()
    m.backward(feat, torch_mod.from_numpy(z["anchor"]).cuda(), accum_scale=0.5)
    assert rel_l2(m.grads.cpu().numpy(), 2 * g1.cpu().numpy()) < 1e-6


def test_adamw_kernel_vs_oracle(torch_mod):
    import ctypes as C
    from leaf_amd import _lib
    from tests.util import ptr, stream
    lib = _lib.lib()
    rng = np.random.default_rng(0)
    n, nd = 4096, 1000
    p = rng.standard_normal(n).astype(np.float32)
    g = (rng.standard_normal(n) * 1e-3).astype(np.float32)
    w = {"a.weight": p[:nd].reshape(10, 100).copy(), "b.bias": p[nd:].copy()}
    gg = {"a.weight": g[:nd].reshape(10, 100).copy(), "b.bias": g[nd:].copy()}
    mm = {k: np.zeros_like(v) for k, v in w.items()}
    vv = {k: np.zeros_like(v) for k, v in w.items()}
    tp, tg = torch_mod.from_numpy(p.copy()).cuda(), torch_mod.from_numpy(g).cuda()
    tm, tv = torch_mod.zeros_like(tp), torch_mod.zeros_like(tp)
    for step in (1, 2, 3):
        O.adamw_step(w, gg, mm, vv, step, lr=1e-3, wd=0.1, beta1=0.9, beta2=0.98, eps=1e-6)
        _lib.check(lib.leaf_adamw_step(ptr(tp), ptr(tg), ptr(tm), ptr(tv), n, nd, 1e-3, 0.9, 0.98, 1e-6, 0.1, step, 1.0, stream()), "adamw")
    torch_mod.cuda.synchronize()
    want = np.concatenate([w["a.weight"].ravel(), w["b.bias"]])
    assert np.abs(tp.cpu().numpy() - want).max() < 1e-6


def test_full_step_moves_loss_down(torch_mod):
    """anchor -> search (k=1, rho=8) -> train fwd/bwd -> AdamW, six times on the tiny model: the TextFARE loss of
    the same adversarial batch must drop after the updates (end-to-end plumbing of utils_AT.py:282-362)."""
    from leaf_amd.model import LeafCLIPText, create_model, get_config
    m = create_model("tiny-test-quickgelu", seed=12, trainable=True)
    frozen = LeafCLIPText(get_config("tiny-test-quickgelu")).copy_from(m)
    base = O.synthetic_tokens(16, seed=9)
    cand = O.synthetic_candidates(base, 8, seed=10)
    anchor = frozen.encode_text(base)
    idx, _ = m.score_candidates(cand.reshape(-1, 77), anchor, 8, "l2", want_features=False)
    adv = cand[np.arange(16), idx.cpu().numpy()]
    losses = []
    for _ in range(6):
        feat = m.forward_train(adv)
        m.zero_grad()
        losses.append(float(m.backward(feat, anchor)))
        m.adamw_step(lr=1e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-4)
    feat = m.encode_text(adv)
    final = float(((feat - anchor) ** 2).sum(-1).mean())
    assert np.isfinite(losses).all() and final < losses[0], (losses, final)


def test_packed_training_matches_dense(torch_mod):
    """Training forward/backward on EOT-trimmed rows gives the same loss and gradients as the dense layout (only the
    order of fp32 atomic adds differs)."""
    from leaf_amd.model import create_model
    toks = O.synthetic_tokens(8, seed=21, min_len=3, max_len=60)
    res = []
    for trim in (True, False):
        m = create_model("tiny-test", seed=11, trainable=True)
        m.trim_rows = trim
        anchor = m.encode_text(toks) + 0.1
        feat = m.forward_train(toks)
        m.zero_grad()
        loss = float(m.backward(feat, anchor))
        res.append((loss, feat.cpu().numpy(), m.grads.cpu().numpy()))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    assert rel_l2(res[0][2], res[1][2]) < 1e-5


def test_normalize_fare_vs_reference_fixture(torch_mod, golden_dir):
    """--normalize_fare (utils_AT.py:296,319): normalised training features, the loss on them and every gradient against
    tests/golden/tiny_normfare.npz (the reference's encode_text(normalize=True) + torch.autograd)."""
    from leaf_amd.model import create_model
    z = np.load(os.path.join(golden_dir, "tiny_normfare.npz"))
    m = create_model("tiny-test-quickgelu", seed=12, trainable=True)
    feat = m.forward_train(z["tokens"], normalize=True)
    assert rel_l2(feat.cpu().numpy(), z["feat"]) < 2e-3
    m.set_option("ln_fold", 0)      # same instruction sequence as the training forward (see test_loss_and_grads_vs_golden)
    assert np.array_equal(feat.cpu().numpy(), m.encode_text(z["tokens"], normalize=True).cpu().numpy())
    m.set_option("ln_fold", 1)
    assert rel_l2(feat.cpu().numpy(), m.encode_text(z["tokens"], normalize=True).cpu().numpy()) < 1.5e-3
    m.zero_grad()
    loss = m.backward(feat, torch_mod.from_numpy(z["anchor"]).cuda())
    torch_mod.cuda.synchronize()
    assert abs(float(loss) - float(z["loss"])) < 3e-3 * float(z["loss"])
    worst = 0.0
    for k, (off, shape) in m.layout.items():
        g = m.grads[off: off + int(np.prod(shape))].view(shape).cpu().numpy()
        if k == "token_embedding.weight":
            r = rel_l2(g[z["tok_rows"]], z["g_tok_rows"])
        elif "g:" + k in z.files:
            r = rel_l2(g, z["g:" + k])
        else:
            continue
        worst = max(worst, r)
        assert r < 1.0e-2, (k, r)
    print(f"normalize_fare: worst gradient rel-L2 {worst:.2e}")
    feat2 = m.forward_train(z["tokens"])                  # switching back gives the un-normalised features again
    assert rel_l2(feat2.cpu().numpy() / np.linalg.norm(feat2.cpu().numpy(), axis=-1, keepdims=True), z["feat"]) < 2e-3


def test_grad_clip_norm_vs_reference_fixture(torch_mod, golden_dir):
    """--grad-clip-norm (utils_AT.py:348-357): total norm and the clipped AdamW step against tests/golden/tiny_clip.npz
    (the reference model + torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW) and against the oracle."""
    from leaf_amd.model import create_model
    z = np.load(os.path.join(golden_dir, "tiny_clip.npz"))
    zq = np.load(os.path.join(golden_dir, "tiny_quickgelu.npz"))
    m = create_model("tiny-test-quickgelu", seed=12, trainable=True)
    toks = zq["tokens"][:8]
    feat = m.forward_train(toks)
    m.zero_grad()
    m.backward(feat, torch_mod.from_numpy(zq["anchor"]).cuda())
    total = m.adamw_step(1e-3, (0.9, 0.98), 1e-6, 0.2, max_norm=float(z["max_norm"]))
    torch_mod.cuda.synchronize()
    assert abs(float(total) - float(z["total_norm"])) < 6e-3 * float(z["total_norm"])      # fp16 gradient path
    # oracle restatement of the same clip on the oracle's own fp32 gradients reproduces the reference's total norm
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=12)
    _, _, g = O.encode_text_backward(w, cfg, toks, zq["anchor"])
    assert abs(O.clip_grad_norm(g, float(z["max_norm"])) - float(z["total_norm"])) < 1e-4 * float(z["total_norm"])
    lr = 1e-3
    for k, (off, shape) in m.layout.items():
        p = m.flat[off: off + int(np.prod(shape))].view(shape).cpu().numpy()
        if k == "token_embedding.weight":
            d = np.abs(p[z["tok_rows"]] - z["after_tok_rows"])
        elif "after:" + k in z.files:
            d = np.abs(p - z["after:" + k])
        else:
            continue
        # after clipping by 0.002 many gradient entries sit near Adam's eps (1e-6), where the first step lr * g / (|g| + eps)
        # amplifies the 16-bit gradient noise: a few elements may differ by up to 2 lr (sign flip), the mean must not
        # (the key third of in_proj_bias has a mathematically zero gradient: pure noise over eps there)
        assert d.max() < 2.5 * lr and d.mean() < 5e-2 * lr, (k, d.max(), d.mean())


def test_micro_batch_clipping_vs_reference_fixture(torch_mod, golden_dir):
    """--grad-clip-norm with --accum-freq 2 (utils_AT.py:348-362): the trainer's MicroClip clips the running sum after each
    micro-batch's backward (leaf_clip_grads_inplace); norms and the gradient the step sees against the reference fixture
    (tests/golden/make_golden_microclip.py).  Under --precision amp the reference's GradScaler refuses the second unscale_."""
    import types
    from leaf_amd.model import create_model
    from leaf_amd.train import MicroClip
    z = np.load(os.path.join(golden_dir, "tiny_microclip.npz"))
    zq = np.load(os.path.join(golden_dir, "tiny_quickgelu.npz"))
    args = types.SimpleNamespace(grad_clip_norm=float(z["max_norm"]), accum_freq=2, precision="amp_bf16")
    with pytest.raises(RuntimeError, match="unscale_"):
        MicroClip.check(types.SimpleNamespace(grad_clip_norm=1.0, accum_freq=2, precision="amp"))
    MicroClip.check(args)
    MicroClip.check(types.SimpleNamespace(grad_clip_norm=1.0, accum_freq=1, precision="amp"))
    m = create_model("tiny-test-quickgelu", seed=12, trainable=True)
    mc = MicroClip(m, args)
    assert mc.active
    toks, anchor = zq["tokens"][:8], torch_mod.from_numpy(zq["anchor"]).cuda()
    m.zero_grad()
    norms = []
    for j in range(2):
        feat = m.forward_train(toks[4 * j: 4 * j + 4])
        mc.backward(feat, anchor[4 * j: 4 * j + 4], j)
        norms.append(float(m._clipi_ws[1]))
        assert abs(float(m._clipi_ws[0]) - float(z["max_norm"]) / (norms[-1] + 1e-6)) < 1e-6
    assert np.allclose(norms, z["norms"], rtol=6e-3), (norms, z["norms"])           # fp16 gradient path
    tot = float(torch_mod.linalg.vector_norm(m.grads))
    assert abs(tot - float(z["final_grad_norm"])) < 6e-3 * float(z["final_grad_norm"]) and tot <= float(z["max_norm"]) * (1 + 1e-5)
    for k, (off, shape) in m.layout.items():
        g = m.grads[off: off + int(np.prod(shape))].view(shape).cpu().numpy()
        if k == "token_embedding.weight":
            assert rel_l2(g[z["tok_rows"]], z["grad_tok_rows"]) < 1e-2, k
        elif "grad:" + k in z.files and np.linalg.norm(z["grad:" + k]) > 1e-3 * float(z["final_grad_norm"]):
            assert rel_l2(g, z["grad:" + k]) < 1e-2, k
    # pre_scale alone (the data-parallel hand-over between micro-batches): an exact power-of-two scaling, no clip
    before = m.grads.clone()
    m.clip_grads_(None, pre_scale=0.5)
    assert torch_mod.equal(m.grads, before * 0.5)


def test_non_finite_gradients_skip_the_step(torch_mod):
    """a NaN / inf anywhere in the gradient buffer must not reach the weights or the AdamW moments (and, through
    the flat all-reduce, the other ranks): the fused step is skipped as a whole, as GradScaler.step does in the reference's
    fp16 regime, and counted; the next finite step runs normally."""
    from leaf_amd.model import create_model
    m = create_model("tiny-test-quickgelu", seed=12, trainable=True)
    toks = O.synthetic_tokens(6, seed=4)
    anchor = m.encode_text(toks) + 0.3
    f = m.forward_train(toks)
    m.zero_grad()
    m.backward(f, anchor)
    good = m.grads.clone()
    p0, m0, v0 = m.flat.clone(), m.exp_avg.clone(), m.exp_avg_sq.clone()
    for poison in (float("nan"), float("inf")):
        m.grads.copy_(good)
        m.grads[12345] = poison
        norm = m.adamw_step(lr=1e-3, weight_decay=0.1)
        assert not bool(torch_mod.isfinite(norm))
        assert torch_mod.equal(m.flat, p0) and torch_mod.equal(m.exp_avg, m0) and torch_mod.equal(m.exp_avg_sq, v0)
    assert m.skipped_steps() == 2
    m.grads.copy_(good)
    norm = m.adamw_step(lr=1e-3, weight_decay=0.1)
    assert bool(torch_mod.isfinite(norm)) and abs(float(norm) - float(good.double().norm())) < 1e-4 * float(norm)
    assert not torch_mod.equal(m.flat, p0) and m.skipped_steps() == 2 and bool(torch_mod.isfinite(m.flat).all())
    # guard off: with the gradient scaler attached (the default for a trainable model) the guard stays on -- a saturated backward
    # poisons the gradient with a NaN that only the guard stops; without the scaler it is the plain kernel
    assert m.adamw_step(lr=1e-3, guard=False) is not None
    m._scaler_attached = False
    assert m.adamw_step(lr=1e-3, guard=False) is None


def test_fp16_gradient_saturation_skips_the_step_and_halves_the_loss_scale(torch_mod):
    """GradScaler semantics for the fp16 gradient path (train_AT_text_only.py:347, utils_AT.py:339-362; VERDICT r2 missing-6): the
    16-bit conversions saturate at 65504 instead of producing inf, so an overflow must be FOUND -- the backward checks its 16-bit
    gradient tensors, poisons the gradient, the guarded step skips (weights, moments and the bias-correction step untouched) and
    the persistent loss-scale factor halves; once the gradients fit, steps apply again, and after `growth interval` clean steps
    the factor grows back.  The planted model has an ln_final gain of 3e4: the features (computed in exact fp32) are simply 3e4 x
    larger, but the gradient entering the last block is 3e4 x larger than the loss gradient the scale S is chosen from, so its
    16-bit copy overflows fp16 at the usual S."""
    from leaf_amd import _lib
    from leaf_amd.model import LeafCLIPText, get_config
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=12)
    w["ln_final.weight"] = (w["ln_final.weight"] * 3e4).astype(np.float32)
    m = LeafCLIPText(get_config("tiny-test-quickgelu"), trainable=True)
    m.load_state_dict({n: torch_mod.from_numpy(v) for n, v in w.items()}, strict=False)
    toks = O.synthetic_tokens(6, seed=4)
    feat_ref = O.encode_text(w, cfg, toks)
    anchor_np = (feat_ref + 0.3 * np.abs(feat_ref).mean()).astype(np.float32)
    anchor = torch_mod.from_numpy(anchor_np).cuda()
    _, _, g_ref = O.encode_text_backward(w, cfg, toks, anchor_np)
    p0 = m.flat.clone()
    m._clip_ws[_lib.SC_INTERVAL] = 2.0          # grow back after two clean steps (GradScaler's growth_interval, default 2000)
    history = []
    for it in range(12):
        f = m.forward_train(toks)
        m.zero_grad()
        m.backward(f, anchor)
        grads = m.grads.clone()
        norm = m.adamw_step(lr=0.0)              # lr 0: the weights stay put, the step bookkeeping runs
        st = m.grad_scaler_state()
        history.append((bool(torch_mod.isfinite(norm)), st["loss_scale_factor"], st["skipped"], st["skipped_saturated"]))
        if history[-1][0] and len([h for h in history if h[0]]) == 1:
            # first applied step: the gradient at the backed-off scale matches the oracle's fp32 backward
            for name in ("transformer.resblocks.1.mlp.c_proj.weight", "transformer.resblocks.0.attn.in_proj_weight", "positional_embedding"):
                off, shape = m.layout[name]
                got = grads[off: off + int(np.prod(shape))].view(shape).cpu().numpy()
                assert rel_l2(got, g_ref[name]) < 2e-2, name
            assert m.applied_steps() == 1 and m.opt_step == it + 1
    print("scaler history (applied, factor, skipped, skipped_saturated):", history)
    assert torch_mod.equal(m.flat, p0)
    assert not history[0][0] and history[0][1] == 0.5 and history[0][3] == 1, "the first step saturates: skipped, scale halved"
    n_skip0 = next(i for i, h in enumerate(history) if h[0])          # leading skipped steps
    assert 1 <= n_skip0 <= 8 and all(history[i][1] == 0.5 ** (i + 1) for i in range(n_skip0))
    assert history[n_skip0][1] == 0.5 ** n_skip0 and history[n_skip0 + 1][0], "two clean steps at the backed-off scale"
    # ... after which the factor doubles and the very next step overflows again, is skipped and halves it back
    assert history[n_skip0 + 1][1] == 0.5 ** (n_skip0 - 1) and not history[n_skip0 + 2][0] and history[n_skip0 + 2][1] == 0.5 ** n_skip0
    assert history[-1][2] == history[-1][3], "every skipped step of this run was a saturation"
    assert m.applied_steps() == sum(1 for h in history if h[0]) and m.opt_step == 12


def test_deep_tower_layernorm_gradients_vs_oracle(torch_mod):
    """A tower with more LayerNorms than one ln_param_reduce launch takes (34 blocks = 68 > 64): every LayerNorm / bias
    gradient through the per-workgroup partial sums + batched reduction (train.hip) against the oracle's fp32 backward."""
    from leaf_amd.model import LeafCLIPText, TextConfig
    cfg = TextConfig(128, 2, 34, 64, quick_gelu=True)
    ocfg = O.TextCfg(128, 2, 34, 64, quick_gelu=True)
    w = O.init_weights(ocfg, seed=21)
    m = LeafCLIPText(cfg, trainable=True)
    m.load_state_dict({k: torch_mod.from_numpy(v) for k, v in w.items()}, strict=False)
    toks = O.synthetic_tokens(6, seed=5)
    feat_ref = O.encode_text(w, ocfg, toks)
    rng = np.random.default_rng(3)
    anchor = (feat_ref + rng.standard_normal(feat_ref.shape).astype(np.float32) * np.linalg.norm(feat_ref, axis=-1, keepdims=True) /
              np.sqrt(feat_ref.shape[-1])).astype(np.float32)
    loss_ref, _, g = O.encode_text_backward(w, ocfg, toks, anchor)
    feat = m.forward_train(toks)
    m.zero_grad()
    loss = m.backward(feat, torch_mod.from_numpy(anchor).cuda())
    torch_mod.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 3e-3 * abs(float(loss_ref))
    worst = 0.0
    for l in (0, 1, 16, 31, 32, 33):
        for name in ("ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias", "attn.out_proj.bias", "mlp.c_fc.weight"):
            k = f"transformer.resblocks.{l}.{name}"
            off, shape = m.layout[k]
            got = m.grads[off: off + int(np.prod(shape))].view(shape).cpu().numpy()
            worst = max(worst, rel_l2(got, g[k]))
    print("deep tower: max grad rel-L2", worst)
    assert worst < 8e-3        # measured 2.6e-3 (width 128: short reductions; ViT-H, 24 blocks of width 1024: 7.8e-3)
