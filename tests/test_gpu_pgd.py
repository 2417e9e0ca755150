"""Optional embedding-space PGD mode on the GPU, through the C ABI, against the fixture the
reference's own pieces produced (tests/golden/make_golden_pgd.py) and against the numpy oracle."""
import os

import numpy as np
import pytest

from oracle import text_oracle as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(golden_dir):
    import torch
    from leaf_amd.model import create_model
    assert torch.cuda.is_available()
    z = np.load(os.path.join(golden_dir, "tiny_pgd.npz"))
    m = create_model("tiny-test-quickgelu", seed=12, trainable=True)
    lens = z["tokens"].argmax(-1) + 1
    keep = np.arange(77)[None, :] < lens[:, None]
    return torch, m, z, lens, keep


def _pack(dense, keep):            # [N, ctx, d] -> packed [rows, d]
    return np.ascontiguousarray(dense[keep])


def _unpack(packed, keep, d):      # packed [rows, d] -> [N, ctx, d] with zeros after EOT
    out = np.zeros(keep.shape + (d,), dtype=np.float32)
    out[keep] = packed
    return out


@pytest.mark.parametrize("norm", ["linf", "l2"])
def test_forward_and_input_grad_vs_reference_fixture(env, norm):
    torch, m, z, lens, keep = env
    toks, anchor = z["tokens"], torch.from_numpy(z["anchor"]).cuda()
    N = toks.shape[0]
    for k in range(3):
        delta = torch.from_numpy(_pack(z[f"{norm}_delta{k}"], keep)).cuda()
        feat = m.forward_train(toks, seq_lens=lens, delta=delta)
        assert rel_l2(feat.cpu().numpy(), z[f"{norm}_feat{k}"]) < 2e-3
        loss, g = m.input_grad(feat, anchor)
        want_loss = float(z[f"{norm}_loss{k}"]) / N          # fixture: sum over captions; TextFARE: mean
        assert abs(float(loss) - want_loss) < 3e-3 * want_loss
        got = _unpack(g.cpu().numpy(), keep, 128) * N
        assert rel_l2(got, z[f"{norm}_grad{k}"]) < 8e-3      # fp16 loss-scaled gradient path


@pytest.mark.parametrize("norm", ["linf", "l2"])
def test_pgd_step_kernel_matches_the_reference_update(env, norm):
    """The fused update on the fixture's OWN gradient: linf is exact, l2 to fp32 rounding."""
    torch, m, z, lens, keep = env
    eps, alpha = float(z[f"{norm}_eps"]), float(z[f"{norm}_alpha"])
    m.forward_train(z["tokens"], seq_lens=lens)               # sets the row plan the step uses
    for k in range(3):
        delta = torch.from_numpy(_pack(z[f"{norm}_delta{k}"], keep)).cuda()
        grad = torch.from_numpy(_pack(z[f"{norm}_grad{k}"], keep)).cuda()
        m.pgd_step(delta, grad, alpha, eps, norm)
        got = _unpack(delta.cpu().numpy(), keep, 128)
        want = z[f"{norm}_delta{k + 1}"]
        if norm == "linf":
            assert np.array_equal(got, want)
        else:
            assert np.abs(got - want).max() < 2e-6
        ora = O.pgd_step(z[f"{norm}_delta{k}"], z[f"{norm}_grad{k}"], alpha, eps, norm)
        assert np.abs(got - ora).max() < 2e-6


@pytest.mark.parametrize("norm", ["linf", "l2"])
def test_attack_embedding_pgd_trajectory(env, norm):
    """k = 3 steps end to end from the fixture's start: the loss rises like the reference's, delta stays in the ball and
    ends near the reference's delta (sign() of near-zero gradient components may differ in 16-bit arithmetic)."""
    torch, m, z, lens, keep = env
    from leaf_amd.attacks import attack_embedding_pgd
    eps, alpha = float(z[f"{norm}_eps"]), float(z[f"{norm}_alpha"])
    toks, anchor = z["tokens"], torch.from_numpy(z["anchor"]).cuda()
    d0 = torch.from_numpy(_pack(z[f"{norm}_delta0"], keep))
    feat, delta = attack_embedding_pgd(m, toks, anchor, eps, alpha, k=3, norm=norm, delta0=d0, seq_lens=lens)
    got = _unpack(delta.cpu().numpy(), keep, 128)
    want = z[f"{norm}_delta3"]
    if norm == "linf":
        assert np.abs(got).max() <= eps * (1 + 1e-6)
        assert (np.abs(got - want) < 1e-6).mean() > 0.97
    else:
        nrm = np.sqrt((got.reshape(got.shape[0], -1) ** 2).sum(1))
        assert (nrm <= eps * (1 + 1e-5)).all()
        assert rel_l2(got, want) < 3e-2
    N = toks.shape[0]
    end_loss = float(((z["anchor"] - feat.cpu().numpy()) ** 2).sum())
    ref_end = float(((z["anchor"] - O.encode_text(O.init_weights(O.TextCfg(128, 2, 2, 64, quick_gelu=True), seed=12),
                                                  O.TextCfg(128, 2, 2, 64, quick_gelu=True), toks, delta=want)) ** 2).sum())
    assert end_loss > float(z[f"{norm}_loss0"])               # the attack increased the distance to the anchor
    assert abs(end_loss - ref_end) < 2e-2 * ref_end


def test_large_tower_input_grad_vs_oracle(env):
    """ViT-L shapes: forward with a perturbation and d loss / d delta against the fp32 oracle (a few captions)."""
    torch, _, _, _, _ = env
    from leaf_amd.model import create_model
    cfg = O.CONFIGS["ViT-L-14"]
    w = O.init_weights(cfg, seed=1)
    m = create_model("ViT-L-14", seed=1, trainable=True)
    toks = O.synthetic_tokens(4, seed=77, min_len=6, max_len=30)
    lens = toks.argmax(-1) + 1
    keep = np.arange(77)[None, :] < lens[:, None]
    rng = np.random.default_rng(3)
    dense = (0.02 * rng.standard_normal((4, 77, cfg.width))).astype(np.float32) * keep[:, :, None]
    anchor = (O.encode_text(w, cfg, toks) + 0.5 * rng.standard_normal((4, cfg.embed_dim))).astype(np.float32)
    loss_o, feat_o, g = O.encode_text_backward(w, cfg, toks, anchor, delta=dense)
    delta = torch.from_numpy(_pack(dense, keep)).cuda()
    feat = m.forward_train(toks, seq_lens=lens, delta=delta)
    assert rel_l2(feat.cpu().numpy(), feat_o) < 1e-3
    loss, gd = m.input_grad(feat, torch.from_numpy(anchor).cuda())
    assert abs(float(loss) - loss_o) < 2e-3 * loss_o
    assert rel_l2(_unpack(gd.cpu().numpy(), keep, cfg.width), g["d_embed"]) < 8e-3
