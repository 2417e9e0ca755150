"""fp32-grade ``encode_text(precise=True)`` (leaf_text_forward_precise, leaf_amd/csrc/precise.hip) through the C ABI
against the fixtures the reference itself produced (``CLIP.encode_text``, src/open_clip/model.py:269-284) and against the fp32
oracle.  Tolerance: 2e-5 rel-L2 PER ROW -- fp32 reorder noise is ~1e-6; the 16-bit forward the search runs is gated at 1e-3."""
import os

import numpy as np
import pytest

from oracle import text_oracle as O
from tests.util import rel_l2, row_rel_l2, trained_like_weights

pytestmark = pytest.mark.gpu
TOL_PRECISE_ROW = 2.0e-5


def _model(name, seed):
    from leaf_amd.model import create_model
    return create_model(name, device="cuda:0", seed=seed)


@pytest.mark.parametrize("fname,model,seed", [("tiny_gelu", "tiny-test", 11), ("tiny_quickgelu", "tiny-test-quickgelu", 12),
                                              ("vitl_gelu", "ViT-L-14", 1), ("vitl_quickgelu", "ViT-L-14-quickgelu", 1)])
def test_precise_encode_text_vs_reference_fixtures(golden_dir, fname, model, seed):
    z = np.load(os.path.join(golden_dir, fname + ".npz"))
    m = _model(model, seed)
    out = m.encode_text(z["tokens"], precise=True).cpu().numpy()
    r = row_rel_l2(out, z["out"])
    print(f"[precise] {model}: row max {r.max():.3e} median {np.median(r):.3e}")
    assert np.isfinite(out).all() and r.max() < TOL_PRECISE_ROW
    if "out_norm" in z.files:
        outn = m.encode_text(z["tokens"], normalize=True, precise=True).cpu().numpy()
        assert row_rel_l2(outn, z["out_norm"]).max() < TOL_PRECISE_ROW


def test_precise_rows_do_not_depend_on_the_layout_or_the_batch():
    """EOT trimming / dense rows, a caption alone or inside a ragged batch (shortest possible, all 77 positions), more sequences than one
    chunk: the same bits per caption -- every kernel of the precise path computes a row from that row's data in a fixed order."""
    import torch
    m = _model("tiny-test-quickgelu", 12)
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=12)
    t = O.synthetic_tokens(700, seed=5, min_len=1, max_len=70).astype(np.int32)      # 700 > the 512-sequence chunk of the precise path
    t[0, :] = 0
    t[0, 0], t[0, 1] = 49406, 49407
    t[1, :76], t[1, 76] = 5, 49407
    got = m.encode_text(t, precise=True).cpu().numpy()
    m.trim_rows = False
    dense = m.encode_text(t, precise=True).cpu().numpy()
    m.trim_rows = True
    assert np.array_equal(got, dense)
    for i in (0, 1, 2, 699):
        assert np.array_equal(m.encode_text(t[i:i + 1], precise=True).cpu().numpy()[0], got[i])
    dev = m.encode_text(torch.from_numpy(t).cuda(), precise=True).cpu().numpy()       # device-resident ids: lengths unknown, dense rows
    assert np.array_equal(dev, got)
    want = O.encode_text(w, cfg, t[:40])
    assert row_rel_l2(got[:40], want).max() < TOL_PRECISE_ROW


@pytest.mark.parametrize("name,seed,n", [("ViT-H-14", 2, 6), ("ViT-bigG-14", 2, 4)])
def test_precise_on_the_larger_towers_vs_oracle(name, seed, n):
    cfg = O.CONFIGS[name]
    w = O.init_weights(cfg, seed=seed)
    m = _model(name, seed)
    toks = O.synthetic_tokens(n, seed=11, min_len=5, max_len=30)
    L = int(toks.argmax(-1).max()) + 1
    want = O.encode_text(w, cfg, toks[:, :L])
    r = row_rel_l2(m.encode_text(toks, precise=True).cpu().numpy(), want)
    print(f"[precise] {name}: row max {r.max():.3e}")
    assert r.max() < TOL_PRECISE_ROW


def test_precise_on_a_trained_like_tower_meets_1e3_per_row_where_16_bit_operands_cannot():
    """The tower of test_vitl_trained_like_spectrum_... (power-law spectra, log-normal LayerNorm gains with outlier channels): 16-bit
    operands are 1e-3 (median) / 2.4e-3 (worst row) from fp32 there.  The precise mode is gated at north_star's 1e-3 per row and in
    fact sits four orders below the 16-bit forward."""
    name = "ViT-L-14-quickgelu"
    cfg = O.CONFIGS[name]
    w = trained_like_weights(cfg, seed=0)
    base = O.synthetic_tokens(12, seed=71, min_len=4, max_len=40)
    L = int(base.argmax(-1).max()) + 1
    want = O.encode_text(w, cfg, base[:, :L])
    m = _model(name, 1)
    m.load_state_dict(w)
    r16 = row_rel_l2(m.encode_text(base).cpu().numpy(), want)
    rp = row_rel_l2(m.encode_text(base, precise=True).cpu().numpy(), want)
    print(f"[trained-like] 16-bit rows max {r16.max():.3e} median {np.median(r16):.3e} | precise rows max {rp.max():.3e} median {np.median(rp):.3e}")
    assert rp.max() < 1e-3 and rp.max() < 1e-4, "precise mode must be far inside north_star's tolerance on the hard tower"
    assert rp.max() < 0.05 * r16.max()


def test_precise_follows_the_current_master_weights_without_a_pack():
    """The precise path reads the fp32 masters: after an optimizer-like change of the weights it needs no pack() (the 16-bit path does)."""
    m = _model("tiny-test-quickgelu", 12)
    t = O.synthetic_tokens(4, seed=2)
    a = m.encode_text(t, precise=True).cpu().numpy()
    m.flat.mul_(1.01)
    b = m.encode_text(t, precise=True).cpu().numpy()
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = {k: v * np.float32(1.01) for k, v in O.init_weights(cfg, seed=12).items()}
    assert not np.array_equal(a, b)
    assert row_rel_l2(b, O.encode_text(w, cfg, t)).max() < TOL_PRECISE_ROW
