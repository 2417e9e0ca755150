"""QKV GEMM -> attention as ONE launch (leaf_amd/csrc/qkv_attn.hip) against the two-kernel path it replaces
(set_option('fuse_attn', 0): LN-folded QKV GEMM -> [rows, 3d] buffer -> attn_fwd_kernel): bit-identical features, losses and
winners in every mode the big forward-only passes run in -- packed rows, dense rows, prefix reuse from a K/V cache, the fused
clean-caption pass, last-layer trimming on / off, tiles full of one-row sequences of many captions.

Reference op: nn.MultiheadAttention inside ResidualAttentionBlock (src/open_clip/transformer.py:225,239-252)."""
import numpy as np
import pytest

from oracle import text_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _model(name, seed, dtype="fp16"):
    from leaf_amd.model import create_model
    return create_model(name, device="cuda:0", dtype=dtype, seed=seed)


def _both(m, fn):
    """fn() with the fused launch and with the two kernels"""
    m.set_option("fuse_attn", 1)
    a = fn()
    m.set_option("fuse_attn", 0)
    b = fn()
    m.set_option("fuse_attn", 1)
    return a, b


def _eq(torch_mod, a, b, what):
    for x, y, name in zip(a, b, ("winners", "features", "losses")):
        if x is None and y is None:
            continue
        if not torch_mod.equal(x, y):
            d = (x.float() - y.float()).abs()
            raise AssertionError(f"{what}: {name} differ in {int((x != y).sum())} of {x.numel()} elements, max |diff| {float(d.max()):.3e}")


def _prefix_lens(cand, base):
    neq = cand != base[:, None, :]
    pl = neq.argmax(-1)
    pl[~neq.any(-1)] = cand.shape[-1]
    return pl.reshape(-1)


@pytest.mark.parametrize("model,dtype", [("ViT-L-14-quickgelu", "fp16"), ("ViT-L-14", "bf16")])
def test_fused_qkv_attention_is_bit_identical_packed_and_prefix(torch_mod, model, dtype):
    m = _model(model, 1, dtype)
    B, rho = 16, 50
    base = O.synthetic_tokens(B, seed=71, min_len=8, max_len=60)
    cand = O.synthetic_candidates(base, rho, seed=72)
    cand[:, 0] = base                          # no-op candidate: one computed row under prefix reuse
    cand[:, 1] = base
    cand[:, 1, 1] = 9                          # edit at position 1: the whole caption is recomputed
    flat = cand.reshape(-1, 77)
    lens = np.repeat(base.argmax(-1) + 1, rho)
    anchor = m.encode_text(base) + 0.2
    # packed rows, no prefix
    a, b = _both(m, lambda: m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens))
    _eq(torch_mod, a, b, "packed rows")
    # prefix reuse out of a cache written by the small-launch kernels
    kv = m.encode_text_kv(base)
    pl = _prefix_lens(cand, base)
    c, d = _both(m, lambda: m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl, kv=kv))
    _eq(torch_mod, c, d, "prefix reuse")
    _eq(torch_mod, a, c, "prefix reuse against full recomputation")
    # the clean captions riding in the first stage's launches; the cache it leaves must serve a later stage identically
    def fused_stage():
        i, f, cache, l = m.score_candidates_fused(base, base.argmax(-1) + 1, flat, anchor, rho, lens, pl, want_features=True, want_loss=True)
        i2, f2, l2 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl, kv=cache)
        # the cache's K and V thirds (layout [layers][rows][q | k | v]): with the fused launch the captions' q rows are not written
        # there at all -- nothing reads them (a caption's own attention runs inside the launch)
        rows, dd = cache["base_rows"], m.cfg.width
        kvv = cache["kv"][: rows * 3 * dd * 2 * m.cfg.layers].view(torch_mod.int16).view(m.cfg.layers, rows, 3 * dd)[:, :, dd:]
        return (i, f, l), (i2, f2, l2), kvv.clone()
    (e1, e2, kvbytes1), (g1, g2, kvbytes0) = _both(m, fused_stage)
    _eq(torch_mod, e1, g1, "fused caption pass")
    _eq(torch_mod, e2, g2, "second stage out of the fused pass's cache")
    _eq(torch_mod, a, e1, "fused caption pass against full recomputation")
    assert torch_mod.equal(kvbytes1, kvbytes0), "K/V cache bytes of the fused caption pass"


def test_fused_qkv_attention_dense_rows_trim_off_and_one_row_sequences(torch_mod):
    m = _model("ViT-L-14-quickgelu", 3)
    toks = O.synthetic_tokens(90, seed=81, min_len=1, max_len=75)
    lens = toks.argmax(-1) + 1
    # dense rows: 77 per sequence, three sequences per M tile
    m.trim_rows = False
    a, b = _both(m, lambda: (m.encode_text(toks),))
    assert torch_mod.equal(a[0], b[0]), "dense rows"
    m.trim_rows = True
    big = np.concatenate([toks] * 4, 0)
    blens = np.concatenate([lens] * 4)
    c, d = _both(m, lambda: (m.encode_text(big, seq_lens=blens), m.encode_text(big, normalize=True, seq_lens=blens)))
    assert torch_mod.equal(c[0], d[0]) and torch_mod.equal(c[1], d[1]), "packed rows"
    assert torch_mod.equal(c[0][:90], a[0]), "packed against dense"
    m.set_option("last_layer_trim", 0)
    e, f = _both(m, lambda: (m.encode_text(big, seq_lens=blens),))
    m.set_option("last_layer_trim", 1)
    assert torch_mod.equal(e[0], f[0]) and torch_mod.equal(e[0], c[0]), "last-layer trimming off"
    # tiles made of one-row sequences of many captions (rejected / duplicate candidates collapse to the no-op edit)
    B, rho = 64, 50
    base = O.synthetic_tokens(B, seed=82, min_len=6, max_len=40)
    cand = np.repeat(base[:, None, :], rho, axis=1)
    rng = np.random.default_rng(5)
    bl = base.argmax(-1) + 1
    for bi in range(B):
        for r in range(0, rho, 5):
            cand[bi, r, int(rng.integers(1, bl[bi] - 1))] = int(rng.integers(1, 49405))
    flat = cand.reshape(-1, 77)
    clens = np.repeat(bl, rho)
    anchor = m.encode_text(base) + 0.1
    kv = m.encode_text_kv(base)
    pl = _prefix_lens(cand, base)
    g, h = _both(m, lambda: m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=clens, prefix_lens=pl, kv=kv))
    _eq(torch_mod, g, h, "one-row sequences")


@pytest.mark.parametrize("model,B", [("ViT-H-14", 10), ("ViT-bigG-14", 8)])
def test_fused_qkv_attention_other_towers(torch_mod, model, B):
    """d = 1024 / 1280 (16 / 20 heads, K loops of 16 / 20 tiles, erf-GELU towers): packed rows and prefix reuse, fused against the two
    kernels, bit for bit."""
    m = _model(model, 2)
    rho = 50
    base = O.synthetic_tokens(B, seed=91, min_len=10, max_len=50)
    cand = O.synthetic_candidates(base, rho, seed=92)
    flat = cand.reshape(-1, 77)
    lens = np.repeat(base.argmax(-1) + 1, rho)
    assert lens.sum() // 256 * m.cfg.heads >= 256        # the fused launch engages
    anchor = m.encode_text(base) + 0.2
    a, b = _both(m, lambda: m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens))
    _eq(torch_mod, a, b, "packed rows")
    kv = m.encode_text_kv(base)
    pl = _prefix_lens(cand, base)
    c, d = _both(m, lambda: m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl, kv=kv))
    _eq(torch_mod, c, d, "prefix reuse")
    _eq(torch_mod, a, c, "prefix reuse against full recomputation")


def test_fused_qkv_attention_long_captions_one_caption_per_tile(torch_mod):
    """Captions of up to 77 tokens: the caption images take 80 rows each (three per tile still fit), prefixes of up to 75 positions,
    and with short suffixes many sequences per tile."""
    m = _model("ViT-L-14-quickgelu", 4)
    B, rho = 24, 50
    base = O.synthetic_tokens(B, seed=93, min_len=60, max_len=75)
    bl = base.argmax(-1) + 1
    cand = np.repeat(base[:, None, :], rho, axis=1)
    rng = np.random.default_rng(7)
    for bi in range(B):
        for r in range(rho):
            cand[bi, r, int(rng.integers(max(1, bl[bi] - 12), bl[bi] - 1))] = int(rng.integers(1, 49405))    # late edits: long prefixes
    flat = cand.reshape(-1, 77)
    lens = np.repeat(bl, rho)
    anchor = m.encode_text(base) + 0.1
    kv = m.encode_text_kv(base)
    pl = _prefix_lens(cand, base)
    a, b = _both(m, lambda: m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl, kv=kv))
    _eq(torch_mod, a, b, "long prefixes")
    full = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens)
    _eq(torch_mod, a, full, "long prefixes against full recomputation")


def test_scoring_passes_take_the_fused_launch(torch_mod):
    """No silent fall-back: in the benchmark's shape every transformer block of a scoring stage is ONE qkv_attn launch (profiler
    family 8) and neither a stand-alone LN-folded QKV GEMM nor an attention launch of its own; with the option off the GEMM is back."""
    import ctypes as C
    from leaf_amd import _lib
    lib = _lib.lib()
    m = _model("ViT-L-14-quickgelu", 1)
    B, rho = 32, 50
    base = O.synthetic_tokens(B, seed=95, min_len=8, max_len=40)
    cand = O.synthetic_candidates(base, rho, seed=96)
    flat = cand.reshape(-1, 77)
    lens = np.repeat(base.argmax(-1) + 1, rho)
    anchor = m.encode_text(base)
    kv = m.encode_text_kv(base)
    pl = _prefix_lens(cand, base)

    def families():
        lib.leaf_prof_begin()
        m.score_candidates(flat, anchor, rho, "l2", seq_lens=lens, prefix_lens=pl, kv=kv)
        ng = 64
        ms, fl, by = (C.c_double * ng)(), (C.c_double * ng)(), (C.c_double * ng)()
        rows, info, n = (C.c_int64 * ng)(), (C.c_int32 * (4 * ng))(), C.c_int(0)
        _lib.check(lib.leaf_prof_end_shapes(ms, fl, by, rows, info, ng, C.byref(n)), "prof")
        out = {}       # (family, epilogue, N) -> launches, summed over K (block 0 of the default arithmetic multiplies K = 3d: split QKV)
        for i in range(n.value):
            key = ((info[4 * i] % 1024) // 32, info[4 * i] % 16, info[4 * i + 1])
            out[key] = out.get(key, 0) + info[4 * i + 3]
        return out
    on = families()
    assert on.get((8, 5, 3 * 768)) == m.cfg.layers and not any(f != 8 and e == 5 and N == 3 * 768 for (f, e, N) in on), on
    m.set_option("fuse_attn", 0)
    off = families()
    m.set_option("fuse_attn", 1)
    assert (8, 5, 3 * 768) not in off and off.get((4, 5, 3 * 768)) == m.cfg.layers, off
