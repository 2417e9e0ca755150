"""Parity of the HIP text path (through the C ABI) against the golden fixtures produced by the reference and
against the CPU oracle.  Tolerance (north_star): embeddings within 1e-3 rel-L2 of the fp32 reference, fp16 MFMA
operands with fp32 accumulation; the per-row bound is north_star's 1e-3 too (tests/util.py:TOL_ROW; measured maxima 9.2e-4 ... 9.9e-4)."""
import json
import os

import numpy as np
import pytest

from oracle import text_oracle as O
from tests.util import TOL_ROW, rel_l2, row_rel_l2

pytestmark = pytest.mark.gpu
TOL_GLOBAL = 1.0e-3


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _model(name, seed, dtype="fp16", **kw):
    from leaf_amd.model import create_model
    return create_model(name, device="cuda:0", dtype=dtype, seed=seed, **kw)


def test_init_matches_oracle_weights(torch_mod):
    m = _model("tiny-test", 11)
    w = O.init_weights(O.TextCfg(128, 2, 2, 64), seed=11)
    for k, v in w.items():
        assert np.array_equal(m.params[k].cpu().numpy(), v), k


@pytest.mark.parametrize("name,seed,model", [("tiny_gelu", 11, "tiny-test"), ("tiny_quickgelu", 12, "tiny-test-quickgelu")])
def test_encode_text_tiny_golden(torch_mod, golden_dir, name, seed, model):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = _model(model, seed)
    out = m.encode_text(torch_mod.from_numpy(z["tokens"].astype(np.int64))).cpu().numpy()
    assert np.isfinite(out).all()
    assert rel_l2(out, z["out"]) < TOL_GLOBAL
    # d = 128: eight times fewer terms per dot product than ViT-L, so the per-row figure scatters more -- the FORMAT's own noise
    # exceeds 1e-3 here: the oracle's emulation of fp16 operands gives row max 9.1e-4 / 1.09e-3 on these two fixtures
    # (tests/test_oracle_golden.py::test_fp16_operand_noise_of_the_tiny_config; BASELINE.md section 4 says why d = 128 is exempt
    # from north_star's per-row figure); measured on the GPU: 1.28e-3 in the round-5 arithmetic, 7.0e-4 / 9.6e-4 on these two fixtures in
    # the default ('rowsafe': block 0's QKV and out-projection on splits) -- 400 synthetic d = 128 rows still reach 1.27e-3, so the
    # config stays exempt; this fixture is held to 1.1e-3.  The BASELINE.json towers keep TOL_ROW.
    assert row_rel_l2(out, z["out"]).max() < 1.1e-3
    outn = m.encode_text(torch_mod.from_numpy(z["tokens"].astype(np.int64)), normalize=True).cpu().numpy()
    assert rel_l2(outn, z["out_norm"]) < TOL_GLOBAL


def test_encode_text_degenerate_batches(torch_mod):
    """empty batch -> empty [0, D] (what the torch module returns); the shortest caption there is (SOT, EOT) and one that fills all
    77 positions, alone and mixed in one ragged batch, against the oracle"""
    m = _model("tiny-test-quickgelu", 12)
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=12)
    assert tuple(m.encode_text(np.zeros((0, 77), np.int32)).shape) == (0, cfg.embed_dim)
    assert tuple(m.encode_text(torch_mod.zeros(0, 77, dtype=torch_mod.int64)).shape) == (0, cfg.embed_dim)
    t = np.zeros((3, 77), np.int32)
    t[0, 0], t[0, 1] = 49406, 49407                                 # SOT EOT
    t[1, :76], t[1, 76] = 5, 49407                                  # EOT at the last position
    t[2, 0], t[2, 1:9], t[2, 9] = 49406, np.arange(1, 9), 49407
    want = O.encode_text(w, cfg, t)
    got = m.encode_text(t).cpu().numpy()
    assert np.isfinite(got).all() and row_rel_l2(got, want).max() < 1.5e-3
    for i in range(3):                                              # each alone gives the same bits as in the ragged batch
        assert np.array_equal(m.encode_text(t[i:i + 1]).cpu().numpy()[0], got[i])


@pytest.mark.parametrize("fname,model", [("vitl_gelu", "ViT-L-14"), ("vitl_quickgelu", "ViT-L-14-quickgelu")])
def test_encode_text_vitl_golden(torch_mod, golden_dir, fname, model):
    z = np.load(os.path.join(golden_dir, fname + ".npz"))
    m = _model(model, 1)
    out = m.encode_text(z["tokens"]).cpu().numpy()
    r = row_rel_l2(out, z["out"])
    print(f"{model}: rel-L2 global {rel_l2(out, z['out']):.3e} row max {r.max():.3e}")
    assert rel_l2(out, z["out"]) < TOL_GLOBAL
    assert r.max() < TOL_ROW


def test_bf16_mode_runs_and_is_coarser(torch_mod, golden_dir):
    z = np.load(os.path.join(golden_dir, "vitl_gelu.npz"))
    m = _model("ViT-L-14", 1, dtype="bf16")
    out = m.encode_text(z["tokens"]).cpu().numpy()
    assert rel_l2(out, z["out"]) < 1.2e-2   # bf16 operands: ~7.5e-3 (precision table)


def test_padding_and_chunking_are_exact(torch_mod):
    """Rows are independent: results must be bit-identical whatever the batch composition / chunk size, and
    tokens after EOT must not matter (causal mask)."""
    from leaf_amd.model import LeafCLIPText, get_config
    toks = O.synthetic_tokens(37, seed=5)
    m = _model("tiny-test", 11)
    a = m.encode_text(toks).cpu().numpy()
    m2 = LeafCLIPText(get_config("tiny-test"), chunk=8).copy_from(m)
    b = m2.encode_text(toks).cpu().numpy()
    assert np.array_equal(a, b)
    c = m.encode_text(toks[5:9]).cpu().numpy()
    assert np.array_equal(a[5:9], c)
    junk = toks.copy()
    eot = toks.argmax(-1)
    for i in range(len(toks)):
        junk[i, eot[i] + 1:] = np.arange(1, 77 - eot[i])   # < EOT id so argmax is unchanged
    d = m.encode_text(junk).cpu().numpy()
    assert np.array_equal(a, d)


@pytest.mark.parametrize("objective", ["l2", "negl2", "sim", "dissim"])
def test_score_candidates_vs_oracle(torch_mod, objective):
    cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=12)
    m = _model("tiny-test-quickgelu", 12)
    B, rho = 6, 50
    base = O.synthetic_tokens(B, seed=2)
    cand = O.synthetic_candidates(base, rho, seed=3)
    cand[:, 7] = cand[:, 3]          # duplicates: the first index must win ties
    anchor = O.encode_text(w, cfg, base)
    if objective in ("sim", "dissim"):
        anchor = anchor / np.linalg.norm(anchor, axis=-1, keepdims=True)
    idx_o, best_o, loss_o = O.score_candidates(w, cfg, cand, anchor, objective)
    idx, feat, loss = m.score_candidates(cand.reshape(-1, 77), torch_mod.from_numpy(anchor).cuda(), rho, objective,
                                         want_loss=True)
    idx, feat, loss = idx.cpu().numpy(), feat.cpu().numpy(), loss.cpu().numpy()
    assert np.allclose(loss, loss_o, rtol=5e-3, atol=5e-3 * np.abs(loss_o).max())
    # margin-aware selection parity: must agree whenever the oracle's top-2 gap exceeds the
    # measured loss error; on disagreement the chosen candidate must be within that error of the optimum
    err = np.abs(loss - loss_o).max(-1)
    srt = np.sort(loss_o, -1)
    gap = srt[:, -1] - srt[:, -2]
    for b in range(B):
        if gap[b] > 4 * err[b]:
            assert idx[b] == idx_o[b] or loss_o[b, idx[b]] == loss_o[b, idx_o[b]]
        assert loss_o[b, idx[b]] >= loss_o[b, idx_o[b]] - 4 * err[b] - 1e-6
        assert idx[b] == int(np.argmax(loss[b]))                 # first maximum of the engine's own loss
        # tiny model (d=128): fewer terms per dot product, so per-row noise is larger than on ViT-L; 2e-3 here
        assert rel_l2(feat[b], O.encode_text(w, cfg, cand[b, idx[b]][None], normalize=objective in ("sim", "dissim"))[0]) < 2e-3


@pytest.mark.parametrize("native", [False, True])
def test_attack_text_replays_reference_trace(torch_mod, golden_dir, native):
    """Same numpy seed -> same candidate strings per stage and same adversarial sentences as the reference run
    (utils_attacks.py:297-393), on the tiny model; also with --constrain on the stub dictionary."""
    from leaf_amd import attacks
    from leaf_amd.tokenizer import SimpleTokenizer
    with open(os.path.join(golden_dir, "attack_trace.json")) as f:
        trace = json.load(f)
    with open(os.path.join(golden_dir, "mutation_kat.json")) as f:
        stub = json.load(f)["stub_words"]
    if native:
        from leaf_amd.native_text import NativeTokenizer
        tok = NativeTokenizer(n_threads=4)
    else:
        tok = SimpleTokenizer()
    m = _model("tiny-test-quickgelu", 12)
    attacks.set_dictionary(attacks.Dictionary(stub))
    for key, t in trace.items():
        z = np.load(os.path.join(golden_dir, f"attack_{key}.npz"))
        anchor = torch_mod.from_numpy(z["anchor"]).cuda()
        got_trace, picks = [], []
        np.random.seed(t["seed"])
        feats, adv = attacks.attack_text_leaf(m, tok, list(t["sentences"]), anchor, objective="l2", n=t["rho"],
                                              k=t["k"], V=attacks.DEFAULT_V, constrain=t["constrain"],
                                              return_trace=got_trace, return_picks=picks)
        assert got_trace[0] == t["stage_candidates"][0], "stage-1 candidates differ: RNG / mutation drift"
        # later stages depend on fp-level arg-max decisions; with identical decisions they are identical
        if adv == t["adv"]:
            assert got_trace == t["stage_candidates"]
            assert rel_l2(feats.cpu().numpy(), z["feats"]) < TOL_GLOBAL
        else:
            # a near-tie flipped somewhere.  The margin rule, for any k: up to the first stage whose candidates differ from the
            # reference's, both runs scored the SAME candidates, so the flip is the pick of the stage before it -- and there the
            # engine's choice must be as good as the best candidate under the fp32 oracle, within fp16 noise, for every caption
            # (after that stage the two runs edit different sentences and are no longer comparable)
            cfg = O.TextCfg(128, 2, 2, 64, quick_gelu=True)
            w = O.init_weights(cfg, seed=12)
            ref = t["stage_candidates"]
            assert len(got_trace) == len(ref) == len(picks) == 2 * t["k"]
            s_div = next((i for i in range(len(ref)) if got_trace[i] != ref[i]), len(ref))
            assert s_div >= 1, "stage-1 candidates are drawn before any decision"
            B, n = len(t["sentences"]), t["rho"]
            cands = got_trace[s_div - 1]
            f = O.encode_text(w, cfg, tok.encode_batch(cands)).reshape(B, n, -1)
            loss_o = ((f - z["anchor"][:, None, :]) ** 2).sum(-1)
            pick = loss_o[np.arange(B), picks[s_div - 1]]
            assert np.all(pick >= loss_o.max(-1) * (1 - 5e-3)), (key, s_div, pick, loss_o.max(-1))
    attacks.set_dictionary(None)


@pytest.mark.parametrize("native", [False, True])
def test_token_identical_candidates_are_computed_once(torch_mod, native):
    """SURVEY 8f-2 (third item): the tokenizer lower-cases and collapses whitespace (src/open_clip/tokenizer.py:83-85,139), so
    'a' / 'A' at the same slot, a space beside a space and with-replacement draws give identical id rows.  With dedupe they are
    computed once and the first-index tie-break of torch.argmax (utils_attacks.py:348,386) is preserved: same picks at every
    stage, same adversarial sentences, bit-identical features, fewer rows."""
    from leaf_amd import attacks
    from leaf_amd.tokenizer import SimpleTokenizer
    if native:
        from leaf_amd.native_text import NativeTokenizer
        tok = NativeTokenizer(n_threads=4)
    else:
        tok = SimpleTokenizer()
    m = _model("tiny-test-quickgelu", 12)
    sents = ["ab", "a  photo of a cat", "two dogs", "x"]          # 'ab': 5 slots < rho -> stage 1 draws with replacement
    anchor = m.encode_text(tok.encode_batch(sents)) + 0.05
    rho, k = 60, 2                                                  # 60 of the 96 stage-2 characters: many 'q' / 'Q' pairs
    out = {}
    for dd in (True, False):
        np.random.seed(7)
        picks, trace = [], []
        rows0 = m.rows_scored
        feats, adv = attacks.attack_text_leaf(m, tok, list(sents), anchor.clone(), objective="l2", n=rho, k=k, V=attacks.DEFAULT_V,
                                              return_picks=picks, return_trace=trace, dedupe=dd)
        out[dd] = (adv, [p.tolist() for p in picks], feats.cpu().numpy(), m.rows_scored - rows0, trace)
    assert out[True][0] == out[False][0] and out[True][1] == out[False][1]
    assert np.array_equal(out[True][2], out[False][2]), "winner features must be the same bits"
    assert out[True][3] < 0.9 * out[False][3], (out[True][3], out[False][3])
    # the planted duplicates really occurred, and a winner is never a later copy of an earlier candidate
    ids = tok.encode_batch(out[True][4][1])                         # stage-2 candidates of the first edit
    dup = attacks.duplicate_map(np.asarray(ids), len(sents), rho)
    assert (dup != np.arange(rho)[None, :]).sum() >= 8
    assert all(dup[b, out[True][1][1][b]] == out[True][1][1][b] for b in range(len(sents)))
    # the grouped pipeline (host preparation of one group of captions beside the GPU scoring of another): captions are
    # independent and rows have the same bits whichever launch computes them -> nothing may depend on the grouping
    for pipe in (1, 2, 3):
        np.random.seed(7)
        picks = []
        feats, adv = attacks.attack_text_leaf(m, tok, list(sents), anchor.clone(), objective="l2", n=rho, k=k, V=attacks.DEFAULT_V,
                                              return_picks=picks, pipeline=pipe)
        assert adv == out[True][0] and [p.tolist() for p in picks] == out[True][1], pipe
        assert np.array_equal(feats.cpu().numpy(), out[True][2]), pipe


def test_packed_rows_are_bit_exact(torch_mod, monkeypatch):
    """EOT trimming (compute only rows <= EOT) is exact work skipping: outputs, search decisions and losses are
    bit-identical to the dense 77-row computation, for host tokens (lengths inferred) and device tokens + lens."""
    from leaf_amd.model import LeafCLIPText, get_config
    toks = O.synthetic_tokens(45, seed=6, min_len=1, max_len=74)
    toks[0, :] = 0; toks[0, 0] = 49406; toks[0, 1] = 49407            # shortest possible caption (len 2)
    toks[1] = np.arange(1, 78); toks[1, 0] = 49406; toks[1, 76] = 49407   # longest (len 77)
    m = _model("tiny-test-quickgelu", 12)
    assert m.trim_rows
    packed = m.encode_text(toks).cpu().numpy()
    m.trim_rows = False
    dense = m.encode_text(toks).cpu().numpy()
    m.trim_rows = True
    assert np.array_equal(packed, dense)
    dev = torch_mod.from_numpy(toks.astype(np.int32)).cuda()
    lens = toks.argmax(-1) + 1
    assert np.array_equal(m.encode_text(dev, seq_lens=lens).cpu().numpy(), dense)
    small = LeafCLIPText(get_config("tiny-test-quickgelu"), chunk=4).copy_from(m)   # chunks split by row budget
    assert np.array_equal(small.encode_text(toks).cpu().numpy(), dense)
    # search stage
    B, rho = 5, 12
    base = O.synthetic_tokens(B, seed=2)
    cand = O.synthetic_candidates(base, rho, seed=3).reshape(-1, 77)
    anchor = m.encode_text(base)
    i1, f1, l1 = m.score_candidates(cand, anchor, rho, "l2", want_loss=True)
    m.trim_rows = False
    i2, f2, l2 = m.score_candidates(cand, anchor, rho, "l2", want_loss=True)
    assert torch_mod.equal(i1, i2) and torch_mod.equal(f1, f2) and torch_mod.equal(l1, l2)


@pytest.mark.parametrize("model", ["ViT-H-14", "ViT-bigG-14"])
def test_encode_text_large_towers_vs_oracle(torch_mod, model):
    """BASELINE.json configs[3]/[4] shapes (d=1024/24 layers, d=1280/32 layers, erf-GELU): a few captions against the
    fp32 oracle, same 1e-3 rel-L2 gate."""
    cfg = O.CONFIGS[model]
    w = O.init_weights(cfg, seed=2)
    m = _model(model, 2)
    toks = O.synthetic_tokens(6, seed=11, min_len=5, max_len=60)
    want = O.encode_text(w, cfg, toks)
    got = m.encode_text(toks).cpu().numpy()
    r = row_rel_l2(got, want)
    print(f"{model}: rel-L2 global {rel_l2(got, want):.3e} row max {r.max():.3e}")
    assert rel_l2(got, want) < TOL_GLOBAL and r.max() < TOL_ROW


def test_prefix_reuse_is_bit_exact(torch_mod):
    """Prefix reuse (K/V of the positions before the first changed token come from the clean caption's cache) returns
    bit-identical losses, winners and features, also for candidates equal to the caption and edits at position 1."""
    m = _model("tiny-test-quickgelu", 12)
    B, rho = 7, 20
    base = O.synthetic_tokens(B, seed=31, min_len=2, max_len=70)
    cand = O.synthetic_candidates(base, rho, seed=32)
    cand[:, 0] = base                      # no-op candidate: nothing changes -> only the EOT row is recomputed
    cand[:, 1] = base
    cand[:, 1, 1] = 7                      # first real token edited -> whole suffix recomputed
    flat = cand.reshape(-1, 77)
    neq = flat.reshape(B, rho, 77) != base[:, None, :]
    pl = neq.argmax(-1)
    pl[~neq.any(-1)] = 77
    anchor = m.encode_text(base) + 0.3
    lens = np.repeat(base.argmax(-1) + 1, rho)
    i0, f0, l0 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens)
    kv, feats = m.encode_text_kv(base, want_features=True)
    assert torch_mod.equal(feats, m.encode_text(base))
    i1, f1, l1 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl.reshape(-1), kv=kv)
    assert torch_mod.equal(l0, l1) and torch_mod.equal(i0, i1) and torch_mod.equal(f0, f1)


def test_last_layer_trim_is_bit_exact(torch_mod):
    """Computing the last block's attention output / out-projection / MLP only for the pooled row changes nothing."""
    for model, seed in (("tiny-test-quickgelu", 12), ("ViT-L-14", 1)):
        m = _model(model, seed)
        toks = O.synthetic_tokens(23, seed=41, min_len=1, max_len=74)
        on = m.encode_text(toks).cpu().numpy()
        onn = m.encode_text(toks, normalize=True).cpu().numpy()
        m.set_option("last_layer_trim", 0)
        off = m.encode_text(toks).cpu().numpy()
        offn = m.encode_text(toks, normalize=True).cpu().numpy()
        assert np.array_equal(on, off) and np.array_equal(onn, offn)
        m.trim_rows = False                      # dense rows + trimming
        m.set_option("last_layer_trim", 1)
        assert np.array_equal(m.encode_text(toks).cpu().numpy(), off)


def test_two_stream_pipeline_is_bit_exact(torch_mod):
    """The optional two-stream chunk pipeline (set_option('streams', 2)) splits a pass into two chunks that run on the
    caller's stream and the handle's side stream: same rows, same kernels, identical results."""
    m = _model("ViT-L-14", 1)
    B, rho = 12, 50
    base = O.synthetic_tokens(B, seed=51, min_len=20, max_len=60)
    cand = O.synthetic_candidates(base, rho, seed=52).reshape(-1, 77)
    lens = np.repeat(base.argmax(-1) + 1, rho)
    assert lens.sum() >= 16384                    # enough rows for the split to engage
    anchor = m.encode_text(base) + 0.1
    i1, f1, l1 = m.score_candidates(cand, anchor, rho, "l2", want_loss=True, seq_lens=lens)
    e1 = m.encode_text(cand[:700], seq_lens=lens[:700])
    m.set_option("streams", 2)
    i2, f2, l2 = m.score_candidates(cand, anchor, rho, "l2", want_loss=True, seq_lens=lens)
    e2 = m.encode_text(cand[:700], seq_lens=lens[:700])
    m.set_option("streams", 1)
    assert torch_mod.equal(i1, i2) and torch_mod.equal(f1, f2) and torch_mod.equal(l1, l2) and torch_mod.equal(e1, e2)


def test_prefix_reuse_is_bit_exact_across_kernels(torch_mod):
    """ViT-L shapes: the clean captions' K/V cache comes from a 12-sequence launch (128^2 GEMM kernel), the candidate
    rows from a 600-sequence launch (256^2 LDS-DMA kernel); prefix reuse must still equal the full recomputation bit
    for bit (tests/test_gpu_kernels.py::test_gemm_rows_do_not_depend_on_the_kernel is the per-GEMM statement)."""
    m = _model("ViT-L-14", 1)
    B, rho = 12, 50
    base = O.synthetic_tokens(B, seed=61, min_len=25, max_len=60)
    cand = O.synthetic_candidates(base, rho, seed=62)
    flat = cand.reshape(-1, 77)
    neq = cand != base[:, None, :]
    pl = neq.argmax(-1)
    pl[~neq.any(-1)] = 77
    anchor = m.encode_text(base) + 0.2
    lens = np.repeat(base.argmax(-1) + 1, rho)
    i0, f0, l0 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens)
    kv = m.encode_text_kv(base)
    i1, f1, l1 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl.reshape(-1), kv=kv)
    assert torch_mod.equal(l0, l1) and torch_mod.equal(i0, i1) and torch_mod.equal(f0, f1)
    # the cache pass can also return the captions' features; its final projection must not touch the cache
    kv2, feats = m.encode_text_kv(base, want_features=True)
    assert torch_mod.equal(feats, m.encode_text(base))
    i2, f2, l2 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl.reshape(-1), kv=kv2)
    assert torch_mod.equal(l0, l2) and torch_mod.equal(i0, i2) and torch_mod.equal(f0, f2)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_shapes_dense_packed_prefix_agree(torch_mod, seed):
    """Randomised edge cases on the tiny config: captions of every length up to the full 77 tokens (EOT in the last
    position), one-token captions, odd B and rho, edits at the first and last token: the dense, the EOT-trimmed and the
    prefix-reuse computation must give bit-identical losses, winners and features."""
    rng = np.random.default_rng(100 + seed)
    m = _model("tiny-test-quickgelu", 12)
    B, rho = int(rng.integers(3, 8)), int(rng.integers(5, 12))
    base = np.concatenate([O.synthetic_tokens(B - 2, seed=200 + seed, min_len=0, max_len=75),
                           O.synthetic_tokens(1, seed=300 + seed, min_len=75, max_len=75),      # EOT at position 76
                           O.synthetic_tokens(1, seed=400 + seed, min_len=0, max_len=0)], 0)    # SOT, EOT only
    lens = base.argmax(-1) + 1
    cand = np.repeat(base[:, None, :], rho, axis=1)
    for b in range(B):
        for r in range(rho):
            if lens[b] > 2 and r % 4 != 3:                       # every 4th candidate stays a no-op
                p = 1 if r == 0 else (lens[b] - 2 if r == 1 else int(rng.integers(1, lens[b] - 1)))
                cand[b, r, p] = int(rng.integers(1, 49405))
    flat = cand.reshape(-1, 77)
    neq = cand != base[:, None, :]
    pl = neq.argmax(-1)
    pl[~neq.any(-1)] = 77
    anchor = m.encode_text(base) + 0.3
    cl = np.repeat(lens, rho)
    m.trim_rows = False
    i0, f0, l0 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True)
    m.trim_rows = True
    i1, f1, l1 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=cl)
    kv = m.encode_text_kv(base)
    i2, f2, l2 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=cl, prefix_lens=pl.reshape(-1), kv=kv)
    assert torch_mod.equal(l0, l1) and torch_mod.equal(i0, i1) and torch_mod.equal(f0, f1)
    assert torch_mod.equal(l0, l2) and torch_mod.equal(i0, i2) and torch_mod.equal(f0, f2)


def test_split_blocks_precision_escape_hatch(torch_mod, golden_dir):
    """``set_split_blocks(n)`` -- hi + lo 16-bit splits of both operands in the four GEMMs of the first n
    transformer blocks of the forward-only passes (three MFMA passes over a 3x longer K through the unchanged kernels; q|k|v,
    attention and hidden rows stay 16-bit).  On the reference-generated ViT-L fixture (tests/golden/vitl_quickgelu.npz) the worst
    row falls with every block (oracle emulation: 9.6e-4 -> 8.3e-4 -> 7.1e-4 for n = 0, 1, 2) and is <= 8e-4 at n = 2; n = 0 gives
    the shipped bits back; EOT trimming, prefix reuse and the fused caption pass stay bit-exact with the option on."""
    from leaf_amd.model import create_model
    z = np.load(os.path.join(golden_dir, "vitl_quickgelu.npz"))
    m = create_model("ViT-L-14-quickgelu", seed=1)
    m.set_precision("fast")                    # no split GEMM at all (round 6: the DEFAULT is the priced policy-1 spend, tested below)
    toks = z["tokens"]
    base_out = m.encode_text(toks).cpu().numpy()
    rows = {0: row_rel_l2(base_out, z["out"])}
    for n in (1, 2):
        m.set_split_blocks(n)                  # all four GEMMs of the first n blocks
        rows[n] = row_rel_l2(m.encode_text(toks).cpu().numpy(), z["out"])
    print("[split blocks] worst / median row rel-L2:", {n: (f"{r.max():.3e}", f"{np.median(r):.3e}") for n, r in rows.items()})
    assert rows[0].max() < TOL_ROW
    assert rows[1].max() < rows[0].max() - 5e-5 and rows[2].max() < rows[1].max() - 5e-5
    assert rows[2].max() <= 8.0e-4 and np.median(rows[2]) < 7.0e-4
    # exactness properties with the option on (ViT-L, 2 split blocks): dense == trimmed == prefix reuse; fused caption pass
    B, rho = 6, 50
    base = O.synthetic_tokens(B, seed=61, min_len=20, max_len=50)
    cand = O.synthetic_candidates(base, rho, seed=62)
    flat = cand.reshape(-1, 77)
    neq = cand != base[:, None, :]
    pl = neq.argmax(-1)
    pl[~neq.any(-1)] = 77
    anchor = m.encode_text(base) + 0.2
    lens = np.repeat(base.argmax(-1) + 1, rho)
    i0, f0, l0 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens)
    kv = m.encode_text_kv(base)
    i1, f1, l1 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl.reshape(-1), kv=kv)
    assert torch_mod.equal(l0, l1) and torch_mod.equal(i0, i1) and torch_mod.equal(f0, f1)
    fused = m.score_candidates_fused(torch_mod.from_numpy(base.astype(np.int32)).cuda(), (base.argmax(-1) + 1).astype(np.int32), flat, anchor,
                                     rho, lens, pl.reshape(-1), want_features=True, want_loss=True)
    assert fused is not None and torch_mod.equal(fused[0], i0) and torch_mod.equal(fused[1], f0) and torch_mod.equal(fused[3], l0)
    m.trim_rows = False
    i2, f2, l2 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True)
    m.trim_rows = True
    assert torch_mod.equal(l0, l2) and torch_mod.equal(i0, i2) and torch_mod.equal(f0, f2)
    # a weight change reaches the split copies through pack(); option off = the shipped bits
    m.params["transformer.resblocks.0.mlp.c_fc.weight"].mul_(1.01)
    m.pack()
    changed = m.encode_text(toks).cpu().numpy()
    m.set_split_blocks(0)
    plain = create_model("ViT-L-14-quickgelu", seed=1)
    plain.set_precision("fast")
    plain.params["transformer.resblocks.0.mlp.c_fc.weight"].mul_(1.01)
    ref2 = plain.encode_text(toks).cpu().numpy()
    assert rel_l2(changed, ref2) < 2e-3 and not np.array_equal(changed, ref2)
    assert np.array_equal(m.encode_text(toks).cpu().numpy(), ref2)
    # tiny config (2 layers: one split block, K = 3 x 128 through the small-launch kernels)
    t = _model("tiny-test-quickgelu", 12)
    t.set_precision("fast")
    zt = np.load(os.path.join(golden_dir, "tiny_quickgelu.npz"))
    e0 = rel_l2(t.encode_text(zt["tokens"]).cpu().numpy(), zt["out"])
    t.set_split_blocks(1)
    e1 = rel_l2(t.encode_text(zt["tokens"]).cpu().numpy(), zt["out"])
    assert e1 < e0 < TOL_GLOBAL
    with pytest.raises(ValueError):
        t.set_split_blocks(2)


def test_default_arithmetic_is_the_priced_split_policy_and_keeps_every_exactness_property(torch_mod, golden_dir):
    """Round 6: the DEFAULT arithmetic of the forward-only passes = 'rowsafe' (leaf_amd.model.PRECISION_MODES):
    QKV of block 0 on hi + lo splits of both operands (through the fused QKV + attention launch, K = 3d) + the out-projection weights
    of blocks 0-3 split, on the 16 + 8-bit residual stream.  Against the reference-generated ViT-L fixtures the worst row sits well
    inside 1e-3 (the large-sample gate is tests/test_gpu_fullshape.py::test_row_error_census_gate), 'fast' gives the round-5 bits
    back, a frozen copy inherits the arithmetic, a K/V cache is tied to the arithmetic it was built under, and dense == EOT-trimmed
    == prefix reuse == fused caption pass bit for bit in BOTH residual formats."""
    from leaf_amd.model import LeafCLIPText, create_model, get_config
    z = np.load(os.path.join(golden_dir, "vitl_quickgelu.npz"))
    m = create_model("ViT-L-14-quickgelu", seed=1)
    from leaf_amd.model import PRECISION_MODES
    assert m.split_masks == PRECISION_MODES["rowsafe"] and m.precision_name() == "rowsafe" and m.arithmetic_tag() == (1, PRECISION_MODES["rowsafe"])
    toks = z["tokens"]
    r_def = row_rel_l2(m.encode_text(toks).cpu().numpy(), z["out"])
    fast = create_model("ViT-L-14-quickgelu", seed=1).set_precision("fast")
    r_fast = row_rel_l2(fast.encode_text(toks).cpu().numpy(), z["out"])
    print(f"[default arithmetic] rows max / median: rowsafe {r_def.max():.3e} / {np.median(r_def):.3e}   fast {r_fast.max():.3e} / {np.median(r_fast):.3e}")
    assert r_def.max() < 8.8e-4 and np.median(r_def) < 0.88 * np.median(r_fast)
    frozen = LeafCLIPText(get_config("ViT-L-14-quickgelu")).copy_from(fast)
    assert frozen.arithmetic_tag() == fast.arithmetic_tag() == (1, ())
    assert np.array_equal(frozen.encode_text(toks).cpu().numpy(), fast.encode_text(toks).cpu().numpy())
    B, rho = 6, 50
    base = O.synthetic_tokens(B, seed=61, min_len=20, max_len=50)
    cand = O.synthetic_candidates(base, rho, seed=62)
    flat = cand.reshape(-1, 77)
    neq = cand != base[:, None, :]
    pl = neq.argmax(-1)
    pl[~neq.any(-1)] = 77
    lens = np.repeat(base.argmax(-1) + 1, rho)
    for compact in (1, 0):
        m.set_option("compact_resid", compact)
        anchor = m.encode_text(base) + 0.2
        i0, f0, l0 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens)
        kv = m.encode_text_kv(base)
        i1, f1, l1 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl.reshape(-1), kv=kv)
        assert torch_mod.equal(l0, l1) and torch_mod.equal(i0, i1) and torch_mod.equal(f0, f1), compact
        fused = m.score_candidates_fused(torch_mod.from_numpy(base.astype(np.int32)).cuda(), (base.argmax(-1) + 1).astype(np.int32), flat, anchor,
                                         rho, lens, pl.reshape(-1), want_features=True, want_loss=True)
        assert fused is not None and torch_mod.equal(fused[0], i0) and torch_mod.equal(fused[1], f0) and torch_mod.equal(fused[3], l0), compact
        m.trim_rows = False
        i2, f2, l2 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True)
        m.trim_rows = True
        assert torch_mod.equal(l0, l2) and torch_mod.equal(i0, i2) and torch_mod.equal(f0, f2), compact
        # a candidate row alone (small-launch kernels) = the same bits as inside the big pass
        one = m.encode_text(flat[7:8]).cpu().numpy()[0]
        allf = m.encode_text(flat).cpu().numpy()
        assert np.array_equal(one, allf[7])
    m.set_option("compact_resid", 1)
    kv = m.encode_text_kv(base)
    m.set_precision("fast")
    with pytest.raises(ValueError, match="K/V cache built under arithmetic"):
        m.score_candidates(flat, m.encode_text(base), rho, "l2", seq_lens=lens, prefix_lens=pl.reshape(-1), kv=kv)


def test_compact_residual_stream_gate_and_exactness(torch_mod, golden_dir):
    """Round 5: the residual stream of the LN-folded forward-only passes in 16 + 8 bits (option 'compact_resid', default on: the
    16-bit copy the next GEMM reads anyway + a block-scaled e4m3 remainder byte, instead of an fp32 row beside that copy).  On the reference-generated
    ViT-L fixture both forms pass the per-row gate at the same error level (the remainder keeps the stream to 2^-16 per store against
    the 2^-11 of every GEMM operand), and the exactness properties of DESIGN section 5 hold in BOTH
    forms: dense == EOT-trimmed == prefix reuse == fused caption pass, bit for bit."""
    from leaf_amd.model import create_model
    z = np.load(os.path.join(golden_dir, "vitl_quickgelu.npz"))
    m = create_model("ViT-L-14-quickgelu", seed=1)
    toks = z["tokens"]
    out = {}
    for on in (1, 0):
        m.set_option("compact_resid", on)
        out[on] = m.encode_text(toks).cpu().numpy()
    r1, r0 = row_rel_l2(out[1], z["out"]), row_rel_l2(out[0], z["out"])
    between = row_rel_l2(out[1], out[0])
    print(f"[compact residual] worst / median row rel-L2 vs the reference: 16+8 {r1.max():.3e} / {np.median(r1):.3e}, fp32 {r0.max():.3e} / "
          f"{np.median(r0):.3e}; between the two {between.max():.3e}")
    assert r1.max() < TOL_ROW and r0.max() < TOL_ROW
    # the remainder's own error is ~2e-5 of a feature row (oracle emulation: fp32 GEMMs on the 16 + 8-bit stream); what the two forms
    # differ by is the 16-bit operand roundings falling differently once the stream differs in its 16th bit, i.e. two draws of
    # the same 9e-4 noise -- so the check is on the error LEVEL against the reference, not on the distance between the two
    assert abs(np.median(r1) - np.median(r0)) < 4e-5 and abs(r1.max() - r0.max()) < 6e-5
    assert between.max() < 1.4 * max(r1.max(), r0.max()) and not np.array_equal(out[0], out[1])
    B, rho = 6, 50
    base = O.synthetic_tokens(B, seed=71, min_len=20, max_len=50)
    cand = O.synthetic_candidates(base, rho, seed=72)
    flat = cand.reshape(-1, 77)
    neq = cand != base[:, None, :]
    pl = neq.argmax(-1)
    pl[~neq.any(-1)] = 77
    lens = np.repeat(base.argmax(-1) + 1, rho)
    for on in (1, 0):
        m.set_option("compact_resid", on)
        anchor = m.encode_text(base) + 0.2
        i0, f0, l0 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens)
        kv = m.encode_text_kv(base)
        i1, f1, l1 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens, prefix_lens=pl.reshape(-1), kv=kv)
        assert torch_mod.equal(l0, l1) and torch_mod.equal(i0, i1) and torch_mod.equal(f0, f1), on
        fused = m.score_candidates_fused(torch_mod.from_numpy(base.astype(np.int32)).cuda(), (base.argmax(-1) + 1).astype(np.int32), flat,
                                         anchor, rho, lens, pl.reshape(-1), want_features=True, want_loss=True)
        assert fused is not None and torch_mod.equal(fused[0], i0) and torch_mod.equal(fused[1], f0) and torch_mod.equal(fused[3], l0), on
        m.trim_rows = False
        i2, f2, l2 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True)
        m.trim_rows = True
        assert torch_mod.equal(l0, l2) and torch_mod.equal(i0, i2) and torch_mod.equal(f0, f2), on
        # last-block trimming on / off: the pooled rows come out of the same stream
        m.set_option("last_layer_trim", 0)
        i3, f3, l3 = m.score_candidates(flat, anchor, rho, "l2", want_loss=True, seq_lens=lens)
        m.set_option("last_layer_trim", 1)
        assert torch_mod.equal(l0, l3) and torch_mod.equal(f0, f3), on
