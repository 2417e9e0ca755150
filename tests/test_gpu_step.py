"""The synthetic-token step used by bench.py (leaf_amd/step.py) on the tiny model: data dependencies and shapes."""
import numpy as np
import pytest

from oracle import text_oracle as O

pytestmark = pytest.mark.gpu


def test_synthetic_step_runs_and_selects_argmax():
    import torch
    from leaf_amd.model import LeafCLIPText, create_model, get_config
    from leaf_amd.step import StepConfig, SyntheticCandidates, search_synthetic, train_step_tokens
    m = create_model("tiny-test-quickgelu", seed=12, trainable=True)
    frozen = LeafCLIPText(get_config("tiny-test-quickgelu")).copy_from(m)
    base = torch.from_numpy(O.synthetic_tokens(8, seed=4).astype(np.int32)).cuda()
    sc = StepConfig(rho=10, k_adv=2, lr=1e-4)
    anchor = frozen.encode_text(base)
    adv = search_synthetic(m, anchor, base, sc, seed=0)
    assert adv.shape == base.shape and adv.dtype == torch.int32
    diff = (adv != base).sum(-1).cpu().numpy()
    assert diff.max() <= sc.k_adv and (adv.argmax(-1) == base.argmax(-1)).all()   # <= k edits, EOT untouched
    # candidates differ from the current row in exactly <= 1 position, inside the caption
    gen = SyntheticCandidates(base, None, 10, m.cfg.vocab_size, 0)
    cand, pos = gen.stage1(base)
    d = (cand != base[:, None, :])
    assert d.sum(-1).max() <= 1
    eot = base.argmax(-1).cpu().numpy()
    assert (pos >= 1).all() and (pos < eot[:, None]).all()
    # prefix reuse and plain EOT trimming pick the same adversarial ids (bit-identical scoring)
    lens = eot + 1
    a1 = search_synthetic(m, anchor, base, sc, seed=3, base_lens=lens, prefix_reuse=True)
    a2 = search_synthetic(m, anchor, base, sc, seed=3, base_lens=lens, prefix_reuse=False)
    assert torch.equal(a1, a2)
    l0 = float(train_step_tokens(m, frozen, base, sc, seed=1))
    l1 = float(train_step_tokens(m, frozen, base, sc, seed=1))
    assert np.isfinite([l0, l1]).all()


def test_overlapped_gradient_reduction_on_rccl_single_rank(tmp_path):
    """The bucketed, event-driven gradient reduction (GradReducer: leaf_textfare_backward_events + RCCL all-reduces on a side
    stream) rehearsed on the one GPU of this box: under torch.distributed.run with one rank and LEAF_BENCH_FORCE_DIST=1 the
    step must give the same loss trajectory as the flat all-reduce and as no process group at all (a one-rank sum is the
    identity), and bench.py must report the rank count it saw.  The N > 1 arithmetic is covered on CPU by tests/test_dp_gloo.py."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--model", "tiny-test-quickgelu", "--batch", "16", "--rho", "8", "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
            "--no-dense-leg", "--gpus", "1"]
    base_env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LEAF_BENCH_FORCE_DIST", "LEAF_DP_OVERLAP")}

    def run(extra_env, launcher):
        env = dict(base_env, **extra_env)
        cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", "29611"] if launcher else [sys.executable]) + [os.path.join(root, "bench.py")] + args
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1, p.stdout[-2000:]
        return json.loads(line[0])
    plain = run({}, False)
    overlap = run({"LEAF_BENCH_FORCE_DIST": "1"}, True)
    flat = run({"LEAF_BENCH_FORCE_DIST": "1", "LEAF_DP_OVERLAP": "0"}, True)
    assert plain["n_ranks_seen"] == 1 and overlap["n_ranks_seen"] == 1
    assert plain["loss"] == overlap["loss"] == flat["loss"], (plain["loss"], overlap["loss"], flat["loss"])


@pytest.mark.parametrize("name,B,rho", [("tiny-test-quickgelu", 8, 10), ("ViT-L-14-quickgelu", 6, 7)])
def test_fused_caption_pass_is_bit_identical(name, B, rho):
    """The clean captions' K/V pass fused into the first stage's scoring pass (leaf_score_candidates_prefix_fused) against the
    two separate calls: same arg-max, same winning features, the same bytes in the K/V cache (GEMM rows do not depend on the
    launch they are computed in), and the later stage scores identically from either cache."""
    import torch
    from leaf_amd.model import create_model
    from leaf_amd.step import SyntheticCandidates
    m = create_model(name, seed=5)
    base = torch.from_numpy(O.synthetic_tokens(B, seed=9).astype(np.int32)).cuda()
    lens = (base.argmax(-1) + 1).cpu().numpy().astype(np.int32)
    anchor = m.encode_text(base, seq_lens=lens) + 0.05
    gen = SyntheticCandidates(base, lens, rho, m.cfg.vocab_size, 3)
    cand, pos = gen.stage1(base)
    cand_lens = np.repeat(lens, rho)
    kv_a = m.encode_text_kv(base, seq_lens=lens)
    need = kv_a["base_rows"] * 3 * m.cfg.width * 2 * m.cfg.layers
    kv_bytes_a = kv_a["kv"][:need].clone()
    i_a, f_a = m.score_candidates(cand.view(B * rho, -1), anchor, rho, "l2", seq_lens=cand_lens, prefix_lens=pos.reshape(-1), kv=kv_a)
    i_a, f_a = i_a.clone(), f_a.clone()
    kv_a["kv"].zero_()
    fused = m.score_candidates_fused(base, lens, cand.view(B * rho, -1), anchor, rho, cand_lens, pos.reshape(-1), want_features=True)
    assert fused is not None
    i_b, f_b, kv_b = fused
    assert torch.equal(i_a, i_b) and torch.equal(f_a, f_b)
    # ... the K and V thirds of every cached row ([layers][rows][q | k | v]; the fused pass does not write the captions' q rows into the
    # cache: no consumer reads them)
    rows_, dd = kv_a["base_rows"], m.cfg.width
    kvv = lambda t: t.view(torch.int16).view(m.cfg.layers, rows_, 3 * dd)[:, :, dd:]
    assert torch.equal(kvv(kv_bytes_a), kvv(kv_b["kv"][:need]))
    assert kv_b["base_rows"] == kv_a["base_rows"] and torch.equal(kv_a["base_cu"].cpu(), kv_b["base_cu"].cpu())
    # a later stage from the fused pass's cache
    cand2 = gen.stage2_device(base, i_b)
    pos2 = gen.stage2_positions(pos, i_b.cpu().numpy())
    i2_b, f2_b = m.score_candidates(cand2.view(B * rho, -1), anchor, rho, "l2", seq_lens=cand_lens, prefix_lens=pos2.reshape(-1), kv=kv_b)
    i2_d, f2_d = m.score_candidates(cand2.view(B * rho, -1), anchor, rho, "l2", seq_lens=cand_lens)   # no prefix reuse at all
    assert torch.equal(i2_b, i2_d) and torch.equal(f2_b, f2_d)


def test_search_with_and_without_the_fused_caption_pass_agree(monkeypatch):
    """search_synthetic, k = 2, ViT-L: the adversarial ids with the caption pass fused into every first stage equal those with
    the separate leaf_text_forward_kv pass (LEAF_FUSE_KV=0) and those without any prefix reuse."""
    import torch
    from leaf_amd.model import create_model
    from leaf_amd.step import StepConfig, search_synthetic
    m = create_model("ViT-L-14-quickgelu", seed=2)
    base = torch.from_numpy(O.synthetic_tokens(6, seed=13).astype(np.int32)).cuda()
    lens = (base.argmax(-1) + 1).cpu().numpy().astype(np.int32)
    anchor = m.encode_text(base, seq_lens=lens) + 0.03
    sc = StepConfig(rho=9, k_adv=2)
    fused = search_synthetic(m, anchor, base, sc, seed=4, base_lens=lens)
    monkeypatch.setenv("LEAF_FUSE_KV", "0")
    separate = search_synthetic(m, anchor, base, sc, seed=4, base_lens=lens)
    plain = search_synthetic(m, anchor, base, sc, seed=4, base_lens=lens, prefix_reuse=False)
    assert torch.equal(fused, separate) and torch.equal(fused, plain)


def test_fused_caption_pass_falls_back_when_the_rows_do_not_fit_one_chunk():
    """With a row budget smaller than captions + candidates the fused entry point declines (returns None, error text set by the
    C ABI) and search_synthetic runs the separate caption pass: same adversarial ids as without any prefix reuse."""
    import torch
    from leaf_amd.model import create_model
    from leaf_amd.step import StepConfig, SyntheticCandidates, search_synthetic
    m = create_model("tiny-test-quickgelu", seed=6)
    base = torch.from_numpy(O.synthetic_tokens(8, seed=2).astype(np.int32)).cuda()
    lens = (base.argmax(-1) + 1).cpu().numpy().astype(np.int32)
    anchor = m.encode_text(base, seq_lens=lens) + 0.02
    sc = StepConfig(rho=10, k_adv=1)
    want = search_synthetic(m, anchor, base, sc, seed=9, base_lens=lens, prefix_reuse=False)
    m.set_option("chunk", 8)            # 8 x 77 rows per pass: the 8 captions alone fit, captions + 80 candidates do not
    gen = SyntheticCandidates(base, lens, 10, m.cfg.vocab_size, 9)
    cand, pos = gen.stage1(base)
    assert m.score_candidates_fused(base, lens, cand.view(80, -1), anchor, 10, np.repeat(lens, 10), pos.reshape(-1)) is None
    got = search_synthetic(m, anchor, base, sc, seed=9, base_lens=lens)
    assert torch.equal(got, want)


def test_bench_fresh_batches_fixed_batch_and_rank_sim():
    """bench.py round 5: by default every step trains on a NEW synthetic batch (seeded by rank and step,
    built on a side stream one step ahead) -- the line says so and two runs give the same loss; --fixed-batch restores the rounds 1-4
    form; --rank-sim R runs R ranks' batches one after another on the same weights (gradient sum, one AdamW) and prints the predicted
    R-GPU efficiency with its ingredients instead of the metric line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    common = ["--model", "tiny-test-quickgelu", "--batch", "16", "--rho", "8", "--no-cpu-baseline", "--no-dense-leg"]

    def run(extra):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common + extra, env=env, capture_output=True, text=True,
                           timeout=600, cwd=root)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1, p.stdout[-2000:]
        return json.loads(line[0])
    fresh, again, fixed = run(["--steps", "6", "--warmup", "2"]), run(["--steps", "6", "--warmup", "2"]), run(["--steps", "6", "--warmup", "2", "--fixed-batch"])
    assert "new batch per step" in fresh["data"] and "fixed batch" in fixed["data"]
    assert fresh["loss"] == again["loss"] and np.isfinite(fresh["loss"]) and fresh["loss"] != fixed["loss"]
    assert fresh["config"]["workload_key"].endswith("freshbatch") and fixed["config"]["workload_key"].endswith("fixedbatch")
    sim = run(["--rank-sim", "3", "--steps", "4", "--warmup", "1"])
    assert sim["rank_sim"] == 3 and len(sim["ms_per_rank_step_by_rank"]) == 3 and sim["steps"] == 4
    assert sim["ms_max_over_ranks_mean"] >= sim["ms_per_rank_step_mean"] > 0 and sim["skew_ratio_max_over_mean"] >= 1.0
    assert 0.0 < sim["pred_eff_with_collective_ESTIMATE"] <= sim["pred_eff_skew_only"] <= 1.0
    coll = sim["exposed_collective"]          # measured one-rank RCCL time of the exposed bucket + a stated link model
    assert coll["bucket_bytes"] > 0 and coll["ranks"] == 3 and coll["link_ms_model"] > 0
    assert coll["one_rank_rccl_ms_measured"] is None or coll["one_rank_rccl_ms_measured"] > 0
    assert sim["scored_rows_per_rank_step_min_mean_max"][0] > 0


def test_bench_two_ranks_on_rccl_when_the_box_has_two_gpus():
    """the first contact of the multi-rank step with RCCL.  ``bench.py --gpus 2 --backend nccl`` (its own launcher:
    torch.distributed.run, one rank per device, bucketed all-reduce behind the backward) on any box with >= 2 GPUs; the one-GPU test
    boxes skip it.  ``torch.cuda.device_count()`` does not initialise the GPU in this process (the ranks are child processes)."""
    import json
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL wants one device per rank); the one-GPU rehearsals are test_gpu_dp2.py and the test above")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LEAF_BENCH_FORCE_DIST", "LEAF_DP_OVERLAP")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "nccl", "--model", "tiny-test-quickgelu",
                        "--batch", "16", "--rho", "8", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-dense-leg"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and np.isfinite(out["loss"])
    assert len(out["per_rank"]["ms_per_step"]) == 2 and out["dp_gradient_buckets"]["active"]
