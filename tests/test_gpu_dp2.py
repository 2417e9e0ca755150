"""Data parallelism with TWO ranks on the one GPU of the test box (gloo transport, CUDA tensors): the bucketed, event-driven
gradient reduction of leaf_amd.step.GradReducer (leaf_textfare_backward_events + all-reduces on a side stream while the
backward of the earlier blocks still runs) against (a) the single flat all-reduce -- bit for bit -- and (b) ONE process on the
concatenated batch ("N-GPU step == 1-GPU step on the concatenated batch").  RCCL itself needs one device per rank
(the driver's scaling run); everything above the transport is the code exercised here."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(name, n):
    sys.path.insert(0, ROOT)
    from oracle import text_oracle as O
    toks = O.synthetic_tokens(n, seed=31, min_len=5, max_len=30).astype(np.int32)
    lens = (toks.argmax(-1) + 1).astype(np.int32)
    return toks, lens


def _worker(rank, world, port, name, n, overlap, out, sparse=0):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LEAF_DP_OVERLAP="1" if overlap else "0", LEAF_DP_SPARSE_EMBED=str(sparse))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from leaf_amd.model import create_model
    from leaf_amd.step import get_reducer
    m = create_model(name, seed=1, trainable=True)
    toks, lens = _inputs(name, n)
    anchor = m.encode_text(toks, seq_lens=lens)
    g = torch.Generator(device="cpu").manual_seed(7)
    anchor = anchor + (anchor.norm(dim=-1, keepdim=True) / anchor.shape[-1] ** 0.5) * torch.randn(anchor.shape, generator=g).to(anchor.device)
    sl = slice(rank * (n // world), (rank + 1) * (n // world))
    m.train()
    feat = m.forward_train(toks[sl], seq_lens=lens[sl])
    m.zero_grad()
    red = get_reducer(m)
    assert red.overlap == bool(overlap) and red.sparse_embed == bool(sparse)
    red.begin_step(int(lens[sl].sum()))          # (a no-op unless the sparse embedding reduction is on)
    loss = red.backward(feat, anchor[sl].contiguous())
    scale = red.finish()
    torch.cuda.synchronize()
    grads = (m.grads * scale).cpu()
    m.adamw_step(1e-4, (0.9, 0.999), 1e-8, 1e-4, grad_scale=scale)
    torch.cuda.synchronize()
    torch.save({"loss": float(loss), "scale": scale, "grads": grads, "flat": m.flat.cpu(), "buckets": len(red.plan)}, f"{out}.{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("name,n", [("tiny-test-quickgelu", 8), ("ViT-L-14-quickgelu", 6)])
def test_two_ranks_overlapped_equals_flat_and_the_concatenated_batch(tmp_path, name, n):
    import torch
    import torch.multiprocessing as mp
    res = {}
    for overlap in (1, 0):
        out = str(tmp_path / f"r{overlap}")
        mp.spawn(_worker, args=(2, _free_port(), name, n, overlap, out), nprocs=2, join=True)
        res[overlap] = [torch.load(out + ".0"), torch.load(out + ".1")]
    a0, a1 = res[1]
    b0, b1 = res[0]
    assert a0["scale"] == a1["scale"] == 0.5
    # every rank ends with the same averaged gradient and the same updated weights ...
    assert torch.equal(a0["grads"], a1["grads"]) and torch.equal(a0["flat"], a1["flat"])
    # ... and the bucketed, overlapped reduction equals the flat all-reduce (a two-term sum has one order).  The two forms are
    # separate runs: the block gradients (deterministic kernels: grouped weight gradients, LayerNorm partial sums) must
    # agree bit for bit, the embedding-table gradient (atomic scatter of repeated tokens) to fp32 re-association noise
    sys.path.insert(0, ROOT)
    from leaf_amd.model import get_config
    from leaf_amd.step import bucket_plan
    lay_cfg = get_config(name)
    assert a0["buckets"] >= 2
    diff = (a0["grads"] - b0["grads"]).norm() / b0["grads"].norm()
    assert float(diff) < 1e-6, float(diff)
    assert float((a0["flat"] - b0["flat"]).abs().max()) < 1e-6
    # one process on the concatenated batch
    sys.path.insert(0, ROOT)
    from leaf_amd.model import create_model
    m = create_model(name, seed=1, trainable=True)
    toks, lens = _inputs(name, n)
    anchor = m.encode_text(toks, seq_lens=lens)
    g = torch.Generator(device="cpu").manual_seed(7)
    anchor = anchor + (anchor.norm(dim=-1, keepdim=True) / anchor.shape[-1] ** 0.5) * torch.randn(anchor.shape, generator=g).to(anchor.device)
    m.train()
    feat = m.forward_train(toks, seq_lens=lens)
    m.zero_grad()
    loss = m.backward(feat, anchor)
    torch.cuda.synchronize()
    ref = m.grads.cpu()
    assert abs(0.5 * (a0["loss"] + a1["loss"]) - float(loss)) < 1e-4 * abs(float(loss))
    # per-tensor against the one-process run (the loss scale is a power of two, so the half-batch backwards round like the
    # whole-batch one; what differs is fp32 summation order: 9.5e-8 measured)
    worst = 0.0
    for k, (off, shape) in m.layout.items():
        cnt = int(np.prod(shape))
        r = ref[off:off + cnt]
        if float(r.norm()) == 0.0:
            continue
        worst = max(worst, float((a0["grads"][off:off + cnt] - r).norm() / r.norm()))
    # the block weights' gradients of the overlapped and the flat run: bit-identical
    for off, numel in [r for ev, rs in bucket_plan(m.layout, m.n_params, lay_cfg.layers)[:-1] for r in rs]:
        assert torch.equal(a0["grads"][off:off + numel], b0["grads"][off:off + numel])
    print("2 ranks vs concatenated batch: worst per-tensor grad rel-L2", worst)
    assert worst < 1e-5


def test_two_ranks_sparse_embedding_reduction_equals_the_dense_one(tmp_path):
    """LEAF_DP_SPARSE_EMBED=1 (opt-in): the exposed last bucket reduces the token-embedding gradient by touched rows
    (id flags all-reduced with MAX, a fixed-size selection, SUM all-reduce of the gathered rows) and the rest of the bucket densely.
    Two ranks on this GPU (gloo transport): every rank ends with the same gradient and weights, equal to the dense overlapped form
    (embedding table to the re-association noise of its atomic scatter between separate runs, everything else bit for bit)."""
    import torch
    import torch.multiprocessing as mp
    name, n = "ViT-L-14-quickgelu", 6
    res = {}
    for sparse in (1, 0):
        out = str(tmp_path / f"s{sparse}")
        mp.spawn(_worker, args=(2, _free_port(), name, n, 1, out, sparse), nprocs=2, join=True)
        res[sparse] = [torch.load(out + ".0"), torch.load(out + ".1")]
    a0, a1 = res[1]
    d0, _ = res[0]
    assert torch.equal(a0["grads"], a1["grads"]) and torch.equal(a0["flat"], a1["flat"])
    sys.path.insert(0, ROOT)
    from leaf_amd.model import get_config
    from leaf_amd.step import bucket_plan
    from tests.test_dp_gloo import _layout_for
    layout, n_params = _layout_for(get_config(name))
    off_t, shape_t = layout["token_embedding.weight"]
    n_t = shape_t[0] * shape_t[1]
    tok_s, tok_d = a0["grads"][off_t:off_t + n_t], d0["grads"][off_t:off_t + n_t]
    assert float(tok_d.norm()) > 0 and float((tok_s - tok_d).norm() / tok_d.norm()) < 1e-6
    assert int((tok_s.view(shape_t).abs().sum(-1) > 0).sum()) == int((tok_d.view(shape_t).abs().sum(-1) > 0).sum())     # same touched rows
    rest_s, rest_d = a0["grads"][off_t + n_t:], d0["grads"][off_t + n_t:]
    assert float((rest_s - rest_d).norm() / rest_d.norm()) < 1e-6
    for off, numel in [r for ev, rs in bucket_plan(layout, n_params, get_config(name).layers)[:-1] for r in rs]:      # block weights: bit for bit
        assert torch.equal(a0["grads"][off:off + numel], d0["grads"][off:off + numel])


def _micro_clip_run(m, toks, lens, anchor, rows_of_micro, max_norm):
    """two micro-batches through the trainer's MicroClip; returns (norm found at each micro-batch, gradient the step sees)"""
    import types
    import torch
    from leaf_amd.train import MicroClip
    mc = MicroClip(m, types.SimpleNamespace(grad_clip_norm=max_norm, accum_freq=len(rows_of_micro), precision="amp_bf16"))
    m.train()
    m.zero_grad()
    norms = []
    for j, sl in enumerate(rows_of_micro):
        feat = m.forward_train(toks[sl], seq_lens=lens[sl])
        mc.backward(feat, anchor[sl].contiguous(), j)
        norms.append(float(m._clipi_ws[1]))
    torch.cuda.synchronize()
    return norms, m.grads.cpu()


def _micro_clip_worker(rank, world, port, name, n, max_norm, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from leaf_amd.model import create_model
    m = create_model(name, seed=1, trainable=True)
    toks, lens = _inputs(name, n)
    anchor = m.encode_text(toks, seq_lens=lens)
    g = torch.Generator(device="cpu").manual_seed(7)
    anchor = anchor + (anchor.norm(dim=-1, keepdim=True) / anchor.shape[-1] ** 0.5) * torch.randn(anchor.shape, generator=g).to(anchor.device)
    q = n // 4                                   # micro-batch j of rank r = rows [(2 j + r) q, (2 j + r + 1) q)
    norms, grads = _micro_clip_run(m, toks, lens, anchor, [slice((2 * j + rank) * q, (2 * j + rank + 1) * q) for j in (0, 1)], max_norm)
    torch.save({"norms": norms, "grads": grads}, f"{out}.{rank}")
    dist.destroy_process_group()


def test_micro_batch_clipping_two_ranks_equals_the_concatenated_micro_batches(tmp_path):
    """--grad-clip-norm with --accum-freq 2 under data parallelism (utils_AT.py:348-362 inside DDP: every micro-batch's backward
    all-reduces, the clip sees the mean over ranks of the running sum): two ranks with half of each micro-batch == one process
    on the whole micro-batches -- the norms found at each clip and the gradient handed to the optimizer step."""
    import torch
    import torch.multiprocessing as mp
    name, n = "tiny-test-quickgelu", 16
    sys.path.insert(0, ROOT)
    from leaf_amd.model import create_model
    m = create_model(name, seed=1, trainable=True)
    toks, lens = _inputs(name, n)
    anchor = m.encode_text(toks, seq_lens=lens)
    g = torch.Generator(device="cpu").manual_seed(7)
    anchor = anchor + (anchor.norm(dim=-1, keepdim=True) / anchor.shape[-1] ** 0.5) * torch.randn(anchor.shape, generator=g).to(anchor.device)
    free_norms, _ = _micro_clip_run(m, toks, lens, anchor, [slice(0, 8), slice(8, 16)], 1e30)
    max_norm = 0.4 * min(free_norms)             # both clips active
    ref_norms, ref = _micro_clip_run(m, toks, lens, anchor, [slice(0, 8), slice(8, 16)], max_norm)
    assert all(x > max_norm for x in ref_norms) and float(ref.norm()) <= max_norm * (1 + 1e-5)
    out = str(tmp_path / "mc")
    mp.spawn(_micro_clip_worker, args=(2, _free_port(), name, n, max_norm, out), nprocs=2, join=True)
    a0, a1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(a0["grads"], a1["grads"]) and a0["norms"] == a1["norms"]           # replicas stay replicas
    assert np.allclose(a0["norms"], ref_norms, rtol=1e-5), (a0["norms"], ref_norms)
    assert float((a0["grads"] - ref).norm() / ref.norm()) < 1e-5


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`bench.py --gpus 2 --backend gloo`: the launcher, the barrier-bracketed timing with the max over ranks, n_ranks_seen and
    the full step (search with per-rank seeds, bucketed reduction beside the backward, AdamW) with two ranks sharing this box's
    GPU.  A functional rehearsal of the driver's N > 1 command; the line says it is not the metric."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LEAF_BENCH_FORCE_DIST", "LEAF_DP_OVERLAP")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--model", "tiny-test-quickgelu",
                        "--batch", "16", "--rho", "8", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-dense-leg"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and "rehearsal" in d
    assert d["value"] > 0 and np.isfinite(d["loss"])
    assert abs(d["value"] - 2 * 16 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]      # whole-job samples / max-over-ranks time
    # per-rank diagnostics: own step time, exposed reduction wait, rows the rank's (differently seeded) search computed
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and all(0 < t <= d["ms_per_step"] * 1.5 for t in pr["ms_per_step"])
    assert all(0 <= e <= t for e, t in zip(pr["exposed_collective_ms_per_step"], pr["ms_per_step"]))
    assert all(r > 0 for r in pr["scored_rows_per_step"]) and d["host_threads_per_rank"] >= 1
