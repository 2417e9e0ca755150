"""Data-parallel path on CPU (gloo, world_size 2): the single flat all-reduce + 1/world scaling of leaf_amd.step
reproduces the gradient of the concatenated batch ("N-GPU step == 1-GPU step on the concatenated
batch"), and the caption loader shards ranks disjointly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from leaf_amd.step import allreduce_grads
    from leaf_amd.train import TextLoader
    from oracle import text_oracle as O
    cfg = O.TextCfg(128, 2, 1, 64, quick_gelu=True)
    w = O.init_weights(cfg, seed=3)
    toks = O.synthetic_tokens(4, seed=8)
    anchor = O.encode_text(w, cfg, toks) + 0.05
    shard = slice(rank * 2, rank * 2 + 2)
    _, _, g = O.encode_text_backward(w, cfg, toks[shard], anchor[shard])
    keys = sorted(k for k in g if k != "token_embedding.weight")

    class FakeModel:   # what allreduce_grads touches: the flat gradient buffer
        grads = torch.from_numpy(np.concatenate([g[k].ravel() for k in keys]))
    scale = allreduce_grads(FakeModel)
    avg = FakeModel.grads.numpy() * scale
    _, _, gfull = O.encode_text_backward(w, cfg, toks, anchor)
    full = np.concatenate([gfull[k].ravel() for k in keys])
    rel = float(np.linalg.norm(avg - full) / np.linalg.norm(full))
    caps = [f"caption {i}" for i in range(40)]
    loader = TextLoader(caps, None, batch_size=5, num_samples=40, rank=rank, world=world, seed=0)
    seen = [t for _, texts in loader for t in texts]
    torch.save({"rel": rel, "scale": scale, "seen": seen, "batches": len(loader)}, f"{out}.{rank}")
    dist.destroy_process_group()


def test_flat_allreduce_equals_concatenated_batch(tmp_path):
    port, out = _free_port(), str(tmp_path / "res")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["scale"] == r1["scale"] == 0.5
    assert r0["rel"] < 1e-5 and r1["rel"] < 1e-5
    assert r0["batches"] == r1["batches"] == 4
    assert not set(r0["seen"]) & set(r1["seen"]) and len(set(r0["seen"]) | set(r1["seen"])) == 40


def _layout_for(cfg):
    """The engine's flat layout via host-only C calls (as tests/test_checkpoint_cpu.py)."""
    import ctypes as C
    from leaf_amd import _lib
    lib = _lib.lib()
    c = _lib.TextCfgC(cfg.layers, cfg.width, cfg.heads, cfg.embed_dim, cfg.context_length, cfg.vocab_size, int(cfg.quick_gelu), cfg.ln_eps)
    h = C.c_void_p()
    _lib.check(lib.leaf_text_create(C.byref(c), 1, C.byref(h)), "create")
    out, name = {}, C.create_string_buffer(128)
    off, rows, cols = C.c_size_t(), C.c_int64(), C.c_int64()
    for i in range(lib.leaf_text_num_tensors(h)):
        lib.leaf_text_param_info(h, i, name, 128, C.byref(off), C.byref(rows), C.byref(cols))
        out[name.value.decode()] = (off.value, (rows.value, cols.value) if cols.value else (rows.value,))
    n = lib.leaf_text_param_count(h)
    lib.leaf_text_destroy(h)
    return out, n


@pytest.mark.parametrize("name", ["tiny-test", "ViT-L-14", "ViT-bigG-14"])
def test_bucket_plan_partitions_the_gradient_buffer(name):
    """GradReducer's buckets (one per transformer block in completion order + the rest) cover every gradient element exactly
    once, each block bucket holds exactly that block's four GEMM weights, and the order is L-1 .. 0 then the remainder."""
    sys.path.insert(0, ROOT)
    from leaf_amd.model import MODEL_CONFIGS
    from leaf_amd.step import bucket_plan
    cfg = MODEL_CONFIGS[name]
    layout, n = _layout_for(cfg)
    plan = bucket_plan(layout, n, cfg.layers)
    evs = [ev for ev, _ in plan]
    assert evs[-1] == cfg.layers and evs[:-1] == sorted(evs[:-1], reverse=True) and evs[-2] == 0, "completion order, block 0 last"
    spans = sorted((off, off + m) for _, ranges in plan for off, m in ranges if m)
    assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:])), "gap or overlap"
    per_block = 12 * cfg.width ** 2
    hi_layer = cfg.layers - 1
    for ev, ranges in plan[:-1]:
        (off, m), = ranges
        nblk = hi_layer - ev + 1                       # blocks ev .. hi_layer, the lowest one finishes last
        assert m == nblk * per_block
        assert m * 4 >= (64 << 20) or ev == 0, "buckets hold at least 64 MB (except the remainder at block 0)"
        for k, (o, shp) in layout.items():
            inside = off <= o < off + m
            blk = int(k.split(".")[2]) if k.startswith("transformer.resblocks.") else -1
            assert inside == (ev <= blk <= hi_layer and k.endswith("weight") and "ln_" not in k), k
        hi_layer = ev - 1
    assert hi_layer == -1
    # the fp16 backward's saturation poison (NaN in gradient element 0 = token_embedding.weight[0, 0], api_train.hip) must ride in
    # the LAST collective: a bucket launched earlier would ship a clean element 0 while another rank skips the step
    assert layout["token_embedding.weight"][0] == 0
    assert any(off == 0 and m > 0 for off, m in plan[-1][1]) and not any(off <= 0 < off + m for _, rs in plan[:-1] for off, m in rs)
    # one block per bucket when asked for
    assert [ev for ev, _ in bucket_plan(layout, n, cfg.layers, min_bytes=1)] == list(reversed(range(cfg.layers))) + [cfg.layers]


def _bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from leaf_amd.model import MODEL_CONFIGS
    from leaf_amd.step import bucket_plan, reduce_buckets
    cfg = MODEL_CONFIGS["tiny-test"]
    layout, n = _layout_for(cfg)
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(n, generator=g)
    flat = mine.clone()
    dist.all_reduce(flat)
    bucketed = mine.clone()
    reduce_buckets(bucketed, bucket_plan(layout, n, cfg.layers))
    torch.save({"equal": bool(torch.equal(flat, bucketed))}, f"{out}.{rank}")
    dist.destroy_process_group()


def test_bucketed_reduction_equals_the_flat_allreduce(tmp_path):
    """Same sums bit for bit (an all-reduce adds the same two operands per element whatever the bucket boundaries), same
    collective order on every rank."""
    port, out = _free_port(), str(tmp_path / "bk")
    mp.spawn(_bucket_worker, args=(2, port, out), nprocs=2, join=True)
    assert torch.load(out + ".0")["equal"] and torch.load(out + ".1")["equal"]


def _sparse_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from leaf_amd.step import _minus, reduce_token_rows
    vocab, d, B, ctx = 4000, 24, 6, 77
    g = torch.Generator().manual_seed(500 + rank)
    toks = torch.zeros(B, ctx, dtype=torch.int32)
    n_mine = 0
    for b in range(B):
        L = int(torch.randint(3, 40, (1,), generator=g))
        toks[b, :L] = torch.randint(1, vocab, (L,), generator=g, dtype=torch.int32)
        n_mine += L
    toks[0, 1] = 77                    # a token every rank has, and (below) the poison element in row 0
    table = torch.zeros(vocab, d)
    ids = toks.reshape(-1).long().unique()
    table[ids] = torch.randn(len(ids), d, generator=g)
    table[0] = 0.0
    if rank == 1:
        table[0, 0] = float("nan")     # the saturation poison of ONE rank must reach every rank
    dense = table.clone()
    dist.all_reduce(dense)
    nmax = torch.tensor([n_mine])
    dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
    sparse = table.clone()
    reduce_token_rows(sparse, toks, world * (int(nmax) + 1))
    same = bool(torch.equal(torch.nan_to_num(dense, nan=7.0), torch.nan_to_num(sparse, nan=7.0))) if world == 2 else \
        bool(torch.allclose(torch.nan_to_num(dense, nan=7.0), torch.nan_to_num(sparse, nan=7.0), rtol=1e-6, atol=1e-6))
    tiny = table.clone()               # a cap below the union would drop rows: the bound is the caller's duty, here it is checked that a generous one is harmless
    reduce_token_rows(tiny, toks, vocab + 10)
    torch.save({"same": same, "nan_everywhere": bool(torch.isnan(sparse[0, 0])), "full_cap_same": bool(torch.allclose(torch.nan_to_num(tiny, nan=7.0),
                torch.nan_to_num(dense, nan=7.0), rtol=1e-6, atol=1e-6)), "minus": _minus([(0, 100), (200, 50)], 10, 30)}, f"{out}.{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_token_row_reduction_equals_the_dense_allreduce(tmp_path, world):
    """LEAF_DP_SPARSE_EMBED (leaf_amd.step.reduce_token_rows): reducing the token-embedding gradient by the rows
    some rank touched -- MAX all-reduce of id flags, a fixed-size stable selection, SUM all-reduce of the gathered block -- gives what the
    dense all-reduce of the whole table gives (bit for bit at two ranks), and a NaN one rank planted in element [0, 0] (the fp16
    backward's saturation poison) reaches every rank."""
    port, out = _free_port(), str(tmp_path / "sp")
    mp.spawn(_sparse_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(f"{out}.{r}")
        assert res["same"] and res["nan_everywhere"] and res["full_cap_same"], (r, res)
        assert res["minus"] == [(0, 10), (40, 60), (200, 50)]
