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
