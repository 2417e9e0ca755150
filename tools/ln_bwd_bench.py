#!/usr/bin/env python3
"""Time the LayerNorm backward kernel alone (C-ABI hook leaf_op_layernorm_bwd) with and without the dg / db reduction."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leaf_amd import _lib

lib = _lib.lib()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for rows, d in [(3219, 768), (3219, 1024), (3219, 1280), (9856, 768)]:
    x, dy, dx = (torch.randn(rows, d, device=dev) for _ in range(3))
    g, dg, db = (torch.randn(d, device=dev) for _ in range(3))
    d16 = torch.zeros(rows, d, dtype=torch.float16, device=dev)
    gs = torch.tensor([8.0, 0.125], device=dev)
    big = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    ws = torch.empty(lib.leaf_op_layernorm_bwd_ws_bytes(rows, d), dtype=torch.uint8, device=dev)
    for name, a, b in (("dx + dg/db", dg, db), ("dx only", None, None)):
        ts = []
        for it in range(12):
            big.zero_()                      # evict the operands from L2 / MALL, as in the step
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.leaf_op_layernorm_bwd(p(dy), p(x), p(g), 1e-5, p(dx), p(d16), 1, p(gs), p(a), p(b), rows, d, p(ws), ws.numel(), st), "ln_bwd")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"rows={rows} d={d} {name:11s}: {min(ts[2:]):7.1f} us (median {sorted(ts[2:])[len(ts[2:]) // 2]:7.1f})", flush=True)
