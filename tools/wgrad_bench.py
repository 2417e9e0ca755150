#!/usr/bin/env python3
"""Time the grouped weight-gradient kernel (C-ABI hook leaf_op_wgrad, one problem per launch) on the four shapes of a block."""
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
rows = int(os.environ.get("ROWS", "3219"))
d = int(os.environ.get("WIDTH", "768"))
tot = 0.0
for name, Nw, Kw in (("c_proj", d, 4 * d), ("c_fc", 4 * d, d), ("out", d, d), ("qkv", 3 * d, d)):
    dy = torch.randn(rows, Nw, device=dev).half()
    x = torch.randn(rows, Kw, device=dev).half()
    dw = torch.zeros(Nw, Kw, device=dev)
    db = torch.zeros(Nw, device=dev)
    al = torch.tensor([0.5], device=dev)
    for _ in range(3):
        _lib.check(lib.leaf_op_wgrad(p(dy), p(x), p(dw), p(db), rows, Nw, Kw, 1, 1, p(al), st), "wgrad")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.leaf_op_wgrad(p(dy), p(x), p(dw), p(db), rows, Nw, Kw, 1, 1, p(al), st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tot += us
    print(f"{name:7s} Nw={Nw} Kw={Kw} tiles={Nw * Kw // 16384}: {us:7.1f} us  {2.0 * rows * Nw * Kw / us / 1e6:6.0f} TFLOP/s", flush=True)
print(f"sum {tot:.1f} us (the grouped launch of a block runs all four at once)")
