#!/usr/bin/env python3
"""In-kernel cycle stamps of the weight-gradient kernel (diagnostic build: make -C leaf_amd/csrc stamps;
LEAF_HIP_LIB=tools/diag/libleaf_hip_stamps.so): cycles per 32-row step and the clock the kernel ran at."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import _lib

lib = _lib.lib()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
rows = int(os.environ.get("ROWS", "3219"))
d = 768
for name, Nw, Kw in (("c_proj", d, 4 * d), ("c_fc", 4 * d, d), ("qkv", 3 * d, d)):
    dy = torch.randn(rows, Nw, device=dev).half(); x = torch.randn(rows, Kw, device=dev).half()
    dw = torch.zeros(Nw, Kw, device=dev); db = torch.zeros(Nw, device=dev); al = torch.tensor([0.5], device=dev)
    nblk = Nw * Kw // 16384
    stamps = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    for _ in range(5):
        lib.leaf_op_wgrad(p(dy), p(x), p(dw), p(db), rows, Nw, Kw, 1, 1, p(al), st)
    lib.leaf_debug_gemm_stamps(p(stamps))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.leaf_op_wgrad(p(dy), p(x), p(dw), p(db), rows, Nw, Kw, 1, 1, p(al), st); e1.record()
    torch.cuda.synchronize()
    lib.leaf_debug_gemm_stamps(None)
    s = stamps.cpu().numpy().reshape(nblk, 8)[:, :4].astype(np.float64)
    seg = np.diff(s, axis=1)
    span = s[:, 3].max() - s[:, 0].min()
    us = e0.elapsed_time(e1) * 1e3
    nk = (rows + 31) // 32
    print(f"{name}: {us:.1f} us, kernel span {span:.0f} ticks -> {span / us / 1e3:.2f} GHz-equivalent; prologue {np.median(seg[:,0]):.0f}, "
          f"loop {np.median(seg[:,1]):.0f} = {np.median(seg[:,1]) / nk:.0f} ticks per step ({nk} steps), epilogue {np.median(seg[:,2]):.0f}")
