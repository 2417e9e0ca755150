#!/usr/bin/env python3
"""Where does a GEMM workgroup spend its cycles?  Needs the diagnostic build (make -C leaf_amd/csrc stamps) and
LEAF_HIP_LIB=tools/diag/libleaf_hip_stamps.so.  Shares only; never quote this build's run time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import _lib

lib = _lib.lib()
dev = torch.device("cuda:0")
M = int(os.environ.get("SEQS", "1024")) * 77
d = 768
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, epi, N, K in [("qkv", 0, 3 * d, d), ("out", 2, d, d), ("fc", 1, 4 * d, d), ("proj", 2, d, 4 * d)]:
    A = (torch.randn(M, K, device=dev) * 0.5).half()
    B = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev)
    Cm = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
    nblk = ((M + 255) // 256) * (N // 256)
    stamps = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    args = (1, epi, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()),
            C.c_void_p(bias.data_ptr()), None, M, N, K, 1, 0.0, 0, st)
    for _ in range(3):
        lib.leaf_op_gemm(*args)
    lib.leaf_debug_gemm_stamps(C.c_void_p(stamps.data_ptr()))
    lib.leaf_op_gemm(*args)
    torch.cuda.synchronize()
    lib.leaf_debug_gemm_stamps(None)
    s = stamps.cpu().numpy().reshape(nblk, 8)[:, :5].astype(np.float64)
    s = s[s[:, 4] > 0]            # the persistent form records the LAST tile of each of its (one per CU) workgroups
    seg = np.diff(s, axis=1)
    tot = s[:, 4] - s[:, 0]
    names = ["first DMA wait", "K loop (main)", "K tail", "epilogue"]
    print(f"{name}: N={N} K={K} tiles={nblk} stamped workgroups={len(s)} median tile {np.median(tot):.0f} ticks; kernel span {(s[:,4].max()-s[:,0].min()):.0f} ticks")
    for i, n in enumerate(names):
        print(f"   {n:22s} median {np.median(seg[:, i]):9.0f}  mean {seg[:, i].mean():9.0f}  ({100 * seg[:, i].sum() / tot.sum():5.1f}%)")
