#!/usr/bin/env python3
"""Do the HBM-heavy epilogues of the residual GEMM (<2>/<7>: fp32 read-modify-write of a 256 x 256 tile = 512 KB per workgroup) of the 256
CUs coincide?  Stamps build (make -C leaf_amd/csrc stamps; LEAF_HIP_LIB=tools/diag/libleaf_hip_stamps.so): per workgroup the start / end
of its epilogue; prints the epilogue duration and how many workgroups are inside an epilogue at the same time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import _lib

lib = _lib.lib()
dev = torch.device("cuda:0")
M = int(os.environ.get("ROWS", "107520"))
d = int(os.environ.get("WIDTH", "768"))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, epi, N, K in [("out", 2, d, d), ("proj", 2, d, 4 * d), ("proj, fp32 store only (no residual read)", 3, d, 4 * d), ("qkv", 0, 3 * d, d)]:
    A = (torch.randn(M, K, device=dev) * 0.5).half()
    B = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev)
    Cm = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.float16)
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
    s = s[s[:, 4] > 0]
    t0 = s[:, 0].min()
    span = s[:, 4].max() - t0
    e0, e1 = s[:, 3] - t0, s[:, 4] - t0
    dur = e1 - e0
    kl = s[:, 3] - s[:, 0]
    # concurrency: sample the kernel span at 2000 points
    ts = np.linspace(0, span, 2000)
    conc = ((e0[None, :] <= ts[:, None]) & (ts[:, None] < e1[None, :])).sum(1)
    busy = conc > 0
    print(f"{name}: N={N} K={K} workgroups stamped {len(s)} span {span:.0f} ticks; start..epilogue median {np.median(kl):.0f}; epilogue median {np.median(dur):.0f} "
          f"p10 {np.percentile(dur, 10):.0f} p90 {np.percentile(dur, 90):.0f} ticks = {100 * np.median(dur) / np.median(dur + kl):.1f} % of a tile")
    print(f"   workgroups inside an epilogue at one time: mean {conc.mean():.1f}, max {conc.max()}, >= 128 for {100 * (conc >= 128).mean():.1f} % of the span, "
          f"none for {100 * (~busy).mean():.1f} % of the span")
    # the same per dispatch round: order workgroups by start time, rounds of 256
    order = np.argsort(s[:, 0])
    for r in range(0, min(len(order), 256 * 3), 256):
        idx = order[r:r + 256]
        print(f"   round {r // 256}: starts within {np.ptp(s[idx, 0]):.0f} ticks, epilogue starts within {np.ptp(e0[idx]):.0f} ticks, epilogue median {np.median(dur[idx]):.0f}")
