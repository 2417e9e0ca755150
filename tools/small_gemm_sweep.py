#!/usr/bin/env python3
"""Which GEMM kernel should take the B-caption launches (3-13k rows)?  Times the layer's GEMM shapes through the C-ABI hooks
with the half-stage 256^2 ring kernel forced on (min tiles 1) and off (min tiles 10^9 -> 64 x 128 ring / two-stage kernels).
Needs the diagnostic build for leaf_debug_gemm_min_tiles: LEAF_HIP_LIB=tools/diag/libleaf_hip_variants.so (make -C leaf_amd/csrc variants)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leaf_amd import _lib


def main():
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = int(os.environ.get("WIDTH", "768"))
    shapes = [("qkv", 0, 3 * d, d), ("out", 2, d, d), ("fc", 1, 4 * d, d), ("proj", 2, d, 4 * d), ("dgrad_fc", 3, d, 4 * d), ("dgrad_qkv", 3, d, 3 * d)]
    for M in [int(x) for x in os.environ.get("MS", "800,1600,3219,4800,6400,9600,12800").split(",")]:
        for name, epi, N, K in shapes:
            A = (torch.randn(M, K, device=dev) * 0.5).half()
            B = (torch.randn(N, K, device=dev) * 0.05).half()
            bias = torch.randn(N, device=dev)
            Cm = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (2, 3) else torch.float16)
            args = (1, epi, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), C.c_void_p(bias.data_ptr()),
                    None, M, N, K, 1, 0.0, 0, st)
            res = []
            for mt in (1, 10 ** 9):
                lib.leaf_debug_gemm_min_tiles(mt)
                for _ in range(3):
                    _lib.check(lib.leaf_op_gemm(*args), "gemm")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                it = 20
                e0.record()
                for _ in range(it):
                    lib.leaf_op_gemm(*args)
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / it * 1e3)
            tiles = ((M + 255) // 256) * (N // 256)
            print(f"M={M:6d} {name:9s} N={N:5d} K={K:5d} tiles256={tiles:4d}: ring256 {res[0]:7.1f} us   small {res[1]:7.1f} us   ratio {res[1] / res[0]:.2f}", flush=True)


if __name__ == "__main__":
    main()
