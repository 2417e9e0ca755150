#!/usr/bin/env python3
"""Time the GEMM kernel alone (C-ABI hook leaf_op_gemm) on the text-tower shapes; random fp16 data."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leaf_amd import _lib


def main():
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    seqs = int(os.environ.get("SEQS", "1024"))
    M = seqs * 77
    d = int(os.environ.get("WIDTH", "768"))
    shapes = [("qkv", 0, 3 * d, d), ("out", 2, d, d), ("fc", 1, 4 * d, d), ("proj", 2, d, 4 * d)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot_ms, tot_fl = 0.0, 0.0
    pad = int(os.environ.get("PAD", "0"))     # extra elements per row (stride experiments)
    zero = os.environ.get("ZERO", "0") == "1"  # zero operands: same instruction stream, far less switching energy (clock probe)
    for name, epi, N, K in shapes:
        A = (torch.randn(M, K + pad, device=dev) * 0.5).half()
        B = (torch.randn(N, K + pad, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev)
        if zero:
            A.zero_(); B.zero_(); bias.zero_()
        Cm = torch.zeros(M, N + pad, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
        args = (1, epi, C.c_void_p(A.data_ptr()), K + pad, C.c_void_p(B.data_ptr()), K + pad, C.c_void_p(Cm.data_ptr()),
                N + pad, C.c_void_p(bias.data_ptr()), None, M, N, K, 1, 0.0, 0, st)
        lib.leaf_op_gemm = lib.leaf_op_gemm_ld
        for _ in range(3):
            _lib.check(lib.leaf_op_gemm(*args), "gemm")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 10
        e0.record()
        for _ in range(it):
            lib.leaf_op_gemm(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        fl = 2.0 * M * N * K
        tot_ms += ms
        tot_fl += fl
        print(f"{name:5s} M={M} N={N} K={K}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s", flush=True)
    print(f"layer GEMMs: {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
