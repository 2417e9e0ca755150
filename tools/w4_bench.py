#!/usr/bin/env python3
"""Round-5 experiment: the four-wave 128 x 128-per-wave GEMM (variants/gemm256w4.hip, LEAF_GEMM_W4=1, diagnostic build) against the
shipped eight-wave kernel on the bias + 16-bit-store epilogue, same operands; prints time, TF/s and a SHA-1 of the output (the two
kernels must produce the same bits).   LEAF_HIP_LIB=tools/diag/libleaf_hip_variants.so [LEAF_GEMM_W4=1] python tools/w4_bench.py"""
import ctypes as C
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leaf_amd import _lib


def main():
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = [int(x) for x in os.environ.get("ROWS", "107520,118016,3219").split(",")]
    print("LEAF_GEMM_W4 =", os.environ.get("LEAF_GEMM_W4", "0"), "| LEAF_GEMM_PERSIST =", os.environ.get("LEAF_GEMM_PERSIST", "1"))
    for M in rows:
        for name, N, K in [("qkv", 2304, 768), ("fc", 3072, 768), ("proj", 768, 3072), ("out", 768, 768)]:
            g = torch.Generator(device=dev).manual_seed(M + N + K)
            A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
            B = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
            bias = torch.randn(N, device=dev, generator=g)
            Cm = torch.zeros(M, N, device=dev, dtype=torch.float16)
            args = (1, 0, C.c_void_p(A.data_ptr()), K, C.c_void_p(B.data_ptr()), K, C.c_void_p(Cm.data_ptr()), N, C.c_void_p(bias.data_ptr()),
                    None, M, N, K, 1, 0.0, 0, st)
            for _ in range(3):
                _lib.check(lib.leaf_op_gemm_ld(*args), "gemm")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 10
            e0.record()
            for _ in range(it):
                lib.leaf_op_gemm_ld(*args)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / it
            ref = (A[:512].float() @ B.float().t() + bias).half()
            err = float((Cm[:512].float() - ref.float()).abs().max())
            sha = hashlib.sha1(Cm.cpu().numpy().tobytes()).hexdigest()[:12]
            print(f"M {M:7d} {name:5s} N={N:5d} K={K:5d}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s  max|err| vs torch (512 rows) {err:.3g}  sha1 {sha}", flush=True)


if __name__ == "__main__":
    main()
