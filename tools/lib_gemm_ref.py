#!/usr/bin/env python3
"""Calibration, not product: what the vendor GEMM library (torch.matmul / torch.addmm -> hipBLASLt or rocBLAS) reaches on this box for the
text tower's four GEMM shapes at a scoring pass's row count, next to this repo's kernels (C-ABI hook leaf_op_gemm) on the SAME random
fp16 operands.  The library call computes the bare product (+ bias for addmm) into fp16 -- none of the fused epilogues (LayerNorm fold,
activation, fp32 residual read-modify-write, row statistics) -- so it is an upper reference for the K loop, not a like-for-like kernel.
  python tools/lib_gemm_ref.py            (SEQS, WIDTH as tools/gemm_bench.py; ROWS overrides the row count; ONLY=fc,proj picks shapes; ITERS)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leaf_amd import _lib


def timed(fn, it=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def main():
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    d = int(os.environ.get("WIDTH", "768"))
    M = int(os.environ.get("ROWS", "107520"))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print(f"rows {M}, width {d}; torch {torch.__version__}, preferred BLAS library: {torch.backends.cuda.preferred_blas_library()}")
    tot = {"lib": 0.0, "ours": 0.0, "fl": 0.0}
    only = [x for x in os.environ.get("ONLY", "").split(",") if x]      # e.g. ONLY=fc: one shape per process (PMC passes: one vendor kernel per run)
    iters = int(os.environ.get("ITERS", "20"))
    for name, epi, N, K in [("qkv", 0, 3 * d, d), ("out", 2, d, d), ("fc", 1, 4 * d, d), ("proj", 2, d, 4 * d)]:
        if only and name not in only:
            continue
        A = (torch.randn(M, K, device=dev) * 0.5).half()
        B = (torch.randn(N, K, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev)
        bias16 = bias.half()
        out16 = torch.empty(M, N, device=dev, dtype=torch.float16)
        Bt = B.t()
        t_mm = timed(lambda: torch.matmul(A, Bt, out=out16), iters)
        t_addmm = timed(lambda: torch.addmm(bias16, A, Bt, out=out16), iters)
        Cm = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
        args = (1, epi, C.c_void_p(A.data_ptr()), K, C.c_void_p(B.data_ptr()), K, C.c_void_p(Cm.data_ptr()), N, C.c_void_p(bias.data_ptr()),
                None, M, N, K, 1, 0.0, 0, st)
        t_own = timed(lambda: lib.leaf_op_gemm_ld(*args), iters)
        fl = 2.0 * M * N * K
        tot["lib"] += min(t_mm, t_addmm); tot["ours"] += t_own; tot["fl"] += fl
        print(f"{name:5s} N={N:5d} K={K:5d}:  library matmul {t_mm:.3f} ms {fl / t_mm / 1e9:6.0f} TF/s | addmm(+bias) {t_addmm:.3f} ms "
              f"{fl / t_addmm / 1e9:6.0f} TF/s | this repo (epilogue {epi}) {t_own:.3f} ms {fl / t_own / 1e9:6.0f} TF/s", flush=True)
    print(f"layer: library {tot['lib']:.3f} ms {tot['fl'] / tot['lib'] / 1e9:.0f} TF/s | this repo {tot['ours']:.3f} ms {tot['fl'] / tot['ours'] / 1e9:.0f} TF/s")


if __name__ == "__main__":
    main()
