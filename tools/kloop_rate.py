#!/usr/bin/env python3
"""In-loop rate and per-tile overhead of a 256 x 256 GEMM kernel from the SLOPE of time over K (round 5): the same M x N at K, 2K, 4K --
t(K) = overhead + K / rate.  Runs this repo's kernel (the bias + 16-bit store epilogue; LEAF_GEMM_W4=1 with the diagnostic build selects
the four-wave experiment) and the vendor library (torch.matmul -> hipBLASLt) on the same operands.
    [LEAF_HIP_LIB=tools/diag/libleaf_hip_variants.so LEAF_GEMM_W4=1] python tools/kloop_rate.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leaf_amd import _lib


def timed(fn, it=8, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
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
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = int(os.environ.get("ROWS", "107520"))
    print("LEAF_GEMM_W4 =", os.environ.get("LEAF_GEMM_W4", "0"), "| LEAF_GEMM_PERSIST =", os.environ.get("LEAF_GEMM_PERSIST", "1"), "| rows", M)
    for N in (768, 3072):
        Ks = [768, 1536, 3072, 6144]
        rows = {"ours": [], "vendor": []}
        for K in Ks:
            A = (torch.randn(M, K, device=dev) * 0.5).half()
            B = (torch.randn(N, K, device=dev) * 0.05).half()
            bias = torch.randn(N, device=dev)
            Cm = torch.zeros(M, N, device=dev, dtype=torch.float16)
            args = (1, 0, C.c_void_p(A.data_ptr()), K, C.c_void_p(B.data_ptr()), K, C.c_void_p(Cm.data_ptr()), N, C.c_void_p(bias.data_ptr()),
                    None, M, N, K, 1, 0.0, 0, st)
            _lib.check(lib.leaf_op_gemm_ld(*args), "gemm")
            rows["ours"].append(timed(lambda: lib.leaf_op_gemm_ld(*args)))
            Bt = B.t()
            rows["vendor"].append(timed(lambda: torch.matmul(A, Bt, out=Cm)))
            del A, B, Cm
        tiles = (M // 256) * (N // 256)
        for who, t in rows.items():
            # slope between the two largest K: ms per unit K -> in-loop TF/s; intercept = everything that does not scale with K
            slope = (t[-1] - t[-2]) / (Ks[-1] - Ks[-2])
            rate = 2.0 * M * N / slope / 1e9
            icpt = t[-1] - slope * Ks[-1]
            print(f"N={N:5d} {who:6s}: " + "  ".join(f"K={k}: {x:.3f} ms ({2.0 * M * N * k / x / 1e9:.0f})" for k, x in zip(Ks, t))
                  + f"  | in-loop {rate:.0f} TF/s, overhead {icpt * 1e3:.0f} us per launch = {icpt * 1e3 / (tiles / 256.0):.1f} us per round of 256 tiles", flush=True)


if __name__ == "__main__":
    main()
