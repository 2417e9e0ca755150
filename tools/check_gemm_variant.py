#!/usr/bin/env python3
"""Correctness of whichever GEMM kernel variant the environment selects (LEAF_GEMM256H, LEAF_GEMM64, LEAF_GEMM64_DEEP,
LEAF_GEMM64_MI, LEAF_GEMM_BM64 are read once per process): a few shapes x epilogues through the C-ABI hook against a
float64 product of the rounded operands.  Used by tests/test_gpu_variants.py in a subprocess per variant; exits non-zero
on the first mismatch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import _lib


def main():
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(0)
    worst = 0.0
    for M, N, K in [(11085, 768, 256), (8300, 1024, 192), (3234, 768, 3072), (2500, 768, 768), (700, 2304, 768), (333, 256, 448)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        a16 = torch.from_numpy(A).to(dev).half()
        b16 = torch.from_numpy(B).to(dev).half()
        ref = a16.double() @ b16.double().T
        tb = torch.from_numpy(bias).to(dev)
        x0 = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(dev)
        for epi in (0, 2, 3):
            c = torch.zeros(M, N, dtype=torch.float16, device=dev) if epi == 0 else x0.clone()
            rc = lib.leaf_op_gemm(1, epi, C.c_void_p(a16.data_ptr()), C.c_void_p(b16.data_ptr()), C.c_void_p(c.data_ptr()),
                                  C.c_void_p(tb.data_ptr()) if epi != 3 else None, None, M, N, K, 0, 1.0 if epi == 3 else 0.0, 0, st)
            if rc:
                print("launch failed:", lib.leaf_last_error().decode())
                return 2
            torch.cuda.synchronize()
            want = ref + tb.double() if epi == 0 else (x0.double() + ref + (tb.double() if epi == 2 else 0.0))
            err = float(((c.double() - want).norm() / want.norm()).item())
            worst = max(worst, err)
            tol = 2e-3 if epi == 0 else 1e-5
            if not np.isfinite(err) or err > tol:
                print(f"MISMATCH M={M} N={N} K={K} epi={epi}: rel-L2 {err:.3e} > {tol}")
                return 1
    print(f"ok worst rel-L2 {worst:.3e}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
