#!/usr/bin/env python3
"""Tail-round quantisation of the N = 768 residual GEMMs.  `<F16,7>` (out_proj K = 768, c_proj K = 3072) runs one
256 x 256 tile per workgroup and one workgroup per CU (160 KiB LDS), i.e. a launch is ceil(M / 256) x 3 tiles on 256 CUs; this sweeps M
across whole and fractional rounds and prints us per launch, us per tile-round and the deviation from the straight line through the
whole-round points.  LEAF_GEMM_TAIL=0|1 (if the library knows it) switches the half-tile tail on / off.
    python tools/tail_sweep.py [--k 768 3072] [--mtiles 320 341 342 ...]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leaf_amd import _lib


def timed(fn, it, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[768, 3072])
    ap.add_argument("--mtiles", type=int, nargs="+",
                    default=[256, 299, 320, 341, 342, 352, 363, 374, 384, 395, 405, 416, 426, 427, 448, 469, 512, 555, 597, 598])
    ap.add_argument("--rows-off", type=int, default=0, help="rows added to every M (ragged last tile)")
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    N = 768
    out = []
    for K in a.k:
        Mmax = max(a.mtiles) * 256 + a.rows_off
        A = (torch.randn(Mmax, K, device=dev) * 0.5).half()
        B = (torch.randn(N, K, device=dev) * 0.05).half()
        bias = torch.randn(N, device=dev)
        Cm = torch.zeros(Mmax, N, device=dev)
        x16 = torch.empty(Mmax, N, device=dev, dtype=torch.float16)
        stat = torch.empty(N // 64, Mmax, 2, device=dev)
        rows = []
        for mt in a.mtiles:
            M = mt * 256 + a.rows_off
            tiles = ((M + 255) // 256) * 3
            args = (1, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), C.c_void_p(bias.data_ptr()),
                    C.c_void_p(x16.data_ptr()), C.c_void_p(stat.data_ptr()), M, N, K, st)
            _lib.check(lib.leaf_op_gemm_resid_ln(*args), "gemm")
            us = timed(lambda: lib.leaf_op_gemm_resid_ln(*args), a.iters)
            rows.append(dict(K=K, M=M, tiles=tiles, rounds=tiles / 256.0, us=us, tflops=2.0 * M * N * K / us / 1e6))
        # straight line through the origin-free fit of the points nearest to whole rounds (fractional part <= 0.02 from below)
        whole = [r for r in rows if 0 <= (-r["rounds"]) % 1.0 <= 0.03]
        if len(whole) >= 2:
            x = torch.tensor([r["rounds"] for r in whole]); y = torch.tensor([r["us"] for r in whole])
            slope = float(((x - x.mean()) * (y - y.mean())).sum() / ((x - x.mean()) ** 2).sum())
            icpt = float(y.mean() - slope * x.mean())
        else:
            slope, icpt = rows[-1]["us"] / rows[-1]["rounds"], 0.0
        print(f"K = {K}: line through the whole-round points: {icpt:.1f} + {slope:.2f} us per round of 256 tiles")
        for r in rows:
            r["linear_us"] = icpt + slope * r["rounds"]
            r["over_linear"] = r["us"] / r["linear_us"] - 1.0
            print(f"  M {r['M']:7d}  tiles {r['tiles']:5d}  rounds {r['rounds']:6.3f}  {r['us']:8.1f} us  {r['tflops']:7.1f} TF/s  "
                  f"vs linear {100 * r['over_linear']:+5.1f} %", flush=True)
        out += rows
    if a.json:
        with open(a.json, "w") as f:
            json.dump(dict(tail=os.environ.get("LEAF_GEMM_TAIL", ""), rows=out), f, indent=1)


if __name__ == "__main__":
    main()
