#!/usr/bin/env python3
"""The fused QKV + attention launch alone (C-ABI hook leaf_op_qkv_attn) on row maps shaped like the two stages of the benchmark's
search: B = 128 captions of 10..42 tokens, rho = 50 candidates each;  stage 1: the edit position is uniform inside the caption,
stage 2: all candidates of a caption share ONE early position.  Prints time, GEMM-equivalent TFLOP/s and -- with the diagnostic
build (make -C leaf_amd/csrc stamps; LEAF_HIP_LIB=tools/diag/libleaf_hip_stamps.so) -- where a workgroup's cycles go.

    python tools/qkv_attn_bench.py [--heads 12 --width 768 --batch 128 --rho 50]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from leaf_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=768)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--rho", type=int, default=50)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda:0")
    d, heads, B, rho = a.width, a.width // 64, a.batch, a.rho
    rng = np.random.default_rng(0)
    cap_len = rng.integers(10, 43, B).astype(np.int32)
    base_cu = np.zeros(B + 1, np.int32); np.cumsum(cap_len, out=base_cu[1:])
    kv = (torch.randn(int(base_cu[-1]), 3 * d, device=dev) * 0.5).half()
    Wp = (torch.randn(3 * d, d, device=dev) * 0.03).half()
    cvec = torch.randn(3 * d, device=dev) * 0.1
    svec = Wp.float().sum(1).contiguous()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    stamps_on = "stamps" in os.environ.get("LEAF_HIP_LIB", "")
    phases_on = "phases" in os.environ.get("LEAF_HIP_LIB", "")      # make -C leaf_amd/csrc qa_phases
    for stage in (1, 2):
        if stage == 1:
            pfx = np.concatenate([rng.integers(1, L - 1, rho) for L in cap_len]).astype(np.int32)
        else:
            pfx = np.concatenate([np.full(rho, max(1, int(rng.integers(1, max(2, L // 3))))) for L in cap_len]).astype(np.int32)
        lens = (np.repeat(cap_len, rho) - pfx).astype(np.int32)
        n = lens.size
        cu = np.zeros(n + 1, np.int32); np.cumsum(lens, out=cu[1:])
        rows = int(cu[-1])
        x16 = (torch.randn(rows, d, device=dev)).half()
        rowstat = torch.stack([x16.float().mean(1), 1.0 / x16.float().std(1)], 1).contiguous()
        out = torch.zeros(rows, d, device=dev, dtype=torch.float16)
        cu_d, pfx_d, bcu_d = (torch.from_numpy(v).to(dev) for v in (cu, pfx, base_cu))
        tile_seq = torch.zeros(2 * (n + 2), dtype=torch.int32, device=dev)
        P = lambda t: C.c_void_p(t.data_ptr())
        args = (1, P(x16), P(Wp), P(cvec), P(svec), P(rowstat), P(out), P(kv), C.c_void_p(lens.ctypes.data), P(cu_d), P(pfx_d), P(bcu_d),
                None, P(tile_seq), n, rows, rho, 77, heads, d, st)
        for _ in range(3):
            _lib.check(lib.leaf_op_qkv_attn(*args), "qkv_attn")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            lib.leaf_op_qkv_attn(*args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        plan = np.zeros(2 * (n + 2), np.int32)
        nt = _lib.diag_lib().leaf_debug_qkv_attn_plan(C.c_void_p(lens.ctypes.data), 77, 0, n, 1, rho, 0, 0, 0, C.c_void_p(plan.ctypes.data))
        per_tile = np.diff(plan[0:2 * (nt + 1):2])
        fl = 2.0 * rows * 3 * d * d
        print(f"stage {stage}: {n} sequences, {rows} rows ({rows / n:.1f} per sequence), {nt} M tiles ({rows / nt:.0f} rows, {per_tile.mean():.1f} "
              f"sequences each, max {per_tile.max()}) x {heads} heads: {ms:.3f} ms, {fl / ms / 1e9:.0f} TFLOP/s of projection work", flush=True)
        if stamps_on:
            nblk = nt * heads
            slots = 16 if phases_on else 8
            stamps = torch.zeros(nblk * slots, dtype=torch.int64, device=dev)
            lib.leaf_debug_gemm_stamps(C.c_void_p(stamps.data_ptr()))
            lib.leaf_op_qkv_attn(*args)
            torch.cuda.synchronize()
            lib.leaf_debug_gemm_stamps(None)
            raw = stamps.cpu().numpy().reshape(nblk, slots)
            s = raw[:, :6].astype(np.float64)
            seg = np.diff(s, axis=1)
            tot = s[:, 5] - s[:, 0]
            names = ["first DMA wait", "K loop", "tables + caption DMA issue + staging", "wait for DMA / barrier", "attention"]
            print(f"   median workgroup {np.median(tot):.0f} ticks ({nblk} workgroups)")
            for i, nm in enumerate(names):
                print(f"   {nm:40s} median {np.median(seg[:, i]):8.0f}  mean {seg[:, i].mean():8.0f}  ({100 * seg[:, i].sum() / tot.sum():5.1f}%)")
            if phases_on:
                ph = raw[:, 8:14].astype(np.float64)
                pn = ["table + K fragments", "Q + S MFMAs + mask + max", "exp2 + sum + rcp", "P + V^T reads + PV MFMAs", "pack + store issue", "loop overhead"]
                print(f"   wave 0's attention stage, phases serialised by waits (shares, not times): total mean {ph.sum(1).mean():.0f} ticks")
                for i, nm in enumerate(pn):
                    print(f"      {nm:36s} mean {ph[:, i].mean():8.0f}  ({100 * ph[:, i].sum() / ph.sum():5.1f}%)")


if __name__ == "__main__":
    main()
