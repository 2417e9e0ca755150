#!/usr/bin/env python3
"""One readable line from a bench.py JSON line on stdin: step time + ms per step of the big launches' GEMM shapes (tools/ab.sh).
    python bench.py ... | tail -1 | python tools/bench_line.py LABEL [--small]     (--small: the B-caption launches instead)"""
import json
import sys

NAMES = {("gemm_nt256_half_kernel<F16,6>", 3072, 768): "c_fc", ("gemm_nt256_half_kernel<F16,5>", 2304, 768): "qkv",
         ("qkv_attn_kernel<F16,5>", 2304, 768): "fused", ("gemm_nt256_half_kernel<F16,7>", 768, 3072): "c_proj",
         ("gemm_nt256_half_kernel<F16,7>", 768, 768): "out_proj", ("gemm_nt256_half_kernel<F16,8>", 768, 3072): "c_proj",
         ("gemm_nt256_half_kernel<F16,8>", 768, 768): "out_proj"}


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else ""
    small = "--small" in sys.argv
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    if "roofline" not in d:
        print(f"{label:28s} {d.get('ms_per_step', float('nan')):7.2f} ms/step", flush=True)
        return
    parts = []
    for s in d["roofline"]["shapes"]:
        if bool(s.get("big_launches")) == small:
            continue
        nm = NAMES.get((s["kernel"], s["N"], s["K"]), f"{s['kernel'].split('_kernel')[0]}<{s['kernel'].split(',')[-1]} {s['N']}x{s['K']}")
        parts.append(f"{nm} {s['ms_per_step']:.2f} ({s['tflops']:.0f})")
    print(f"{label:28s} {d['ms_per_step']:7.2f} ms/step  {d['value']:7.1f} samples/s | " + "  ".join(parts[:8]), flush=True)


if __name__ == "__main__":
    main()
