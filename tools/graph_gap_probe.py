#!/usr/bin/env python3
"""What a captured graph would save on the B-caption chains (round 5): N dependent small launches queued one by one on a stream against
the same launches replayed as ONE hipGraph.  The kernels are stand-ins (in-place elementwise passes over a 3,219 x 768 fp32 tensor and
3,219-row fp16 GEMMs through torch), the question is the GAP between dependent launches, not the kernels.
    python tools/graph_gap_probe.py [N]"""
import sys
import time

import torch


def run(kind, n):
    dev = torch.device("cuda:0")
    x = torch.randn(3219, 768, device=dev)
    a = torch.randn(3219, 768, device=dev, dtype=torch.float16)
    w = torch.randn(768, 768, device=dev, dtype=torch.float16) * 0.02

    def body():
        if kind == "elementwise":
            for _ in range(n):
                x.mul_(1.0001)
        else:
            y = a
            for _ in range(n):
                y = torch.matmul(y, w)
            return y
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            body()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            body()
        e1.record(s)
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) / 5
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            body()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
        graph = e0.elapsed_time(e1) / 5
    # one launch alone, to split kernel time from gap
    with torch.cuda.stream(s):
        e0.record(s)
        if kind == "elementwise":
            x.mul_(1.0001)
        else:
            torch.matmul(a, w)
        e1.record(s)
        torch.cuda.synchronize()
        one = e0.elapsed_time(e1)
    print(f"{kind:12s} {n} dependent launches: stream {eager * 1e3 / n:6.2f} us per launch, graph replay {graph * 1e3 / n:6.2f} us per launch "
          f"(one launch alone, event to event: {one * 1e3:.1f} us)", flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for kind in ("elementwise", "gemm"):
        run(kind, n)
