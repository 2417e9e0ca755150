#!/usr/bin/env python3
"""GPU busy share from a rocprofv3 kernel trace (csv): kernel intervals of all streams merged over the LAST `frac` of the traced span
(the steady iterations of a tool that has no per-step marker), gaps listed by size class.   usage: trace_busy.py <kernel_trace.csv> [frac]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
t0, t1 = iv[0][0], max(e for _, e in iv)
a = t1 - frac * (t1 - t0)
busy_ns, cur, gaps = 0, a, []
for s, e in iv:
    if e <= a:
        continue
    s = max(s, a)
    if s > cur:
        gaps.append(s - cur)
        cur = s
    if e > cur:
        busy_ns += e - cur
        cur = e
span = t1 - a
print(f"last {frac:.0%} of the trace: {span / 1e6:.1f} ms, a kernel running {100 * busy_ns / span:.1f} % of it; {len(gaps)} gaps, "
      f"{sum(gaps) / 1e6:.2f} ms in all: > 1 ms: {sum(g for g in gaps if g > 1e6) / 1e6:.2f} ms ({sum(g > 1e6 for g in gaps)}), "
      f"0.1-1 ms: {sum(g for g in gaps if 1e5 < g <= 1e6) / 1e6:.2f} ms ({sum(1e5 < g <= 1e6 for g in gaps)}), "
      f"10-100 us: {sum(g for g in gaps if 1e4 < g <= 1e5) / 1e6:.2f} ms ({sum(1e4 < g <= 1e5 for g in gaps)}), "
      f"< 10 us: {sum(g for g in gaps if g <= 1e4) / 1e6:.2f} ms ({sum(g <= 1e4 for g in gaps)})")
