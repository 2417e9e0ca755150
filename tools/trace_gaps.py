#!/usr/bin/env python3
"""GPU idle time inside steady bench steps, from a rocprofv3 kernel trace (csv of `rocprofv3 --kernel-trace --output-format csv --
python3 bench.py --steps 6 --warmup 3 ...`): the kernel intervals of all streams are merged over the last 3 steps (delimited by
the AdamW launches) and the gaps in which NO kernel runs are listed by the kernels on either side.
usage: trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))


def nm(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"at::native::", "", n)
    return n.split("(")[0][:44]


iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm(r["Kernel_Name"])) for r in rows)
ad = [i for i, x in enumerate(iv) if x[2].startswith("adamw_kernel")]
a, b = iv[ad[-4]][1], iv[ad[-1]][1]
sel = [x for x in iv if x[0] >= a and x[1] <= b]
busy, last, gaps = a, "adamw_kernel", []
for s, e, n in sel:
    if s > busy:
        gaps.append((s - busy, last, n, (busy - a) / 1e6))
    if e > busy:
        busy, last = e, n
tot = sum(g[0] for g in gaps)
print(f"3 steps: {(b - a) / 3e6:.2f} ms per step, {len(sel) / 3:.0f} kernels per step, idle {tot / 3e6:.2f} ms per step in {len(gaps) / 3:.0f} gaps")
by = collections.defaultdict(lambda: [0, 0])
for g, x, y, _ in gaps:
    by[(x, y)][0] += g
    by[(x, y)][1] += 1
for k, (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"{g / 3e3:8.1f} us/step in {c / 3:6.1f} gaps (avg {g / c / 1e3:6.1f} us)  after {k[0]:44s} before {k[1]}")
print("largest:")
for g, x, y, at in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:8.1f} us at {at:7.2f} ms  after {x} before {y}")
