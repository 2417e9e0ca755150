#!/usr/bin/env python3
"""Static check of the persistent GEMM epilogue's inline-asm LDS flush (gemm256h.hip, flush16): in every persistent
instantiation each of the eight ds_read_b128 destinations must be stored only after the s_waitcnt that releases it, and must
not be copied in between (nothing ties an in-flight LDS read to its registers except program order).
usage: check_flush_asm.py <gemm256h.s>   (hipcc -S --cuda-device-only ... gemm256h.hip)"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
names = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN.*gemm_nt256_half_kernel\w*Lb1E\w*:", l)]
bad = 0
for start, name in names:
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    i, nblocks = 0, 0
    while i + 7 < len(body):
        if "ds_read_b128" in body[i] and "offset:7168" in body[i + 7]:
            regs = [re.search(r"ds_read_b128 (v\[\d+:\d+\])", body[i + k]).group(1) for k in range(8)]
            nblocks += 1
            seen, j, waited = 0, i + 8, set()
            while seen < 8 and j < len(body):
                l = body[j]
                m = re.search(r"s_waitcnt .*lgkmcnt\((\d+)\)", l)
                if m:
                    waited |= {k for k in range(8) if 7 - k >= int(m.group(1))}
                m = re.search(r"global_store_dwordx4 v\[\d+:\d+\], (v\[\d+:\d+\])", l)
                if m:
                    seen += 1
                    if m.group(1) not in regs or regs.index(m.group(1)) not in waited:
                        bad += 1
                        print(name[-40:], "store of", m.group(1), "before its wait / of an unexpected register")
                for k, r in enumerate(regs):
                    lo = int(re.search(r"v\[(\d+):", r).group(1))
                    ops = l.split(",")[1:] if "," in l else []
                    if k not in waited and re.search(r"\bv_(mov|pk_mov|accvgpr)", l) and any(re.search(rf"\bv{lo + t}\b|v\[{lo + t}:", o) for o in ops for t in range(4)):
                        bad += 1
                        print(name[-40:], "copy of an in-flight register:", l.strip())
                j += 1
            i = j
        else:
            i += 1
    print(name[-36:], "flush blocks", nblocks)
print("OK" if not bad else f"{bad} PROBLEMS")
sys.exit(1 if bad else 0)
