#!/usr/bin/env python3
"""Table for tools/vendor_counters.sh: per (shape, rows) one row for the vendor library's kernel and one for this repo's, from the
rocprofv3 --pmc passes under OUTDIR/<shape>_<rows>/<pass>/.   python tools/vendor_counters.py OUTDIR [--json out.json]
Columns as tools/pmc_util.py (MI355X_MICROARCH.md: effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy =
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8); FETCH_SIZE doubled on gfx950, both in KB)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

SHAPES = {"fc": (3072, 768), "proj": (768, 3072), "qkv": (2304, 768), "out": (768, 768)}


def load(d):
    by = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, n]
    dur = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if not ("Cijk" in k or "gemm_nt" in k):
                continue
            by[k][r["Counter_Name"]][0] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:          # one duration per dispatch (a dispatch has one csv line per counter and per XCD/instance)
                seen.add(key)
                dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                dur[k][1] += 1
    out = {}
    for k, cs in by.items():
        n = dur[k][1]
        out[k] = dict(n=n, us=dur[k][0] / n / 1e3, **{c: s / n for c, (s, _) in cs.items()})
    return out


def label(k):
    if "Cijk" in k:
        return "vendor " + ("(+bias) " if "_Bias" in k else "") + re.sub(r"_(?:SK|UserArgs|shortname).*", "", k)[:46]
    m = re.search(r"(\w+)<(\w+), (\d+)(?:, (\w+))*>", k)
    return f"this repo {m.group(1)}<{m.group(2)},{m.group(3)}{',P' if ', true>' in k else ''}>" if m else k


def main():
    root = sys.argv[1]
    rows = []
    for tagdir in sorted(glob.glob(os.path.join(root, "*_*"))):
        if not os.path.isdir(tagdir):
            continue
        shape, M = os.path.basename(tagdir).rsplit("_", 1)
        N, K = SHAPES[shape]
        fl = 2.0 * int(M) * N * K
        passes = {p: load(os.path.join(tagdir, p)) for p in ("sq1", "sq2", "tcc", "fetch", "write")}
        for k in sorted(passes["sq1"], key=lambda k: passes["sq1"][k]["us"]):
            a = passes["sq1"][k]
            if a["us"] < 50:
                continue
            gui = a.get("GRBM_GUI_ACTIVE", 0.0)
            clk = gui / 8.0 / (a["us"] * 1e3)
            busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * gui / 8.0) if gui else 0.0
            b, t = passes["sq2"].get(k, {}), passes["tcc"].get(k, {})
            fe, wr = passes["fetch"].get(k, {}).get("FETCH_SIZE", 0.0), passes["write"].get(k, {}).get("WRITE_SIZE", 0.0)
            hit = t.get("TCC_HIT_sum", 0.0) / max(1.0, t.get("TCC_HIT_sum", 0.0) + t.get("TCC_MISS_sum", 0.0))
            rows.append(dict(shape=shape, M=int(M), N=N, K=K, kernel=label(k), dispatches=a["n"], us=a["us"], tflops=fl / a["us"] / 1e6,
                             eff_clock_ghz=clk, mfma_busy=busy, busy_x_clock=busy * clk, frac_of_2p5pf=busy * clk / 2.4,
                             lds_conflict=b.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, b.get("SQ_LDS_IDX_ACTIVE", 0.0)),
                             valu_insts=b.get("SQ_INSTS_VALU", 0.0), lds_insts=b.get("SQ_INSTS_LDS", 0.0),
                             wait_share=a.get("SQ_WAIT_ANY", 0.0) / max(1.0, a.get("SQ_WAVE_CYCLES", 0.0)),
                             stall_share=a.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, a.get("SQ_WAVE_CYCLES", 0.0)),
                             waves=a.get("SQ_WAVES", 0.0), l2_hit=hit, fetch_mb=2 * fe * 1024 / 1e6, write_mb=wr * 1024 / 1e6))
    print("| shape | rows | kernel | n | us (under PMC) | TF/s | clock GHz | MFMA busy | busy x clock | of 2.5 PF | LDS confl | wait / stall | waves | L2 hit | fetch MB | write MB |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['shape']} {r['N']}x{r['K']} | {r['M']} | {r['kernel']} | {r['dispatches']} | {r['us']:.1f} | {r['tflops']:.0f} | {r['eff_clock_ghz']:.2f} | "
              f"{100 * r['mfma_busy']:.1f} % | {r['busy_x_clock']:.3f} | {r['frac_of_2p5pf']:.3f} | {100 * r['lds_conflict']:.1f} % | "
              f"{100 * r['wait_share']:.0f} / {100 * r['stall_share']:.0f} % | {r['waves']:.0f} | {100 * r['l2_hit']:.1f} % | {r['fetch_mb']:.0f} | {r['write_mb']:.0f} |")
    if "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
