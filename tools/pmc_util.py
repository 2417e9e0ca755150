"""MFMA utilisation / effective clock / L2 hit rate / LDS conflicts per kernel from rocprofv3 PMC passes.

    python tools/pmc_util.py OUT.json DIR [DIR ...]

Every DIR is the `-d` directory of ONE pass `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -- python3 bench.py
--steps 2 --warmup 1 --no-cpu-baseline --no-dense-leg` (program directly behind `--`; SQ / GRBM / TCC counters in passes of
their own, never together with other trace domains).  For every kernel family of the step the JSON holds the mean per
dispatch of each counter, the mean duration of the same dispatches, and what follows from them
(MI355X_MICROARCH.md: 'DVFS give-back', 'rocprofv3 PMC slots', 'Per-instruction cycle constants'):

  eff_clock_ghz   GRBM_GUI_ACTIVE / 8 XCDs / duration  (reads high below ~0.3 ms per dispatch);
  mfma_busy_frac  SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): share of the chip's MFMA pipe-cycles in use;
  mfma_peak_frac_at_clock   the same x eff_clock / 2.4 GHz = fraction of the NOMINAL 2.5 PF the MFMA pipes delivered;
  l2_hit          TCC_HIT / (TCC_HIT + TCC_MISS);
  lds_conflict    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;
  wait_share / issue_stall_share / active_share   SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+)<(\w+), (\d+)(?:, (\w+))*>", name)
    if m and ("gemm" in m.group(1)):
        tail = ",P" if name.find(", true>") >= 0 else ""
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}{tail}>"
    m = re.search(r"(?:\(anonymous namespace\)::)?(\w+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def load(dirs):
    """One record per dispatch and pass: kernel, grid, duration, the pass's counters."""
    recs = []
    for d in dirs:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            by_id = {}
            for r in csv.DictReader(open(f)):
                e = by_id.get(r["Dispatch_Id"])
                if e is None:
                    e = by_id[r["Dispatch_Id"]] = {"kernel": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]),
                                                   "ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), "c": {}}
                e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            recs.extend(by_id.values())
    return recs


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    groups = defaultdict(list)
    for r in load(dirs):
        groups[(r["kernel"], r["grid"])].append(r)
    # one kernel name can carry two GEMM shapes on the same grid (out_proj K = 768 and c_proj K = 3072 of the residual epilogue):
    # split such a group at the geometric mean of its extreme durations
    split = {}
    for (k, g), rs in groups.items():
        lo, hi = min(r["ns"] for r in rs), max(r["ns"] for r in rs)
        if "gemm" in k and hi > 1.8 * lo and len(rs) >= 8:
            thr = (lo * hi) ** 0.5
            split.setdefault(k + "/short", []).extend(r for r in rs if r["ns"] < thr)
            split.setdefault(k + "/long", []).extend(r for r in rs if r["ns"] >= thr)
        else:
            split.setdefault(k, []).extend(rs)
    res = {}
    for k, rs in split.items():
        cs = defaultdict(lambda: [0.0, 0])
        for r in rs:
            for n, v in r["c"].items():
                cs[n][0] += v; cs[n][1] += 1
        c = {n: s_ / cnt for n, (s_, cnt) in cs.items()}
        mean_ns = sum(r["ns"] for r in rs) / len(rs)
        e = {"dispatches": len(rs), "mean_us": mean_ns / 1e3, "counters": c}
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui:
            own = [r["ns"] for r in rs if "GRBM_GUI_ACTIVE" in r["c"]]      # durations of the passes that carried the counter
            e["eff_clock_ghz"] = gui / 8.0 / (sum(own) / len(own))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * gui / 8.0)
                e["mfma_peak_frac_at_clock"] = e["mfma_busy_frac"] * e["eff_clock_ghz"] / 2.4
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
        w = c.get("SQ_WAVE_CYCLES")
        if w:
            for nm, key in (("SQ_WAIT_ANY", "wait_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"), ("SQ_ACTIVE_INST_ANY", "active_share")):
                if nm in c:
                    e[key] = c[nm] / w
        res[k] = e
    keep = sorted(res.items(), key=lambda kv: -kv[1]["mean_us"] * kv[1]["dispatches"])
    json.dump({"source": "rocprofv3 --pmc passes (tools/pmc_util.py); counters are means per dispatch; a kernel name that carries two GEMM "
                         "shapes on one grid is split by duration (/short, /long)", "passes": dirs, "kernels": dict(keep[:32])},
              open(out, "w"), indent=1)
    for k, e in keep[:20]:
        print(f"{k:48s} n={e['dispatches']:5d} {e['mean_us']:8.1f} us  clk {e.get('eff_clock_ghz', 0):.2f} GHz  "
              f"mfma {100 * e.get('mfma_busy_frac', 0):5.1f} %  L2 {100 * e.get('l2_hit', 0):5.1f} %  "
              f"lds-confl {100 * e.get('lds_conflict', 0):4.1f} %  wait {100 * e.get('wait_share', 0):4.1f} % stall {100 * e.get('issue_stall_share', 0):4.1f} %")


if __name__ == "__main__":
    main()
