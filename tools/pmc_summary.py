"""Summarise rocprofv3 PMC passes into the per-kernel CSV / traffic JSON kept under profiles/.

    python tools/pmc_summary.py FETCH_DIR WRITE_DIR OUT_PREFIX [KERNEL_SUBSTR] [BENCH_JSON]

BENCH_JSON: the line bench.py printed in the FETCH pass (tools/pmc_passes.sh sets LEAF_BENCH_JSON_OUT): the algorithmic bytes per
launch of the same kernel in the SAME run are stored beside the measured traffic (the step is data dependent, a short PMC run
and a long timing run process different row counts).

FETCH_DIR / WRITE_DIR are the `-d` directories of two SEPARATE passes
(`rocprofv3 --pmc FETCH_SIZE --kernel-trace …` and `--pmc WRITE_SIZE --kernel-trace …`) of the same bench command.
Writes OUT_PREFIX_pmc_fetch.csv, OUT_PREFIX_pmc_write.csv (mean KB per dispatch, per kernel) and OUT_PREFIX_traffic.json
for the kernel with the largest summed fetch (or the one matching KERNEL_SUBSTR).  gfx950 correction
(MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests at 64 B, so fetch bytes are doubled.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return acc


def write_csv(path, acc, counter):
    rows = sorted(acc.items(), key=lambda kv: -kv[1][0])
    with open(path, "w") as f:
        f.write("kernel,counter,mean_value_KB,dispatches,sum_KB\n")
        for k, (s, n) in rows:
            f.write(f'"{k}",{counter},{s / n:.1f},{n},{s:.1f}\n')


def short(name):
    """'void (anonymous namespace)::gemm_nt256_half_kernel<F16, 7, false>(GemmArgs)' -> 'gemm_nt256_half_kernel<F16,7>' (the name
    bench.py builds from its profiler key: kernel family, operand type, epilogue id)."""
    m = re.search(r"(\w+)<(\w+), (\d+)(?:, \w+)*>", name)
    return f"{m.group(1)}<{m.group(2)},{m.group(3)}>" if m else name


def main():
    fd, wd, out = sys.argv[1:4]
    want = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] else None
    bench_json = sys.argv[5] if len(sys.argv) > 5 else None
    fe, wr = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    write_csv(out + "_pmc_fetch.csv", fe, "FETCH_SIZE")
    write_csv(out + "_pmc_write.csv", wr, "WRITE_SIZE")
    gemms = {k: v for k, v in fe.items() if ("gemm_nt" in k or "qkv_attn" in k) and (want is None or want in k)}
    from bench import kernel_sources_hash
    algo, bj = None, None
    if bench_json and os.path.exists(bench_json):
        bj = json.load(open(bench_json))
    # the kernel the bench line names as dominant (largest summed time), else the one with the largest summed fetch
    dom = bj["roofline"]["kernel"].split(" ")[0] if bj else None
    named = [k for k in gemms if short(k) == dom]
    k = named[0] if named else max(gemms, key=lambda x: gemms[x][0])
    fkb, wkb = fe[k][0] / fe[k][1], wr[k][0] / wr[k][1]
    if bj and dom == short(k):
        algo = bj["roofline"]["algorithmic_bytes_per_launch"]
    traffic = (2 * fkb + wkb) * 1024
    # every big kernel of the step beside it: measured bytes per launch against the algorithmic bytes of the SAME run (mean over
    # the kernel's shapes, weighted by launches), and what of it was fetched from beyond L2
    kernels = []
    if bj:
        per = defaultdict(lambda: [0.0, 0])
        for sh in bj["roofline"]["shapes"]:
            if sh.get("algorithmic_bytes_per_launch") and sh.get("big_launches"):
                per[sh["kernel"]][0] += sh["algorithmic_bytes_per_launch"] * sh["launches"]
                per[sh["kernel"]][1] += sh["launches"]
        for kk in gemms:
            nm = short(kk) if "qkv_attn" not in kk else "qkv_attn_kernel<%s,5>" % ("F16" if "F16" in kk else "BF16")
            big_disp = fe[kk][0] / fe[kk][1] > 20 * 1024             # > 20 MB fetched per dispatch: the scoring passes' launches
            if nm in per and per[nm][1] and big_disp:
                f_, w_ = fe[kk][0] / fe[kk][1] * 1024, wr[kk][0] / wr[kk][1] * 1024
                a_ = per[nm][0] / per[nm][1]
                kernels.append({"kernel": nm, "dispatches": fe[kk][1], "fetch_bytes_per_launch": 2 * f_, "write_bytes_per_launch": w_,
                                "algorithmic_bytes_per_launch": a_, "traffic_over_algorithmic": (2 * f_ + w_) / a_})
        kernels.sort(key=lambda r: -r["fetch_bytes_per_launch"] * r["dispatches"])
        seen = set()          # persistent and plain instantiations share a short name: keep the one that moved the most bytes
        kernels = [r for r in kernels if not (r["kernel"] in seen or seen.add(r["kernel"]))]
    json.dump({
        "kernel": short(k),
        "kernel_sources_sha16": kernel_sources_hash(),   # bench.py attaches this summary only to lines from the same kernel sources
        "workload_key": bj["config"].get("workload_key") if bj else None,   # ... and the same workload
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-dense-leg`, mean over that kernel's dispatches (tools/pmc_summary.py)",
        "fetch_kb_mean": fkb, "write_kb_mean": wkb,
        "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM)",
        "traffic_bytes_per_launch": traffic,
        "algorithmic_bytes_per_launch": algo,      # of the SAME run (bench.py's line of the FETCH pass), or null
        "traffic_over_algorithmic": traffic / algo if algo else None,
        "kernels": kernels,
    }, open(out + "_traffic.json", "w"), indent=1)
    print(short(k), "fetch KB", fkb, "write KB", wkb, "traffic B/launch", (2 * fkb + wkb) * 1024)


if __name__ == "__main__":
    main()
