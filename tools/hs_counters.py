"""Per-launch beyond-L2 fetch bytes (2 x FETCH_SIZE KB, the gfx950 correction of MI355X_MICROARCH.md) and L2 hit rate of the big
kernels of one `rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum` pass:  python tools/hs_counters.py <dir> <label>"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "qkv_attn" not in k and "half_kernel" not in k:
            continue
        k = k.split("(anonymous namespace)::")[-1].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k] += r["Counter_Name"] == "FETCH_SIZE"
for k, a in sorted(acc.items()):
    print("HSPLIT=%s %-44s n=%4d  fetched beyond L2 %7.0f MB/launch   L2 hit %.1f %%" % (
        sys.argv[2], k, n[k], 2 * a["FETCH_SIZE"] * 1024 / n[k] / 1e6, 100 * a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])))
